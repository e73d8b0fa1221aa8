"""dev: which side of `self |= other` differs from the oracle, per configuration."""
import random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import cbl_amd
from oracle import Oracle

def rs(rng, n): return bytes(rng.choice(b"ACGT") for _ in range(n))
rng = random.Random(21)
for k, pb, n, canonical in ((31, 24, 20000, False), (9, 4, 9000, False), (9, 4, 40000, False), (11, 8, 60000, False), (13, 10, 150000, True),
                            (15, 12, 400000, False), (35, 6, 3000, False), (35, 10, 100000, False), (59, 28, 30000, True)):
    s1, s2 = rs(rng, n), rs(rng, n // 2) + rs(rng, 64)
    if n >= 100000: s2 = s2 + s1[n // 3 : n // 3 + n // 4]
    g1, g2 = cbl_amd.CBL(k, pb, canonical=canonical, profile=True), cbl_amd.CBL(k, pb, canonical=canonical)
    o1, o2 = Oracle(k, pb, canonical), Oracle(k, pb, canonical)
    g1.insert_seq(s1), o1.insert_seq(s1); g2.insert_seq(s2), o2.insert_seq(s2)
    g1.flush(); g1.stage_times_reset() if hasattr(g1, "stage_times_reset") else None
    g1 |= g2
    o1.merge(o2)
    st = g1.stage_times()
    print(k, pb, n, canonical, "self", g1.serialize() == o1.serialize(), "other", g2.serialize() == o2.serialize(),
          "count", g1.count(), o1.count(), {a: b for a, b in st.items() if b[1]})
