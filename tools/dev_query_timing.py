"""Dev: batched query (join path) timing at cfg 2, hits vs misses, with per-stage device times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbl_amd
from cbl_amd import synth

NR, L, K, PB = 10_000_000, 150, 31, 24
d_b, d_o = synth.reads_torch(42, NR, L, device="cuda:0")
m_b, m_o = synth.reads_torch(77, NR, L, device="cuda:0")
g = cbl_amd.CBL(K, PB, device=0, profile=True)
g.insert_seqs_device(d_b, d_o, NR)
for name, (b, o) in (("hits", (d_b, d_o)), ("misses", (m_b, m_o)), ("hits", (d_b, d_o))):
    g.stage_times_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = g.contains_seqs_device(b, o, NR)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = {k: round(v[0], 2) for k, v in g.stage_times().items() if v[0] > 0}
    print(name, r, f"{dt * 1e3:.1f} ms", st, "unaccounted (join + host): %.1f ms" % (dt * 1e3 - sum(st.values())))

d_f = torch.zeros(NR * (L - K + 1) + 8, dtype=torch.uint8, device="cuda:0")
for name, (b, o) in (("hits+flags", (d_b, d_o)), ("misses+flags", (m_b, m_o)), ("hits+flags", (d_b, d_o))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = g.contains_seqs_device(b, o, NR, d_f, NR * (L - K + 1))
    torch.cuda.synchronize()
    print(name, r, f"{(time.perf_counter() - t0) * 1e3:.1f} ms", "flags set:", int(d_f.sum(dtype=torch.int64)))
