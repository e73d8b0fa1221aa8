"""Dev probe: build from a FASTA file (cfg-2 reads, single-line records on tmpfs), k-mers/s, for CBLX_PARSE_THREADS values."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

NR, L, K = 10_000_000, 150, 31
fa = "/dev/shm/cblx_rate.fa"
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import cbl_amd
    g = cbl_amd.CBL(K, 24)
    best = 1e9
    for rep in range(3):
        g.clear()
        time.sleep(0.3)  # the previous mapping is torn down by a helper thread: let it finish
        t0 = time.perf_counter()
        assert g.insert_fastx_file(fa) == NR
        t1 = time.perf_counter()
        g.flush()
        t2 = time.perf_counter()
        print("  insert_fastx_file %.1f ms, flush %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), file=sys.stderr)
        best = min(best, time.perf_counter() - t0)
    print("threads", os.environ.get("CBLX_PARSE_THREADS", "default"), "ms %.1f" % (best * 1e3), "G k-mers/s %.2f" % (NR * (L - K + 1) / best / 1e9), "count", g.count(), flush=True)
    sys.exit(0)
from cbl_amd import synth
h_bases, _ = synth.reads(42, NR, L)
with open(fa, "wb") as f:
    step = 1_000_000
    for a0 in range(0, NR, step):
        n = min(step, NR - a0)
        rec = np.empty((n, 11 + L + 1), dtype=np.uint8)
        rec[:, 0], rec[:, 1], rec[:, 10], rec[:, -1] = ord(">"), ord("r"), 10, 10
        ids = np.arange(a0, a0 + n)
        for d in range(8):
            rec[:, 9 - d] = 48 + (ids // 10**d) % 10
        rec[:, 11:11 + L] = h_bases[a0 * L:(a0 + n) * L].reshape(n, L)
        f.write(rec.tobytes())
for t in sys.argv[1:] or ["16"]:
    env = dict(os.environ, CBLX_PARSE_THREADS=t)
    subprocess.run([sys.executable, __file__, "child"], env=env, check=False)
os.remove(fa)
