"""Dev probe (round 5): stage times of ONE fine-bins build at cfg 3 / cfg 4 size for the library given by CBLX_LIB_PATH. A timing-probe
library (-DCBLX_ENC_PROBE=2: KRN-1 without the flush of its tile counts) fails behind KRN-1 by design; its encode time is still recorded."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbl_amd
from cbl_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
k, pb, n, L = {"cfg3": (31, 28, 12_500_000, 150), "cfg4": (59, 28, 6_250_000, 250), "cfg2": (31, 24, 10_000_000, 150)}[cfg]
d_b, d_o = synth.reads_torch(42, n, L, device="cuda")
g = cbl_amd.CBL(k, pb, profile=True)
res = {}
for i in range(4):
    g.clear()
    if i == 1:
        g.stage_times_reset()
    try:
        g.insert_seqs_device(d_b, d_o, n)
    except cbl_amd.CblxError as e:
        res["error"] = str(e)[:80]
    torch.cuda.synchronize()
st = g.stage_times()
print(cfg, os.environ.get("CBLX_LIB_PATH", "default"), "fine_bins", os.environ.get("CBLX_FINE_BINS", "1"),
      {k_: round(v[0] / 3, 3) for k_, v in st.items() if v[0] > 0}, res)
