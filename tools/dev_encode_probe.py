"""Dev probe (not shipped): time KRN-1 alone through cblx_seq_words_device, for a library given by CBLX_LIB_PATH."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbl_amd
from cbl_amd import synth

k, pb, n, L = 31, 24, 10_000_000, 150
d_b, d_o = synth.reads_torch(42, n, L, device="cuda")
g = cbl_amd.CBL(k, pb, profile=True)
N = n * (L - k + 1)
lo = torch.empty(N + 64, dtype=torch.int64, device="cuda")
hi = torch.empty(N + 64, dtype=torch.uint8, device="cuda")
for i in range(6):
    if i == 2:
        g.stage_times_reset()
    assert g.seq_words_device(d_b, d_o, n, lo, hi, N) == N
st = g.stage_times()
print(os.environ.get("CBLX_LIB_PATH", "default"), {k_: round(v[0] / 4, 3) for k_, v in st.items() if v[0] > 0})
