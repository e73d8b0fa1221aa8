#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (made by tools/collect_profiles.sh on the GPU box) into the committed profiles/<round>_* files.
Usage: tools/write_profiles.py <tag> <round>      e.g.  tools/write_profiles.py r01h r01"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402

tag, rnd = sys.argv[1], sys.argv[2]
base = f"gpurun_out/{tag}/"
line = open(base + "bench_line.json").read().strip().splitlines()[-1]
bl = json.loads(line)


def parse(md):
    out = {}
    for l in open(md):
        m = re.match(r"\| (.+?) \| (\w+) \| (\d+) \| ([\d.e+]+) \| ([\d.e+]+) \|", l)
        if m:
            out[m.group(1)] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
    return out


f, w = parse(base + "pmc_fetch.md"), parse(base + "pmc_write.md")
kA = [k for k in f if "k_radix_scatter<unsigned char" in k][0]
kL = [k for k in f if "k_radix_scatter<cblx::NoHi, cblx::NoHi, cblx::DigitBits" in k][0]
NK = bl["config"]["reads_per_gpu"] * (bl["config"]["read_len"] - bl["config"]["k"] + 1)
la = {"fetch_kb_raw": f[kA][2], "write_kb_raw": w[kA][2], "alg_bytes": 18 * NK}
ll = {"fetch_kb_raw": f[kL][2], "write_kb_raw": w[kL][2], "alg_bytes": int(16.5 * NK)}
hbm = lambda d: (2 * d["fetch_kb_raw"] + d["write_kb_raw"]) * 1024
avg = (hbm(la) + 2 * hbm(ll)) / 3
def group(pred):
    """sum over the launches of a step of every kernel whose name satisfies pred: (launches, fetch KB, write KB)"""
    ks = [k for k in f if pred(k)]
    return sum(f[k][0] for k in ks), sum(f[k][1] for k in ks), sum(w[k][1] for k in ks if k in w)


K_, PB_, L_ = bl["config"]["k"], bl["config"]["prefix_bits"], bl["config"]["read_len"]
sys.path.insert(0, ".")
import bench as _b  # noqa: E402  (the price list the line uses)

alg = _b.stage_alg_bytes(K_, PB_, L_)
per_kernel = {}
for stage, pred in (("encode", lambda k: "k_encode<" in k), ("bucket_medium", lambda k: "k_bucket_msd<" in k or "k_bucket_sorted<" in k or "k_bucket_claim<" in k),
                    ("radix_scatter", lambda k: "k_radix_scatter<" in k), ("radix_hist", lambda k: "k_radix_hist" in k)):
    n, fk, wk = group(pred)
    if n:
        per_kernel[stage] = {"launches_per_step": n, "fetch_kb_raw_per_step": fk, "write_kb_raw_per_step": wk,
                             "hbm_bytes_per_step": int((2 * fk + wk) * 1024), "algorithmic_bytes_per_step": int(alg[stage] * NK)}
tr = {
    "config": {**{k: bl["config"][k] for k in ("k", "prefix_bits", "reads_per_gpu", "read_len")}, "kind": "build"},
    "kernel": "k_radix_scatter",
    # hash of cbl_amd/csrc AS IT WAS ON THE GPU BOX when the counters were taken; bench.py reports `traffic` only while the
    # tree still hashes to this
    "src_sha": (open(base + "src_sha.txt").read().strip() if os.path.exists(base + "src_sha.txt") else source_hash()),
    "launches": {"k_radix_scatter<u8,NoHi> (pass A: 9 B in, 8 B + 1 B digit out)": la,
                 "k_radix_scatter<NoHi,NoHi> (LSD passes, x2: 8 B in, 8 B (+1 B digit) out)": ll},
    "hbm_bytes_per_launch": int(avg),
    "algorithmic_bytes_per_launch": int(51 * NK / 3),
    # the same counters for the other kernel groups that matter, per STEP (all launches of the group): roofline.kernels[].traffic
    "kernels": per_kernel,
    "source": f"profiles/{rnd}_pmc_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE in separate passes, KB per dispatch x 1024; "
              "FETCH doubled per the gfx950 note); average over the 3 scatter launches of a step",
}
json.dump(tr, open(f"profiles/{rnd}_traffic.json", "w"), indent=1)
json.dump(tr, open("profiles/traffic.json", "w"), indent=1)  # the copy bench.py reads
hdr = (f"# {rnd} (final) — rocprofv3 --kernel-trace --stats, bench.py --steps 3 --warmup 1 (cfg 2: K=31, PB=24, 10M x 150 bp, 1 x MI355X)\n\n"
       f"Command: `bash tools/collect_profiles.sh {tag}` on the GPU box (`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 "
       f"--no-cpu-baseline`; 4 steps incl. warm-up; averages are per launch). Same build and box: profiles/{rnd}_bench_line.json "
       f"({bl['ms_per_step']:.1f} ms/step, {bl['value'] / 1e9:.1f} G k-mers/s).\n\n")
rows = open(base + "kernel_stats.md").read().splitlines()
open(f"profiles/{rnd}_kernel_stats.md", "w").write(hdr + "\n".join(rows[:48]) + "\n")
h2 = (f"# {rnd} (final) — HBM traffic counters, separate rocprofv3 --pmc passes (kernel-trace only), bench.py --steps 1 --warmup 0\n\n"
      "FETCH_SIZE / WRITE_SIZE in KB per dispatch (`bash tools/collect_profiles.sh`). On gfx950 FETCH_SIZE reports half of a wide coalesced "
      "streaming read (MI355X_MICROARCH.md, HBM section): double it. k_radix_scatter, average of the three launches of a step: "
      f"{avg / 1e9:.2f} GB of HBM traffic per launch against {51 * NK / 3 / 1e9:.2f} GB algorithmic (records + the 1-byte next-digit side channel).\n\n")
open(f"profiles/{rnd}_pmc_hbm_traffic.md", "w").write(h2 + open(base + "pmc_fetch.md").read() + "\n" + open(base + "pmc_write.md").read())
open(f"profiles/{rnd}_bench_line.json", "w").write(line + "\n")
print(f"hbm {avg / 1e9:.2f} GB/launch, algorithmic {51 * NK / 3 / 1e9:.2f} GB/launch, {bl['ms_per_step']} ms/step")
