#!/usr/bin/env python3
"""gpurun_out/<tag>/{sq1,sq2,sq3,grbm}.md (tools/collect_counters.sh) -> profiles/<round>_sq_counters.md: the SQ counters of the
compute-bound kernels per dispatch, with the derived figures the design discussion uses.
Usage: tools/write_sq_profile.py <tag> <round> <kmers_per_step>"""
import re
import sys

tag, rnd, nk = sys.argv[1], sys.argv[2], float(sys.argv[3])
base = f"gpurun_out/{tag}/"
vals, res = {}, {}
for f in ("sq1", "sq2", "sq3", "grbm"):
    try:
        for line in open(base + f + ".md"):
            m = re.match(r"\| (.+?) \| ([A-Za-z]\w+) \| (\d+) \| ([\d.e+]+) \| ([\d.e+]+) \|", line)
            if m:
                vals.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(5))
                continue
            m = re.match(r"\| (void )?(cblx::.+?) \| (\d+) \| 1 \| 1 \| (\d+) \| 1 \| 1 \| (\d+) \| (\d+) \| (\d+) \| (\d+) \| (\d+) \|", line)
            if m:
                res[("void " if m.group(1) else "") + m.group(2).strip()] = dict(grid=int(m.group(3)), wg=int(m.group(4)), lds=int(m.group(5)), scratch=int(m.group(6)),
                                                                                  vgpr=int(m.group(7)), agpr=int(m.group(8)), sgpr=int(m.group(9)))
    except OSError:
        pass
CU = 256
out = [f"# {rnd} — SQ counters of the compute-bound kernels (rocprofv3 --pmc, one pass per group of <= 8 counters, kernel-trace only)\n",
       f"Command: `bash tools/collect_counters.sh {tag} cfg2` on the GPU box = `rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --config cfg2 --steps 1 "
       "--warmup 0 --no-cpu-baseline --no-h2d`, program directly after `--`, four passes (sq1 / sq2 / sq3 / grbm). Values are per dispatch, summed over the "
       "whole chip (8 XCDs, 256 CUs). `*_CYCLES` / `WAIT_*` / `ACTIVE_INST_*` count quad-cycles of waves (MI355X_MICROARCH.md); LDS_IDX_ACTIVE / LDS_BANK_CONFLICT are "
       "LDS-array cycles summed over the CUs. Workload: cfg 2, 1.2 G k-mers per step.\n"]
keys = [k for k in vals if any(t in k for t in ("k_encode<", "k_bucket_msd<", "k_bucket_sorted<", "k_radix_scatter<"))]
for k in sorted(keys):
    v, r = vals[k], res.get(k, {})
    out.append(f"\n## `{k.replace('void ', '')}`\n")
    if r:
        waves_wg = r["wg"] // 64
        lds_wgs = (160 * 1024) // max(r["lds"], 1)
        out.append(f"workgroup {r['wg']} threads ({waves_wg} waves), LDS {r['lds']} B per workgroup (=> at most {min(lds_wgs, 32 // waves_wg)} workgroups = "
                   f"{min(lds_wgs, 32 // waves_wg) * waves_wg} waves per CU by LDS / wave slots), arch VGPRs {r['vgpr']}, SGPRs {r['sgpr']}, scratch {r['scratch']} B, grid {r['grid']} threads.\n")
    out.append("| counter | per dispatch |\n|---|---|")
    for c in sorted(v):
        out.append(f"| {c} | {v[c]:.4g} |")
    d = []
    if "SQ_WAVES" in v and "SQ_INSTS_VALU" in v:
        d.append(f"VALU instructions per wave {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:.0f}, SALU {v.get('SQ_INSTS_SALU', 0) / v['SQ_WAVES']:.0f}, LDS {v.get('SQ_INSTS_LDS', 0) / v['SQ_WAVES']:.0f}")
    if "SQ_LDS_IDX_ACTIVE" in v and "GRBM_GUI_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
        busy = v["SQ_LDS_IDX_ACTIVE"] / CU / cyc
        conf = v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v["SQ_LDS_IDX_ACTIVE"], 1)
        d.append(f"kernel {cyc:.3g} shader cycles; LDS array busy {100 * busy:.0f} % of them per CU, {100 * conf:.0f} % of the LDS cycles are bank / address conflicts "
                 f"(address conflicts alone: {100 * v.get('SQ_LDS_ADDR_CONFLICT', 0) / max(v['SQ_LDS_IDX_ACTIVE'], 1):.0f} %)")
        if "SQ_INSTS_VALU" in v:
            d.append(f"VALU issue time if every instruction took one 2-cycle slot: {100 * v['SQ_INSTS_VALU'] * 2 / (CU * 4) / cyc:.0f} % of the kernel")
    if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v:
        wc = v["SQ_WAVE_CYCLES"]
        d.append(f"wave time: {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f} % issuing, {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} % stalled at issue "
                 f"(of which LDS {100 * v.get('SQ_WAIT_INST_LDS', 0) / wc:.0f} %), {100 * v['SQ_WAIT_ANY'] / wc:.0f} % parked on s_waitcnt / barriers")
    if "k_encode" in k and "SQ_INSTS_VALU" in v:
        d.append(f"VALU instructions per k-mer (64 k-mers per wave instruction): {v['SQ_INSTS_VALU'] * 64 / nk:.0f}")
    if d:
        out.append("\nDerived: " + "; ".join(d) + ".")
open(f"profiles/{rnd}_sq_counters.md", "w").write("\n".join(out) + "\n")
print("wrote", f"profiles/{rnd}_sq_counters.md", len(keys), "kernels")
