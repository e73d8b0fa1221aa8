#!/usr/bin/env python3
"""profiles/<round>_wire_emulated.md from the JSON lines of tools/emulate_wire.py (one file per rehearsal: configuration, world size, rehearsed rank,
protocol). Usage: tools/write_wire_profile.py out.md <round> wire_cfg3.json [wire_cfg3_rank7.json wire_w2_bins.json ...]"""
import json
import sys


def main():
    out, rnd, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    L = [f"# {rnd} — ONE rank of a W-GPU job on ONE MI355X with the wire PACED (tools/emulate_wire.py, cblx_comm_init_sim)", "",
         "No multi-GPU node was available. The other W - 1 ranks run one after the other on recording communicators (what each would send the rehearsed rank stays",
         "in HBM: headers and record bytes, per exchange); the rehearsed rank then runs the REAL sharded insert — its own KRN-1 + first pass, the receiver's passes,",
         "directory and bucket kernels at the bucket depth of the W-GPU job — on a replaying communicator whose exchanges are D2D copies on a side stream, held back by",
         "a host function until a wire of `link` GB/s per source rank (W - 1 links in parallel) would have delivered them. ms per step, best of 3; `no wire` = the",
         "copies at their own speed (what the kernels alone take). Not emulated: the CUs RCCL's kernels occupy, link contention, the peers' own pace. Grouped runs",
         "with three slices use 50 / 30 / 20 % of the reads (`ShardedBuilder.GROUPED_WEIGHTS`). Round 5: PREFIX_BITS > 24 runs on FINE bins (DESIGN_HISTORY.md §3.12, §5.8:",
         "`fine groups` = groups of the rehearsed rank that sort 16 prefix bits behind the first pass, in two LSD passes), any rank can be rehearsed (rank 0 = the",
         "densest prefix range, rank W - 1 = the sparse tail), and the rank bounds are cost-weighted quantiles (the tail's histogram cells count 1.20 x); 8 bytes per word on the wire (the digit byte stays home).",
         "Round 6: runs that end up sorted take `k_bucket_sorted` (DESIGN.md §3.1); protocol \"replicate\" (DESIGN.md §5.1): the READS cross as bit planes (0.3 bytes per k-mer, one",
         "grouped all-gather up front), the rehearsed rank transforms every rank's reads and keeps its prefix range — `bytes received` are planes and offsets.", ""]
    for f in files:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        W, rk, proto = o["world"], o.get("rank", 0), o.get("protocol", "bins")
        L += [f"## {o['config']}, W = {W}, rank {rk}, protocol \"{proto}\": K={o['k']} PREFIX_BITS={o['prefix_bits']}, {o['reads_per_rank']} x {o['read_len']} bp per rank = "
              f"{o['kmers_per_rank'] / 1e9:.2f} G k-mers per rank", ""]
        if "direct_one_gpu_ms" in o:
            L += [f"The same reads as a one-GPU build (no exchange): {o['direct_one_gpu_ms']} ms.", ""]
        rates = sorted({r["link_gbps"] for r in o["runs"]}, key=lambda x: (x == 0, x))
        modes = []
        for r in o["runs"]:
            key = (r["mode"], r.get("groups", 0), r["slices"])
            if key not in modes:
                modes.append(key)
        L += ["| receiver | " + " | ".join(("no wire" if x == 0 else f"{x:g} GB/s per link") for x in rates) + " | words in the rank's index | bytes received per step | wire alone at 55 GB/s |",
              "|---|" + "---|" * (len(rates) + 3)]
        for m in modes:
            row = {r["link_gbps"]: r for r in o["runs"] if (r["mode"], r.get("groups", 0), r["slices"]) == m}
            any_r = next(iter(row.values()))
            if m[0] == "grouped":
                name = f"grouped, {any_r['groups_used']} groups ({any_r.get('groups_fine', 0)} fine), {m[2]} slice{'s' if m[2] > 1 else ''}"
            elif m[0] == "sorted":
                name = f"\"sorted\" protocol, {m[2]} slices"
            elif m[0] == "replicate":
                name = f"\"replicate\" protocol (reads as bit planes), {any_r['groups_used']} groups ({any_r.get('groups_fine', 0)} fine), {m[2]} slice{'s' if m[2] > 1 else ''}"
            else:
                name = f"ungrouped (round 3), {m[2]} slices"
            L.append(f"| {name} | " + " | ".join(f"**{row[x]['ms_best']:.1f}**" if x in row else "—" for x in rates) +
                     f" | {any_r['words_in_index'] / 1e9:.3f} G | {any_r['recv_bytes_per_step'] / 1e9:.2f} GB | {any_r['recv_bytes_per_step'] / max(W - 1, 1) / 55e9 * 1e3:.1f} ms |")
        L.append("")
        km = o["kmers_per_rank"]
        for m in modes:
            row = {r["link_gbps"]: r for r in o["runs"] if (r["mode"], r.get("groups", 0), r["slices"]) == m}
            if 55.0 in row:
                ms = row[55.0]["ms_best"]
                one = f" = {W * km / ms / (km / o['direct_one_gpu_ms']):.2f} x the one-GPU rate of the same configuration" if "direct_one_gpu_ms" in o else ""
                L.append(f"* {m[0]}{' ' + str(m[1]) + ' groups, ' + str(m[2]) + ' slice(s)' if m[0] == 'grouped' else ''} at 55 GB/s: {W} x {km / 1e9:.2f} G / {ms:.1f} ms = "
                         f"{W * km / ms / 1e6:.0f} G k-mers/s if every rank kept this rank's pace{one}.")
        g = [r for r in o["runs"] if r["mode"] in ("grouped", "sorted", "replicate") and r["link_gbps"] == 0]
        u = [r for r in o["runs"] if r["mode"] == "ungrouped" and r["link_gbps"] == 0]
        if g:
            L += ["", "Stage times of the last step without a wire (HIP events on the rank's stream; ms):", "", "| stage | " + g[0]["mode"] + (" | ungrouped |" if u else " |"), "|---|---|" + ("---|" if u else "")]
            for st in sorted(set(g[0]["stage_ms_last_step"]) | (set(u[0]["stage_ms_last_step"]) if u else set())):
                L.append(f"| {st} | {g[0]['stage_ms_last_step'].get(st, 0)} |" + (f" {u[0]['stage_ms_last_step'].get(st, 0)} |" if u else ""))
        L.append("")
    open(out, "w").write("\n".join(L) + "\n")


if __name__ == "__main__":
    main()
