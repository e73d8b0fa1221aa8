#!/usr/bin/env python3
"""profiles/r04_wire_emulated.md from the JSON lines of tools/emulate_wire.py (one file per configuration).
Usage: tools/write_wire_profile.py out.md wire_cfg3.json [wire_cfg2.json ...]"""
import json
import sys


def main():
    out, files = sys.argv[1], sys.argv[2:]
    L = ["# r04 — rank 0 of an 8-GPU job on ONE MI355X with the wire PACED (tools/emulate_wire.py, cblx_comm_init_sim)", "",
         "No multi-GPU node was available. Ranks 1..7 run one after the other on recording communicators (what each would send rank 0 stays in HBM:",
         "headers and record bytes, per exchange); rank 0 then runs the REAL sharded insert — its own KRN-1 + pass A, the receiver's passes, directory and",
         "bucket kernels at the bucket depth of the 8-GPU job — on a replaying communicator whose exchanges are D2D copies on a side stream, held back by a",
         "host function until a wire of `link` GB/s per source rank (7 links in parallel) would have delivered them. ms per step of rank 0 (the densest",
         "prefix range), best of 3; `0` = the copies at their own speed (no wire: what the kernels alone take). Not emulated: the CUs RCCL's kernels",
         "occupy, link contention, the other ranks being slower than rank 0. Grouped runs with three slices use 50 / 30 / 20 % of the reads",
         "(`ShardedBuilder.GROUPED_WEIGHTS`: only the first group's share of the LAST slice is exposed).", ""]
    for f in files:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        W = o["world"]
        L += [f"## {o['config']}: K={o['k']} PREFIX_BITS={o['prefix_bits']}, {o['reads_per_rank']} x {o['read_len']} bp per rank = {o['kmers_per_rank'] / 1e9:.2f} G k-mers per rank, W = {W}", "",
              f"The same reads as a one-GPU build (no exchange): {o['direct_one_gpu_ms']} ms.", ""]
        rates = sorted({r["link_gbps"] for r in o["runs"]}, key=lambda x: (x == 0, x))
        modes = []
        for r in o["runs"]:
            key = (r["mode"], r.get("groups", 0), r["slices"])
            if key not in modes:
                modes.append(key)
        L += ["| receiver | " + " | ".join(("no wire" if x == 0 else f"{x:g} GB/s per link") for x in rates) + " | bytes received per step | wire alone at 55 GB/s |", "|---|" + "---|" * (len(rates) + 2)]
        for m in modes:
            row = {r["link_gbps"]: r for r in o["runs"] if (r["mode"], r.get("groups", 0), r["slices"]) == m}
            any_r = next(iter(row.values()))
            name = (f"grouped, {any_r['groups_used']} groups, {m[2]} slice{'s' if m[2] > 1 else ''}" if m[0] == "grouped" else f"ungrouped (round 3), {m[2]} slices")
            L.append(f"| {name} | " + " | ".join(f"**{row[x]['ms_best']:.1f}**" if x in row else "—" for x in rates) +
                     f" | {any_r['recv_bytes_per_step'] / 1e9:.2f} GB | {any_r['recv_bytes_per_step'] / (W - 1) / 55e9 * 1e3:.1f} ms |")
        L.append("")
        km = o["kmers_per_rank"]
        for m in modes:
            row = {r["link_gbps"]: r for r in o["runs"] if (r["mode"], r.get("groups", 0), r["slices"]) == m}
            if 55.0 in row:
                ms = row[55.0]["ms_best"]
                L.append(f"* {m[0]}{' ' + str(m[1]) + ' groups, ' + str(m[2]) + ' slice(s)' if m[0] == 'grouped' else ''} at 55 GB/s: {W} x {km / 1e9:.2f} G / {ms:.1f} ms = {W * km / ms / 1e6:.0f} G k-mers/s if every rank kept rank 0's pace "
                         f"= {W * km / ms / (km / o['direct_one_gpu_ms']):.2f} x the one-GPU rate of the same configuration.")
        g = [r for r in o["runs"] if r["mode"] == "grouped" and r["link_gbps"] == 0]
        u = [r for r in o["runs"] if r["mode"] == "ungrouped" and r["link_gbps"] == 0]
        if g and u:
            L += ["", "Stage times of the last step without a wire (HIP events on rank 0's stream; ms):", "", "| stage | grouped | ungrouped |", "|---|---|---|"]
            for st in sorted(set(g[0]["stage_ms_last_step"]) | set(u[0]["stage_ms_last_step"])):
                L.append(f"| {st} | {g[0]['stage_ms_last_step'].get(st, 0)} | {u[0]['stage_ms_last_step'].get(st, 0)} |")
        L.append("")
    open(out, "w").write("\n".join(L) + "\n")


if __name__ == "__main__":
    main()
