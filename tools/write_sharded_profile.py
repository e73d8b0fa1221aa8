#!/usr/bin/env python3
"""profiles/<round>_sharded_per_rank.{md,json} from gpurun_out/<tag>/sharded_*.json (tools/r3_lines.sh sharded).
Usage: tools/write_sharded_profile.py <tag> <round>"""
import glob
import json
import os
import sys

tag, rnd = sys.argv[1], sys.argv[2]
rows, raw = [], {}
for f in sorted(glob.glob(f"gpurun_out/{tag}/sharded_*.json")):
    name = os.path.basename(f)[len("sharded_"):-len(".json")]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except (ValueError, IndexError):
        continue
    proto, cfg = name.rsplit("_", 1)
    so = d["sharded_overhead"]
    raw[name] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "sharded_overhead": so, "exchange": d.get("exchange"), "config": d["config"]}
    extra = {k: round(v - so["direct_stage_ms"].get(k, 0.0), 2) for k, v in so["sharded_stage_ms"].items() if abs(v - so["direct_stage_ms"].get(k, 0.0)) >= 0.3}
    rows.append((cfg, proto, so["direct_ms"], so["sharded_ms"], so["ratio"], extra))
rows.sort()
md = [f"# {rnd} — what the N-GPU code path costs ONE rank (1-rank RCCL group on one MI355X)\n",
      "`python bench.py --gpus 1 --force-sharded --config <cfg> --protocol <p> --transport native|torch --steps 5 --warmup 2` (tools/r3_lines.sh sharded).",
      "`direct` = `cblx_insert_seqs_device` of the same reads in the same process, after the timed region. One rank: every word is an \"own\" word, so the",
      "figures are the protocol's fixed costs (slices, bins / batches, receive arena, piece tables, merge), not the wire.\n",
      "| config | protocol | direct ms | sharded ms | ratio | stages that differ by >= 0.3 ms (sharded - direct) |", "|---|---|---|---|---|---|"]
for cfg, proto, di, sh, ra, extra in rows:
    md.append(f"| {cfg} | {proto} | {di:.2f} | {sh:.2f} | {ra:.3f} | {extra} |")
md.append("\n`bins` (round 3): the exchange sits between the first and the second partition pass; `sorted` (round 2): full partition on the sender, packed suffixes on the wire,")
md.append("run merge on the receiver; `words` (round 1, torch transport): KRN-1 + one destination pass, the whole pipeline on the receiver.")
open(f"profiles/{rnd}_sharded_per_rank.md", "w").write("\n".join(md) + "\n")
json.dump(raw, open(f"profiles/{rnd}_sharded_per_rank.json", "w"), indent=1)
print("\n".join(md))
