#!/usr/bin/env python3
"""profiles/<round>_emulated_rank.md from profiles/<round>_emulated_rank.json (tools/refresh_profiles.sh; tools/emulate_rank.py --protocol words).
Usage: tools/write_emulated_rank_md.py r04 [r03]   (second argument: the round to compare with)"""
import json
import sys

rnd = sys.argv[1]
prev = sys.argv[2] if len(sys.argv) > 2 else None
cur = json.load(open(f"profiles/{rnd}_emulated_rank.json"))
old = json.load(open(f"profiles/{prev}_emulated_rank.json")) if prev else {}
L = [f"# {rnd} — one rank of an 8-GPU job at full bucket depth, emulated on ONE MI355X (tools/emulate_rank.py --protocol words)", "",
     "No multi-GPU node was available. What rank 0 (the densest prefix range) of a W = 8 job RECEIVES is produced on one GPU: eight shards of reads are encoded and",
     "split by destination one after the other, the words of rank 0's range are kept in (slice, source) order, and the receiver's direct pipeline runs on them",
     "(`cblx_insert_words_device`: pass A included, histograms from records — the wire and its overlap are measured separately, `" + rnd + "_wire_emulated.md`).", "",
     "| workload (rank 0 of 8) | words received | buckets | mean bucket | receiver ms" + (f" ({prev})" if prev else "") + " | partition (hist + scan + scatter) | directory | bucket stage |", "|---|---|---|---|---|---|---|---|"]
for k, v in cur.items():
    b = v["build"]
    st = b["receiver_stage_ms"]
    part = sum(st.get(x, 0) for x in ("radix_hist", "radix_scan", "radix_scatter"))
    bk = {x: round(y, 1) for x, y in st.items() if x.startswith("bucket")}
    pv = old.get(k, {}).get("build", {}).get("receiver_ms")
    L.append(f"| {k} | {b['words_received'] / 1e6:.0f} M | {b['buckets']} | {b['bucket_len_mean']:.0f} | {b['receiver_ms']:.1f}" + (f" ({pv:.1f})" if pv else "") +
             f" | {part:.1f} | {st.get('directory', 0):.2f} | **{sum(bk.values()):.1f}** ({', '.join(f'{x} {y}' for x, y in bk.items())}) |")
L += ["", "`A |= B` of two such shares (every both-sided bucket is Trie |= Trie at this depth: `k_bucket_union`, merge path straight from the two arenas; round 3 sent them through",
      "the long-run path: one more partition pass into a twin buffer + sub-range sorts):", "", "| workload | words self + other | ms (3 runs)" + (f" | {prev}" if prev else "") + " | stage split |", "|---|---|---|" + ("---|" if prev else "") + "---|"]
for k, v in cur.items():
    m = v.get("merge")
    if not m:
        continue
    pm = (old.get(k, {}).get("merge") or {}).get("ms")
    L.append(f"| {k} | {m['self_words'] / 1e6:.0f} M + {m['other_words'] / 1e6:.0f} M | {m['ms']}" + (f" | {pm}" if prev else "") + f" | {m['stage_ms']} |")
open(f"profiles/{rnd}_emulated_rank.md", "w").write("\n".join(L) + "\n")
print("wrote", f"profiles/{rnd}_emulated_rank.md")
