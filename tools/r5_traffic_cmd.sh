#!/bin/bash
# round 5: HBM requests by size (TCC_EA0_RDREQ_64B / _128B, TCC_EA0_WRREQ / _64B) and durations of every cblx kernel of one command.
# Usage: gpurun -- 'bash tools/r5_traffic_cmd.sh <tag> tools/emulate_rank.py --protocol words ...'   (the python script and its arguments)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o r -- python3 $R/$1 "${@:2}" > /dev/null 2> $OUT/g$i.err
  python3 $R/tools/rocpd_summary.py $OUT/g$i/r_results.db | sed -n '/counter/,$p' | grep -E "cblx::" > $OUT/g$i.md
  rm -rf $OUT/g$i
done
rocprofv3 --kernel-trace --stats -d $OUT/ks -o r -- python3 $R/$1 "${@:2}" > /dev/null 2> $OUT/ks.err
python3 $R/tools/rocpd_summary.py $OUT/ks/r_results.db | grep -E "cblx::" > $OUT/ks.md; rm -rf $OUT/ks
python3 - <<PY
import re
rd, wr, ms = {}, {}, {}
for f in ("g1", "g2"):
    for l in open("$OUT/%s.md" % f):
        c = [x.strip() for x in l.strip().strip("|").split("|")]
        if len(c) < 5: continue
        k, cn, nd, tot = c[0][-60:], c[1], int(c[2]), float(c[3])
        if cn == "TCC_EA0_RDREQ_64B_sum": rd[k] = rd.get(k, 0) + 64 * tot
        if cn == "TCC_EA0_RDREQ_128B_sum": rd[k] = rd.get(k, 0) + 128 * tot
        if cn == "TCC_EA0_WRREQ_sum": wr[k] = wr.get(k, 0) + 64 * tot
for l in open("$OUT/ks.md"):
    c = [x.strip() for x in l.strip().strip("|").split("|")]
    if len(c) >= 3: ms[c[0][-60:]] = (int(c[1]), float(c[2]))
print("| kernel | calls | total ms | GB read | GB written | TB/s |")
print("|---|---|---|---|---|---|")
for k in sorted(ms, key=lambda k: -ms[k][1]):
    r, w = rd.get(k, 0) / 1e9, wr.get(k, 0) / 1e9
    if ms[k][1] < 0.05: continue
    print(f"| {k[-90:]} | {ms[k][0]} | {ms[k][1]:.3f} | {r:.2f} | {w:.2f} | {(r + w) / ms[k][1]:.2f} |")
PY
