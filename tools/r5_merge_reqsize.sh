#!/bin/bash
# round 5: request sizes behind the HBM traffic of a bench configuration's kernels (TCC_EA0_RDREQ by size, WRREQ / WRREQ_64B, L2 hits / misses;
# one --pmc pass per group, kernel-trace only). Usage: gpurun -- 'bash tools/r5_merge_reqsize.sh <tag> [config]'
TAG=${1:-r5rq}; CFG=${2:-merge}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o r -- python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/g$i.err
  python3 $R/tools/rocpd_summary.py $OUT/g$i/r_results.db | sed -n '/counter/,$p' | grep -E "counter|---|k_bucket_msd|k_bucket_union|k_radix_scatter" > $OUT/g$i.md
  rm -rf $OUT/g$i
done
cat $OUT/g*.md
