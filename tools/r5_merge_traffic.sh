#!/bin/bash
# round 5: HBM traffic (FETCH_SIZE / WRITE_SIZE, one --pmc pass each, kernel-trace only) of the merge bench's kernels.
# Usage: gpurun -- 'bash tools/r5_merge_traffic.sh <tag>'
TAG=${1:-r5mt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cn in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $cn -d $OUT/$cn -o r -- python3 $R/bench.py --config merge --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/$cn.err
  python3 $R/tools/rocpd_summary.py $OUT/$cn/r_results.db | sed -n '/counter/,$p' | grep -E "counter|---|k_bucket|k_classify_merge|k_merge" > $OUT/$cn.md
  rm -rf $OUT/$cn
done
cat $OUT/FETCH_SIZE.md $OUT/WRITE_SIZE.md
