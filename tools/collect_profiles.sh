#!/bin/bash
# Run on the GPU box from the repo root (gpurun -- 'bash tools/collect_profiles.sh r01'): rocprofv3 kernel stats and the
# two HBM-traffic counter passes of the default bench.py workload. Summaries land in gpurun_out/<tag>/ ; copy the ones
# to be judged into profiles/.
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$R'); from bench import source_hash; print(source_hash())" > $OUT/src_sha.txt
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample-reads 1000000 > $OUT/bench_line.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o r -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d > $OUT/stats_bench.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o r -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-h2d > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o r -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-h2d > /dev/null 2> $OUT/write.err
cd $R
python3 tools/rocpd_summary.py $OUT/stats/r_results.db > $OUT/kernel_stats.md
python3 tools/rocpd_summary.py $OUT/fetch/r_results.db | sed -n '/counter/,$p' > $OUT/pmc_fetch.md
python3 tools/rocpd_summary.py $OUT/write/r_results.db | sed -n '/counter/,$p' > $OUT/pmc_write.md
rm -rf $OUT/stats $OUT/fetch $OUT/write
tail -1 $OUT/bench_line.json | cut -c1-400
