#!/bin/bash
# One GPU-box session: the -m gpu suite, then the bench lines of every workload. Usage: gpurun -- 'bash tools/gpu_round.sh <tag> [what...]'
TAG=${1:-run}; shift
WHAT=${@:-tests bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for w in $WHAT; do
  case $w in
    tests)   timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -25 $OUT/pytest.log ;;
    bench)   for c in cfg2 cfg3 cfg4 merge; do
               timeout 900 python bench.py --config $c --steps 5 --warmup 2 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c rc=$?"; tail -1 $OUT/bench_$c.json | cut -c1-600
             done ;;
    shared)  timeout 900 python bench.py --gpus 2 --shared-gpu --reads 2000000 --steps 2 --warmup 1 > $OUT/bench_shared2.json 2> $OUT/bench_shared2.err; echo "shared rc=$?"; tail -1 $OUT/bench_shared2.json | cut -c1-1500
             timeout 900 python bench.py --gpus 2 --shared-gpu --config merge --reads 1000000 --steps 2 --warmup 1 > $OUT/bench_shared2m.json 2> $OUT/bench_shared2m.err; echo "shared merge rc=$?"; tail -1 $OUT/bench_shared2m.json | cut -c1-1500 ;;
  esac
done
