#!/bin/bash
# End-of-round evidence on the GPU box: the -m gpu suite, kernel stats + HBM traffic of the default workload, the SQ counter
# passes, and the bench line of every workload. Usage: gpurun -- 'bash tools/final_round.sh <tag>'
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/collect_profiles.sh $TAG
bash tools/collect_counters.sh ${TAG}_sq cfg2
for c in cfg3 cfg4 merge dup; do timeout 900 python bench.py --config $c --steps 5 --warmup 2 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c rc=$?"; done
timeout 900 python tools/emulate_rank.py --merge --serialize > $OUT/emul_cfg5.json 2> $OUT/emul_cfg5.err; echo "emul rc=$?"
