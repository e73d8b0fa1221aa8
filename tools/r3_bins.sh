#!/bin/bash
# round 3: the "bins" protocol — its parity tests, then the per-rank cost of the N-GPU code path (1-rank RCCL group) for both
# native protocols. Usage: gpurun -- 'bash tools/r3_bins.sh <tag> [tests] [bench]'
TAG=${1:-r3bins}; shift
WHAT=${@:-tests bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for w in $WHAT; do
  case $w in
    tests) timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_sharded or sharded_builder_single_rank" --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log ;;
    bench) for c in cfg2 cfg3 cfg4; do
             for p in bins sorted; do
               timeout 600 python bench.py --gpus 1 --force-sharded --config $c --protocol $p --transport native --steps 5 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/native_${p}_$c.json 2> $OUT/native_${p}_$c.err; echo "native $p $c rc=$?"
               tail -1 $OUT/native_${p}_$c.json | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['stage_ms_per_step'])
except Exception as e: print('ERR', e)"
             done
           done ;;
  esac
done
