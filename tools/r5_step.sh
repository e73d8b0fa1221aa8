#!/bin/bash
# round 5: one GPU session after a change to the fine-bins route: parity of every path that takes it, then the A/B numbers.
# Usage: gpurun -- 'bash tools/r5_step.sh <tag> [tests] [probe] [wire]'
TAG=${1:-r5s}; shift
WHAT=${@:-tests probe wire}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for w in $WHAT; do
  case $w in
    tests) timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_sharded or rehearsal or fine_bins or streamed_insert" --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log ;;
    probe) for c in cfg3 cfg4; do for f in 1 0; do CBLX_FINE_BINS=$f python tools/dev_fine_probe.py $c 2>&1 | tail -1; done; done ;;
    wire)  CBLX_FINE_BINS=1 timeout 900 python tools/emulate_wire.py --config cfg3 --groups 4 --wire-gbps 55,0 --no-ungrouped --no-direct > $OUT/wire_cfg3.json 2> $OUT/wire_cfg3.err; echo "wire rc=$?"
           grep -o '"link_gbps": [0-9.]*, "ms": \[[^]]*\], "ms_best": [0-9.]*\|"groups_fine": [0-9]*\|"stage_ms_last_step": {[^}]*}' $OUT/wire_cfg3.err | tr '\n' ' '; echo ;;
  esac
done
