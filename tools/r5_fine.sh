#!/bin/bash
# round 5: FINE bins (PREFIX_BITS > 24) — parity of the sharded paths, then rank 0 of 8 at cfg 3 with the bins on / off on the same box.
# Usage: gpurun -- 'bash tools/r5_fine.sh <tag> [tests] [wire]'
TAG=${1:-r5fine}; shift
WHAT=${@:-tests wire}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for w in $WHAT; do
  case $w in
    tests) timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_sharded or rehearsal" --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log ;;
    wire)  for f in 1 0; do
             CBLX_FINE_BINS=$f timeout 900 python tools/emulate_wire.py --config cfg3 --groups 4 --wire-gbps 55,0 --no-ungrouped $([ $f = 0 ] && echo --no-direct) > $OUT/wire_cfg3_fine$f.json 2> $OUT/wire_cfg3_fine$f.err; echo "wire fine=$f rc=$?"
             grep -o '"link_gbps": [0-9.]*, "ms": \[[^]]*\], "ms_best": [0-9.]*\|"groups_fine": [0-9]*\|"stage_ms_last_step": {[^}]*}' $OUT/wire_cfg3_fine$f.err | tr '\n' ' '; echo
           done ;;
    wire4) for f in 1 0; do
             CBLX_FINE_BINS=$f timeout 900 python tools/emulate_wire.py --config cfg4 --groups 4 --wire-gbps 55,0 --no-ungrouped $([ $f = 0 ] && echo --no-direct) > $OUT/wire_cfg4_fine$f.json 2> $OUT/wire_cfg4_fine$f.err; echo "wire4 fine=$f rc=$?"
             grep -o '"link_gbps": [0-9.]*, "ms": \[[^]]*\], "ms_best": [0-9.]*\|"groups_fine": [0-9]*\|"stage_ms_last_step": {[^}]*}' $OUT/wire_cfg4_fine$f.err | tr '\n' ' '; echo
           done ;;
    one)   timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fine_bins or index_matches or incremental or golden" --durations=5 > $OUT/pytest_one.log 2>&1; echo "pytest one rc=$?"; tail -6 $OUT/pytest_one.log
           for c in cfg3 cfg4; do for f in 1 0; do
             CBLX_FINE_BINS=$f timeout 900 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-h2d --no-fasta --no-per-record --no-serialize > $OUT/bench_${c}_fine$f.json 2> $OUT/bench_${c}_fine$f.err; echo "bench $c fine=$f rc=$?"
             python3 -c "import json,sys; d=json.loads(open('$OUT/bench_${c}_fine$f.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
           done; done ;;
  esac
done
