#!/bin/bash
# round 5 dev: groups x slice taper of the grouped receiver at 55 GB/s per link (rank 0 of 8, cfg 3). Usage: gpurun -- 'bash tools/r5_sweep.sh <tag>'
TAG=${1:-r5sw}; CFG=${2:-cfg3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for t in "0.5,0.8,1" "0.6,0.9,1" "0.4,0.7,0.9,1" "0.35,0.6,0.8,0.93,1"; do
  ns=$(echo $t | tr ',' '\n' | wc -l)
  timeout 900 python tools/emulate_wire.py --config $CFG --groups 4,6,8 --grouped-slices $ns --taper $t --wire-gbps 55,0 --no-ungrouped --no-direct --steps 3 > $OUT/sweep_${ns}_$t.json 2> $OUT/sweep_$t.err
  echo "taper $t:"; grep -o '"groups": [0-9]*\|"link_gbps": [0-9.]*\|"ms_best": [0-9.]*' $OUT/sweep_$t.err | paste - - - | tr '\t' ' '
done
