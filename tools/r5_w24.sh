#!/bin/bash
# round 5: the N = 2 and N = 4 points rehearsed — rank 0 of W against a paced wire, "bins" (grouped / ungrouped) vs "sorted", cfg 3.
# Usage: gpurun -- 'bash tools/r5_w24.sh <tag>'
TAG=${1:-r5w}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for W in 2 4; do
  for P in bins sorted; do
    timeout 1200 python tools/emulate_wire.py --world $W --config cfg3 --protocol $P --groups 4 --wire-gbps 40,55,75,0 $([ $P = sorted ] && echo --no-direct) > $OUT/wire_w${W}_$P.json 2> $OUT/wire_w${W}_$P.err; echo "W=$W $P rc=$?"
    grep -o '"mode": "[a-z]*"\|"link_gbps": [0-9.]*\|"ms_best": [0-9.]*' $OUT/wire_w${W}_$P.err | paste - - - | tr '\t' ' '
  done
done
