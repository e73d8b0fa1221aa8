#!/bin/bash
# round 5: workgroup shape of k_bucket_union after the staging rings (tools/libcblx_u<threads>x<items>.so: UNI_THREADS / UNI_ITEMS edited for the build):
# the unions at the 8-GPU depth (tools/emulate_rank.py --merge: every both-sided bucket is a union). Usage: gpurun -- 'bash tools/r5_union_sweep.sh <tag> name...'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for v in "$@"; do
  if [ "$v" = main ]; then unset CBLX_LIB_PATH; else export CBLX_LIB_PATH=$R/tools/libcblx_$v.so; fi
  timeout 120 python tools/emulate_rank.py --protocol words --merge > $OUT/${v}_emul.json 2> $OUT/${v}_emul.err
  python3 - <<PY
import json
try:
    e = json.loads(open("$OUT/${v}_emul.json").read().strip().splitlines()[-1])
    print("$v", "8-GPU depth:", e["merge"]["ms"], e["merge"]["stage_ms"]["bucket_big"], e["merge"]["union"], e["merge"]["validate"])
except Exception as ex:
    print("$v failed", ex)
PY
done
