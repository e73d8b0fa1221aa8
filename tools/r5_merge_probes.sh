#!/bin/bash
# round 5: where the merge's counting-sort classes spend their time — bench.py --config merge on the timing-probe builds
# tools/libcblx_mp<N>.so (-DCBLX_TIMING_PROBES -DCBLX_MSD_PROBE=N [-DCBLX_MSD_PROBE_MERGE=1]; results are WRONG by construction).
# Usage: gpurun -- 'bash tools/r5_merge_probes.sh <tag> name...'   ("main" = the product library)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for v in "$@"; do
  if [ "$v" = main ]; then unset CBLX_LIB_PATH; else export CBLX_LIB_PATH=$R/tools/libcblx_$v.so; fi
  timeout 300 python bench.py --config merge --steps 5 --warmup 2 --no-cpu-baseline > $OUT/$v.json 2> $OUT/$v.err; rc=$?
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/$v.json").read().strip().splitlines()[-1])
    print("$v", d["ms_per_step"], {k["stage"]: k["ms_per_step"] for k in d["roofline"]["kernels"]})
except Exception as e:
    print("$v failed rc=$rc", e, open("$OUT/$v.err").read()[-300:])
PY
done
