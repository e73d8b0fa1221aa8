#!/bin/bash
# bench.py stage times for every tuning variant tools/libcblx_<name>.so (built with -D switches). Usage: variants.sh <tag> <config> name...
TAG=$1; CFG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do
  CBLX_LIB_PATH=$R/tools/libcblx_$v.so timeout 600 python bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/$v.json 2> $OUT/$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$v.json").read().strip().splitlines()[-1])
    print("$v", d["ms_per_step"], {k["stage"]: k["ms_per_step"] for k in d["roofline"]["kernels"]}, d["distinct_kmers_in_index"])
except Exception as e:
    print("$v failed", e, open("$OUT/$v.err").read()[-400:])
PY
done
