#!/bin/bash
# round 3: deep buckets. Usage: gpurun -- 'bash tools/r3_deep.sh <tag> [tests] [emul] [lim]'
TAG=${1:-r3deep}; shift
WHAT=${@:-tests emul}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
show() { tail -1 $1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); b = d['build']; print(round(b['receiver_ms'], 2), b['receiver_stage_ms'], 'validate', b['validate'], (d.get('merge') or {}).get('ms'), (d.get('merge') or {}).get('stage_ms'))
except Exception as e: print('ERR', e)"; }
for w in $WHAT; do
  case $w in
    tests) timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deep_buckets or threshold or merge or repeat" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log ;;
    emul)  timeout 800 python tools/emulate_rank.py --reads 10000000 --protocol words --merge > $OUT/cfg2x8_words.json 2> $OUT/cfg2x8_words.err; echo "cfg2x8 rc=$?"; show $OUT/cfg2x8_words.json
           timeout 800 python tools/emulate_rank.py --protocol words --merge > $OUT/cfg5_words.json 2> $OUT/cfg5_words.err; echo "cfg5 rc=$?"; show $OUT/cfg5_words.json ;;
    ws)    for v in 4096 2048 1024; do
             CBLX_LDS_MAX_WS=$v timeout 800 python tools/emulate_rank.py --k 59 --prefix-bits 28 --reads 6250000 --read-len 250 --protocol words > $OUT/cfg4_ws_$v.json 2> $OUT/cfg4_ws_$v.err; echo "cfg4 depth lds_max=$v rc=$?"; show $OUT/cfg4_ws_$v.json
           done ;;
    lim)   for v in "" lim96 lim160; do
             L=""; [ -n "$v" ] && L=$R/tools/libcblx_$v.so
             CBLX_LIB_PATH=$L timeout 800 python tools/emulate_rank.py --k 59 --prefix-bits 28 --reads 6250000 --read-len 250 --protocol words > $OUT/cfg4_words_$v.json 2> $OUT/cfg4_words_$v.err; echo "cfg4 depth [$v] rc=$?"; show $OUT/cfg4_words_$v.json
           done ;;
  esac
done
