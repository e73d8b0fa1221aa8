#!/bin/bash
# round 3: bench lines. Usage: gpurun -- 'bash tools/r3_lines.sh <tag> [default] [sharded] [configs]'
TAG=${1:-r3lines}; shift
WHAT=${@:-default sharded}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
show() { tail -1 $1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['stage_ms_per_step'], d.get('sharded_overhead', {}).get('ratio'), d.get('serialize'), (d.get('cpu_baseline') or {}).get('value'), d.get('value_h2d_inclusive'))
except Exception as e: print('ERR', e)"; }
for w in $WHAT; do
  case $w in
    default) timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"; show $OUT/bench_default.json ;;
    sharded) for c in cfg2 cfg3 cfg4; do for p in bins sorted; do
               timeout 900 python bench.py --gpus 1 --force-sharded --config $c --protocol $p --transport native --steps 5 --warmup 2 --no-cpu-baseline --no-h2d --no-serialize > $OUT/sharded_${p}_$c.json 2> $OUT/sharded_${p}_$c.err; echo "sharded $p $c rc=$?"; show $OUT/sharded_${p}_$c.json
             done; done
             for c in cfg2 cfg3 cfg4; do
               timeout 900 python bench.py --gpus 1 --force-sharded --config $c --protocol words --transport torch --steps 5 --warmup 2 --no-cpu-baseline --no-h2d --no-serialize > $OUT/sharded_words_$c.json 2> $OUT/sharded_words_$c.err; echo "sharded words $c rc=$?"; show $OUT/sharded_words_$c.json
             done ;;
    configs) for c in cfg3 cfg4 merge dup; do timeout 900 python bench.py --config $c --steps 5 --warmup 2 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c rc=$?"; show $OUT/bench_$c.json; done ;;
  esac
done
