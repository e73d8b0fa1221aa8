#!/bin/bash
# round 3, first GPU call: the per-rank cost of the N-GPU code path as round 2 left it (1-rank groups on one GPU), both protocols,
# and the emulated 8-GPU rank at PREFIX_BITS = 28. Usage: gpurun -- 'bash tools/r3_baseline.sh <tag>'
TAG=${1:-r3base}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for c in cfg2 cfg3 cfg4; do
  timeout 600 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/direct_$c.json 2> $OUT/direct_$c.err; echo "direct $c rc=$?"
  timeout 600 python bench.py --gpus 1 --force-sharded --config $c --protocol sorted --transport native --steps 5 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/native_sorted_$c.json 2> $OUT/native_sorted_$c.err; echo "native sorted $c rc=$?"
  timeout 600 python bench.py --gpus 1 --force-sharded --config $c --protocol words --transport torch --steps 5 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/torch_words_$c.json 2> $OUT/torch_words_$c.err; echo "torch words $c rc=$?"
done
timeout 900 python tools/emulate_rank.py --reads 12500000 --prefix-bits 28 > $OUT/emul_cfg3.json 2> $OUT/emul_cfg3.err; echo "emul cfg3 rc=$?"
timeout 900 python tools/emulate_rank.py --k 59 --prefix-bits 28 --reads 6250000 --read-len 250 > $OUT/emul_cfg4.json 2> $OUT/emul_cfg4.err; echo "emul cfg4 rc=$?"
for f in $OUT/*.json; do echo "== $f"; tail -1 $f | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    if 'ms_per_step' in d: print(d['ms_per_step'], d['value'], d['roofline']['stage_ms_per_step'])
    else: print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ('receiver_ms','receiver_stage_ms','words_received','buckets','bucket_len_mean','senders_ms_total')}) for k, v in d.items() if k != 'config'})
except Exception as e: print('ERR', e)
"; done
