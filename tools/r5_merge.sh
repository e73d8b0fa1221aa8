#!/bin/bash
# round 5: cblx_merge_from (parity) and the merge bench line with / without the clone. Usage: gpurun -- 'bash tools/r5_merge.sh <tag>'
TAG=${1:-r5m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in "" "--merge-clone"; do
  timeout 900 python bench.py --config merge --steps 5 --warmup 2 $v > $OUT/bench_merge$v.json 2> $OUT/bench_merge$v.err; echo "bench merge $v rc=$?"
  python3 -c "import json; d=json.loads(open('$OUT/bench_merge$v.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['whole_path'], [(k['stage'], k['ms_per_step'], k['frac'], k.get('words_per_step')) for k in d['roofline']['kernels']])"
done
