#!/bin/bash
# Round 6, last evidence call: the three `--cpu-full` identities (cfg 3, cfg 2, cfg 4: every byte of the GPU index of ALL reads against the oracle's, SHA-256
# of both in the line) side by side — the oracle is one host thread per run and ten minutes of it; the GPU phases of the three are started 100 s apart so that
# their timed steps do not overlap (the lines' ms_per_step are NOT the ones to quote: profiles/r06_bench_cfg*.json are). Usage: gpurun -- 'bash tools/r6_cpufull_parallel.sh <tag>'
TAG=${1:-r6final}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
avail=$(awk '/MemAvailable/ {print int($2/1048576)}' /proc/meminfo); echo "host MemAvailable ${avail} GiB, cores $(nproc)"
if [ "$avail" -lt 300 ]; then echo "not enough host memory for three oracles side by side: cfg3 only"; CFGS="cfg3"; else CFGS="cfg3 cfg2 cfg4"; fi
pids=""
for c in $CFGS; do
  if [ $c = cfg2 ]; then f=$OUT/bench_cpufull; a=""; else f=$OUT/bench_cpufull_$c; a="--config $c"; fi
  timeout 2400 python bench.py $a --cpu-full --steps 5 --warmup 1 --no-h2d --no-fasta --no-per-record > $f.json 2> $f.err &
  pids="$pids $!"
  sleep 100
done
for p in $pids; do wait $p; echo "pid $p rc=$?"; done
for c in $CFGS; do if [ $c = cfg2 ]; then f=$OUT/bench_cpufull; else f=$OUT/bench_cpufull_$c; fi
  tail -1 $f.json | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d.get('parity_full_size') or {}
print('$c', d['ms_per_step'], {k: (v if not isinstance(v, (dict, list)) else '...') for k, v in p.items()})"; done
