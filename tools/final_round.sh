#!/bin/bash
# End-of-round evidence on the GPU box: the -m gpu suite, kernel stats + HBM traffic of the default workload, the SQ counter
# passes, the bench line of every workload, the per-rank cost of the N-GPU code path, the emulated 8-GPU rank.
# Usage: gpurun -- 'bash tools/final_round.sh <tag>'
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
timeout 3000 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/collect_profiles.sh $TAG
python3 - <<PY
import json
v = json.loads(open("$OUT/bench_line.json").read().strip().splitlines()[-1])["value"]
print("REGRESSION GUARD cfg 2:", "ok" if v >= 41e9 else "BELOW 41 G k-mers/s", v)
PY
bash tools/collect_counters.sh ${TAG}_sq cfg2
bash tools/collect_counters.sh ${TAG}_sq4 cfg4   # K = 59: VALU per k-mer of the 128-bit k_encode
bash tools/r3_lines.sh $TAG configs sharded
for w in "--reads 10000000" "" "--reads 12500000 --prefix-bits 28" "--k 59 --prefix-bits 28 --reads 6250000 --read-len 250"; do
  n=$(echo "$w" | tr -d ' -' | cut -c1-24); [ -z "$n" ] && n=cfg5
  m=""; case "$n" in reads10000000|cfg5) m="--merge";; esac
  timeout 900 python tools/emulate_rank.py --protocol words $m $w > $OUT/emul_$n.json 2> $OUT/emul_$n.err; echo "emul [$w] rc=$?"
done
# round 4: rank 0 of 8 against a paced wire, grouped receiver on / off (DESIGN_HISTORY.md §5.7)
for c in cfg3 cfg2 cfg4; do
  timeout 1200 python tools/emulate_wire.py --config $c --groups 4,8 --wire-gbps 40,55,75,0 > $OUT/wire_$c.json 2> $OUT/wire_$c.err; echo "wire $c rc=$?"
done
timeout 1800 python bench.py --cpu-full --steps 5 --warmup 1 --no-h2d --no-fasta --no-per-record > $OUT/bench_cpufull.json 2> $OUT/bench_cpufull.err; echo "cpu-full rc=$?"
bash tools/fuzz_campaign.sh ${TAG}_fuzz 1101 1110
