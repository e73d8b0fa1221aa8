#!/bin/bash
# Round 5, end-of-round evidence on the GPU box (tools/final_round.sh's legs + this round's): the -m gpu suite, kernel stats + HBM traffic of cfg 2,
# SQ counters, the bench line of every workload, kernel tables of cfg 3 (FINE bins) and the merge, the rehearsals (rank 0 / 3 / 6 / 7 of 8 at cfg 3,
# cfg 2 shape, cfg 4, W = 2 / 4 on both protocols), the full-size parity legs, fuzz seeds. Usage: gpurun -- 'bash tools/r5_final.sh <tag> [legs...]'
TAG=${1:-r5final}; shift
LEGS=${@:-tests profiles counters lines kstats emul wire cpufull fuzz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for leg in $LEGS; do
  case $leg in
    tests) timeout 3000 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -1 ;;
    profiles) bash tools/collect_profiles.sh $TAG
      python3 - <<PY
import json
v = json.loads(open("$OUT/bench_line.json").read().strip().splitlines()[-1])["value"]
print("REGRESSION GUARD cfg 2:", "ok" if v >= 41e9 else "BELOW 41 G k-mers/s", v)
PY
      ;;
    counters) bash tools/collect_counters.sh ${TAG}_sq cfg2; bash tools/collect_counters.sh ${TAG}_sq4 cfg4 ;;
    lines) bash tools/r3_lines.sh $TAG configs sharded ;;
    kstats) cd /tmp && export TMPDIR=/tmp
      for c in cfg3 cfg4 merge; do
        rocprofv3 --kernel-trace --stats -d $OUT/ks_$c -o r -- python3 $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-serialize --no-fasta --no-per-record > /dev/null 2> $OUT/ks_$c.err
        python3 $R/tools/rocpd_summary.py $OUT/ks_$c/r_results.db > $OUT/kernel_stats_$c.md; rm -rf $OUT/ks_$c; echo "kstats $c: $(wc -l < $OUT/kernel_stats_$c.md) rows"
      done; cd $R ;;
    merge) timeout 600 python bench.py --config merge --steps 10 --warmup 3 > $OUT/bench_merge.json 2> $OUT/bench_merge.err; echo "bench merge rc=$?"
      cd /tmp && export TMPDIR=/tmp
      rocprofv3 --kernel-trace --stats -d $OUT/ks_merge -o r -- python3 $R/bench.py --config merge --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/ks_merge.err
      python3 $R/tools/rocpd_summary.py $OUT/ks_merge/r_results.db > $OUT/kernel_stats_merge.md; rm -rf $OUT/ks_merge; cd $R
      bash tools/r5_merge_traffic.sh ${TAG}_mt > /dev/null 2>&1; bash tools/collect_counters.sh ${TAG}_msq merge > /dev/null 2>&1 ;;
    emul) for w in "--reads 10000000" "" "--reads 12500000 --prefix-bits 28" "--k 59 --prefix-bits 28 --reads 6250000 --read-len 250"; do
        n=$(echo "$w" | tr -d ' -' | cut -c1-24); [ -z "$n" ] && n=cfg5
        m=""; case "$n" in reads10000000|cfg5) m="--merge";; esac
        timeout 900 python tools/emulate_rank.py --protocol words $m $w > $OUT/emul_$n.json 2> $OUT/emul_$n.err; echo "emul [$w] rc=$?"
      done ;;
    wire) for c in cfg3 cfg2 cfg4; do
        timeout 1200 python tools/emulate_wire.py --config $c --groups 4,8 --wire-gbps 40,55,75,0 > $OUT/wire_$c.json 2> $OUT/wire_$c.err; echo "wire $c rc=$?"
      done
      for r in 3 6 7; do
        timeout 900 python tools/emulate_wire.py --config cfg3 --rank $r --groups 4 --wire-gbps 40,55,75,0 --no-ungrouped --no-direct > $OUT/wire_cfg3_rank$r.json 2> $OUT/wire_cfg3_rank$r.err; echo "wire cfg3 rank $r rc=$?"
      done
      timeout 900 python tools/emulate_wire.py --config cfg4 --rank 7 --groups 4 --wire-gbps 55,0 --no-ungrouped --no-direct > $OUT/wire_cfg4_rank7.json 2> $OUT/wire_cfg4_rank7.err; echo "wire cfg4 rank 7 rc=$?"
      for W in 2 4; do for P in bins sorted; do
        timeout 1200 python tools/emulate_wire.py --world $W --config cfg3 --protocol $P --groups 4 --wire-gbps 40,55,75,0 $([ $P = sorted ] && echo --no-direct) > $OUT/wire_w${W}_$P.json 2> $OUT/wire_w${W}_$P.err; echo "wire W=$W $P rc=$?"
      done; done ;;
    cpufull) timeout 1800 python bench.py --cpu-full --steps 5 --warmup 1 --no-h2d --no-fasta --no-per-record > $OUT/bench_cpufull.json 2> $OUT/bench_cpufull.err; echo "cpu-full cfg2 rc=$?"
      for c in cfg3 cfg4 merge; do
        timeout 2400 python bench.py --config $c --cpu-full --steps 5 --warmup 1 --no-h2d --no-fasta --no-per-record > $OUT/bench_cpufull_$c.json 2> $OUT/bench_cpufull_$c.err; echo "cpu-full $c rc=$?"
      done ;;
    cpu23) timeout 1800 python bench.py --cpu-full --steps 5 --warmup 1 --no-h2d --no-fasta --no-per-record > $OUT/bench_cpufull.json 2> $OUT/bench_cpufull.err; echo "cpu-full cfg2 rc=$?"
      timeout 2400 python bench.py --config cfg3 --cpu-full --steps 5 --warmup 1 --no-h2d --no-fasta --no-per-record > $OUT/bench_cpufull_cfg3.json 2> $OUT/bench_cpufull_cfg3.err; echo "cpu-full cfg3 rc=$?" ;;
    cpu4) timeout 2400 python bench.py --config cfg4 --cpu-full --steps 5 --warmup 1 --no-h2d --no-fasta --no-per-record > $OUT/bench_cpufull_cfg4.json 2> $OUT/bench_cpufull_cfg4.err; echo "cpu-full cfg4 rc=$?" ;;
    cpumerge) timeout 2400 python bench.py --config merge --cpu-full --steps 5 --warmup 1 --no-h2d --no-fasta --no-per-record > $OUT/bench_cpufull_merge.json 2> $OUT/bench_cpufull_merge.err; echo "cpu-full merge rc=$?" ;;
    fuzz) bash tools/fuzz_campaign.sh ${TAG}_fuzz ${FUZZ_FROM:-1211} ${FUZZ_TO:-1220} ;;
  esac
done
