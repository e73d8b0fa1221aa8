R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/collect_profiles.sh r02p
bash tools/collect_counters.sh r02p_sq cfg2
OUT=$R/gpurun_out/r02p
for c in cfg3 cfg4 merge; do timeout 900 python bench.py --config $c --steps 5 --warmup 2 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c rc=$?"; done
timeout 1500 python bench.py --config cfg2 --steps 2 --warmup 1 --cpu-full --no-h2d > $OUT/bench_cpufull.json 2> $OUT/bench_cpufull.err; echo "cpufull rc=$?"
tail -1 $OUT/bench_cpufull.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cpu_baseline'])"
