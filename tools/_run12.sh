R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02q; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 900 python tests/fuzz_parity.py --cases 200 --seed 201 > $OUT/fuzz_201.log 2>&1; echo "fuzz 201 rc=$?"; tail -2 $OUT/fuzz_201.log
CBLX_BATCH_MAX_BASES=3000 timeout 900 python tests/fuzz_parity.py --cases 200 --seed 202 > $OUT/fuzz_202.log 2>&1; echo "fuzz 202 (sub-batches) rc=$?"; tail -2 $OUT/fuzz_202.log
CBLX_FORCE_PEER_COPY=1 timeout 900 python tests/fuzz_parity.py --cases 200 --seed 203 > $OUT/fuzz_203.log 2>&1; echo "fuzz 203 (peer copy) rc=$?"; tail -2 $OUT/fuzz_203.log
timeout 600 python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; tail -1 $OUT/bench_cfg2.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value_h2d_inclusive'], d['h2d_inclusive'])"
