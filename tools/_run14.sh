R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02s; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 900 python tests/fuzz_parity.py --cases 150 --seed 301 > $OUT/fuzz_301.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_301.log
timeout 900 python tools/emulate_rank.py --merge > $OUT/emul_cfg5.json 2> $OUT/emul_cfg5.err; echo "rc=$?"; tail -1 $OUT/emul_cfg5.json | cut -c1-2500; tail -2 $OUT/emul_cfg5.err
timeout 900 python tools/emulate_rank.py --reads 10000000 > $OUT/emul_cfg2x8.json 2> $OUT/emul_cfg2x8.err; echo "rc=$?"; tail -1 $OUT/emul_cfg2x8.json | cut -c1-1500
timeout 600 python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; tail -1 $OUT/bench_cfg2.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k['stage']: k['ms_per_step'] for k in d['roofline']['kernels']})"
