R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02v; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 900 python tests/fuzz_parity.py --cases 150 --seed 401 > $OUT/fuzz_401.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_401.log
timeout 1200 python tools/emulate_rank.py --serialize > $OUT/emul_ser.json 2> $OUT/emul_ser.err; echo "rc=$?"; tail -1 $OUT/emul_ser.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('serialize'), d['build']['receiver_ms'])"; tail -2 $OUT/emul_ser.err
