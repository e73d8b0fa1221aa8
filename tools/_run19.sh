R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02y; mkdir -p $OUT
for seed in 401 402 403 404; do CBLX_FUZZ_DIAG=1 timeout 900 python tests/fuzz_parity.py --cases 200 --seed $seed > $OUT/fuzz_$seed.log 2>&1; echo "fuzz $seed rc=$?"; tail -1 $OUT/fuzz_$seed.log; done
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
