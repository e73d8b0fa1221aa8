CBLX_FUZZ_DIAG=1 timeout 900 python tests/fuzz_parity.py --cases 60 --seed 401 2>&1 | grep -v "^START" | tail -12
echo "== r1 lib"; CBLX_LIB_PATH=$PWD/tools/libcblx_r1.so CBLX_FUZZ_DIAG=1 timeout 900 python tests/fuzz_parity.py --cases 60 --seed 401 2>&1 | grep -v "^START" | tail -6
