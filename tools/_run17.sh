echo "== current"; timeout 600 python tools/repro_pb25.py 7 2>&1 | tail -8
echo "== r1 library"; CBLX_LIB_PATH=$PWD/tools/libcblx_r1.so timeout 600 python tools/repro_pb25.py 7 2>&1 | tail -8
