R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02u; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 1200 python tools/emulate_rank.py --serialize > $OUT/emul_ser.json 2> $OUT/emul_ser.err; echo "rc=$?"; tail -1 $OUT/emul_ser.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('serialize'), d['build']['receiver_ms'])"; tail -2 $OUT/emul_ser.err
timeout 900 python -m pytest tests -m gpu -x -q -k "serializ or cli or load" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
