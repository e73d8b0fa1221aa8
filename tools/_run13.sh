R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02r; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 900 python tools/emulate_rank.py --merge > $OUT/emul_cfg5.json 2> $OUT/emul_cfg5.err; echo "rc=$?"; tail -1 $OUT/emul_cfg5.json | cut -c1-2500; tail -3 $OUT/emul_cfg5.err
timeout 900 python tools/emulate_rank.py --reads 10000000 > $OUT/emul_cfg2x8.json 2> $OUT/emul_cfg2x8.err; echo "rc=$?"; tail -1 $OUT/emul_cfg2x8.json | cut -c1-1500; tail -3 $OUT/emul_cfg2x8.err
