R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02i; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "sorted or shim or native or sharded" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python3 $R/bench.py --config cfg2 --force-sharded --transport native --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fs_native.json 2> $OUT/prof.err
cd $R; python3 tools/rocpd_summary.py $OUT/prof/r_results.db | head -40 > $OUT/fs_native_kernels.md; rm -rf $OUT/prof; head -12 $OUT/fs_native_kernels.md | cut -c1-160; tail -1 $OUT/fs_native.json | cut -c1-200
