R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02k; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python3 $R/bench.py --config cfg3 --force-sharded --transport native --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fs_cfg3.json 2> $OUT/prof.err
cd $R; python3 tools/rocpd_summary.py $OUT/prof/r_results.db | head -60 > $OUT/fs_cfg3_kernels.md; rm -rf $OUT/prof; head -45 $OUT/fs_cfg3_kernels.md | cut -c1-150
