R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02h; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "sorted or shim or native or sharded" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for t in torch native; do timeout 600 python bench.py --config cfg2 --force-sharded --transport $t --steps 5 --warmup 2 --no-cpu-baseline > $OUT/fs_$t.json 2> $OUT/fs_$t.err; echo "fs $t rc=$?"; done
python - <<'PY'
import json
for n in ("torch","native"):
    try:
        d=json.loads(open("gpurun_out/r02h/fs_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], {k["stage"]: k["ms_per_step"] for k in d["roofline"]["kernels"]}, d.get("exchange"))
    except Exception as e: print(n, "failed", e, open("gpurun_out/r02h/fs_%s.err"%n).read()[-600:])
PY
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python3 $R/bench.py --config cfg2 --force-sharded --transport native --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/prof.err
cd $R; python3 tools/rocpd_summary.py $OUT/prof/r_results.db | head -40 > $OUT/fs_native_kernels.md; rm -rf $OUT/prof; head -30 $OUT/fs_native_kernels.md | cut -c1-160
