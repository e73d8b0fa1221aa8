R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02l; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "sorted or shim or native or sharded or merge or incremental or load_then or install" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for c in cfg2 cfg3 cfg4; do timeout 600 python bench.py --config $c --force-sharded --transport native --steps 4 --warmup 2 --no-cpu-baseline > $OUT/fs_$c.json 2> $OUT/fs_$c.err; echo "fs $c rc=$?"; done
timeout 600 python bench.py --config merge --steps 5 --warmup 2 > $OUT/merge.json 2> $OUT/merge.err
python - <<'PY'
import json
for n in ("fs_cfg2","fs_cfg3","fs_cfg4","merge"):
    try:
        d=json.loads(open("gpurun_out/r02l/%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], {k["stage"]: k["ms_per_step"] for k in d["roofline"]["kernels"]})
    except Exception as e: print(n, "failed", e)
PY
