bash tools/gpu_round.sh r02f tests
for c in merge cfg2; do timeout 900 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02f/bench_$c.json 2> gpurun_out/r02f/bench_$c.err; echo "$c rc=$?"; done
timeout 900 python bench.py --config cfg2 --force-sharded --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02f/bench_forcesharded.json 2> gpurun_out/r02f/bench_forcesharded.err; echo "fs rc=$?"
python - <<'PY'
import json
for n in ("merge","cfg2","forcesharded"):
    try:
        d=json.loads(open("gpurun_out/r02f/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], {k["stage"]: k["ms_per_step"] for k in d["roofline"]["kernels"]}, d.get("exchange"))
    except Exception as e: print(n, "failed", e)
PY
