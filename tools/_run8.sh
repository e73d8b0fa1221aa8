R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02j; mkdir -p $OUT
run() { n=$1; shift; timeout 900 python bench.py "$@" > $OUT/$n.json 2> $OUT/$n.err; echo "$n rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("$OUT/$n.json").read().strip().splitlines()[-1])
    print("  ", d["config"]["name"], d["n_gpus"], d["ms_per_step"], "ms", "%.2f G/s" % (d["value"]/1e9), {k["stage"]: k["ms_per_step"] for k in d["roofline"]["kernels"]}, d.get("exchange") and {k: d["exchange"][k] for k in ("transport","outstanding_ms_per_step","effective_gbps_per_link")})
except Exception as e: print("   failed", e, open("$OUT/$n.err").read()[-500:])
PY
}
run s2_cfg3 --gpus 2 --shared-gpu --steps 2 --warmup 1
run s2_cfg3_native --gpus 2 --shared-gpu --transport native --steps 2 --warmup 1
run s2_cfg4 --gpus 2 --shared-gpu --config cfg4 --steps 2 --warmup 1
run s2_merge --gpus 2 --shared-gpu --config merge --steps 2 --warmup 1
run s4_cfg3 --gpus 4 --shared-gpu --reads 3000000 --steps 2 --warmup 1
run fs_cfg3 --config cfg3 --force-sharded --transport native --steps 4 --warmup 2 --no-cpu-baseline
run fs_cfg2 --config cfg2 --force-sharded --transport native --steps 4 --warmup 2 --no-cpu-baseline
