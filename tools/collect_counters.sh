#!/bin/bash
# SQ counter passes of the default bench.py workload (run on the GPU box: gpurun -- 'bash tools/collect_counters.sh <tag>').
# One rocprofv3 --pmc pass per counter group (8 SQ slots per pass), kernel-trace only, program directly after `--`.
TAG=${1:-sq}
CFG=${2:-cfg2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1   # build outside the profiler
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
have() { grep -qw "$1" $OUT/counters_available.txt; }
pass() {  # name, counters...
  local name=$1; shift
  local list=""
  for c in "$@"; do if have $c; then list="$list $c"; else echo "counter $c not available" >> $OUT/skipped.txt; fi; done
  [ -z "$list" ] && return
  rocprofv3 --kernel-trace --pmc $list -d $OUT/$name -o r -- python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-h2d > /dev/null 2> $OUT/$name.err
  python3 $R/tools/rocpd_summary.py $OUT/$name/r_results.db | sed -n '/counter/,$p' | grep -E "counter|---|cblx::" > $OUT/$name.md
  python3 - <<PY >> $OUT/$name.md
import sqlite3
db = sqlite3.connect("$OUT/$name/r_results.db")
try:
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    want = [c for c in cols if any(k in c.lower() for k in ("vgpr", "sgpr", "lds", "scratch", "workgroup", "grid"))]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    if want:
        print("\n| kernel | " + " | ".join(want) + " |")
        print("|---|" + "---|" * len(want))
        for row in db.execute(f"select {namecol}, " + ", ".join(want) + f" from kernels where {namecol} like '%cblx::%' group by {namecol}"):
            print("| " + row[0].split("(")[0][-70:] + " | " + " | ".join(str(x) for x in row[1:]) + " |")
except Exception as e:
    print("no resource columns:", e)
PY
  rm -rf $OUT/$name
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq3 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls $OUT
