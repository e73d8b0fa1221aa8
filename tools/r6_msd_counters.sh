#!/bin/bash
# SQ counters of the bucket kernels alone (tools/dev_msd_bench.cpp binary directly behind `--`): gpurun -- 'bash tools/r6_msd_counters.sh <binary> <tag> [args]'
BIN=$1; TAG=${2:-msdsq}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o r -- $R/$BIN ${ARGS:-10000000 24 1} > $OUT/$name.out 2> $OUT/$name.err
  python3 $R/tools/rocpd_summary.py $OUT/$name/r_results.db | sed -n '/counter/,$p' | grep -E "counter|---|k_bucket" > $OUT/$name.md
  rm -rf $OUT/$name
}
ARGS="$@"
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq3 SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE
cat $OUT/sq1.md $OUT/sq2.md $OUT/sq3.md
