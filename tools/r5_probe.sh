#!/bin/bash
# round 5 dev: where the fine-bins build spends its first pass and KRN-1 (same box A/B). Usage: gpurun -- 'bash tools/r5_probe.sh <tag>'
TAG=${1:-r5p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fine_bins" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python tools/dev_dbg15.py 2>&1 | grep -v "^\[cblx" | tail -8
for c in cfg3; do
  python tools/dev_fine_probe.py $c 2>&1 | tail -1
  CBLX_FINE_REDIR=1 python tools/dev_fine_probe.py $c 2>&1 | tail -1
  CBLX_FINE_BINS=0 python tools/dev_fine_probe.py $c 2>&1 | tail -1
  CBLX_LIB_PATH=$R/tools/libcblx_noflush.so python tools/dev_fine_probe.py $c 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python3 $R/tools/dev_fine_probe.py cfg3 > $OUT/prof.log 2>&1; echo "rocprof rc=$?"
cd $R
DB=$(find $OUT/prof -name "*.db" | head -1); python3 tools/rocpd_summary.py $DB 2>/dev/null | head -45 > $OUT/kernels.md; head -30 $OUT/kernels.md
ls $OUT/prof | head
