#!/bin/bash
# round 5 dev: kernel table of the rehearsal (rank 0 of 8 at cfg 3, no wire). Usage: gpurun -- 'bash tools/r5_prof_wire.sh <tag> [cfg]'
TAG=${1:-r5pw}; CFG=${2:-cfg3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python3 $R/tools/emulate_wire.py --config $CFG --groups 4 --wire-gbps 0 --no-ungrouped --no-direct --steps 2 > $OUT/wire.json 2> $OUT/wire.err; echo "rocprof rc=$?"
cd $R
DB=$(find $OUT/prof -name "*.db" | head -1); python3 tools/rocpd_summary.py $DB 2>/dev/null | head -70 > $OUT/kernels.md; head -64 $OUT/kernels.md
rm -f $DB
