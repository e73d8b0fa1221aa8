#!/bin/bash
# a longer differential-fuzz campaign on the GPU box: seeds $2..$3 (default 701..720), 300 cases each
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-fuzzc}; mkdir -p $OUT
A=${2:-701}; B=${3:-720}
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bad=0
for seed in $(seq $A $B); do
  CBLX_FUZZ_DIAG=1 timeout 900 python tests/fuzz_parity.py --cases 300 --seed $seed > $OUT/fuzz_$seed.log 2>&1; rc=$?
  echo "fuzz $seed rc=$rc $(tail -1 $OUT/fuzz_$seed.log)"
  [ $rc -ne 0 ] && bad=1
  [ $rc -eq 0 ] && rm -f $OUT/fuzz_$seed.log
done
exit $bad
