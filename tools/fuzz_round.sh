#!/bin/bash
# several seeds of the differential fuzzer on the GPU box (also with the sub-batch cut and the peer-copy merge forced)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-fuzz}; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for seed in 501 502 503; do CBLX_FUZZ_DIAG=1 timeout 900 python tests/fuzz_parity.py --cases 200 --seed $seed > $OUT/fuzz_$seed.log 2>&1; echo "fuzz $seed rc=$?"; tail -1 $OUT/fuzz_$seed.log; done
CBLX_BATCH_MAX_BASES=3000 CBLX_FUZZ_DIAG=1 timeout 900 python tests/fuzz_parity.py --cases 200 --seed 504 > $OUT/fuzz_504.log 2>&1; echo "fuzz 504 (sub-batches) rc=$?"; tail -1 $OUT/fuzz_504.log
CBLX_FORCE_PEER_COPY=1 CBLX_FUZZ_DIAG=1 timeout 900 python tests/fuzz_parity.py --cases 200 --seed 505 > $OUT/fuzz_505.log 2>&1; echo "fuzz 505 (peer copy) rc=$?"; tail -1 $OUT/fuzz_505.log
grep -h "^FAIL" $OUT/fuzz_50*.log | head
