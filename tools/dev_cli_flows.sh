#!/bin/bash
# The reference CLI's flows (examples/cbl.rs: build / count / query / merge) end to end through `python -m cbl_amd`, on cfg 2-sized files in
# /dev/shm: wall time of each command (process start, library load and context creation included). Usage: gpurun -- 'bash tools/dev_cli_flows.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
D=/dev/shm/cblx_cli_$$; mkdir -p $D
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$R")
from cbl_amd import synth
for seed, name, NR in ((42, "a", 10_000_000), (43, "b", 10_000_000)):
    L = 150
    hb, _ = synth.reads(seed, NR, L)
    with open("$D/%s.fa" % name, "wb") as f:
        for a0 in range(0, NR, 1_000_000):
            n = min(1_000_000, NR - a0)
            rec = np.empty((n, 11 + L + 1), dtype=np.uint8)
            rec[:, 0], rec[:, 1], rec[:, 10], rec[:, -1] = ord(">"), ord("r"), 10, 10
            ids = np.arange(a0, a0 + n)
            for d in range(8):
                rec[:, 9 - d] = 48 + (ids // 10 ** d) % 10
            rec[:, 11:11 + L] = hb[a0 * L:(a0 + n) * L].reshape(n, L)
            f.write(rec.tobytes())
PY
t() { local s=$(date +%s%N); "$@" > $D/out.txt 2>&1; local rc=$?; local e=$(date +%s%N); local ms=$(( (e - s) / 1000000 )); echo "$ms ms  rc=$rc  $*  | $(tail -1 $D/out.txt | cut -c1-70)"; }
t python -m cbl_amd -k 31 build $D/a.fa -o $D/a.cbl
t python -m cbl_amd -k 31 build $D/b.fa -o $D/b.cbl
t python -m cbl_amd -k 31 count $D/a.cbl
t python -m cbl_amd -k 31 query $D/a.cbl $D/b.fa
t python -m cbl_amd -k 31 merge $D/a.cbl $D/b.cbl -o $D/ab.cbl
ls -la $D | awk '{print $5, $9}' | tail -6
rm -rf $D
