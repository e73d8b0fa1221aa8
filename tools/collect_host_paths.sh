#!/bin/bash
# Evidence for the paths either side of the kernels (profiles/<round>_host_paths.md): index bytes out (emitter kernels under rocprofv3,
# host-buffer / file times), index file in, a host batch (kernel time line), a FASTA file (phase trace), one call per record.
# Usage: gpurun -- 'bash tools/collect_host_paths.sh <tag>'
TAG=${1:-hostpaths}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --stats -d $OUT/ser -o s -- python3 tools/dev_serialize_rate.py > $OUT/serialize_prof.log 2>&1
python3 tools/rocpd_summary.py $OUT/ser/s_results.db | grep -i "serde\|kernel \|---" > $OUT/serialize_kernels.md
rm -rf $OUT/ser
python3 tools/dev_serialize_rate.py > $OUT/serialize.log 2>&1
python3 tools/dev_load_threads.py > $OUT/load.log 2>&1
rocprofv3 --kernel-trace -d $OUT/tl -o h -- python3 tools/dev_h2d_timeline.py run > $OUT/h2d_prof.log 2>&1
python3 tools/dev_h2d_timeline.py report $OUT/tl/h_results.db > $OUT/h2d_timeline.txt
rm -rf $OUT/tl
REPS=8 CBLX_TRACE_H2D=1 python3 tools/dev_h2d_timeline.py run > $OUT/h2d.log 2>&1
CBLX_INGEST_TRACE=1 python3 tools/dev_fasta_rate.py 16 > $OUT/fasta.log 2>&1
g++ -O2 -std=c++17 -I include -o /tmp/prr tools/dev_insert_seq_rate.cpp -L cbl_amd -lcblx -Wl,-rpath,$R/cbl_amd && /tmp/prr 10000000 > $OUT/per_record.log 2>&1
tail -3 $OUT/serialize.log; tail -2 $OUT/load.log; tail -1 $OUT/h2d.log; tail -1 $OUT/fasta.log; head -3 $OUT/per_record.log
