#!/bin/bash
# dev: rebuild libcblx with other scatter tile shapes (same 4096-record tile) and time the default bench. GPU box only.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/cbl_amd/csrc
cp ../libcblx.so /tmp/libcblx.keep
for cfg in "512 8" "1024 4" "256 16"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -DCBLX_RDX_THREADS=$1 -DCBLX_RDX_ITEMS=$2 -o ../libcblx.so cblx.cpp 2>/dev/null || { echo "build failed $cfg"; continue; }
  echo "== threads=$1 items=$2"
  (cd $R && python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['stage_ms_per_step']['radix_scatter'], d['distinct_kmers_in_index'])")
done
cp /tmp/libcblx.keep ../libcblx.so
