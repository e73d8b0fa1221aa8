#!/bin/bash
# Round 6, end-of-round evidence on the GPU box: tools/r5_final.sh's legs plus this round's — the "replicate" protocol rehearsed at W = 2 / 3 / 4
# beside "sorted" and "bins" (wire6), the bucket kernels alone with their SQ counters (msd). Usage: gpurun -- 'bash tools/r6_final.sh <tag> [legs...]'
TAG=${1:-r6final}; shift
LEGS=${@:-tests profiles counters lines kstats emul wire wire6 msd cpufull fuzz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for leg in $LEGS; do
  case $leg in
    wire6) for W in 2 3 4; do
        timeout 1200 python tools/emulate_wire.py --world $W --config cfg3 --protocol replicate --groups 2 --grouped-slices 1 --wire-gbps 40,55,75,0 > $OUT/wire_w${W}_replicate.json 2> $OUT/wire_w${W}_replicate.err; echo "wire W=$W replicate rc=$?"
      done
      for P in bins sorted; do
        timeout 1200 python tools/emulate_wire.py --world 3 --config cfg3 --protocol $P --groups 4 --wire-gbps 40,55,75,0 --no-direct --no-ungrouped > $OUT/wire_w3_$P.json 2> $OUT/wire_w3_$P.err; echo "wire W=3 $P rc=$?"
      done
      timeout 1200 python tools/emulate_wire.py --world 2 --config cfg4 --protocol replicate --groups 2 --grouped-slices 1 --wire-gbps 55,0 > $OUT/wire_w2_replicate_cfg4.json 2> $OUT/wire_w2_replicate_cfg4.err; echo "wire W=2 replicate cfg4 rc=$?" ;;
    msd) hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cbl_amd/csrc -I include tools/dev_msd_bench.cpp -L cbl_amd -lcblx -Wl,-rpath,$R/cbl_amd -o tools/dev_msd_bench.bin 2> $OUT/msd_build.err
      timeout 600 tools/dev_msd_bench.bin 10000000 24 5 > $OUT/msd_bench_cfg2.txt 2>&1; echo "msd bench rc=$?"; tail -9 $OUT/msd_bench_cfg2.txt
      bash tools/r6_msd_counters.sh tools/dev_msd_bench.bin ${TAG}_msdsq 10000000 24 1 > /dev/null 2>&1; ls $R/gpurun_out/${TAG}_msdsq ;;
    *) bash tools/r5_final.sh $TAG $leg ;;
  esac
done
