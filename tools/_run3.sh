bash tools/gpu_round.sh r02c tests
timeout 900 python bench.py --config merge --steps 5 --warmup 2 > gpurun_out/r02c/bench_merge.json 2> gpurun_out/r02c/bench_merge.err; echo "merge rc=$?"; tail -1 gpurun_out/r02c/bench_merge.json | cut -c1-400
