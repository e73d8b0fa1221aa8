bash tools/gpu_round.sh r02b bench
bash tools/collect_counters.sh r02b_sq cfg2
