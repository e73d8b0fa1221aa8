bash tools/variants.sh r02e cfg2 base uni runs all base all
bash tools/gpu_round.sh r02e tests
