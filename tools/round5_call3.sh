#!/bin/bash
# where the chained launch loses its tail-filling gain: ablations of the chained kernel (timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c3"; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/ab_env.sh chain "LS3D_TILE_CHAIN=1" nowait "LS3D_CHAIN_ABLATE=2" cachedstores "LS3D_CHAIN_ABLATE=4" costorder "LS3D_CHAIN_ABLATE=8" nowait_cached "LS3D_CHAIN_ABLATE=6" all3 "LS3D_CHAIN_ABLATE=14" chain0 "LS3D_TILE_CHAIN=0" | tee $OUT/ab.txt
cd /tmp
for A in 0 2 4 6 14; do
  LS3D_CHAIN_ABLATE=$A timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a$A -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-modes --no-train-leg --precision bf16x6 --steps 10 --warmup 3 > $OUT/prof_a$A.log 2>&1
  echo "ablate $A: $(grep -h 'k_tile_conv<4, 6, false, 1, true>\|k_tile_conv<2, 6, false, 1, true>' $(find /tmp/prof_a$A -name 'bench_kernel_stats.csv' | head -1) | awk -F'",' '{print $1}' | cut -c1-40 | tr '\n' ' ') $(grep -h 'true>' $(find /tmp/prof_a$A -name 'bench_kernel_stats.csv' | head -1) | awk -F, '{print $(NF-4)}' | tr '\n' ' ')" | tee -a $OUT/ab.txt
done
