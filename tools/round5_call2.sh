#!/bin/bash
# second GPU call of round 5: chained launches v2 (cached loads behind written-through stores, ticket drawn during the epilogue, chained only
# where it pays) - parity tests, A/B on the bench, kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c2"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "chained_tile or point_mlp_fed or lazy_encoded or batch_norm_train_kernels or linear_weight_gradient" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
bash tools/ab_env.sh chain_default "LS3D_TILE_CHAIN=1" chain0 "LS3D_TILE_CHAIN=0" chain_all "LS3D_CHAIN_MIN_TILES=1 LS3D_CHAIN_MIN_COUT=32" chain_l234 "LS3D_CHAIN_MIN_TILES=1" chain_default_b "LS3D_TILE_CHAIN=1" chain0_b "LS3D_TILE_CHAIN=0" | tee $OUT/ab.txt
EXTRA="--model mseg3d" bash tools/ab_env.sh m_chain1 "LS3D_TILE_CHAIN=1" m_chain0 "LS3D_TILE_CHAIN=0" | tee -a $OUT/ab.txt
cd /tmp
for C in 1 0; do
  LS3D_TILE_CHAIN=$C timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_chain$C -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-modes --precision bf16x6 --steps 10 --warmup 3 > $OUT/prof_chain$C.log 2>&1
  cp $(find /tmp/prof_chain$C -name 'bench_kernel_stats.csv' | head -1) $OUT/kernel_stats_chain$C.csv
done
grep -h "k_tile_conv\|k_gather_gemm" $OUT/kernel_stats_chain1.csv | cut -c1-60,200-400 | head -12
