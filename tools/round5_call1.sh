#!/bin/bash
# first GPU call of round 5: the chained tile launches on the device - parity (bit-identity with the layer-by-layer schedule, watchdog flags),
# the two new fixture tests, A/B of the bench with the chain on / off, kernel stats of both
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c1"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "chained_tile or point_mlp_fed or lazy_encoded" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log
bash tools/ab_env.sh chain1 "LS3D_TILE_CHAIN=1" chain0 "LS3D_TILE_CHAIN=0" chain1_m3split "LS3D_TILE_CHAIN=1 LS3D_TILE_FLAGS=128" chain1_b "LS3D_TILE_CHAIN=1" | tee $OUT/ab.txt
EXTRA="--model mseg3d" bash tools/ab_env.sh m_chain1 "LS3D_TILE_CHAIN=1" m_chain0 "LS3D_TILE_CHAIN=0" | tee -a $OUT/ab.txt
cd /tmp
for C in 1 0; do
  LS3D_TILE_CHAIN=$C timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_chain$C -o bench -- python $R/bench.py --no-cpu-baseline --no-extra-modes --precision bf16x6 --steps 10 --warmup 3 > $OUT/prof_chain$C.log 2>&1
  cp $(find /tmp/prof_chain$C -name 'bench_kernel_stats.csv' | head -1) $OUT/kernel_stats_chain$C.csv
  tail -1 $OUT/prof_chain$C.log | head -c 400; echo
done
head -12 $OUT/kernel_stats_chain1.csv
