#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c12"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "tile or chained or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
bash tools/ab_env.sh color1 "LS3D_TILE_COLOR=1" color0 "LS3D_TILE_COLOR=0" color2 "LS3D_TILE_COLOR=2" color1_b "LS3D_TILE_COLOR=1" color0_b "LS3D_TILE_COLOR=0" | tee $OUT/ab.txt
EXTRA="--model mseg3d" bash tools/ab_env.sh m_color1 "LS3D_TILE_COLOR=1" m_color0 "LS3D_TILE_COLOR=0" | tee -a $OUT/ab.txt
cd /tmp
for C in 1 0; do
LS3D_TILE_COLOR=$C timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc$C -o b -- python $R/bench.py --steps 10 --warmup 3 --no-extra-modes --no-cpu-baseline --no-train-leg --no-graph > /dev/null 2>&1
cp $(find /tmp/pc$C -name b_kernel_stats.csv | head -1) $OUT/kernel_stats_color$C.csv
grep "k_tile_conv\|k_tile_build" $OUT/kernel_stats_color$C.csv | cut -c1-60,150-260
done
echo finished
