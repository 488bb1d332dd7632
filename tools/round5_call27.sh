#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c27"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "tile or chained or golden or schedule_switch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
bash tools/ab_env.sh head "LS3D_TILE_COLOR=0" head_b "LS3D_TILE_COLOR=0" | tee $OUT/ab.txt
EXTRA="--model mseg3d" bash tools/ab_env.sh m_head "LS3D_TILE_COLOR=0" | tee -a $OUT/ab.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/bench.py --steps 10 --warmup 3 --no-extra-modes --no-cpu-baseline --no-train-leg > /dev/null 2>&1
grep "k_tile_build\|k_tile_order" $(find /tmp/pb -name b_kernel_stats.csv | head -1) | cut -c1-150
echo finished
