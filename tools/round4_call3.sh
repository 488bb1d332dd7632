#!/bin/bash
# third GPU call of round 4: A/B of the early conv2.0 order + coordinate-class orders of the transposed tables, the decoder's K / V staging
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/call3"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "sffm or frame_graph or capacity_mode or end_to_end_vs_oracle or tile_conv_full_size" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
bash tools/ab_env.sh base "LS3D_NOP=1" no_orders "LS3D_EARLY_ORDER=0 LS3D_PARITY_ORDER=0" no_early "LS3D_EARLY_ORDER=0" no_parity "LS3D_PARITY_ORDER=0" tb512 "LS3D_TARGET_BLOCKS=512" base2 "LS3D_NOP=2" | tee $OUT/ab.txt
EXTRA="--model mseg3d" bash tools/ab_env.sh m_base "LS3D_NOP=1" m_no_orders "LS3D_EARLY_ORDER=0 LS3D_PARITY_ORDER=0" m_base2 "LS3D_NOP=2" | tee -a $OUT/ab.txt
timeout 200 python tools/bench_layers.py --out $OUT/layers.json 2>/dev/null | grep -E "gather|sum of" > $OUT/layers.txt; cat $OUT/layers.txt
