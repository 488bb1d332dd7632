#!/bin/bash
# First GPU call of the next round: everything that was built after round 3's GPU budget was spent (bit-identical / pinned on tests/hipsim, never
# run or timed on the device).  One gpurun call, ~4 min:
#   /usr/local/graft/bin/gpurun --timeout 400 -- 'bash tools/untimed_checks.sh'
# 1. device tests of the touched kernels (deferred points of the devoxelization, exact-f32 gather-GEMM, end-to-end parity);
# 2. the inference bench without extras (value before these changes: 143.7 frames/s) and the per-layer table (gather layers: 689 us per frame);
# 3. ls3d_sffm_memory against the layer-by-layer form + A/B timing (then: LS3D_FUSED_SFFM_MEMORY default -> 1 in point_heads.py);
# 4. kernel statistics of a frame: k_devox_hard was 217 us per frame before the pruned scan.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/untimed"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "devoxelize or three_nn or gather_gemm or sdseg3d_end_to_end or mseg3d_end_to_end or rulebooks_bit_exact" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
timeout 120 python bench.py --steps 20 --warmup 5 --no-extra-modes --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; head -c 400 $OUT/bench.json; echo
timeout 120 python bench.py --model mseg3d --steps 20 --warmup 5 --no-extra-modes --no-cpu-baseline > $OUT/bench_mseg3d.json 2>> $OUT/bench.err; head -c 300 $OUT/bench_mseg3d.json; echo
timeout 60 python tools/bench_layers.py --reps 20 --out $OUT/layers.json > $OUT/layers.txt 2>&1; grep -E "gather|sum of" $OUT/layers.txt
timeout 60 python tools/check_sffm_memory.py > $OUT/sffm_memory.json 2> $OUT/sffm_memory.err; cat $OUT/sffm_memory.json
timeout 60 python tools/check_sffm_memory.py --cls 23 --batch 2 --points 90000 >> $OUT/sffm_memory.json 2>> $OUT/sffm_memory.err; tail -1 $OUT/sffm_memory.json
(cd /tmp; timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pu -o u -- python $R/bench.py --steps 10 --warmup 3 --no-extra-modes --no-cpu-baseline > /dev/null 2>&1; cp $(find /tmp/pu -name u_kernel_stats.csv | head -1) $OUT/kernel_stats.csv)
grep -E "k_devox|k_gather_gemm" $OUT/kernel_stats.csv | cut -c1-60,200-260 | head
