#!/bin/bash
# Round-end record on the MI355X box (one gpurun call): profiles of HEAD (tools/collect_profiles.sh), kernel timeline of one frame, the
# full bench line (reads the fresh PMC record), training-step timings, then the whole -m gpu suite with durations.
ROUND=${ROUND:-5}; R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; OUT="$R/gpurun_out"; P3="$OUT/profiles_r$ROUND"; mkdir -p $P3
export TMPDIR=/tmp
COMMIT=${COMMIT:-unknown} bash tools/collect_profiles.sh > $P3/collect.log 2>&1
cp $P3/round${ROUND}_pmc.json $P3/round${ROUND}_pmc.md $P3/round${ROUND}_pmc_sq.md profiles/ 2>/dev/null   # bench.py reads profiles/round${ROUND}_pmc.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_g -o bench -- python $R/bench.py --steps 5 --warmup 3 --no-extra-modes --no-cpu-baseline --no-train-leg > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_e -o bench -- python $R/bench.py --steps 5 --warmup 3 --no-extra-modes --no-cpu-baseline --no-train-leg --no-graph > /dev/null 2>&1
cd "$R"
python tools/timeline.py $(find /tmp/tl_g -name bench_kernel_trace.csv | head -1) > $P3/round${ROUND}_timeline_graph.txt 2>&1
python tools/timeline.py $(find /tmp/tl_e -name bench_kernel_trace.csv | head -1) > $P3/round${ROUND}_timeline_eager_final.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $P3/round${ROUND}_bench.json 2> $P3/bench.err; echo "bench rc=$?" >> $P3/summary.txt
timeout 400 python bench.py --model mseg3d --no-cpu-baseline --no-extra-modes --steps 20 --warmup 5 > $P3/round${ROUND}_bench_mseg3d.json 2>> $P3/bench.err
timeout 200 python tools/bench_decoder.py --out $P3/round${ROUND}_decoder.json > $P3/round${ROUND}_decoder.txt 2>&1
timeout 200 python tools/probe_stack_brackets.py > $P3/round${ROUND}_stack_brackets.txt 2>&1
if [ "${TRAIN:-1}" = "1" ]; then
for P in bf16x6 f32; do
  timeout 400 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision $P > $P3/round${ROUND}_train_step_mseg3d_waymo_2frames_$P.json 2>> $P3/train.err
done
LS3D_EXPERIMENT="losses.FUSED=0,ops._FAST_LAYERNORM=0" timeout 400 python tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 > $P3/round${ROUND}_train_step_mseg3d_waymo_2frames_bf16x6_torch_loss_and_layernorm.json 2>> $P3/train.err
(cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptr -o tr -- python $R/tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 > /dev/null 2>&1; cp $(find /tmp/ptr -name tr_kernel_stats.csv | head -1) $P3/round${ROUND}_train_step_mseg3d_waymo_kernel_stats.csv)
timeout 300 python tools/bench_train_step.py --steps 5 --warmup 2 --precision bf16x6 > $P3/round${ROUND}_train_step_sdseg3d_nusc_bf16x6.json 2>> $P3/train.err
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 tools/bench_train_step.py --model mseg3d --geometry waymo --points 180000 --frames 2 --steps 5 --warmup 2 --precision bf16x6 --ddp --syncbn 2>> $P3/train.err | grep "^{" | tail -1 > $P3/round${ROUND}_train_step_mseg3d_waymo_2frames_ddp_syncbn_bf16x6.json
fi
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 ${PYTEST_ARGS:-} > $P3/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $P3/summary.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $P3/smoke.log 2>&1; echo "smoke rc=$?" >> $P3/summary.txt
fi
cat $P3/summary.txt; tail -25 $P3/pytest_gpu.log; tail -3 $P3/smoke.log; head -c 900 $P3/round${ROUND}_bench.json; tail -12 $P3/summarize.log
