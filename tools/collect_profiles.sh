#!/bin/bash
# Round-end profile record of HEAD on the MI355X box (one gpurun call): rocprofv3 kernel stats of the bench command (SDSeg3D bf16x6 / f32,
# MSeg3D), the HBM-side counter passes (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, --kernel-trace only) and the SQ pass, summarised
# ON the box by profiles/summarize_pmc.py into gpurun_out/profiles_r$ROUND/ (the raw counter CSVs exceed gpurun's pull limit).  Copy that
# directory's files into profiles/ afterwards.   usage: COMMIT=<sha> bash tools/collect_profiles.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-6}
cd "$R"; mkdir -p gpurun_out/profiles_r$ROUND; OUT="$R/gpurun_out"; P3="$OUT/profiles_r$ROUND"
export TMPDIR=/tmp
RAW=/tmp/ls3d_prof; rm -rf $RAW; mkdir -p $RAW
cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-extra-modes --no-train-leg"
for P in bf16x6 f32; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/prof_$P -o bench -- $BENCH --precision $P --steps 10 --warmup 3 > $P3/prof_$P.log 2>&1
  echo "rocprof stats $P rc=$?" >> $P3/summary.txt
  cp $(find $RAW/prof_$P -name 'bench_kernel_stats.csv' | head -1) $P3/round${ROUND}_bench_${P}_kernel_stats.csv 2>/dev/null
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/prof_mseg3d -o bench -- $BENCH --model mseg3d --steps 10 --warmup 3 > $P3/prof_mseg3d.log 2>&1
cp $(find $RAW/prof_mseg3d -name 'bench_kernel_stats.csv' | head -1) $P3/round${ROUND}_bench_mseg3d_kernel_stats.csv 2>/dev/null
for P in ${PMC_PRECISIONS:-bf16x6 f32}; do
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $RAW/pmc_SQ_$P -o bench -- $BENCH --precision $P --steps 3 --warmup 2 > $P3/pmc_SQ_$P.log 2>&1
  echo "pmc SQ $P rc=$?" >> $P3/summary.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $RAW/pmc_${c}_$P -o bench -- $BENCH --precision $P --steps 3 --warmup 2 > $P3/pmc_${c}_$P.log 2>&1
    echo "pmc $c $P rc=$?" >> $P3/summary.txt
  done
done
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $RAW/pmc_SQ_mseg3d -o bench -- $BENCH --model mseg3d --steps 3 --warmup 2 > $P3/pmc_SQ_mseg3d.log 2>&1
echo "pmc SQ mseg3d rc=$?" >> $P3/summary.txt
# rocprofv3 nests its output under <dir>/<hostname>/: flatten for the summariser
for d in $RAW/pmc_*; do f=$(find $d -name 'bench_counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $d/; f=$(find $d -name 'bench_kernel_trace.csv' | head -1); [ -n "$f" ] && cp "$f" $d/; done
cd "$R"
LS3D_PROFILE_OUT=$P3 python profiles/summarize_pmc.py $RAW $ROUND "${COMMIT:-unknown}" > $P3/summarize.log 2>&1
echo "summarize rc=$?" >> $P3/summary.txt
# in-kernel trace of the tile kernel and the per-launch table (tools/trace_tile.py, tools/bench_layers.py)
timeout 300 python tools/trace_tile.py --flags 0 --out $P3/round${ROUND}_trace_tile_pipelined > $P3/round${ROUND}_trace_tile_pipelined.txt 2>&1
rm -f $P3/round${ROUND}_trace_tile_pipelined_flags0.npz
timeout 300 python tools/bench_layers.py --out $P3/round${ROUND}_layers.json > $P3/round${ROUND}_layers.txt 2>&1
# when every ops call of a captured frame starts and ends during a replay, stream by stream (ls3d_stamp nodes; rocprofv3 serialises the side streams)
timeout 300 python tools/probe_graph_timeline.py --min-us 6 > $P3/round${ROUND}_graph_timeline.txt 2>&1
timeout 300 python tools/probe_graph_timeline.py --model mseg3d --min-us 6 > $P3/round${ROUND}_graph_timeline_mseg3d.txt 2>&1
cat $P3/summary.txt; tail -30 $P3/summarize.log
