R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; OUT=$R/gpurun_out; export TMPDIR=/tmp
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mseg3d -o bench -- python $R/bench.py --model mseg3d --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode > $OUT/prof_mseg3d.log 2>&1
tail -2 $OUT/prof_mseg3d.log | cut -c1-300
