R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; OUT=$R/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "mseg3d" 2>&1 | tail -3
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mseg3d -o bench -- python $R/bench.py --model mseg3d --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_mseg3d.log 2>&1
grep '"metric"' $OUT/prof_mseg3d.log | cut -c1-2500
