#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c24"; mkdir -p $OUT; export TMPDIR=/tmp
S=$(date +%s.%N)
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?"
E=$(date +%s.%N)
echo "python bench.py (default flags) wall seconds: $(python -c "print(round($E-$S,1))")"
head -c 400 $OUT/bench_default.json; echo
echo finished
