#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; OUT="$R/gpurun_out/r5c18"; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/probe_train_sum_sites.py aten::sum aten::add aten::mul aten::copy_ aten::add_ > $OUT/sites.txt 2>&1
grep " x " $OUT/sites.txt | head -40
echo finished
