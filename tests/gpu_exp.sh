#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "training or backward" 2>&1 | grep -n "Error\|assert \|passed\|failed" | head -20
