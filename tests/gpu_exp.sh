#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "camera_sfam or points_cp" 2>&1 | tail -4
