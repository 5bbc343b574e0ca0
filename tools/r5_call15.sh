#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5o; mkdir -p $O
timeout 600 python tools/g9_native_check.py --shipped vigs 2>&1 | grep -v Warning | tee $O/g9D_vigs.txt | cut -c1-220
timeout 600 python -m pytest tests/test_gpu_golden_slam.py -q -k "shipped" 2>&1 | tail -5
