#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4d; mkdir -p $O
timeout 600 python tools/world_sweep.py 6 > $O/world_sweep.txt 2>&1; grep -E "seed" $O/world_sweep.txt | cut -c1-260
timeout 300 python tools/g9_native_check.py no_transform sh2_python > $O/g9_small_world.txt 2>&1; grep -E "frame|RNG|FAILED" $O/g9_small_world.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -15 $O/tests.log
