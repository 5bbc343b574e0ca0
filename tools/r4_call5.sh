#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4e; mkdir -p $O
timeout 300 python tools/g9_native_check.py no_transform > $O/g9_small_world.txt 2>&1; grep -E "frame|RNG|FAILED" $O/g9_small_world.txt
timeout 300 python tools/g9_native_check.py --large no_transform > $O/g9_large_world.txt 2>&1; grep -E "frame|RNG|FAILED" $O/g9_large_world.txt
timeout 300 python tools/world_conic_diag.py 6 > $O/conic.txt 2>&1; grep seed $O/conic.txt
timeout 900 python -m pytest tests/test_gpu_fused.py -q -k "float64_oracle or packed_bins_bit or bundle or tile_table" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_golden_slam.py -q > $O/tests_g9.log 2>&1; tail -8 $O/tests_g9.log
