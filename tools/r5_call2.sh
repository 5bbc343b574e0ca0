#!/bin/bash
# round 5, GPU call 2: wave -> SIMD placement, SIMD-level balance model on the benchmark map, skeleton probes of the mapping iteration,
# the round's new / tightened GPU tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
tools/ubench/simd_placement.bin > $O/simd_placement.txt 2>&1; cat $O/simd_placement.txt
timeout 300 python tools/list_balance.py 8 2>/dev/null > $O/list_balance.txt; cat $O/list_balance.txt
PROBE_EXPS="0 512 1024 32 1536 1568 1" timeout 600 bash tools/skeleton_probe.sh 8 > $O/skeleton.txt 2>&1; cat $O/skeleton.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_rccl.py tests/test_gpu_golden_slam.py -q -k "best_candidate or rccl or g9L" 2>&1 | tail -30 > $O/tests.log; tail -15 $O/tests.log
