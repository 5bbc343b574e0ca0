#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c; mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 --profile 0"
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_golden_slam.py > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 1500 python -m pytest tests/test_gpu_golden_slam.py -q > $O/tests_g9.log 2>&1; tail -25 $O/tests_g9.log
timeout 900 python tools/g9_native_check.py --large > $O/g9L.txt 2>&1; grep -E "frame 1:|frame 7:|frame 3:|RNG|FAILED" $O/g9L.txt
echo "== load-cut spans"; timeout 300 bash tools/kstats_cmd.sh cut $B | head -6; grep -o '"value": [0-9.]*' /tmp/ks_cut.out | head -1
echo "== equal spans"; MM3DGS_LIB=$PWD/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_eqspans.so timeout 300 bash tools/kstats_cmd.sh eq $B | head -6; grep -o '"value": [0-9.]*' /tmp/ks_eq.out | head -1
echo "== desk load-cut"; timeout 300 python tools/moving_run.py --motion desk --frames 60 --every 100 2>&1 | tail -1
echo "== desk equal"; MM3DGS_LIB=$PWD/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_eqspans.so timeout 300 python tools/moving_run.py --motion desk --frames 60 --every 100 2>&1 | tail -1
echo "== desk load-cut kstats"; timeout 400 bash tools/kstats_cmd.sh deskcut python tools/moving_run.py --motion desk --frames 60 --every 100 | head -8
cp gpurun_out/kstats/*.csv $O/ 2>/dev/null
