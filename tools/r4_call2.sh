#!/bin/bash
# round-4 GPU call 2: tests of the new host paths + empty-pair dropping, XCD balance model, desk run after the native graph update
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b; mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --profile 0"
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_golden_slam.py -q -k "g9-" > $O/tests_g9.log 2>&1; tail -5 $O/tests_g9.log
timeout 600 python tools/g9_native_check.py --large vigs > $O/g9L_vigs.txt 2>&1; tail -12 $O/g9L_vigs.txt
echo "== base (empty pairs dropped)"; timeout 300 bash tools/kstats_cmd.sh drop $B | head -9; grep -o '"value": [0-9.]*' /tmp/ks_drop.out | head -1
echo "== xcd balance bounded 8"; timeout 300 python tools/xcd_balance.py 8 bounded 2>&1 | tail -4 | tee $O/xcd_bounded.txt
echo "== xcd balance desk 45"; timeout 400 python tools/xcd_balance.py 45 desk 2>&1 | tail -9 | tee $O/xcd_desk.txt
echo "== desk"; timeout 300 python tools/moving_run.py --motion desk --frames 60 --phases --every 10 2>&1 | tail -22 | tee $O/moving_desk.txt
echo "== pair stats 8"; timeout 300 python tools/pair_stats.py 8 2>&1 | tail -24 | head -8 | tee $O/pair_stats_8.txt
cp gpurun_out/kstats/*.csv $O/ 2>/dev/null
