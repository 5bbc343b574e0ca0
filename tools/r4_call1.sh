#!/bin/bash
# round-4 GPU call 1: sanity of the new tests, instruction-rate probes, baseline / variants A/B, list statistics, moving-camera traces
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a; mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --profile 0"
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k "overflow or void or tile_table or permutation" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 120 tools/ubench/valu_rate.bin > $O/valu_rate.txt 2>&1; grep -E "mad_u64|lshl_add|mul_lo|ds_add|bpermute|permlane16|v_fma_f32 " $O/valu_rate.txt | grep "SIMD 4"
echo "== base"; timeout 300 bash tools/kstats_cmd.sh base $B | head -14; tail -c 400 /tmp/ks_base.out | grep -o '"value": [0-9.]*'
echo "== off32"; MM3DGS_LIB=$PWD/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_off32.so timeout 300 bash tools/kstats_cmd.sh off32 $B | head -8; grep -o '"value": [0-9.]*' /tmp/ks_off32.out | head -1
for pad in 12288 24576 36000; do
  echo "== pad $pad"; MM3DGS_SLAM_LDS_PAD=$pad timeout 300 bash tools/kstats_cmd.sh pad$pad $B | head -6; grep -o '"value": [0-9.]*' /tmp/ks_pad$pad.out | head -1
done
echo "== pair stats 8"; timeout 300 python tools/pair_stats.py 8 2>&1 | tail -24 | tee $O/pair_stats_8.txt
echo "== pair stats 100"; timeout 400 python tools/pair_stats.py 100 2>&1 | tail -24 | tee $O/pair_stats_100.txt
echo "== moving (r03 trajectory)"; timeout 300 python tools/moving_run.py --motion moving --frames 60 --phases --every 10 2>&1 | tail -22 | tee $O/moving_old.txt
echo "== desk"; timeout 300 python tools/moving_run.py --motion desk --frames 60 --phases --every 5 2>&1 | tail -40 | tee $O/moving_desk.txt
echo "== desk kstats"; timeout 400 bash tools/kstats_cmd.sh desk python tools/moving_run.py --motion desk --frames 60 --every 100 | head -30 | tee $O/desk_kstats.txt
cp gpurun_out/kstats/*.csv $O/ 2>/dev/null
