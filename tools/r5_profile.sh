#!/bin/bash
# round-5 profiles: the round profile (bench lines, kernel stats, PMC passes per workload), the driver's command line, idle traces early / late,
# the forced-collectives lines (both optimiser forms), SQ counters of the hot kernels
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_r05; mkdir -p $O
timeout 2400 bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null > $O/bench_steps20_warmup5.json; cut -c1-200 $O/bench_steps20_warmup5.json
timeout 300 bash tools/idle_trace.sh > $O/idle_trace.log 2>&1; head -3 $O/idle_trace.log
timeout 400 bash tools/idle_trace.sh --warmup 100 > $O/idle_trace_after_100_frames.log 2>&1; head -3 $O/idle_trace_after_100_frames.log
timeout 300 python bench.py --force-collectives --optimizer allreduce --no-cpu-baseline 2>/dev/null > $O/bench_rccl_world1.json; cut -c1-120 $O/bench_rccl_world1.json
timeout 300 python bench.py --force-collectives --optimizer reduce_scatter --no-cpu-baseline 2>/dev/null > $O/bench_rccl_world1_sharded.json; cut -c1-120 $O/bench_rccl_world1_sharded.json
timeout 300 python bench.py --workload c3 --grow-to 0 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --mono-frames 0 2>/dev/null > $O/bench_c3_reference_seeding.json; cut -c1-120 $O/bench_c3_reference_seeding.json
timeout 300 python bench.py --workload c4 --grow-to 0 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null > $O/bench_c4_reference_seeding.json; cut -c1-120 $O/bench_c4_reference_seeding.json
bash tools/sq_counters_cmd.sh r5 "composite|track|bwd_project|project_bin|ssim" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 --profile 0 > $O/sq_counters.txt 2>&1
cut -c1-200 $O/sq_counters.txt | head -20
ls $O
