#!/bin/bash
# round-4 profiles: the round profile (bench lines, kernel stats, PMC passes per workload), the driver's command line, idle traces early / late
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_r04; mkdir -p $O
timeout 2400 bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null > $O/bench_steps20_warmup5.json; cut -c1-200 $O/bench_steps20_warmup5.json
timeout 300 bash tools/idle_trace.sh > $O/idle_trace.log 2>&1; head -3 $O/idle_trace.log
timeout 400 bash tools/idle_trace.sh --warmup 100 > $O/idle_trace_after_100_frames.log 2>&1; head -3 $O/idle_trace_after_100_frames.log
timeout 300 python bench.py --force-collectives --no-cpu-baseline 2>/dev/null > $O/bench_rccl_world1.json; cut -c1-120 $O/bench_rccl_world1.json
ls $O
