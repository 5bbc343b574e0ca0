#!/bin/bash
# round-6 profiles: the round profile (bench lines, kernel stats, PMC passes per workload), the driver's command line, the idle trace, kernel stats of the c3 / c4
# lines, the forced-collectives lines (both optimiser forms), SQ counters of the hot kernels
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_r06; mkdir -p $O
timeout 2400 bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null > $O/bench_steps20_warmup5.json; cut -c1-200 $O/bench_steps20_warmup5.json
timeout 300 bash tools/idle_trace.sh > $O/idle_trace.log 2>&1; head -3 $O/idle_trace.log
for w in c3 c4; do
  rm -rf /tmp/p_$w
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$w -o $w -- python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --steady-frames 0 --profile 0 > /dev/null 2>&1
  cp $(find /tmp/p_$w -name "*kernel_stats.csv" | head -1) $O/bench_${w}_kernel_stats.csv
done
timeout 300 python bench.py --force-collectives --optimizer allreduce --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_rccl_world1.json; cut -c1-120 $O/bench_rccl_world1.json
timeout 300 python bench.py --force-collectives --optimizer reduce_scatter --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_rccl_world1_sharded.json; cut -c1-120 $O/bench_rccl_world1_sharded.json
timeout 300 python bench.py --force-collectives --scaling strong --window-batch 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_rccl_world1_strong_batch2.json; cut -c1-120 $O/bench_rccl_world1_strong_batch2.json
bash tools/sq_counters_cmd.sh r6 "composite|track|bwd_project|project_bin|ssim" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 --profile 0 > $O/sq_counters.txt 2>&1
cut -c1-200 $O/sq_counters.txt | head -20
ls $O
