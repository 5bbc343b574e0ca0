#!/bin/bash
# round 5, GPU call 4: aligned-record experiment, forward / tracking skeleton probes, configs[2] / configs[3] lines at their stated sizes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
timeout 500 bash tools/ab_lib.sh rec48 > $O/ab_rec48.txt 2>&1; cat $O/ab_rec48.txt
PROBE_EXPS="0 2 4 8 512 32 544 1" timeout 700 bash tools/skeleton_probe.sh 8 > $O/skeleton2.txt 2>&1; cat $O/skeleton2.txt
timeout 600 python bench.py --workload c3 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --mono-frames 0 2> $O/c3.err | tee $O/bench_c3.json | cut -c1-600; tail -3 $O/c3.err
timeout 900 python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline 2> $O/c4.err | tee $O/bench_c4.json | cut -c1-600; tail -3 $O/c4.err
timeout 300 python -m pytest tests/test_gpu_fused.py -q -k "best_candidate" 2>&1 | tail -3
