#!/bin/bash
# round-4 closing run on the GPU box: the round's profiles + the SQ counter table of the hot kernels
cd "$GRAFT_REPO_ROOT"
bash tools/r4_profile.sh
bash tools/sq_counters_cmd.sh r4 "composite|track|bwd_project|project_bin|ssim" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 --profile 0 > gpurun_out/profiles_r04/sq_counters.txt 2>&1
cat gpurun_out/profiles_r04/sq_counters.txt | cut -c1-220
