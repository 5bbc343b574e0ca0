#!/bin/bash
# Same-box A/B of library builds on the configs[4] pass (1080p, 3 M Gaussians, SH 3, forward + backward):  bash tools/ab_c5.sh <tag> [<tag> ...]
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for t in product "$@"; do
    if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
    MM3DGS_LIB=$L python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', round(d['value'], 1), {k: round(v, 1) for k, v in d['kernel_us'].items()})"
  done
done
