#!/bin/bash
# Same-box A/B of the product library against variant build "base" on the headline workload (+ 20 frames of the hand-held sweep):  bash tools/ab_c2.sh
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for t in product base; do
    if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
    MM3DGS_LIB=$L python bench.py --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 20 --mono-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', round(d['value'], 3), {k: round(v, 1) for k, v in d['kernel_us_whole_run'].items()}, 'moving', round(d.get('moving', {}).get('value', 0), 2))"
  done
done
