#!/bin/bash
# Same-box frame-rate A/B of the product library against variant builds, without a profiler:  bash tools/ab_fps.sh <tag> [<tag> ...]
cd "$GRAFT_REPO_ROOT"
A="--steps 10 --warmup 3 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 --profile 0"
for rep in 1 2 3; do
  for t in product "$@"; do
    if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
    v=$(MM3DGS_LIB=$L python bench.py $A 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$t $v"
  done
done
