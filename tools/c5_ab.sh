#!/bin/bash
# Developer A/B of the generic path at configs[4] (1080p, 3 M Gaussians, SH 3): the bench's own per-kernel event timings per library variant.
#   bash tools/c5_ab.sh product ppb1 ppb2 ...      (variants built with tools/build_variant.sh <tag> <flags>)
cd "$GRAFT_REPO_ROOT"
# a variant may carry a probe word: probes:32 = the probe build (-DMM3DGS_PROBES) with MM3DGS_EXP=32
for arg in "$@"; do
  v=${arg%%:*}; unset MM3DGS_EXP
  case "$arg" in *:*) export MM3DGS_EXP=${arg##*:};; esac
  if [ "$v" = product ]; then unset MM3DGS_LIB; else export MM3DGS_LIB=$PWD/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$v.so; fi
  python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernel_us']
print('$arg'.ljust(12), '%7.1f Mpix/s' % j['value'], ' | '.join('%s %7.1f' % (a, b) for a, b in k.items()))"
done
