#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; mkdir -p $O
for rep in 1 2; do for t in product base; do
  if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
  MM3DGS_LIB=$L timeout 300 python bench.py --workload c4 --grow-to 0 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t c4', round(d['value'], 2), 'frames/s', {k: round(v, 1) for k, v in d['kernel_us'].items()})" | tee -a $O/c4_lpt.txt
done; done

