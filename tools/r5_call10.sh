#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; mkdir -p $O
timeout 600 bash tools/ab_lib.sh prio2 prio3 prio4 prio5 > $O/ab_prio.txt 2>&1; cat $O/ab_prio.txt
for rep in 1 2; do for t in product prio2 prio3 prio4 prio5; do
  if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
  MM3DGS_LIB=$L timeout 300 python tools/moving_run.py --frames 60 --every 100 2>/dev/null | tail -1 | sed "s/^/$t /" | tee -a $O/moving_prio.txt
done; done
for t in product prio2 prio3 prio4 prio5; do
  if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
  MM3DGS_LIB=$L timeout 300 python bench.py --workload c4 --grow-to 0 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t c4', round(d['value'], 2), 'frames/s', {k: round(v, 1) for k, v in d['kernel_us'].items()})" | tee -a $O/c4_prio.txt
  MM3DGS_LIB=$L timeout 300 python bench.py --workload c3 --grow-to 0 --steps 6 --warmup 2 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --mono-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t c3', round(d['value'], 2), 'frames/s', {k: round(v, 1) for k, v in d['kernel_us'].items()})" | tee -a $O/c4_prio.txt
done
