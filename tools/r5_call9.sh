#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p $O
timeout 500 bash tools/ab_lib.sh prio > $O/ab_prio.txt 2>&1; cat $O/ab_prio.txt
for rep in 1 2; do for t in product prio; do
  if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
  MM3DGS_LIB=$L timeout 300 python tools/moving_run.py --frames 60 --every 100 2>/dev/null | tail -1 | sed "s/^/$t /" | tee -a $O/moving_prio.txt
done; done
