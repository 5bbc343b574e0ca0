#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p $O
PROBE_EXPS="0 16 2048 2064 32" timeout 700 bash tools/skeleton_probe.sh 8 > $O/skeleton3.txt 2>&1; cat $O/skeleton3.txt
