#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5n; mkdir -p $O
timeout 500 bash tools/ab_lib.sh cprio > $O/ab_cprio.txt 2>&1; cat $O/ab_cprio.txt
