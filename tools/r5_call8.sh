#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p $O
timeout 500 bash tools/ab_lib.sh nt > $O/ab_nt.txt 2>&1; cat $O/ab_nt.txt
