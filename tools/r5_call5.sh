#!/bin/bash
# round 5, GPU call 5: four-lanes-per-pair combine against the lane-per-pair one; whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
timeout 500 bash tools/ab_lib.sh lanecombine > $O/ab_coop.txt 2>&1; cat $O/ab_coop.txt
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/tests.log; tail -6 $O/tests.log
