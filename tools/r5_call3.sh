#!/bin/bash
# round 5, GPU call 3: A/B of the combine changes (table prefetch before the barrier; private pairs before the barrier) + the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p $O
timeout 700 bash tools/ab_lib.sh base prefetch > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/tests.log; tail -8 $O/tests.log
