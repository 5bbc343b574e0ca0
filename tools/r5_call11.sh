#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/tests.log; tail -5 $O/tests.log
bash tools/r5_profile.sh
