#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5s; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/tests.log; tail -4 $O/tests.log
bash tools/r5_profile.sh
