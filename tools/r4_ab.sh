#!/bin/bash
# same-box A/B of the round-3 tree (ab_r03/, extracted from commit f0bb5e1) against the current one: the headline frames only
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4ab; mkdir -p $O
A="--steps 20 --warmup 5 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0"
for rep in 1 2; do
  (cd ab_r03 && python bench.py $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r03', round(d['value'],2), 'frames/s', {k: round(v,1) for k,v in d.get('kernel_us',{}).items()})")
  python bench.py $A --mono-frames 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04', round(d['value'],2), 'frames/s', {k: round(v,1) for k,v in d.get('kernel_us',{}).items()})"
done
B="--steps 5 --warmup 2 --no-cpu-baseline --full-seed-steps 0 --steady-frames 100 --moving-frames 0"
(cd ab_r03 && python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r03 steady', round(d['steady_state']['value'],2))")
python bench.py $B --mono-frames 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 steady', round(d['steady_state']['value'],2))"
