#!/bin/bash
# Round 3's tree (ab_r03/, extracted from commit f0bb5e1) against this one on ONE box, no profiler, three alternating repeats
A="--steps 10 --warmup 3 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --profile 0"
for rep in 1 2 3; do
  for t in r03 r04; do
    if [ $t = r03 ]; then D=$GRAFT_REPO_ROOT/ab_r03; X=""; else D=$GRAFT_REPO_ROOT; X="--mono-frames 0"; fi
    v=$(cd $D && python bench.py $A $X 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$t $v"
  done
done
