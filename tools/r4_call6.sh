#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4f; mkdir -p $O
timeout 600 python bench.py --steps 3 --warmup 1 --steady-frames 5 --full-seed-steps 2 --moving-frames 8 --mono-frames 5 --no-cpu-baseline > $O/bench_small.json 2> $O/bench_small.err; echo "rc $?"; tail -3 $O/bench_small.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4f/bench_small.json").read().strip().splitlines()[-1])
print({k: (round(v["value"], 2) if isinstance(v, dict) and "value" in v else None) for k, v in d.items() if k in ("steady_state", "full_seed", "moving", "mono_depth")}, round(d["value"], 2))
print(d["roofline"].get("traffic"), d["roofline"].get("traffic_kind"), d["roofline"].get("fabric_GBps"))
PY
timeout 900 python -m pytest tests/test_gpu_fused.py -q -k "world_frame or packed_bins_bit" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 600 python -m pytest tests/test_gpu_golden_slam.py -q > $O/tests_g9.log 2>&1; tail -3 $O/tests_g9.log
MM3DGS_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc $?"; tail -2 $O/bench_gloo2.err; cut -c1-400 $O/bench_gloo2.json
timeout 600 python bench.py --force-collectives --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_force.json 2> $O/bench_force.err; echo "force rc $?"; python -c "
import json; d=json.loads(open('$O/bench_force.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['multi_gpu'])"
