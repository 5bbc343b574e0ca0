#!/bin/bash
# Idle time between consecutive kernels of the SLAM loops from a rocprofv3 kernel trace (GPU box, repo root).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/p_gap
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_gap -o kt -- python bench.py --steps 2 --warmup 1 --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 --no-cpu-baseline --profile 0 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/p_gap/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(f))))
gaps = collections.defaultdict(list)
for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g < 100000:          # ignore host-side pauses
        gaps[(k0, k1)].append(g)
tot = sum(sum(v) for v in gaps.values())
ours = ("composite", "preprocess", "scatter", "ssim", "loss_", "sort_", "pose_finish")
gaps = {k: v for k, v in gaps.items() if any(t in k[0] for t in ours) and any(t in k[1] for t in ours)}
print("pairs with the largest total idle time (ns):")
for (k0, k1), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(f"{k0:46s} -> {k1:46s} n={len(v):5d} mean {sum(v)/len(v):8.0f} total {sum(v)/1e6:7.2f} ms")
print("total idle (gaps < 100 us)", tot / 1e6, "ms;  busy", sum(e - s for s, e, _ in rows) / 1e6, "ms")
PY
