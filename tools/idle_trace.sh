#!/bin/bash
# Where the GPU idles inside a SLAM frame: every gap > 5 us between consecutive kernels of the bench run, grouped by the kernels on
# either side (GPU box, repo root).    bash tools/idle_trace.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/p_idle; mkdir -p /tmp/p_idle
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_idle -o kt -- python bench.py --steps 6 --warmup 2 --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 --no-cpu-baseline --profile 0 "$@" > /tmp/p_idle/bench.json 2>/tmp/p_idle/bench.log
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/p_idle/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]) for r in csv.DictReader(open(f))))
# the last 6 frames: find the tracking loops (runs of sort_composite_fwd_bwd_track) -> frame boundaries
t_end = rows[-1][1]
track_starts = []
prev_track = False
for i, (s, e, k) in enumerate(rows):
    is_track = "fwd_bwd_track" in k
    if is_track and not prev_track and (not track_starts or s - rows[track_starts[-1]][0] > 5_000_000):
        track_starts.append(i)
    prev_track = is_track or ("slam_preprocess_bwd_kernel<true" in k) or ("pose_finish" in k) or ("project_bin" in k and prev_track)
frames = track_starts[-6:]
lo = frames[0]
sel = rows[lo:]
span = sel[-1][1] - sel[0][0]
busy = sum(e - s for s, e, _ in sel)
print(f"last {len(frames)} frames: span {span/1e6:.2f} ms, kernels busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms = {(span-busy)/span*100:.1f} % ({(span-busy)/len(frames)/1e6:.2f} ms / frame)")
gaps = collections.defaultdict(list)
for (s0, e0, k0), (s1, e1, k1) in zip(sel, sel[1:]):
    g = s1 - e0
    if g > 5000:
        gaps[(k0, k1)].append(g)
print("gaps > 5 us by neighbours (per frame):")
for (k0, k1), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"  {k0:48s} -> {k1:48s} n/frame {len(v)/len(frames):6.1f} mean {sum(v)/len(v)/1e3:8.1f} us  total/frame {sum(v)/len(frames)/1e6:6.3f} ms")
small = sum(s1 - e0 for (s0, e0, k0), (s1, e1, k1) in zip(sel, sel[1:]) if 0 < s1 - e0 <= 5000)
print(f"gaps <= 5 us: {small/len(frames)/1e6:.3f} ms / frame")
ks = collections.defaultdict(lambda: [0, 0])
for s, e, k in sel:
    ks[k][0] += 1; ks[k][1] += e - s
print("kernel time per frame:")
for k, (n, t) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {k:48s} n/frame {n/len(frames):7.1f}  avg {t/n/1e3:7.1f} us  total/frame {t/len(frames)/1e6:6.3f} ms")
PY
tail -2 /tmp/p_idle/bench.log
# SEQ=1: the kernel sequence of one timed frame with the native loops collapsed (what the host does between them, and how long the GPU waits for it)
if [ -n "$SEQ" ]; then python - <<'PY'
import csv, glob
f = glob.glob("/tmp/p_idle/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]) for r in csv.DictReader(open(f))))
hot = ("composite", "slam_", "ssim_maps", "loss_finish", "sort_tiles")
starts, last_track = [], -10**18
for i, (s, e, k) in enumerate(rows):
    if "fwd_bwd_track" in k:
        if s - last_track > 2_000_000:
            starts.append(i)
        last_track = e
a, b = starts[-4], starts[-3]
t0 = rows[a][0]
run, last_end = None, None
for s, e, k in rows[a:b]:
    gap = (s - last_end) / 1e3 if last_end else 0.0
    if any(h in k for h in hot):
        if run is None or gap > 5:
            if run: print(f"   [{run[0]} hot kernels, {run[1]/1e3:.0f} us]")
            run = [0, 0]
            if gap > 5: print(f"  +{(s-t0)/1e3:9.1f} us  gap {gap:7.1f}  -> {k}")
        run[0] += 1; run[1] += e - s
    else:
        if run: print(f"   [{run[0]} hot kernels, {run[1]/1e3:.0f} us]"); run = None
        print(f"  +{(s-t0)/1e3:9.1f} us  gap {gap:7.1f}  {k}  ({(e-s)/1e3:.1f} us)")
    last_end = e
if run: print(f"   [{run[0]} hot kernels, {run[1]/1e3:.0f} us]")
print(f"frame: {(rows[b][0]-t0)/1e6:.2f} ms")
PY
fi
