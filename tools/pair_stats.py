"""Developer diagnostic (GPU box): what an LDS-resident gradient combine would have to hold.  Runs the bench scene for a few frames, then reads
the last render's bins (image_state ranges + binning payloads: 16-bit block mask per (tile, splat) pair) and prints, per tile: pairs, listed
4x4 blocks per pair, (8x8 sub-tile, splat) entries -- the per-wave accumulator slots a deterministic wave-level combine needs -- and how many
tiles fit a given LDS budget at 40 B (mapping) / 28 B (tracking) per slot.    python tools/pair_stats.py [frames] [motion]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import _engine
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
motion = sys.argv[2] if len(sys.argv) > 2 else "bounded"
H, W = 480, 640
cfg = default_config(device="cuda", height=H, width=W, mapping={"seed_fraction": 0.51})
torch.manual_seed(0); random.seed(0); np.random.seed(0)
seq = SyntheticSequence(cfg, frames + 1, 150000, seed=0, motion=motion)
slam = SLAM(cfg, seq)
for i in range(frames):
    slam.step(i)
eng = _engine(slam.renderer)
with torch.no_grad():
    eng.forward(slam.estimate_pose_list[frames - 1].detach().float().contiguous(), slam.gaussians)
torch.cuda.synchronize()
T = ((W + 15) // 16) * ((H + 15) // 16)
up = lambda x: (x + 255) // 256 * 256
hdr = eng.img_state[:40].view(torch.int32).cpu().numpy()
cap = int(hdr[7])
assert cap > 0, "direct bins expected"
r_off = 256 + up(T * 4)
lens = np.minimum(eng.img_state[r_off:r_off + 4 * T].view(torch.int32).cpu().numpy().astype(np.int64), cap)
N = eng.n_cap
p_off = up(N * 8) + up(N * 16 * 8) + up(N * 2)
pl = eng.binning[p_off:p_off + 8 * T * cap].view(torch.int64).cpu().numpy().reshape(T, cap)
idx = np.arange(cap)[None, :] < lens[:, None]
mask = (pl & 0xffff)[idx]
tile_of = np.repeat(np.arange(T), lens)
blocks = np.array([bin(int(m)).count("1") for m in np.unique(mask)])
lut = np.zeros(65536, dtype=np.int64); lut[np.unique(mask)] = blocks
nb = lut[mask]
sub = ((mask & 0xf) != 0).astype(np.int64) + ((mask & 0xf0) != 0) + ((mask & 0xf00) != 0) + ((mask & 0xf000) != 0)
print(f"P {slam.gaussians.get_xyz.shape[0]}  pairs {mask.size}  tiles {T}  per-tile span {cap}")
print(f"pairs per tile: mean {lens.mean():.0f} p50 {np.percentile(lens, 50):.0f} p90 {np.percentile(lens, 90):.0f} p99 {np.percentile(lens, 99):.0f} max {lens.max()}")
print(f"listed 4x4 blocks per pair {nb.mean():.2f} (records {nb.sum()});  8x8 sub-tiles per pair {sub.mean():.2f} (wave records {sub.sum()});  pairs with no block {np.mean(nb == 0) * 100:.1f} %")
per_tile_sub = np.bincount(tile_of, weights=sub, minlength=T)
per_tile_blk = np.bincount(tile_of, weights=nb, minlength=T)
for name, v in (("wave-level slots per tile (sum over the 4 waves)", per_tile_sub), ("block records per tile", per_tile_blk)):
    print(f"{name}: mean {v.mean():.0f} p50 {np.percentile(v, 50):.0f} p90 {np.percentile(v, 90):.0f} p99 {np.percentile(v, 99):.0f} max {v.max():.0f}")
# per-wave maximum (a fixed per-wave region must hold the largest of the four)
wave_cnt = np.stack([np.bincount(tile_of, weights=((mask >> (4 * w)) & 0xf) != 0, minlength=T) for w in range(4)], 1)
print(f"per-wave slots: mean {wave_cnt.mean():.0f} p90 {np.percentile(wave_cnt, 90):.0f} p99 {np.percentile(wave_cnt, 99):.0f} max {wave_cnt.max():.0f}")
for kb in (8, 12, 16, 20, 24, 32, 48):
    for rec, nm in ((40, "map"), (28, "track")):
        fit_w = np.mean(per_tile_sub * rec <= kb * 1024) * 100
        fit_t = np.mean(lens * rec <= kb * 1024) * 100
        pairs_w = lens[per_tile_sub * rec <= kb * 1024].sum() / lens.sum() * 100
        print(f"  {kb:2d} KB {nm:5s}: tiles whose wave-level slots fit {fit_w:5.1f} % (holding {pairs_w:5.1f} % of the pairs); tile-level slots fit {fit_t:5.1f} %")
