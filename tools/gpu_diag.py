"""Developer diagnostic: prints parity metrics for a handful of cases (run on the GPU box)."""
import json, sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import parity_util as pu

cases = {
    "rgb": dict(P=3000, H=100, W=130, seed=0),
    "posed": dict(P=3000, H=96, W=128, seed=1, posed=True),
    "sh3": dict(P=2000, H=80, W=112, seed=5, sh_degree=3, posed=True),
    "cov": dict(P=2000, H=80, W=112, seed=7, cov_precomp=True, posed=True),
    "six": dict(P=2500, H=90, W=120, seed=8, sh_degree=0, extras=3),
    "long": dict(P=6000, H=48, W=48, seed=11, log_scale=-1.2, spread=1.0),
}
out = {}
for name, kw in cases.items():
    try:
        t0 = time.time()
        out[name] = pu.compare(pu.make_case(**kw), verbose=True)
        out[name]["sec"] = time.time() - t0
    except Exception as e:
        traceback.print_exc()
        out[name] = {"error": repr(e)}
    torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/diag.json", "w"), indent=1)
