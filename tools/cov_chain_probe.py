"""Developer probe (CPU, no GPU needed): the backward projection's covariance chain -- dL/dconic -> dL/dSigma2 -> ... -> dL/d(log-scale) --
for ONE Gaussian of the stress scenes (tests/test_gpu_fused._setup), evaluated from the float64 oracle's exact dL/dconic (i) as the kernels' expanded closed form
(rounds 1-5) in float32 numpy, (ii) by torch autograd in float32, (iii) in the two-term form the kernels use since round 6.  On the big thin splats that
carried the native path's d_scaling excess (tools/dscale_bisect.py: radius >= 32 px, the long axis) the expanded form is 10 - 100x less accurate than the others.
    python tools/cov_chain_probe.py <scene seed> <Gaussian index>"""
import sys, os, copy, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tests.test_gpu_fused as tf
tf.DEV = "cpu"
import mm3dgs_slam_amd.pose_utils as P_
import mm3dgs_slam_amd.renderer as rmod
from mm3dgs_slam_amd.renderer import Renderer
import oracle.raster_ref as orr
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 27
gid = int(sys.argv[2]) if len(sys.argv) > 2 else 2980
cfg, g, R, pose, color, depth = tf._setup(P=3000, H=120, W=160, seed=seed)
keys = ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation")
store = {}
orig_pre = orr.preprocess_ref
def pre_hook(*a, **k):
    pre = orig_pre(*a, **k)
    pre["conic"].retain_grad(); pre["xy"].retain_grad()
    store["pre"] = pre
    return pre
orr.preprocess_ref = pre_hook
def run(dt):
    class PC: active_sh_degree = 0; max_sh_degree = 0
    pc = PC()
    leaf = {k: getattr(g, k).detach().to(dt).cpu().requires_grad_(True) for k in keys}
    pc._xyz, pc._scaling, pc._rotation = leaf["_xyz"], leaf["_scaling"], leaf["_rotation"]
    pc.get_xyz, pc.get_opacity, pc.get_scaling = leaf["_xyz"], torch.sigmoid(leaf["_opacity"]), torch.exp(leaf["_scaling"])
    pc.get_rotation, pc.get_features = torch.nn.functional.normalize(leaf["_rotation"]), leaf["_features_dc"]
    ccfg = copy.deepcopy(cfg); ccfg["device"] = "cpu"
    Rc = Renderer(ccfg, rasterizer_cls=orr.RefRasterizer)
    Rc.projection_matrix, Rc.background, Rc._eye = Rc.projection_matrix.to(dt), Rc.background.to(dt), Rc._eye.to(dt)
    orig = rmod.get_camera_from_tensor
    def cam(t):
        return torch.cat([torch.cat([P_.quad2rotation(t[None, :4])[0], t[4:7, None]], 1), torch.tensor([[0.0, 0, 0, 1]], dtype=t.dtype)], 0)
    rmod.get_camera_from_tensor = cam
    try:
        p_ = pose.detach().to(dt).cpu().requires_grad_(True)
        r_ = Rc.render(pc, p_)
        ref_ = torch.cat([r_["render"], r_["depth"]], 0)
        w = torch.randn(6, 120, 160, generator=torch.Generator().manual_seed(1)).to(dt)
        (ref_ * w).sum().backward()
    finally:
        rmod.get_camera_from_tensor = orig
    return leaf, store["pre"], Rc, p_
leaf, pre, Rc, p_ = run(torch.float64)
dcon = pre["conic"].grad[gid].numpy()      # (dL/dA, dL/dB, dL/dC) of the conic
dls64 = leaf["_scaling"].grad[gid].numpy()
print("Gaussian", gid, "oracle d_log_scale", dls64, "dconic", dcon)
# ---- the kernel's chain (fused.hip slam_bwd_body, transform mode) from the oracle's EXACT dconic, in float32 and float64
def chain(dt):
    f = lambda v: np.asarray(v, dtype=dt)
    Wd, Hd = 160, 120
    fx, fy = f(Rc.fovx), f(Rc.fovy)
    tanx, tany = f(Wd / (2 * Rc.fovx)), f(Hd / (2 * Rc.fovy))
    q = f(g._rotation[gid].detach().numpy()); ls = f(g._scaling[gid].detach().numpy())
    x = f(g._xyz[gid].detach().numpy())
    pq = f(pose.detach().numpy())
    qn = pq[:4] / np.sqrt((pq[:4] ** 2).sum(dtype=dt))
    def quatR(q):
        r, xx, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (xx * y - r * z), 2 * (xx * z + r * y)], [2 * (xx * y + r * z), 1 - 2 * (xx * xx + z * z), 2 * (y * z - r * xx)], [2 * (xx * z - r * y), 2 * (y * z + r * xx), 1 - 2 * (xx * xx + y * y)]], dtype=dt)
    Rp = quatR(qn); p = Rp @ x + pq[4:]
    qg = q / max(np.sqrt((q ** 2).sum(dtype=dt)), dt(1e-12)); Rg = quatR(qg)
    sm = np.exp(ls).astype(dt)
    Mx = Rg * sm[None, :]; S3 = (Mx @ Mx.T).astype(dt)
    tz = p[2]; limx, limy = dt(1.3) * tanx, dt(1.3) * tany
    txtz, tytz = p[0] / tz, p[1] / tz
    in_x, in_y = abs(txtz) <= limx, abs(tytz) <= limy
    txc = p[0] if in_x else np.clip(txtz, -limx, limx) * tz; tyc = p[1] if in_y else np.clip(tytz, -limy, limy) * tz
    itz = dt(1) / tz; itz2 = itz * itz; itz3 = itz2 * itz
    J00, J02, J11, J12 = fx * itz, -fx * txc * itz2, fy * itz, -fy * tyc * itz2
    A = np.array([[J00, 0, J02], [0, J11, J12]], dtype=dt)
    AS = (A @ S3).astype(dt)
    a = (AS[0] * A[0]).sum(dtype=dt) + dt(0.3); b = (AS[0] * A[1]).sum(dtype=dt); c = (AS[1] * A[1]).sum(dtype=dt) + dt(0.3)
    gA, gB, gC = f(dcon[0]), f(dcon[1]), f(dcon[2])
    det = a * c - b * b; id2 = dt(1) / (det * det)
    da = (-c * c * gA + b * c * gB - b * b * gC) * id2
    db = (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gC) * id2
    dcc = (-b * b * gA + a * b * gB - a * a * gC) * id2
    G2 = np.array([[da, dt(0.5) * db], [dt(0.5) * db, dcc]], dtype=dt)
    GA = (G2 @ A).astype(dt)
    dS = (A.T @ GA).astype(dt); dS = ((dS + dS.T) * dt(0.5)).astype(dt)
    dM = (2 * (dS @ Rg) * sm[None, :]).astype(dt)
    ds = (dM * Rg).sum(0, dtype=dt)
    return (ds * sm).astype(dt), dict(a=a, b=b, c=c, det=det, da=da, db=db, dcc=dcc)
d32, i32 = chain(np.float32); d64, i64 = chain(np.float64)
print("chain from exact dconic: float64", d64, " float32", d32, " rel err of float32 chain", np.abs(d32 - d64) / np.abs(d64).max())
print("cov2D", i64["a"], i64["b"], i64["c"], "det", i64["det"], "ac/det", i64["a"] * i64["c"] / i64["det"])
print("da db dcc f64", i64["da"], i64["db"], i64["dcc"], " f32 rel", abs(i32["da"] - i64["da"]) / abs(i64["da"]), abs(i32["db"] - i64["db"]) / abs(i64["db"]), abs(i32["dcc"] - i64["dcc"]) / abs(i64["dcc"]))
print("NOTE the position part of d_log_scale: oracle total", dls64, "vs chain-from-dconic", d64)
leaf32, pre32, _, _ = run(torch.float32)
d32o = leaf32["_scaling"].grad[gid].double().numpy()
print("float32 ORACLE d_log_scale", d32o, "rel err", np.abs(d32o - dls64) / np.abs(dls64).max())
print("float32 oracle dconic", pre32["conic"].grad[gid].numpy(), "rel err", np.abs(pre32["conic"].grad[gid].double().numpy() - dcon) / np.abs(dcon).max())
# torch autograd chain in float32 FROM THE EXACT dconic (isolates the chain from the rest of the float32 evaluation)
def torch_chain(dt):
    ls = g._scaling[gid].detach().to(dt).clone().requires_grad_(True)
    q = torch.nn.functional.normalize(g._rotation[gid:gid+1].detach().to(dt))
    Rg = orr.quat_to_rot_ref(q)[0]
    sm = torch.exp(ls)
    M = Rg * sm[None, :]
    S3 = M @ M.t()
    pq = pose.detach().to(dt)
    Rp = P_.quad2rotation(pq[None, :4])[0]
    p = Rp @ g._xyz[gid].detach().to(dt) + pq[4:]
    fx, fy = Rc.fovx, Rc.fovy
    tz = p[2]
    J = torch.stack([torch.stack([fx / tz, torch.zeros((), dtype=dt), -fx * p[0] / (tz * tz)]), torch.stack([torch.zeros((), dtype=dt), fy / tz, -fy * p[1] / (tz * tz)])])
    cov2 = J @ S3 @ J.t()
    a, b, c = cov2[0, 0] + 0.3, cov2[0, 1], cov2[1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det])
    (conic * torch.tensor(dcon, dtype=dt)).sum().backward()
    return ls.grad.double().numpy()
t64, t32 = torch_chain(torch.float64), torch_chain(torch.float32)
print("torch autograd chain from exact dconic: float64", t64, "float32", t32, "rel err", np.abs(t32 - t64) / np.abs(t64).max())
def chain2(dt):
    # as chain(), with G2 = (1/det) [[gC, -gB/2], [-gB/2, gA]] + kappa [[c, -b], [-b, a]],  kappa = -(c gA - b gB + a gC) / det^2
    f = lambda v: np.asarray(v, dtype=dt)
    fx, fy = f(Rc.fovx), f(Rc.fovy)
    q = f(g._rotation[gid].detach().numpy()); ls = f(g._scaling[gid].detach().numpy()); x = f(g._xyz[gid].detach().numpy())
    pq = f(pose.detach().numpy())
    qn = pq[:4] / np.sqrt((pq[:4] ** 2).sum(dtype=dt))
    def quatR(q):
        r, xx, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (xx * y - r * z), 2 * (xx * z + r * y)], [2 * (xx * y + r * z), 1 - 2 * (xx * xx + z * z), 2 * (y * z - r * xx)], [2 * (xx * z - r * y), 2 * (y * z + r * xx), 1 - 2 * (xx * xx + y * y)]], dtype=dt)
    Rp = quatR(qn); p = Rp @ x + pq[4:]
    qg = q / np.sqrt((q ** 2).sum(dtype=dt)); Rg = quatR(qg)
    sm = np.exp(ls).astype(dt)
    Mx = Rg * sm[None, :]; S3 = (Mx @ Mx.T).astype(dt)
    tz = p[2]; itz = dt(1) / tz; itz2 = itz * itz
    A = np.array([[fx * itz, 0, -fx * p[0] * itz2], [0, fy * itz, -fy * p[1] * itz2]], dtype=dt)
    AS = (A @ S3).astype(dt)
    a = (AS[0] * A[0]).sum(dtype=dt) + dt(0.3); b = (AS[0] * A[1]).sum(dtype=dt); c = (AS[1] * A[1]).sum(dtype=dt) + dt(0.3)
    gA, gB, gC = f(dcon[0]), f(dcon[1]), f(dcon[2])
    det = a * c - b * b; idet = dt(1) / det
    kappa = -(c * gA - b * gB + a * gC) * idet * idet
    da = gC * idet + kappa * c
    hb = dt(-0.5) * gB * idet - kappa * b          # = db / 2
    dcc = gA * idet + kappa * a
    G2 = np.array([[da, hb], [hb, dcc]], dtype=dt)
    GA = (G2 @ A).astype(dt)
    dS = (A.T @ GA).astype(dt); dS = ((dS + dS.T) * dt(0.5)).astype(dt)
    dM = (2 * (dS @ Rg) * sm[None, :]).astype(dt)
    ds = (dM * Rg).sum(0, dtype=dt)
    return (ds * sm).astype(dt)
c32, c64 = chain2(np.float32), chain2(np.float64)
print("two-term G2: float64", c64, "float32", c32, "rel err", np.abs(c32 - c64) / np.abs(c64).max())
