import sys; sys.path.insert(0, '.')
from tests import parity_util as pu
for deg, seed in ((2, 4), (2, 14), (2, 24), (2, 34), (3, 5), (3, 16)):
    m = pu.compare(pu.make_case(P=2000, H=80, W=112, seed=seed, sh_degree=deg, posed=True))
    print(deg, seed, {k: float('%.3g' % v) for k, v in m.items() if k in ('img', 'd_view', 'd_proj', 'd_campos', 'd_means3D')})
