import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, time
from banded_setup import BandedInstance
from ipsolver import device as dv, projector, qp, cg_fused
n, m = 125000, 12500
inst = BandedInstance(n, m)
A = dv.DeviceCSR.from_scipy(inst.A); H = dv.DeviceCSR.from_scipy(inst.H)
b = np.zeros(m)
Z, LS, Y = projector.projections(A)
for kw in (dict(tol=0, max_iter=41), dict(tol=1e-12), dict(tol=0, max_iter=30, trust_radius=1e300), dict(tol=0, max_iter=1)):
    t0 = time.time()
    x, info = qp.projected_cg(H, inst.c, Z, Y, b, **kw)
    print(kw, info, "fallbacks", cg_fused.STATS["resident_fallbacks"], "resident_calls", cg_fused.STATS["resident_calls"], "%.3f s" % (time.time() - t0), flush=True)
