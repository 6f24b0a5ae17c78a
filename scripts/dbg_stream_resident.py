"""Dev check of the streamed form of the resident kernel (two workgroups per compute unit):
the loop at n = 1e6 in its two forms, same iterates, time per iteration.
    python scripts/dbg_stream_resident.py [n m]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np
import torch
from ipsolver import _hip, cg_fused, projector
from ipsolver import device as dv
from ipsolver.operators import DeviceHessian
from ipsolver.synthetic import CenteredBandedNLP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n // 10
lib = _hip.load()
st = dv.stream_ptr()
prob = CenteredBandedNLP(n, m, seed=0)
x = prob.x0
v = 0.1 * np.random.default_rng(7).standard_normal(m)
A = dv.DeviceCSR.from_scipy(prob.constr_jac(x))
H = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(prob.hess(x)),
                  diag=dv.DVec.from_host(prob.kappa * prob.Wt.dot(v)))
c = dv.DVec.from_host(prob.grad(x))
Z, LS, Y = projector.projections(A)
P = Z.projector
x0 = Y.dot(-dv.DVec.zeros(m))
r0 = Z.dot(H.dot(x0) + c)
g0 = Z.dot(r0)
rt_g = g0.sumsq_amax()[0]
SEG = 200
out = {}
for form, kw in (("three", {"resident": False}), ("resident", {"resident": True})):
    L = cg_fused._Loop(H, P, None, None, **kw)
    print(form, "resident flag", L.args.resident, "form",
          int(lib.ipx_cg_resident_ok(L.ref())) if L.args.resident else 0, flush=True)
    if form == "resident" and not L.args.resident:
        continue
    init = np.zeros(L.state.numel())
    init[cg_fused.ST_RTG0], init[cg_fused.ST_RADIUS] = rt_g, 1e300
    init[cg_fused.ST_ORTH_RHS] = P.orth_tol * P.norm_A
    init_d = torch.from_numpy(init).to(L.state.device)

    def run(k):
        it = 0
        while it < k:
            if it % SEG == 0:
                L.x.copy_(x0.t)
                L.r.copy_(r0.t)
                _hip.call("ipx_axpby", n, -1.0, dv._p(g0.t), 0.0, None, dv._p(L.p), st)
                L.state.copy_(init_d)
                _hip.check(lib.ipx_cg_hp(L.ref(), st), "ipx_cg_hp")
            end = min(k, it - it % SEG + SEG)
            _hip.check(lib.ipx_cg_iterate(L.ref(), it % SEG, it % SEG + end - it, st), "iterate")
            it = end
    run(40)
    torch.cuda.synchronize()
    s = L.state.cpu().numpy()
    print(form, "after 40: stop", s[cg_fused.ST_STOP], "niter", s[cg_fused.ST_NITER], "viol", s[cg_fused.ST_VIOL], flush=True)
    out[form] = (L.x.cpu().numpy().copy(), L.r.cpu().numpy().copy(), L.p.cpu().numpy().copy(), s.copy())
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(200)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 200 * 1e6)
    print(form, "us per iteration", ["%.2f" % t for t in ts], flush=True)
if len(out) == 2:
    for k, name in enumerate(("x", "r", "p")):
        a, b = out["three"][k], out["resident"][k]
        print(name, "rel diff", np.max(np.abs(a - b)) / max(np.max(np.abs(a)), 1e-300))
    print("state diff", np.max(np.abs(out["three"][3] - out["resident"][3])))
