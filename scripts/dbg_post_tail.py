"""dev tool: the barrier projection at config 5's size, the back substitution as the tail of the
Schur solve's kernel against the separate launch (IPX_DEBUG_FORMS=no-post-tail): g bit for bit?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, scipy.sparse as sps, torch
from ipsolver import _hip, device as dv, projector
from ipsolver.synthetic import CenteredBandedNLP
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500000, 50000)
prob = CenteredBandedNLP(n, m, eps=1.0)
J = prob.constr_jac(prob.x0)
rng = np.random.default_rng(7)
s = rng.uniform(1e-6, 2.0, m + 2 * n)
I = sps.eye(n, format="csr")
A = sps.bmat([[J, sps.diags(s[:m]), None, None], [-I, None, sps.diags(s[m:m + n]), None],
              [I, None, None, sps.diags(s[m + n:])]], format="csr")
A.sort_indices()
N, M = A.shape[1], A.shape[0]
r = rng.standard_normal(N)
out = {}
for form in ("", "no-post-tail"):
    if form:
        os.environ["IPX_DEBUG_FORMS"] = form
    Z, LS, Y = projector.projections(dv.DeviceCSR.from_scipy(A))
    solver = Z.projector.solver
    args = solver.c_args()
    lib = _hip.load()
    nblk = lib.ipx_boxschur_project_count(ctypes.byref(args))
    g = torch.empty(N, dtype=torch.float64, device="cuda")
    pg = torch.zeros(2 * nblk + 16, dtype=torch.float64, device="cuda")
    pres = torch.zeros(M // 256 + 2, dtype=torch.float64, device="cuda")
    n3, n4 = ctypes.c_int32(0), ctypes.c_int32(0)
    rd = dv.DVec.from_host(r)
    _hip.call("ipx_boxschur_project", ctypes.byref(args), dv._p(rd.t), dv._p(g), dv._p(pg),
              ctypes.byref(n3), dv._p(pres), ctypes.byref(n4), None, dv.stream_ptr())
    out[form] = (g.cpu().numpy(), float(pg[:nblk].sum()), bool(args.post_own_g), n3.value, n4.value,
                 float(pres[:n4.value].sum()))
    print(form or "default", "tail tables:", bool(args.post_own_g), "n3", n3.value, "n4", n4.value,
          "sum g^2 partials %.17g" % out[form][1], "direct %.17g" % float(out[form][0] @ out[form][0]),
          "resid partials %.6g" % out[form][5])
a, b = out[""][0], out["no-post-tail"][0]
bad = np.flatnonzero(a != b)
print("entries that differ:", bad.size, "of", N, "max abs diff", float(np.max(np.abs(a - b))) if bad.size else 0.0)
if bad.size:
    print("first bad indices:", bad[:20], "segments: x<%d, s_nl<%d, s_lb<%d" % (n, n + m, n + m + n))
