import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, scipy.linalg
import oracle
from ipsolver import projector, device as dv
rng = np.random.default_rng(0)
m, n = 200, 1000
for cond in (1e1, 1e3, 1e5, 1e6, 1e7):
    U, _ = np.linalg.qr(rng.standard_normal((m, m)))
    V, _ = np.linalg.qr(rng.standard_normal((n, m)))
    s = np.logspace(0, -np.log10(cond), m)
    A = (U * s) @ V.T
    x = rng.standard_normal(n); b = rng.standard_normal(m)
    # exact-ish via extended precision QR (longdouble) is overkill: use oracle QR as the reference's answer
    Zo, LSo, Yo = oracle.projections(A)
    Z, LS, Y = projector.projections(A)
    G = A @ A.T
    c = scipy.linalg.cho_factor(G)
    ls_trsv = scipy.linalg.cho_solve(c, A @ x)
    y_trsv = A.T @ scipy.linalg.cho_solve(c, b)
    def rel(a, b): return np.max(np.abs(a - b)) / np.max(np.abs(b))
    print("cond %.0e  Z %.1e  LS inv %.1e trsv %.1e   Y inv %.1e trsv %.1e  refin %d" % (cond, rel(Z.dot(x).to_host(), Zo.dot(x)), rel(LS.dot(x).to_host(), LSo.dot(x)), rel(ls_trsv, LSo.dot(x)), rel(Y.dot(b).to_host(), Yo.dot(b)), rel(y_trsv, Yo.dot(b)), Z.projector.stats["refinements"]))
