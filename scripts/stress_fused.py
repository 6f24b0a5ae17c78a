"""Random banded problems: fused kernels vs one launch per step (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, scipy.sparse as sps
import ipsolver.qp as qp, ipsolver.projector as proj, ipsolver.device as dv, ipsolver.cg_fused as cf

def host(v): return v.to_host()
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(700, 60000)); m = max(8, n // int(rng.integers(6, 14)))
    hbw = int(rng.integers(1, 7)); abw = int(rng.integers(3, 24))
    offs = list(range(-hbw, hbw + 1))
    Hm = sps.diags([rng.uniform(-1, 1, n - abs(o)) for o in offs], offs, format="csr")
    Hm = sps.csr_matrix(0.5 * (Hm + Hm.T) + sps.diags(np.full(n, 2.0 * hbw + 1.0)))
    starts = np.maximum.accumulate(np.minimum(np.arange(m) * (n // m) + rng.integers(0, 3, m), n - abw))
    lens = rng.integers(2, abw + 1, m)
    rows = np.repeat(np.arange(m), lens)
    cols = np.concatenate([s + np.arange(k) for s, k in zip(starts, lens)])
    Am = sps.csr_matrix((rng.standard_normal(len(cols)), (rows, cols)), shape=(m, n))
    try:
        A = dv.DeviceCSR.from_scipy(Am); H = dv.DeviceCSR.from_scipy(Hm)
        Z, LS, Y = proj.projections(A)
    except Exception as exc:
        print(seed, "skip:", type(exc).__name__); continue
    if not cf.supports(H, Z, Y):
        print(seed, "skip: not on the fused loop"); continue
    c = rng.standard_normal(n)
    x_free, _ = qp.projected_cg(H, c, Z, Y, np.zeros(m), tol=1e-12, max_iter=80)
    amax = float(np.max(np.abs(host(x_free))))
    variants = {"plain": dict(tol=1e-13, max_iter=80),
                "sphere": dict(trust_radius=0.6 * dv.norm(x_free)),
                "box": dict(lb=np.full(n, -0.5 * amax), ub=np.full(n, 0.5 * amax), max_iter=60)}
    for name, kw in variants.items():
        res = []
        for nf in ("", "1"):
            if nf: os.environ["IPX_DEBUG_FORMS"] = "no-fuse"
            else: os.environ.pop("IPX_DEBUG_FORMS", None)
            x, info = qp.projected_cg(H, c, Z, Y, np.zeros(m), **kw)
            res.append((host(x), info))
        (x1, i1), (x2, i2) = res
        ok = (i1["niter"], i1["stop_cond"], i1["hits_boundary"]) == (i2["niter"], i2["stop_cond"], i2["hits_boundary"]) \
            and np.max(np.abs(x1 - x2)) <= 1e-11 * max(np.max(np.abs(x2)), 1e-300)
        if not ok:
            bad += 1
            print("MISMATCH seed %d %s n=%d m=%d hbw=%d: %s vs %s, dx=%.2e" % (seed, name, n, m, hbw, i1, i2, np.max(np.abs(x1 - x2))))
    os.environ.pop("IPX_DEBUG_FORMS", None)
    L = cf._Loop(H, Z.projector, None, None)
    from ipsolver import _hip
    import ctypes
    dec = _hip.load().ipx_banded_decoupled(ctypes.c_void_p(Z.projector.solver.handle))
    # the solver against a direct sparse solve of (A A') v = w
    w = rng.standard_normal(m)
    v = Z.projector.solver.solve(dv.DVec.from_host(w)).to_host()
    S = (Am @ Am.T).tocsc()
    import scipy.sparse.linalg as spla
    vref = spla.spsolve(S, w)
    serr = np.max(np.abs(v - vref)) / np.max(np.abs(vref))
    if serr > 1e-9:
        bad += 1
        print("SOLVE MISMATCH seed %d: %.2e" % (seed, serr))
    print(seed, "n=%d m=%d hbw=%d k=%d decoupled=%d solve_err=%.1e  fused: H %d A %d Atv %d" % (n, m, hbw, Z.projector.solver.k if hasattr(Z.projector.solver, "k") else -1, dec, serr, L.args.H_hmax, L.args.A_span, L.args.At_qv))
print("mismatches:", bad)
