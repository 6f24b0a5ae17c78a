"""Randomised cross-check of the device loop's fused kernels (step1 + A.r, cyclic reduction +
g = r - A'v from the ELL(2) form, step2 + H.p, the resident launch) against the loop without
fused kernels (IPX_DEBUG_FORMS=no-fuse) and, for small cases, the host oracle: random row
lengths 2..16, random overlaps (tridiagonal or diagonal A A'), row counts around the
workgroup size of the solve (260), unconstrained variables at the end and in between.

    python tests/fuzz_fused_loop.py [cases] [seed]      (tests/test_gpu_qp.py runs 24 cases)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
from ipsolver import device as dv, projector, qp, cg_fused
from ipsolver.operators import DeviceHessian
import oracle

def run(cases, seed, verbose=True, only=None, fixed=None):
    rng = np.random.default_rng(seed)
    worst, done = 0.0, 0
    for case in range(cases):
        rl = int(rng.integers(2, 17))
        shift = int(rng.integers((rl + 1) // 2, rl + 2))          # rl + 1: a free variable between rows
        if (rl, shift) == (2, 1):       # (a bidiagonal chain: cond(A) grows like exp(sqrt(m)) --
            shift = 2                   #  3.7e9 at m = 5000, the singular-Jacobian path on both sides)
        m = int(rng.choice([1, 2, 37, 259, 260, 261, 519, 520, 521, 1300, 5000, 26001, 60000]))
        tail = int(rng.integers(0, 40))
        if fixed is not None:           # (a given shape instead of the drawn one: regression cases)
            rl, shift, m, tail = fixed
        n = (m - 1) * shift + rl + tail
        n += n % 2
        if n - m < 3:
            continue
        rows = np.repeat(np.arange(m), rl)
        cols = (np.arange(m)[:, None] * shift + np.arange(rl)[None, :]).ravel()
        Ah = sp.csr_matrix((rng.uniform(0.5, 1.5, m * rl) * rng.choice([-1.0, 1.0], m * rl), (rows, cols)),
                           shape=(m, n))
        # one case in five: the rows in random order -- A A' is banded only after a row
        # permutation, which the projector applies once (projector.projections) so that the
        # loop still sees a banded matrix; b follows the rows
        shuffled = bool(rng.random() < 0.2) and m > 2
        if shuffled:
            Ah = Ah[rng.permutation(m)]
            Ah.sort_indices()
        off = rng.uniform(-0.4, 0.4, n - 1)
        Hh = sp.diags([off, rng.uniform(1.5, 2.5, n), off], [-1, 0, 1], format="csr")
        c = rng.standard_normal(n)
        b = rng.standard_normal(m) * (rng.random() < 0.5)
        K = int(min(12, n - m))
        radius = float(rng.choice([1e300, np.inf, 5.0]))
        if radius == 5.0:       # (a radius the start lies inside of: else the call raises, as the reference's)
            radius = 5.0 + 2.0 * float(np.linalg.norm(b)) * 3.0
        kw = dict(tol=0, max_iter=K, trust_radius=radius)
        if only is not None and case != only:      # (replay one case of a longer run)
            continue
        runs = []
        try:
            for flag in ("", "no-fuse", "no-resident"):
                if flag:
                    os.environ["IPX_DEBUG_FORMS"] = flag
                else:
                    os.environ.pop("IPX_DEBUG_FORMS", None)
                A = dv.DeviceCSR.from_scipy(Ah)
                H = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(Hh))
                Z, LS, Y = projector.projections(A)
                before = dict(cg_fused.STATS)
                x, info = qp.projected_cg(H, c, Z, Y, b, **kw)
                runs.append((x.to_host() if hasattr(x, "to_host") else np.asarray(x), info,
                             cg_fused.STATS["resident_calls"] - before["resident_calls"]))
        finally:
            os.environ.pop("IPX_DEBUG_FORMS", None)
        (x1, i1, res1), (x2, i2, _), (x3, i3, _) = runs
        scale = max(np.max(np.abs(x2)), 1e-300)
        d12, d13 = np.max(np.abs(x1 - x2)) / scale, np.max(np.abs(x1 - x3)) / scale
        line = "case %2d rl=%2d shift=%2d m=%6d n=%7d b%d%s radius=%-6g resident=%d  fused-vs-plain %.1e  vs-three-launch %.1e  %s" % (
            case, rl, shift, m, n, int(np.any(b)), " shuffled" if shuffled else "", kw["trust_radius"], res1, d12, d13, i1)
        if n <= 30000:
            Zo, LSo, Yo = oracle.projections(Ah)
            xo, io = oracle.projected_cg(Hh, c, Zo, Yo, b, **kw)
            do = np.max(np.abs(x1 - xo)) / max(np.max(np.abs(xo)), 1e-300)
            line += "  vs-oracle %.1e" % do
            assert io["niter"] == i1["niter"] and io["stop_cond"] == i1["stop_cond"], (line, io)
            assert do <= 1e-9, line
        if verbose:
            print(line, flush=True)
        assert (i1["niter"], i1["stop_cond"]) == (i2["niter"], i2["stop_cond"]) == (i3["niter"], i3["stop_cond"]), line
        assert d12 <= 1e-11 and d13 <= 1e-11, line
        worst = max(worst, d12, d13)
    return worst


if __name__ == "__main__":
    w = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
            only=int(sys.argv[3]) if len(sys.argv) > 3 else None)
    print("ok, worst relative deviation between the forms %.1e" % w)
