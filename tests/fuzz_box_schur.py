"""Randomised cross-check of the barrier-shaped projected CG (the augmented Jacobian
[[J, S_nl, 0, 0], [-I_L, 0, S_lb, 0], [I_U, 0, 0, S_ub]] of tr_interior_point.py:141-194, slack
bounds on the CG step: the box-Schur elimination of csrc/boxschur.hip inside the device loop)
against the same loop with the general group tables (IPX_DEBUG_FORMS=no-compact-groups) and,
for small cases, the host oracle's projected CG (qp_subproblem.py:332-637): random row lengths
of J (15 is the one that takes the solve-forms-its-own-right-hand-side launch), random overlaps,
bounds on every variable / a ragged mix / lower only, slacks down to 1e-6.

    python tests/fuzz_box_schur.py [cases] [seed]        (tests/test_gpu_qp.py runs 16 cases)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
from ipsolver import device as dv, projector, cg_fused
import oracle


def run(cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(cases):
        rl = int(rng.choice([3, 7, 8, 15, 15, 16, 11]))
        shift = int(rng.integers((rl + 1) // 2, rl + 1))
        m = int(rng.choice([37, 259, 261, 520, 1300, 3000, 20000]))
        n = (m - 1) * shift + rl + int(rng.integers(0, 9))
        rows = np.repeat(np.arange(m), rl)
        cols = (np.arange(m)[:, None] * shift + np.arange(rl)[None, :]).ravel()
        J = sp.csr_matrix((rng.uniform(0.5, 1.5, m * rl) * rng.choice([-1.0, 1.0], m * rl), (rows, cols)),
                          shape=(m, n))
        mode = str(rng.choice(["all", "ragged", "lower"]))
        if mode == "all":
            L = U = np.arange(n)
        elif mode == "lower":
            L, U = np.arange(n), np.arange(0)
        else:
            kind = rng.integers(0, 4, n)
            L, U = np.flatnonzero(kind & 1), np.flatnonzero(kind & 2)
        nl, nu = len(L), len(U)
        I = sp.eye(n, format="csr")
        s = rng.uniform(1e-6, 2.0, m + nl + nu)
        blocks = [[J, sp.diags(s[:m]), None, None],
                  [-I[L], None, sp.diags(s[m:m + nl]), None]]
        if nu:
            blocks.append([I[U], None, None, sp.diags(s[m + nl:])])
        else:
            blocks = [row[:3] for row in blocks]
        A = sp.bmat(blocks, format="csr")
        A.sort_indices()
        M, N = A.shape
        off = rng.uniform(-0.4, 0.4, n - 1)
        Hx = sp.diags([off, rng.uniform(1.5, 2.5, n), off], [-1, 0, 1], format="csr")
        Hz = sp.block_diag([Hx, sp.diags(rng.uniform(0.5, 2.0, N - n))], format="csr")
        c = rng.standard_normal(N)
        b = np.zeros(M)
        lb = np.concatenate((np.full(n, -np.inf), np.full(N - n, -0.995)))
        kw = dict(trust_radius=float(rng.choice([5.0, 50.0, np.inf])), tol=1e-10, max_iter=40)
        runs = []
        try:
            for flag in ("", "no-compact-groups"):
                if flag:
                    os.environ["IPX_DEBUG_FORMS"] = flag
                else:
                    os.environ.pop("IPX_DEBUG_FORMS", None)
                Ad = dv.DeviceCSR.from_scipy(A)
                Z, LS, Y = projector.projections(Ad)
                Hd = dv.DeviceCSR.from_scipy(Hz)
                fused = cg_fused.supports(Hd, Z, Y)
                before = cg_fused.STATS["calls"]
                x, info = cg_fused.projected_cg(Hd, dv.DVec.from_host(c), Z, Y, dv.DVec.from_host(b),
                                                lb=dv.DVec.from_host(lb), **kw)
                runs.append((x.to_host(), info, type(Z.projector.solver).__name__,
                             fused and cg_fused.STATS["calls"] == before + 1))
        finally:
            os.environ.pop("IPX_DEBUG_FORMS", None)
        (x1, i1, solver, dev1), (x2, i2, _, _) = runs
        scale = max(np.max(np.abs(x1)), 1e-300)
        d12 = np.max(np.abs(x1 - x2)) / scale
        line = "case %2d rl=%2d shift=%2d m=%5d n=%6d %-6s radius=%-4g %s loop=%d  compact-vs-general %.1e  %s" % (
            case, rl, shift, m, n, mode, kw["trust_radius"], solver, dev1, d12, i1)
        if N <= 40000:
            Zo, _, Yo = oracle.projections(A)
            xo, io = oracle.projected_cg(Hz, c, Zo, Yo, b, lb=lb, ub=np.full(N, np.inf), **kw)
            do = np.max(np.abs(x1 - xo)) / max(np.max(np.abs(xo)), 1e-300)
            line += "  vs-oracle %.1e (%d its)" % (do, io["niter"])
            assert (io["niter"], io["stop_cond"], io["hits_boundary"]) == \
                (i1["niter"], i1["stop_cond"], i1["hits_boundary"]), (line, io)
            assert do <= 1e-9, line
        if verbose:
            print(line, flush=True)
        assert i1 == i2 and d12 <= 1e-12, line
        worst = max(worst, d12)
    return worst


if __name__ == "__main__":
    w = run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("ok, worst relative deviation between the table forms %.1e" % w)
