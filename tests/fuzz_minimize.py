"""Randomised end-to-end cross-check: ``minimize_constrained`` on small random NLPs -- a
strictly convex quartic objective under a random MIX of the reference's constraint classes
(dense or sparse linear equalities, interval / one-sided linear inequalities, nonlinear ball
constraints, ragged boxes; _constraints.py, _canonical_constraint.py) -- on the HIP backend
against the same call on the host oracle's backend (oracle/numpy_backend.py: the reference's
algorithms, tr_interior_point.py / equality_constrained_sqp.py, over numpy + SuperLU).  The two
factor their projections differently, so traces drift at the 1e-10 level; compared are the
first outer iterations row by row (up to the first long CG call: there the problem itself
moves by 1e-5 under one ulp) and the end points (a
unique minimiser).

    python tests/fuzz_minimize.py [cases] [seed]         (tests/test_gpu_e2e.py runs 12 cases)"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sps
import ipsolver
from ipsolver import backend
import oracle.numpy_backend as nb


def problem(rng):
    n = int(rng.integers(4, 41))
    # one case in seven: a dense equality-constrained problem of a few hundred variables (the
    # shape of BASELINE config 2: dense Jacobian and Hessian on the device-resident dense loop,
    # the Gram matrix on the matrix cores)
    big_dense = bool(rng.random() < 0.15)
    if big_dense:
        n = int(rng.integers(120, 500))
    B = rng.standard_normal((n, n)) / np.sqrt(n)
    Q = B @ B.T + np.diag(rng.uniform(0.5, 2.0, n))
    c = rng.standard_normal(n)
    x0 = rng.uniform(-0.3, 0.3, n)
    sparse = bool(rng.random() < 0.5) and not big_dense
    fun = lambda x: 0.5 * x @ Q @ x + c @ x + 0.05 * np.sum(x ** 4)
    grad = lambda x: Q @ x + c + 0.2 * x ** 3
    hess = (lambda x: sps.csr_matrix(Q + np.diag(0.6 * x ** 2))) if sparse else \
        (lambda x: Q + np.diag(0.6 * x ** 2))
    cons, tags = [], []
    # one case in five: the reference's DEFAULT Hessian, finite differences of the gradient (an
    # operator: applied between the device loop's iterations, _numdiff.py:403-441)
    if rng.random() < 0.2 and not big_dense:
        hess = str(rng.choice(["2-point", "3-point"]))
        tags.append("fd-" + hess)
    m_eq = int(rng.integers(0, max(1, n // 3) + 1))
    if big_dense:
        m_eq = max(m_eq, 8)
    if m_eq:
        A = rng.standard_normal((m_eq, n)) * (rng.random((m_eq, n)) < (0.4 if sparse else 1.0))
        A[np.arange(m_eq), rng.permutation(n)[:m_eq]] += 2.0          # full row rank
        cons.append(ipsolver.LinearConstraint(sps.csr_matrix(A) if sparse else A, ("equals", A @ x0)))
        tags.append("eq%d" % m_eq)
    if rng.random() < 0.6 and not big_dense:
        k = int(rng.integers(1, max(2, n // 4) + 1))
        C = rng.standard_normal((k, n)) * (rng.random((k, n)) < 0.5)
        C[np.arange(k), rng.permutation(n)[:k]] += 1.0
        mid = C @ x0
        kind = str(rng.choice(["interval", "less", "greater"]))
        spec = {"interval": ("interval", mid - rng.uniform(0.2, 1.0, k), mid + rng.uniform(0.2, 1.0, k)),
                "less": ("less", mid + rng.uniform(0.2, 1.0, k)),
                "greater": ("greater", mid - rng.uniform(0.2, 1.0, k))}[kind]
        cons.append(ipsolver.LinearConstraint(sps.csr_matrix(C) if sparse else C, spec))
        tags.append("lin-%s%d" % (kind, k))
    if rng.random() < 0.5 and not big_dense:
        S = rng.permutation(n)[:max(2, n // 2)]
        r2 = float(np.sum(x0[S] ** 2)) + rng.uniform(0.3, 1.5)

        def ball(x, S=S):
            return np.array([np.sum(x[S] ** 2)])

        def ball_jac(x, S=S):
            J = np.zeros((1, n)); J[0, S] = 2 * x[S]
            return sps.csr_matrix(J) if sparse else J

        def ball_hess(x, v, S=S):
            d = np.zeros(n); d[S] = 2 * v[0]
            return sps.diags(d).tocsr() if sparse else np.diag(d)
        cons.append(ipsolver.NonlinearConstraint(ball, ("less", r2), ball_jac, ball_hess))
        tags.append("ball")
    if rng.random() < 0.6 and not big_dense:
        kind = rng.integers(0, 4, n)                       # 0 none, 1 lower, 2 upper, 3 both
        lo = np.where(kind & 1, x0 - rng.uniform(0.1, 1.0, n), -np.inf)
        hi = np.where(kind & 2, x0 + rng.uniform(0.1, 1.0, n), np.inf)
        cons.append(ipsolver.BoxConstraint(("interval", lo, hi)))
        tags.append("box")
    ineq = any(not t.startswith(("eq", "fd-")) for t in tags)
    has_eq = any(t.startswith("eq") for t in tags)
    methods = ["tr_interior_point"] + ([] if ineq or not has_eq else ["equality_constrained_sqp"])
    return dict(n=n, fun=fun, grad=grad, hess=hess, x0=x0, cons=cons, tags=tags, sparse=sparse,
                methods=methods)


def solve(P, method):
    rows = []

    def record(state):
        rows.append([int(state.niter), int(state.cg_niter), float(state.optimality),
                     float(state.constr_violation)])
        return False
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = ipsolver.minimize_constrained(P["fun"], P["x0"], P["grad"], P["hess"], P["cons"],
                                            method=method, sparse_jacobian=P["sparse"] or None,
                                            callback=record)
    return res, np.array(rows)


def run(cases, seed, verbose=True, only=None):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(cases):
        P = problem(rng)
        if only is not None and case != only:      # (replay one case of a longer run)
            continue
        for method in P["methods"]:
            got, rows = solve(P, method)
            with backend.use(nb):
                want, wrows = solve(P, method)
            # rows compared: the first six, but only up to the first long CG call -- one
            # box-constrained call of 22 iterations moved 1e-5 between the two sides on
            # identical inputs, and as far on the oracle alone when its gradient moved by one
            # ulp (the ``last_feasible_x`` bookkeeping of qp_subproblem.py:599-616 is discrete)
            k = min(6, len(rows), len(wrows))
            long_cg = np.flatnonzero(np.diff(wrows[:k, 1]) > 12)
            if len(long_cg):
                k = int(long_cg[0]) + 1
            if any(t.startswith("fd-") for t in P["tags"]):
                # (a finite-difference Hessian product carries ~1e-8 of rounding noise, another
                # draw of it on each side, and every CG iteration amplifies it: seed 505 case 83
                # agrees to 1e-16 / 3e-9 / 1e-5 / 2e-4 after 1 / 4 / 13 / 18 CG iterations --
                # rows are compared while the oracle's count is at most 8)
                k = max(min(k, int(np.searchsorted(wrows[:k, 1], 8, side="right"))), min(k, 2))
            dx = float(np.max(np.abs(got.x - want.x)) / max(1.0, np.max(np.abs(want.x))))
            line = "case %2d n=%2d %-5s %-24s %-34s status %d/%d  %3d/%3d outer  |dx| %.1e  opt %.1e/%.1e" % (
                case, P["n"], "csr" if P["sparse"] else "dense", method, "+".join(P["tags"]) or "-",
                got.status, want.status, got.niter, want.niter, dx, got.optimality, want.optimality)
            if verbose:
                print(line, flush=True)
            if max(got.niter, want.niter) > 300:
                # a degenerate instance: hundreds of outer iterations at the last barrier
                # parameters on BOTH sides, the optimality hovering between 2e-8 and 1e-6 (one
                # case of seed 61: the oracle's run got below gtol after 591, this side had not
                # after 1000 -- from a trajectory that differs from row 13 on by 1e-11).  Only
                # the end point says anything there.
                assert dx <= 1e-4, line
                continue
            # (1: gtol, 2: xtol -- an end game on the merit function's rounding floor may end
            # either way on either side, _minimize_constrained.py:395-407)
            assert got.status in (1, 2) and want.status in (1, 2), line
            assert np.array_equal(rows[:k, :2], wrows[:k, :2]), line
            # (a finite-difference Hessian product carries ~1e-8 of rounding noise, another draw
            # of it on each side)
            rtol = 1e-4 if any(t.startswith("fd-") for t in P["tags"]) else 1e-6
            assert np.allclose(rows[:k, 2:], wrows[:k, 2:], rtol=rtol, atol=1e-10), line
            # (late barrier subproblems amplify 1e-10 differences into other iteration counts --
            # 198 against 443 outer iterations on one of these problems, the same end point:
            # DESIGN.md section 7 -- so the counts are printed, not compared)
            # (weakly active constraints leave a barrier run's end point determined to ~sqrt(mu))
            assert dx <= 1e-4 and got.constr_violation <= 1e-8, line
            worst = max(worst, dx)
    return worst


if __name__ == "__main__":
    w = run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
            only=int(sys.argv[3]) if len(sys.argv) > 3 else None)
    print("ok, worst end-point deviation %.1e" % w)
