"""Single ``projected_cg`` calls out of the REFERENCE's barrier runs on the config-5 style
problem (box on every variable + nonlinear inequalities), from the first barrier parameter to
the last (mu <= 1e-6: slacks of active bounds ~1e-8, the regime the cancellation-free box-Schur
formulas of csrc/boxschur.hip exist for).  tests/golden/make_golden.py --late-barrier recorded
what the reference passed to ``projected_cg`` (equality_constrained_sqp.py:125-132, assembled
by tr_interior_point.py:141-241) and what it returned; here the product solves the same
subproblems through its own path -- augmented Jacobian, z-space Hessian operator, box-Schur
projections, device-resident loop -- on one GPU and row-sharded over two processes."""
import os

import numpy as np
import pytest
import scipy.sparse as sps

from banded_setup import load_synthetic
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

SIZES = {400: 6, 12000: 4}


def _gold(n):
    return dict(np.load(os.path.join(GOLDEN, "late_barrier_n%d.npz" % n)))


def _pieces(gold, j):
    """The data of call j as the reference's SQP held it."""
    n, m = (int(v) for v in gold["n"])
    g = lambda k: gold["c%d_%s" % (j, k)]
    return dict(n=n, m=m, x=g("x_vars"), s=g("s"), v_nl=g("v_nl"), Hs=g("Hs"), c=g("c"),
                lb=g("lb"), radius=float(g("radius")), mu=float(g("mu")), x_out=g("x"),
                info=[int(v) for v in g("info")], allvecs=g("allvecs"),
                stride=int(gold["stride"][0]))


def close_rel(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.max(np.abs(b)), 1e-300)
    err = np.max(np.abs(a - b)) / scale
    assert err <= tol, "relative deviation %.3e > %.1e" % (err, tol)
    return err


def _solve_single(d, return_all=False):
    """The product's own assembly (backend_hip: augmented Jacobian with the slack entries,
    Hessian terms merged into one fused SpMV + diagonal) and solve."""
    from ipsolver import backend_hip as xp
    from ipsolver import cg_fused
    from ipsolver.device import DVec
    n, m = d["n"], d["m"]
    n_ineq = m + 2 * n
    prob = load_synthetic().CenteredBandedNLP(n, m, eps=1.0)
    eye = sps.identity(n, format="csr")
    J_ineq = sps.vstack([prob.constr_jac(d["x"]), -eye, eye], format="csr")
    A = xp.augmented_jacobian(sps.csr_matrix((0, n)), J_ineq, DVec.from_host(d["s"]), n, 0, n_ineq)
    H = xp.hessian_operator([prob.hess(d["x"]), prob.constr_hess(d["x"], d["v_nl"])], n,
                            DVec.from_host(d["Hs"]))
    Z, LS, Y = xp.projections(A)
    assert type(Z.projector.solver).__name__ == "BoxSchurNormalSolver"
    assert cg_fused.supports(H, Z, Y)
    N = n + n_ineq
    calls = cg_fused.STATS["calls"]
    x, info = xp.projected_cg(H, DVec.from_host(d["c"]), Z, Y, DVec.zeros(n_ineq), d["radius"],
                              DVec.from_host(d["lb"]), DVec.full(N, np.inf),
                              **({"return_all": True} if return_all else {}))
    if not return_all:
        assert cg_fused.STATS["calls"] == calls + 1, "device-resident loop not taken"
    return x, info


@pytest.mark.parametrize("n", sorted(SIZES))
def test_late_barrier_subproblems_single_gpu(n):
    """Integers exact (niter, stop_cond, hits_boundary), the returned step and the first 20
    iterates to 1e-10 of their norms -- every recorded call, the last ones included."""
    gold = _gold(n)
    assert len(gold["picks"]) == SIZES[n]
    worst = 0.0
    for j in range(SIZES[n]):
        d = _pieces(gold, j)
        x, info = _solve_single(d)
        got = [info["niter"], info["stop_cond"], int(info["hits_boundary"])]
        assert got == d["info"], (j, d["mu"], got, d["info"])
        worst = max(worst, close_rel(x.to_host(), d["x_out"], 1e-10))
        # the general driver (same kernels, host-side control flow) for the iterates
        xg, ig = _solve_single(d, return_all=True)
        assert [ig["niter"], ig["stop_cond"], int(ig["hits_boundary"])] == d["info"]
        st = d["stride"]
        for k, want in enumerate(d["allvecs"]):
            close_rel(ig["allvecs"][k].to_host()[::st], want, 1e-10)
    print("late barrier n=%d: worst relative deviation of the returned step %.2e" % (n, worst))
