"""Randomised cross-check of the projections Z, LS, Y (projections.py:14-290) on sparse
Jacobians of many structures -- random sparsity of several densities, a band with its rows
shuffled, block diagonal, a band plus a few dense rows, dense storage -- so that every
``(A A')^-1`` solver of the product takes its turn (banded with and without reordering, dense
Cholesky, preconditioned CG, box-Schur), against the host oracle's projections on the same
seeded inputs; then the modified dogleg step (qp_subproblem.py:320-413) on the same Jacobian.

    python tests/fuzz_projections.py [cases] [seed]      (tests/test_gpu_qp.py runs 20 cases of <= 1500 rows;
                                                          the 6000-row cases take minutes of host time)"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd")); sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp
from ipsolver import device as dv, projector
from ipsolver.dense import DeviceDense
import oracle


def jacobian(rng, max_m):
    kind = str(rng.choice(["random", "random", "shuffled-band", "blocks", "band+dense-rows", "dense"]))
    if kind == "dense":
        # (a third of them large enough for the tiled Gram kernel on the fp64 matrix cores)
        big = rng.random() < 0.33 and max_m >= 300
        m = int(rng.integers(100, 500)) if big else int(rng.integers(1, 60))
        n = m + (int(rng.integers(50, 1500)) if big else int(rng.integers(1, 80)))
        return kind, rng.standard_normal((m, n))
    m = int(rng.choice([k for k in (1, 3, 40, 300, 1500, 6000) if k <= max_m]))
    n = m + int(rng.integers(1, 3 * m + 5))
    if kind == "random":
        dens = float(rng.choice([0.5, 3.0, 8.0])) / n * 2
        A = sp.random(m, n, density=min(1.0, dens + 2.0 / n), random_state=int(rng.integers(1 << 30)),
                      data_rvs=rng.standard_normal, format="lil")
    elif kind in ("shuffled-band", "band+dense-rows"):
        w = int(rng.integers(2, 9))
        A = sp.lil_matrix((m, n))
        step = n / m
        for i in range(m):
            j0 = int(i * step)
            cols = np.arange(j0, min(n, j0 + w))
            A[i, cols] = rng.standard_normal(len(cols))
        if kind == "band+dense-rows" and m > 5:
            for i in rng.permutation(m)[:2]:
                cols = rng.permutation(n)[:max(3, n // 10)]
                A[i, cols] = rng.standard_normal(len(cols))
    else:
        A = sp.block_diag([rng.standard_normal((bm, bm + int(rng.integers(1, 6))))
                           for bm in rng.integers(1, 7, max(1, m // 4))], format="lil")
        m, n = A.shape
    # full row rank: a dominant entry per row in a column of its own
    own = rng.permutation(n)[:m]
    for i in range(m):
        A[i, own[i]] = A[i, own[i]] + 4.0 * (1 if rng.random() < 0.5 else -1)
    A = sp.csr_matrix(A)
    if kind == "shuffled-band":
        A = A[rng.permutation(m)]
    A.sort_indices()
    return kind, A


def run(cases, seed, verbose=True, max_m=6000):
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(cases):
        kind, A = jacobian(rng, max_m)
        m, n = A.shape
        x, y = rng.standard_normal(n), rng.standard_normal(m)
        # every third sparse case: no dense Cholesky (as for > 16384 rows), so that what is not
        # banded goes to the preconditioned CG on A A'
        no_dense = kind != "dense" and case % 3 == 2
        import ipsolver.dense as dense
        keep = dense.DenseNormalSolver.MAX_ROWS_FROM_SPARSE
        try:
            if no_dense:
                dense.DenseNormalSolver.MAX_ROWS_FROM_SPARSE = 0
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                Ad = DeviceDense.from_host(A) if kind == "dense" else dv.DeviceCSR.from_scipy(A)
                Z, LS, Y = projector.projections(Ad)
                Zo, LSo, Yo = oracle.projections(A)
        finally:
            dense.DenseNormalSolver.MAX_ROWS_FROM_SPARSE = keep
        solver = type(Z.projector.solver).__name__
        inner = type(getattr(Z.projector.solver, "inner", None)).__name__
        errs = []
        for name, op, oop, v in (("Z", Z, Zo, x), ("LS", LS, LSo, x), ("Y", Y, Yo, y)):
            got = op.dot(dv.DVec.from_host(v)).to_host()
            want = oop.dot(v)
            errs.append(float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-300)))
        # modified_dogleg (qp_subproblem.py:320-413) on the same Jacobian: a radius / box drawn so
        # that the Newton point is accepted, cut by the sphere, or cut by the box
        from ipsolver import qp
        yb = Yo.dot(y)
        scale = float(np.linalg.norm(yb)) or 1.0
        mode = int(rng.integers(0, 4))
        radius = scale * [2.0, 0.5, 2.0, 0.05][mode]
        if mode == 2:
            lo_, hi_ = np.full(n, -0.3 * np.max(np.abs(yb))), np.full(n, 0.4 * np.max(np.abs(yb)))
        else:
            lo_, hi_ = np.full(n, -np.inf), np.full(n, np.inf)
        got = qp.modified_dogleg(Ad, Y, y, radius, lo_, hi_).to_host()
        want = oracle.modified_dogleg(A, Yo, y, radius, lo_, hi_)
        errs.append(float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-300)))
        line = "case %2d %-16s m=%5d n=%6d nnz=%7d %-22s Z %.1e  LS %.1e  Y %.1e  dogleg(%d) %.1e" % (
            case, kind, m, n, (A != 0).sum(), solver + ("/" + inner if inner != "NoneType" else ""),
            errs[0], errs[1], errs[2], mode, errs[3])
        if verbose:
            print(line, flush=True)
        assert max(errs) <= 1e-9, line
        worst = max(worst, *errs)
    return worst


if __name__ == "__main__":
    w = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("ok, worst relative deviation from the oracle's projections %.1e" % w)
