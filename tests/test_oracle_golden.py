"""Pin the CPU oracle against outputs of the reference itself.

Every expected value below comes from ``tests/golden/*`` which
``tests/golden/make_golden.py`` produced by running the reference on the
inputs in ``tests/cases_small.py`` / ``tests/banded_setup.py``.
"""
import numpy as np
import pytest
import scipy.sparse as sps

import cases_small as cs
from banded_setup import BandedInstance
from conftest import unjson, rel_err, close_projection
from conftest import close as _close
import oracle

RTOL = 1e-12


def close(a, b, tol=RTOL, zero_scale=None):
    _close(a, b, tol, zero_scale)


def check_interval(got, want):
    want = unjson(want)
    assert bool(got[2]) == bool(want[2])
    close(got[:2], want[:2])


def test_sphere_intersections(qp_small):
    it = iter(qp_small["sphere"])
    for z, d, r in cs.SPHERE:
        for line in (False, True):
            check_interval(oracle.sphere_intersections(z, d, r, line), next(it))


def test_box_intersections(qp_small):
    it = iter(qp_small["box"])
    for z, d, lb, ub in cs.BOX:
        for line in (False, True):
            check_interval(oracle.box_intersections(z, d, lb, ub, line),
                           next(it))


def test_box_sphere_intersections(qp_small):
    it = iter(qp_small["box_sphere"])
    for z, d, lb, ub, r in cs.BOX_SPHERE:
        for line in (False, True):
            check_interval(
                oracle.box_sphere_intersections(z, d, lb, ub, r, line),
                next(it))


def test_modified_dogleg_small(qp_small):
    for (A, b, r, lb, ub), want in zip(cs.DOGLEG, qp_small["dogleg"]):
        A = np.array(A, dtype=float)
        _, _, Y = oracle.projections(A)
        close(oracle.modified_dogleg(A, Y, np.array(b, float), r, lb, ub),
              unjson(want))


@pytest.mark.parametrize("idx", range(len(cs.PCG)))
def test_projected_cg_small(qp_small, idx):
    case, want = cs.PCG[idx], qp_small["pcg"][idx]
    H = sps.csc_matrix(np.array(case["H"], dtype=float))
    A = sps.csc_matrix(np.array(case["A"], dtype=float))
    c = np.array(case["c"], dtype=float)
    b = np.array(case["b"], dtype=float)
    Z, _, Y = oracle.projections(A)
    if case.get("raises"):
        assert "raises" in want
        with pytest.raises(ValueError) as err:
            oracle.projected_cg(H, c, Z, Y, b, **case["kw"])
        assert str(err.value)[:20] == want["raises"][:20]
        return
    x, info = oracle.projected_cg(H, c, Z, Y, b, return_all=True, **case["kw"])
    assert info["niter"] == want["niter"]
    assert info["stop_cond"] == want["stop_cond"]
    assert info["hits_boundary"] == want["hits_boundary"]
    close(x, unjson(want["x"]), 1e-11)
    assert len(info["allvecs"]) == len(want["allvecs"])
    for a, w in zip(info["allvecs"], want["allvecs"]):
        close(a, unjson(w), 1e-11)


def test_known_answers():
    """Literal answers quoted by the reference's tests
    (test_qp_subproblem.py:24-34, :356, :376-377)."""
    H = sps.csc_matrix(np.array(cs.PCG[0]["H"], dtype=float))
    A = sps.csc_matrix(np.array(cs.PCG[0]["A"], dtype=float))
    x, lam = oracle.eqp_kktfact(H, np.array([-8., -3, -3]), A,
                                np.array([-3., 0]))
    np.testing.assert_allclose(x, [2, -1, 1], atol=1e-12)
    np.testing.assert_allclose(lam, [3, -2], atol=1e-12)
    A1 = np.array([[1., 8]])
    _, _, Y = oracle.projections(A1)
    np.testing.assert_allclose(-Y.dot(np.array([-16.])),
                               [0.24615385, 1.96923077], atol=1e-8)


@pytest.mark.parametrize("method", ["AugmentedSystem", "QRFactorization",
                                    "SVDFactorization", "NormalEquation"])
def test_projections_3x8(qp_small, method):
    A38 = np.array(cs.A38, dtype=float)
    sparse = method in ("AugmentedSystem", "NormalEquation")
    A = sps.csc_matrix(A38) if sparse else A38
    Z, LS, Y = oracle.projections(A, method)
    # NormalEquation has no reference run here (no scikit-sparse): it is held
    # to the AugmentedSystem outputs -- same operators, test_projections.py
    # pins only the operators.
    want = qp_small["proj38"][method if method != "NormalEquation"
                              else "AugmentedSystem"]
    for p, wz, wl in zip(cs.A38_POINTS_N, want["Z"], want["LS"]):
        p = np.array(p, float)
        close_projection(Z.dot(p), unjson(wz), p, 1e-12)
        close(LS.matvec(p), unjson(wl), 1e-10)
        assert np.max(np.abs(A38.dot(Z.dot(p)))) < 1e-8
    for p, wy in zip(cs.A38_POINTS_M, want["Y"]):
        close(Y.dot(np.array(p, float)), unjson(wy), 1e-11)


def test_orthogonality(qp_small):
    A38 = np.array(cs.A38, dtype=float)
    for v, want in zip(cs.ORTH_VECTORS, qp_small["orth"]):
        got = oracle.orthogonality(A38, np.array(v))
        assert abs(got - want) < 1e-15
        assert abs(oracle.orthogonality(sps.csc_matrix(A38), np.array(v))
                   - want) < 1e-15


@pytest.mark.parametrize("key", ["diag4", "diag3"])
def test_dense_vs_sparse(qp_small, key):
    A = cs.diag4_matrix() if key == "diag4" else cs.diag3_matrix()
    m, n = A.shape
    rng = np.random.RandomState(0)
    Zs, LSs, Ys = oracle.projections(sps.csc_matrix(A))
    Zd, LSd, Yd = oracle.projections(A)
    want = qp_small[key]
    for k in range(3):
        z, x = rng.normal(size=n), rng.normal(size=m)
        close(Zs.dot(z), want["Z_sparse"][k], 1e-11)
        close(Zd.dot(z), want["Z_dense"][k], 1e-11)
        close(LSs.dot(z), want["LS_sparse"][k], 1e-11)
        close(LSd.dot(z), want["LS_dense"][k], 1e-11)
        close(Ys.dot(x), want["Y_sparse"][k], 1e-11)
        close(Yd.dot(x), want["Y_dense"][k], 1e-11)


def test_projection_errors():
    A38 = np.array(cs.A38, dtype=float)
    with pytest.raises(ValueError):
        oracle.projections(A38, "AugmentedSystem")
    with pytest.raises(ValueError):
        oracle.projections(sps.csc_matrix(A38), "QRFactorization")
    Z, LS, Y = oracle.projections(np.empty((0, 5)))
    np.testing.assert_array_equal(Z.dot(np.arange(5.0)), np.arange(5.0))


@pytest.mark.parametrize("size", ["n2000", "n20000"])
@pytest.mark.parametrize("method", ["AugmentedSystem", "NormalEquation"])
def test_banded_traces(size, method, banded2000, banded20000):
    gold = banded2000 if size == "n2000" else banded20000
    n, m = (2000, 200) if size == "n2000" else (20000, 2000)
    inst = BandedInstance(n, m)
    s = int(gold["stride"][0])
    Z, LS, Y = oracle.projections(inst.A, method)
    for p, w in zip(inst.probes_n, gold["Z"]):
        close(Z.dot(p)[::s], w, 1e-12)
    for p, w in zip(inst.probes_n, gold["LS"]):
        close(LS.dot(p)[::s], w, 1e-12)
    for p, w in zip(inst.probes_m, gold["Y"]):
        close(Y.dot(p)[::s], w, 1e-12)

    gnorm = float(gold["gnorm"][0])
    for name, kw in inst.pcg_variants(gnorm).items():
        x, info = oracle.projected_cg(inst.H, inst.c, Z, Y, np.zeros(m),
                                      return_all=True, **kw)
        want = gold["pcg_%s_info" % name]
        assert [info["niter"], info["stop_cond"],
                int(info["hits_boundary"])] == list(want), name
        close(x[::s], gold["pcg_%s_x" % name], 1e-10)
        for a, w in zip(info["allvecs"], gold["pcg_%s_allvecs" % name]):
            close(a[::s], w, 1e-10)

    y_b = Y.dot(inst.b)
    x, info = oracle.projected_cg(inst.H, inst.c, Z, Y, inst.b, tol=0,
                                  max_iter=10,
                                  trust_radius=10 * np.linalg.norm(y_b))
    assert [info["niter"], info["stop_cond"],
            int(info["hits_boundary"])] == list(gold["pcg_rowstart_info"])
    close(x[::s], gold["pcg_rowstart_x"], 1e-10)

    for (radius, lo, hi), w in zip(inst.dogleg_cfg(y_b), gold["dogleg"]):
        got = oracle.modified_dogleg(inst.A, Y, inst.b, radius,
                                     np.full(n, lo), np.full(n, hi))
        close(got[::s], w, 1e-10)


# ---- projection refinement (reference test_projections.py:48-65,141-156) -----
@pytest.mark.parametrize("method", ["AugmentedSystem", "NormalEquation", "QRFactorization",
                                    "SVDFactorization"])
def test_projections_refinement(qp_extra, method):
    A38 = np.array(cs.A38, dtype=float)
    sparse = method in ("AugmentedSystem", "NormalEquation")
    gold = qp_extra["proj38_refine"][method if method != "NormalEquation" else "AugmentedSystem"]
    Z, _, _ = oracle.projections(sps.csc_matrix(A38) if sparse else A38, method,
                                 orth_tol=1e-18, max_refin=gold["max_refin"])
    for i, (p, want) in enumerate(zip(cs.A38_POINTS_N, gold["Z"])):
        p = np.array(p, float)
        z = Z.dot(p)
        # the reference's own assertions
        assert np.max(np.abs(A38.dot(z))) < 1.5e-14 * max(1.0, np.max(np.abs(p)))
        assert oracle.orthogonality(A38, z) < 1.5e-16
        if i < 3:
            close(z, want, 1e-12)
        else:
            # p = row 3 of A + 1e-10 e_8: Z p is 1e-10-sized, what is left after cancelling
            # 16 digits of p; the reference's own three methods differ by 2e-5 on it
            close(z, want, 1e-4)
            assert np.max(np.abs(z - np.array(want))) <= 1e-15 * np.max(np.abs(p))


@pytest.mark.parametrize("name,kind", [("zero_row", "sparse"), ("zero_row", "dense"),
                                       ("sum_row", "sparse")])
def test_rank_deficient_fallback(qp_extra, name, kind):
    """Rank-deficient Jacobian: SVD fallback with the reference's warning
    (projections.py:101-108,181-187,236-287).  (sum_row / dense is not compared: there the
    reference's pivoted-QR rank test does not fire and its operators return 1e13-sized noise.)"""
    A = np.array(cs.RANK_DEFICIENT[name], dtype=float)
    gold = qp_extra["rank_deficient"][name][kind]
    assert len(gold["warnings"]) == 1 and gold["warnings"][0].startswith("Singular Jacobian")
    if name == "sum_row" and not np.linalg.svd(A, compute_uv=False)[-1] <= 1e-15:
        pytest.skip("this LAPACK leaves the rounding-level singular value above tol=1e-15")
    with pytest.warns(UserWarning, match="Singular Jacobian matrix"):
        Z, LS, Y = oracle.projections(sps.csc_matrix(A) if kind == "sparse" else A)
    for p, wz, wl in zip(cs.A38_POINTS_N[:3], gold["Z"], gold["LS"]):
        close(Z.dot(np.array(p, float)), wz, 1e-10)
        close(LS.dot(np.array(p, float)), wl, 1e-9)
    for p, wy in zip(cs.A38_POINTS_M, gold["Y"]):
        close(Y.dot(np.array(p, float)), wy, 1e-10)


@pytest.mark.parametrize("method", ["AugmentedSystem", "NormalEquation"])
@pytest.mark.parametrize("max_refin", [1, 3])
def test_banded_traces_with_refinement(banded_refine2000, method, max_refin):
    """projected_cg with projections that refine on every application."""
    gold = banded_refine2000
    n, m = 2000, 200
    inst = BandedInstance(n, m)
    Z, LS, Y = oracle.projections(inst.A, method, orth_tol=1e-30, max_refin=max_refin)
    for p, w in zip(inst.probes_n, gold["refine%d_Z" % max_refin]):
        close(Z.dot(p), w, 1e-12)
    for name, kw in inst.pcg_variants(1.0).items():
        if name not in ("free", "box"):
            continue
        x, info = oracle.projected_cg(inst.H, inst.c, Z, Y, np.zeros(m), **kw)
        assert [info["niter"], info["stop_cond"], int(info["hits_boundary"])] == \
            list(gold["refine%d_pcg_%s_info" % (max_refin, name)])
        close(x, gold["refine%d_pcg_%s_x" % (max_refin, name)], 1e-10)
