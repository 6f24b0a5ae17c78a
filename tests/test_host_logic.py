"""Host logic of the drop-in boundary (driver, constraint classes, canonical
form, SQP / barrier control flow, counters, stopping rules) against the
reference's golden end-to-end traces -- on the CPU, by injecting the oracle's
numpy backend in place of the HIP one.  The HIP backend itself is checked in
tests/test_gpu_e2e.py against the same goldens."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sps

import ipsolver
from ipsolver import backend
import oracle.numpy_backend as npb
import problems
from banded_setup import load_synthetic
from conftest import unjson

TRACE_COLS = ("niter", "cg_niter", "trust_radius", "penalty", "barrier_parameter",
              "optimality", "constr_violation", "nfev")


def run(fun, x0, grad, hess, constraints, **kw):
    rows = []

    def cb(state):
        rows.append([int(state.niter), int(state.cg_niter), float(state.trust_radius),
                     float(state.penalty), float(getattr(state, "barrier_parameter", np.nan)),
                     float(state.optimality), float(state.constr_violation), int(state.nfev)])
        return False
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = ipsolver.minimize_constrained(fun, x0, grad, hess, constraints, callback=cb, **kw)
    return res, rows


EPS = np.finfo(float).eps


def compare(res, rows, gold, rtol=1e-10, prefix=None, x_rtol=1e-10, amplify=10.0):
    """The product's trace against the reference's, held to what the reference's OWN trace is
    determined to.  tests/golden/make_golden.py re-ran every reference trace with each
    component of the objective gradient moved by one unit in the last place (three seeded
    sign patterns) and recorded ``one_ulp``: the leading rows whose integer columns (niter,
    cg_niter, nfev) do not move -- the whole trace for the well-conditioned problems, a prefix
    where thousands of CG iterations or finite-difference quotients amplify the last bit
    (elec: 40 of 55 rows, the barrier run at n = 400: 23 of 86) -- and, per row and float
    column, how far the value moves.  On those rows:

      * integer columns: exact;
      * float columns: ``|got - want| <= rtol |want| + amplify x (the reference's own movement)
        + 256 eps x (largest value of the column)``, rtol = 1e-10 -- the last term is the
        rounding floor of quantities formed as differences of O(scale) numbers: optimality and
        constraint violation fall to 1e-9 of their first values while the iterates they are
        formed from stay O(1), and iterates that agree to 1e-14 (the bar is 1e-10) leave them
        an absolute difference of ~100 eps (measured: 2.2e-14 on a violation of 4e-6, README
        example, row 8);
      * when the whole trace is stable: counters, status, result keys exact and the final x to
        ``x_rtol`` + amplify x its own movement.

    ``prefix`` (optional) shortens the compared rows further (callers that stop a run early)."""
    want = np.array([[np.nan if isinstance(v, str) and v == "nan" else v for v in r]
                     for r in unjson(gold["trace"])], dtype=float)
    got = np.array(rows, dtype=float)
    ulp = gold.get("one_ulp")
    stable = len(want) if ulp is None else int(ulp["stable_rows"])
    sens = np.zeros((len(want), 8)) if ulp is None else \
        np.array(unjson(ulp["rows"]), dtype=float).reshape(stable, 8)
    whole = stable == len(want) and prefix is None
    k = min(stable, len(got)) if prefix is None else min(prefix, stable, len(got))
    assert k >= min(stable, 8), (k, stable, len(got))
    if whole:
        assert len(got) == len(want)
        for key in ("status", "niter", "cg_niter", "nfev", "ngev", "nhev", "ncev", "njev"):
            assert int(res[key]) == gold[key], key
        assert res.method == gold["method"]
        assert sorted(res.keys()) == gold["keys"]
    for col in (0, 1, 7):
        assert np.array_equal(got[:k, col], want[:k, col]), TRACE_COLS[col]
    for col in (2, 3, 4, 5, 6):
        a, b = got[:k, col], want[:k, col]
        ok = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), ok)
        if not ok.any():
            continue
        floor = 256 * EPS * np.max(np.abs(want[:, col][np.isfinite(want[:, col])]))
        err = np.abs(a[ok] - b[ok])
        bound = rtol * np.abs(b[ok]) + amplify * sens[:k, col][ok] + floor
        assert np.all(err <= bound), (TRACE_COLS[col], int(np.argmax(err / bound)),
                                      float(np.max(err / bound)))
    if whole:
        gx = np.asarray(unjson(gold["x"]), dtype=float)
        x = np.asarray(res.x)
        if x.size != gx.size:
            x = x[::max(1, x.size // 50)]
        x_move = 0.0 if ulp is None or ulp["x"] is None else float(ulp["x"])
        assert np.max(np.abs(x - gx)) <= (x_rtol + amplify * x_move) * np.max(np.abs(gx))


def compare_rows(rows, gold, x=None, rtol=1e-10, amplify=10.0, min_rows=8):
    """``compare`` for runs that only carry their trace rows (and optionally the final x): the
    multi-process tests.  Compares the rows the reference's own trace is stable on (golden
    ``one_ulp``; fewer when the run was stopped earlier) and returns how many."""
    want = np.array([[np.nan if isinstance(v, str) and v == "nan" else v for v in r]
                     for r in unjson(gold["trace"])], dtype=float)
    got = np.array(rows, dtype=float)
    ulp = gold["one_ulp"]
    stable = int(ulp["stable_rows"])
    sens = np.array(unjson(ulp["rows"]), dtype=float).reshape(stable, 8)
    k = min(stable, len(got))
    assert k >= min(stable, min_rows), (k, stable, len(got))
    for col in (0, 1, 7):
        assert np.array_equal(got[:k, col], want[:k, col]), TRACE_COLS[col]
    for col in (2, 3, 4, 5, 6):
        a, b = got[:k, col], want[:k, col]
        ok = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), ok)
        if not ok.any():
            continue
        floor = 256 * EPS * np.max(np.abs(want[:, col][np.isfinite(want[:, col])]))
        err = np.abs(a[ok] - b[ok])
        bound = rtol * np.abs(b[ok]) + amplify * sens[:k, col][ok] + floor
        assert np.all(err <= bound), (TRACE_COLS[col], int(np.argmax(err / bound)),
                                      float(np.max(err / bound)))
    if x is not None and k == len(want) and ulp["x"] is not None:
        gx = np.asarray(unjson(gold["x"]), dtype=float)
        xs = np.asarray(x)
        if xs.size != gx.size:
            xs = xs[::max(1, xs.size // 50)]
        assert np.max(np.abs(xs - gx)) <= (rtol + amplify * float(ulp["x"])) * np.max(np.abs(gx))
    return k


ALL = problems.exact_hessian_problems() + problems.fd_hessian_problems()


def trace_policy(name):
    """Keyword arguments of ``compare`` per golden trace.  The compared rows and bounds come
    from the trace's ``one_ulp`` record; one class needs more: forward-difference Hessians
    (``hess='2-point'``) divide the rounding of TWO gradient evaluations by h ~ 1.5e-8, so
    every H.p carries a relative noise of eps / h ~ 1e-8 that depends on the last bits of p --
    it does not show in the one-ulp record (the probe scales both evaluations alike) and any
    two implementations see it differently: those traces are held to 1e-6."""
    if name.endswith("_fd2"):
        return dict(rtol=1e-6)
    return {}


@pytest.mark.parametrize("prob", ALL, ids=[p.name for p in ALL])
def test_textbook_problems(prob, e2e_golden):
    with backend.use(npb):
        res, rows = run(prob.fun, prob.x0, prob.grad, prob.hess_arg(), prob.constraints(ipsolver))
    compare(res, rows, e2e_golden[prob.name], **trace_policy(prob.name))
    if prob.x_opt is not None:
        np.testing.assert_array_almost_equal(res.x, prob.x_opt, decimal=5)


def test_readme_example(e2e_golden):
    """BASELINE config 1: x = [1.9528219624212824, 0.0886559778265458]."""
    p = problems.HyperbolicIneq()
    with backend.use(npb):
        res, _ = run(p.fun, p.x0, p.grad, p.hess, p.constraints(ipsolver))
    assert res.status == 1 and res.niter == 21 and res.cg_niter == 23 and res.nfev == 14
    np.testing.assert_allclose(res.x, [1.9528219624212824, 0.0886559778265458], rtol=1e-9)
    assert set(("s", "barrier_parameter", "tolerance")) <= set(res.keys())


def test_constant_hessian_option_with_finite_difference_hessian():
    """``options={'constant_hessian': True}`` with a ``hess`` that is not callable: the
    option has nothing to keep and must not travel on into the outer loop's keyword arguments
    (``TypeError: unexpected keyword 'constant_hessian'``)."""
    p = problems.HyperbolicIneq()
    with backend.use(npb):
        res, _ = run(p.fun, p.x0, p.grad, "2-point", p.constraints(ipsolver),
                     options={"constant_hessian": True})
        ref, _ = run(p.fun, p.x0, p.grad, "2-point", p.constraints(ipsolver))
    assert res.status == ref.status and res.niter == ref.niter
    np.testing.assert_array_equal(res.x, ref.x)


def test_maratos_sqp_method_names(e2e_golden):
    p = problems.Maratos()
    for name in ("equality_constrained_sqp", "equality-constrained-sqp"):
        with backend.use(npb):
            res, rows = run(p.fun, p.x0, p.grad, p.hess, p.constraints(ipsolver), method=name)
        compare(res, rows, e2e_golden["maratos_sqp"])
    with pytest.raises(ValueError, match="Unknown optimization"):
        with backend.use(npb):
            ipsolver.minimize_constrained(p.fun, p.x0, p.grad, p.hess, p.constraints(ipsolver),
                                          method="newton")
    q = problems.IneqRosenbrock()
    with pytest.raises(ValueError, match="does not support inequality"):
        with backend.use(npb):
            ipsolver.minimize_constrained(q.fun, q.x0, q.grad, q.hess, q.constraints(ipsolver),
                                          method="equality_constrained_sqp")


@pytest.mark.parametrize("method", ["tr_interior_point", "equality_constrained_sqp"])
def test_banded_equality_nlp(method, e2e_golden):
    syn = load_synthetic()
    prob = syn.CenteredBandedNLP(2000, 200, eps=1e-3)
    with backend.use(npb):
        res, rows = run(prob.fun, prob.x0, prob.grad, prob.hess, prob.constraints(ipsolver),
                        method=method)
    compare(res, rows, e2e_golden["banded_eq_n2000_%s" % method])


def test_banded_box_inequality_nlp(e2e_golden):
    syn = load_synthetic()
    prob = syn.CenteredBandedNLP(400, 40, eps=1.0)
    cons = (prob.constraints(ipsolver, ("less", 0.0)),
            ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
    with backend.use(npb):
        res, rows = run(prob.fun, prob.x0, prob.grad, prob.hess, cons)
    gold = e2e_golden["banded_ineq_n400"]
    assert res.status == gold["status"]
    # Thousands of CG iterations on an ill-conditioned barrier problem amplify
    # last-bit differences until an accept/reject branch flips (hard
    # part 3): the reference's own trace keeps its integer columns for 23 of 86 rows under
    # one ulp in the gradient (golden ``one_ulp``); both runs end with status 1 at the same point.
    compare(res, rows, gold)
    gx = np.asarray(unjson(gold["x"]))
    assert np.allclose(np.asarray(res.x)[::max(1, 400 // 50)], gx, atol=1e-5)


def test_dense_nonlinear_equality_nlp():
    """Dense NONLINEAR equality constraints (synthetic.CenteredDenseNLP, n = 300, m = 60): the
    Jacobian changes at every accepted step, so every one of them refactors (the reference: a
    pivoted QR each, projections.py:179).  Host logic on the oracle's backend against the
    reference's trace (tests/golden/e2e_dense_nl.json, ``make_golden.py --dense-nl``)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "e2e_dense_nl.json")) as f:
        gold = json.load(f)["dense_nl_n300"]
    prob = load_synthetic().CenteredDenseNLP(300, 60)
    with backend.use(npb):
        res, rows = run(prob.fun, prob.x0, prob.grad, prob.hess, prob.constraints(ipsolver),
                        method="equality_constrained_sqp")
    compare(res, rows, gold)


def test_dense_equality_qp(e2e_golden):
    rng = np.random.default_rng(0)
    n, m = 60, 12
    A = rng.standard_normal((m, n))
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    Hd = G.dot(G.T) + np.eye(n)
    c = rng.standard_normal(n)
    bq = A.dot(rng.standard_normal(n))
    with backend.use(npb):
        res, rows = run(lambda x: 0.5 * x.dot(Hd.dot(x)) + c.dot(x), np.zeros(n),
                        lambda x: Hd.dot(x) + c, lambda x: Hd,
                        ipsolver.LinearConstraint(A, ("equals", bq)),
                        method="equality_constrained_sqp")
    # the reference ends by xtol on the merit-function noise floor (hard
    # part 4): under one ulp its own trace is stable for 13 of 26 rows (golden ``one_ulp``)
    compare(res, rows, e2e_golden["dense_eq_qp_n60"])


def test_sphere_intersection_with_a_radius_whose_square_overflows():
    """qp_subproblem.py:99-149 squares the radius; for a finite radius beyond 1.3e154 -- 1e300 is
    a common way to say "no trust region" -- the reference's ``trust_radius**2`` raises
    OverflowError as soon as a box event calls the routine.  The product treats such a sphere
    as the infinite one (no double-precision step reaches it)."""
    from ipsolver import qp
    assert qp._sphere_from_scalars(2.0, 0.3, 1.0, 1e300, False) == (0, 1, True)
    assert qp._sphere_from_scalars(2.0, 0.3, 1.0, 1e300, True) == (-np.inf, np.inf, True)
    assert qp._sphere_from_scalars(2.0, 0.3, 1.0, np.inf, False) == (0, 1, True)
    ta, tb, hit = qp._sphere_from_scalars(1.0, 0.0, 0.25, 1.0, False)      # z = 0.5, d = 1
    assert hit and ta == 0 and abs(tb - np.sqrt(0.75)) < 1e-15


def test_return_all_and_callback_stop():
    p = problems.HyperbolicIneq()
    with backend.use(npb):
        res = ipsolver.minimize_constrained(p.fun, p.x0, p.grad, p.hess, p.constraints(ipsolver),
                                            options={"return_all": True})
        assert len(res.allvecs) == len(res.allslack) == len(res.allmult)
        assert res.allvecs[0].shape == (2,) and res.allslack[0].shape == (3,)
        stop = ipsolver.minimize_constrained(p.fun, p.x0, p.grad, p.hess,
                                             p.constraints(ipsolver),
                                             callback=lambda st: st.niter >= 3)
    assert stop.status == 3 and stop.message.startswith("`callback`")


def test_constant_jacobian_is_flagged():
    """Linear and box constraints canonicalise to a Jacobian that is the same
    pair of matrices for every x (the solver then factors it once: SURVEY.md
    section 8(f) N1); a nonlinear constraint in the mix clears the flag."""
    from ipsolver.canonical import to_canonical
    x0 = np.array([0.5, 0.5, 0.5])
    lin = ipsolver.LinearConstraint(np.array([[1.0, 1.0, 1.0], [1.0, -1.0, 0.0]]),
                                    ("interval", [0, -1], [2, np.inf]))
    box = ipsolver.BoxConstraint(("greater", 0.0))
    nl = ipsolver.NonlinearConstraint(lambda x: np.array([x.dot(x)]), ("less", 4.0),
                                      lambda x: 2 * x[None, :], lambda x, v: 2 * v[0] * np.eye(3))
    for c in (lin, box, nl):
        c.evaluate_and_initialize(x0)
    both = to_canonical([lin, box])
    assert both.constant_jac
    J1, J2 = both.jac(x0), both.jac(x0 + 1.0)
    assert J1[0] is J2[0] and J1[1] is J2[1]
    dense = lambda M: M.toarray() if sps.issparse(M) else np.asarray(M)
    assert np.array_equal(dense(J1[0]), dense(both.J_ineq0))
    mixed = to_canonical([lin, nl])
    assert not mixed.constant_jac
    assert not to_canonical(nl).constant_jac


def _pattern(A):
    class P:
        pass
    p = P()
    A = sps.csr_matrix(A)
    A.sort_indices()
    p.shape, p.indptr_h, p.indices_h, p.nnz = A.shape, A.indptr, A.indices, A.nnz
    return p


def test_half_bandwidth_of_aat_from_the_columns():
    """projector.half_bandwidth_of_aat reads the half bandwidth of A A' off A's columns (widest
    last row - first row) instead of forming the product pattern (reference: projections.py
    factors A A' / the augmented system whatever its band).  Against the product, on random
    patterns incl. empty rows and columns, and on the barrier problem's augmented Jacobian."""
    from ipsolver.projector import half_bandwidth_of_aat
    rng = np.random.default_rng(3)
    mats = [sps.random(int(rng.integers(1, 50)), int(rng.integers(1, 70)),
                       density=float(rng.uniform(0.01, 0.3)), format="csr", random_state=k)
            for k in range(30)]
    n, m = 300, 30
    J = sps.csr_matrix((np.ones(15 * m), (np.repeat(np.arange(m), 15),
                                           np.minimum((9 * np.arange(m))[:, None].repeat(15, 1).ravel()
                                                      + np.tile(np.arange(15), m), n - 1))),
                       shape=(m, n))
    I = sps.eye(n, format="csr")
    mats.append(sps.bmat([[J, sps.eye(m), None, None], [-I, None, sps.eye(n), None],
                          [I, None, None, sps.eye(n)]], format="csr"))
    mats.append(sps.csr_matrix((4, 5)))
    for A in mats:
        B = sps.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=A.shape)
        S = (B @ B.T).tocoo()
        want = int(np.max(np.abs(S.row - S.col))) if S.nnz else 0
        assert half_bandwidth_of_aat(_pattern(A)) == want


def test_box_row_analysis_groups():
    """boxschur.BoxRowAnalysis on the barrier problem's augmented Jacobian [J S; -I S_l; I S_u]:
    every bound row is simple, grouped by its variable (lower row before upper), the general
    rows are J's; ragged bounds give single-row groups; a column shared by three simple rows is
    left to the general rows."""
    from ipsolver.boxschur import BoxRowAnalysis
    n, m = 40, 4
    rng = np.random.default_rng(0)
    J = sps.random(m, n, density=0.4, format="csr", random_state=1)
    J.data[:] = 1.0
    I = sps.eye(n, format="csr")
    L, U = np.arange(n), np.arange(0, n, 2)
    A = sps.bmat([[J, sps.eye(m), None, None], [-I[L], None, sps.eye(len(L)), None],
                  [I[U], None, None, sps.eye(len(U))]], format="csr")
    an = BoxRowAnalysis(_pattern(A))
    touched = np.unique(J.indices)                    # columns shared with a row of J
    # a variable with ONE bound row that no row of J touches: its column is private to that
    # row, which then has no shared entry -- a general row
    lone = [j for j in range(n) if j % 2 == 1 and j not in touched]
    assert list(an.general) == list(range(m)) + [m + j for j in lone]
    assert an.n_simple == len(L) + len(U) - len(lone)
    cols = {int(c): (int(p), int(q)) for c, p, q in zip(an.col, an.rowp, an.rowq)}
    assert sorted(cols) == [j for j in range(n) if j not in lone]
    for j, (p, q) in cols.items():
        assert p == m + j                             # the lower-bound row comes first
        assert q == (m + len(L) + j // 2 if j % 2 == 0 else -1)
    # three simple rows on one column: not a group
    A3 = sps.vstack([A, sps.csr_matrix(([2.0], ([0], [0])), shape=(1, A.shape[1]))], format="csr")
    an3 = BoxRowAnalysis(_pattern(A3))
    assert 0 not in set(an3.col.tolist()) and {m, m + len(L), A3.shape[0] - 1} <= set(an3.general.tolist())
