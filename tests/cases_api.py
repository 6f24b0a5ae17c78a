"""Cases for the host-side API either side of the hot path: constraint classes, the canonical
form, finite-difference operators (reference ipsolver/_constraints.py, _canonical_constraint.py,
_numdiff.py).  ``run(api)`` evaluates every case with the classes / functions of ``api`` and
returns plain lists and numbers; ``tests/golden/make_golden.py --api`` runs it on the REFERENCE
(-> tests/golden/api.json), ``tests/test_constraints_api.py`` on the product, and compares.

The situations are those the reference's own tests exercise (ipsolver/tests/test_constraints.py,
test_canonical_constraint.py, test__numdiff.py: kind grammar and its errors, feasibility
enforcement, box -> linear -> nonlinear conversions, row selection / re-signing of every kind,
concatenation under every mix of sparse / dense Jacobians, multiplier re-signing in the
Hessian, operator-mode differences) plus ragged variants of each (infinite bounds in every
position, scalar broadcasts, empty selections).  Expected values are never written here: they
are whatever the reference returns.
"""
import warnings
from copy import deepcopy
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sps

INF = np.inf


def reference_api():
    """The reference's names (build container only)."""
    import ipsolver._constraints as c
    import ipsolver._canonical_constraint as k
    import ipsolver._numdiff as nd
    return SimpleNamespace(NonlinearConstraint=c.NonlinearConstraint,
                           LinearConstraint=c.LinearConstraint, BoxConstraint=c.BoxConstraint,
                           check_kind=c._check_kind,
                           check_enforce_feasibility=c._check_enforce_feasibility,
                           reinforce_box=c._reinforce_box_constraint,
                           parse_constraint=k._parse_constraint, to_canonical=k.to_canonical,
                           empty_canonical_constraint=k.empty_canonical_constraint,
                           approx_derivative=nd.approx_derivative)


def _plain(v):
    """JSON-able: arrays -> nested lists (inf / nan as strings), matrices densified."""
    if v is None or isinstance(v, (bool, str)):
        return v
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        return float(v) if np.isfinite(v) else repr(float(v))
    if sps.issparse(v):
        return _plain(v.toarray())
    if isinstance(v, np.matrix):
        return _plain(np.asarray(v))
    if isinstance(v, np.ndarray):
        if v.dtype == bool:
            return [bool(t) for t in v.ravel()] if v.ndim == 1 else [_plain(r) for r in v]
        return [_plain(t) for t in v]
    if isinstance(v, (list, tuple)):
        return [_plain(t) for t in v]
    if isinstance(v, dict):
        return {k: _plain(t) for k, t in v.items()}
    raise TypeError(type(v))


def _guard(f):
    """Value of f(), or the name of the exception it raises."""
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return _plain(f())
    except (ValueError, RuntimeError) as e:
        return {"raises": type(e).__name__}


# ---------------------------------------------------------------- kind grammar
KINDS = [
    (1, "bla"), (1, []), (1, 7), (3, ["interval", [1, 2, 3]]),
    (3, ["interval", [1, 2, 3], [1, 2]]), (3, ["interval", [1, 2, 3], [1, 2, 1]]),
    (3, "greater"), (3, "less"), (3, "equals"), (3, ("greater", 1)), (3, ("less", [1, 2, 3])),
    (3, ("equals", [4, 5, 6])), (2, ("interval", -1, 1)), (2, ("interval", [-INF, 0], 5)),
    (3, ("interval", [1, 2, 3], [1, INF, 3])), (3, ("greater", [1, 2])), (3, ("less", 1, 2)),
    (3, ("equals", 1, 2)), (3, ("interval",)), (3, ("between", 0, 1)), (1, ("greater", -INF)),
    (4, ("interval", 0, [1, 2, 3, 4])),
]

ENFORCE = [([True, True], 3), (True, 3), (False, 2), ([True, False, True], 3), ([False], 1)]

REINFORCE = [
    (("interval", [0, 20, 30], [0.5, INF, 70]), [True, False, True], [1, 2, 3]),
    (("interval", [0, 20, 30], [0.5, INF, 70]), [True, True, True], [1, 2, 3]),
    (("interval", [-INF, -1, 0], [0, 1, 1e-3]), [True, True, True], [5, -5, 0.5]),
    (("greater", [0, 0, 0]), [True, True, False], [-1, 3, -2]),
    (("less", [0, 0, 0]), [True, True, False], [1, -3, 2]),
    (("interval", [0, 0], [1000, 1e-6]), [True, True], [-1, -1]),
]

PARSE = [
    ("equals", [10, 20, 30]), ("greater", [10, 20, 30]), ("greater", [10, -INF, 30]),
    ("less", [10, 20, 30]), ("less", [10, INF, 30]), ("interval", [10, 20, 30], [50, 60, 70]),
    ("interval", [10, 20, 30], [50, 20, 70]), ("interval", [10, 20, 30], [50, 20, INF]),
    ("interval", [-INF, 20, 30], [50, 20, INF]), ("interval", [-INF, -INF], [INF, INF]),
    ("interval", [1, 1, 1], [1, 1, 1]), ("greater", [-INF, -INF]), ("equals", [0.0]),
]


def _checked(api, kind, m):
    k = api.check_kind(kind, m)
    return [k[0]] + [np.asarray(b, dtype=float) for b in k[1:]]


# ---------------------------------------------------------------- constraint objects
A34 = np.array([[1, 2, 3, 4], [5, 0, 0, 6], [7, 0, 8, 0]], dtype=float)
F = (10.0, 1.0, 12.0)
G = (np.array([1.0, 2, 3, 4]), np.array([1.0, 1, 1, 1]), np.array([1.0, 0, 0, 1]))
HS = (np.eye(4), np.zeros((4, 4)), np.diag([1.0, 2, 3, 4]))


def quad_fun(x):
    return np.array([f + g.dot(x) + 0.5 * H.dot(x).dot(x) for f, g, H in zip(F, G, HS)])


def quad_jac(x):
    return np.vstack([g + H.dot(x) for g, H in zip(G, HS)])


def quad_jac_sparse(x):
    return sps.csr_matrix(quad_jac(x))


def quad_hess(x, v):
    return v[0] * HS[0] + v[1] * HS[1] + v[2] * HS[2]


def quad_hess_sparse(x, v):
    return sps.csr_matrix(quad_hess(x, v))


def _canonical_record(canonical, xs, multipliers, rng):
    rec = {"n_eq": canonical.n_eq, "n_ineq": canonical.n_ineq, "n_vars": canonical.n_vars,
           "sparse_jacobian": bool(canonical.sparse_jacobian),
           "enforce_feasibility": np.asarray(canonical.enforce_feasibility, dtype=bool),
           "c_ineq0": canonical.c_ineq0, "c_eq0": canonical.c_eq0,
           "J_ineq0": canonical.J_ineq0, "J_eq0": canonical.J_eq0,
           "J_ineq0_sparse": bool(sps.issparse(canonical.J_ineq0)),
           "has_hess": canonical.hess is not None, "points": []}
    for x in xs:
        c_ineq, c_eq = canonical.constr(np.asarray(x, dtype=float))
        J_ineq, J_eq = canonical.jac(np.asarray(x, dtype=float))
        rec["points"].append({"c_ineq": c_ineq, "c_eq": c_eq, "J_ineq": J_ineq, "J_eq": J_eq})
    if canonical.hess is not None:
        rec["hess"] = []
        for v_eq, v_ineq in multipliers:
            H = canonical.hess(np.asarray(xs[0], dtype=float), np.asarray(v_eq, dtype=float),
                               np.asarray(v_ineq, dtype=float))
            ps = rng.uniform(-1, 1, size=(3, canonical.n_vars))
            rec["hess"].append([np.asarray(H.dot(p)).ravel() for p in ps])
    return rec


def _nonlinear(api, kind, enforce=False, sparse=False, hess=quad_hess):
    return api.NonlinearConstraint(quad_fun, kind, quad_jac_sparse if sparse else quad_jac,
                                   hess, enforce)


def _conversions(api):
    out = {}
    box = api.BoxConstraint(("interval", [10, 20, 30], [50, INF, 70]))
    x0 = box.evaluate_and_initialize(np.array([1, 2, 3]))
    lin = box.to_linear()
    nl = box.to_nonlinear()
    out["box"] = {"x0": x0, "A": lin.A, "A_sparse": bool(sps.issparse(lin.A)),
                  "fun": nl.fun(np.array([4.0, 5, 6])), "jac": nl.jac(np.array([4.0, 5, 6])),
                  "hess_is_none": nl.hess is None, "kind": _checked(api, nl.kind, 3),
                  "enforce": np.asarray(nl.enforce_feasibility, dtype=bool)}
    for label, A in (("dense", A34), ("sparse", sps.csr_matrix(A34)), ("coo", sps.coo_matrix(A34))):
        for flag in (None, True, False):
            lin = api.LinearConstraint(A, ("less",), [False, False, False])
            x0 = lin.evaluate_and_initialize(np.array([1, 2, 3, 4]), flag)
            nl = lin.to_nonlinear()
            x = np.array([0.5, -1, 2, 3])
            out["linear_%s_%s" % (label, flag)] = {
                "x0": x0, "fun": nl.fun(x), "jac": nl.jac(x),
                "jac_sparse": bool(sps.issparse(nl.jac(x))), "f0": lin.f0,
                "sparse_jacobian": bool(lin.sparse_jacobian)}
    return out


def _infeasible(api):
    out = {}
    lb, ub = np.array([0, 20, 30.0]), np.array([0.5, INF, 70])
    for name, enforce in (("some", [False, True, True]), ("all", True), ("none", False)):
        box = api.BoxConstraint(("interval", lb, ub), enforce)
        out["box_" + name] = _guard(lambda: box.evaluate_and_initialize(np.array([1, 2, 3])))
    for name, kind, enforce in (("less_all", ("less",), [True, True, True]),
                                ("less_none", ("less",), False),
                                ("greater_one", ("greater", 0), [False, True, False]),
                                ("equals", ("equals", [30, 29, 31]), [True, True, True]),
                                ("equals_off", ("equals", [30, 0, 31]), [True, True, True])):
        lin = api.LinearConstraint(A34, kind, enforce)
        out["linear_" + name] = _guard(lambda: lin.evaluate_and_initialize(np.array([1, 2, 3, 4])))
        nl = api.NonlinearConstraint(lambda x: A34.dot(x), kind, lambda x: A34, None, enforce)
        out["nonlinear_" + name] = _guard(
            lambda: nl.evaluate_and_initialize(np.array([1, 2, 3, 4])))
    return out


def _canonical_cases(api):
    rng = np.random.RandomState(3)
    out = {}
    x = [1, 2, 3]
    e = api.empty_canonical_constraint(x, 3)
    c_ineq, c_eq = e.constr(x)
    J_ineq, J_eq = e.jac(x)
    out["empty"] = {"n_eq": e.n_eq, "n_ineq": e.n_ineq, "c_ineq": c_ineq, "c_eq": c_eq,
                    "J_ineq_shape": list(J_ineq.shape), "J_eq_shape": list(J_eq.shape),
                    "hess_is_none": e.hess is None,
                    "enforce": np.asarray(e.enforce_feasibility, dtype=bool)}
    for flag in (None, False):
        e = api.empty_canonical_constraint(x, 3, flag)
        out["empty_%s" % flag] = {"sparse": bool(sps.issparse(e.jac(x)[0]))}

    xs3 = [[1, 2, 3], [11, 25, 69], [-1, 0.5, 100]]
    for name, kind, enforce in (("box", ("interval", [10, 20, 30], [50, INF, 70]), False),
                                ("box_ragged", ("interval", [-INF, 20, -INF], [INF, 20, 7]), False),
                                ("box_enforced", ("interval", [0, 0, 0], [5, 5, 5]),
                                 [True, False, True]),
                                ("box_greater", ("greater", [0, -INF, 1]), False)):
        for flag in (None, True, False):
            box = api.BoxConstraint(kind, enforce)
            box.evaluate_and_initialize(np.array([1, 2, 3]), flag)
            out["%s_%s" % (name, flag)] = _canonical_record(api.to_canonical(box), xs3, [], rng)

    xs4 = [[1, 2, 3, 4], [0.5, -1, 2, 0], [3, 3, 3, 3]]
    for name, kind in (("interval", ("interval", [10, 20, 30], [10, INF, 70])),
                       ("less", ("less", [100, INF, 100])), ("equals", ("equals", [1, 2, 3])),
                       ("greater", ("greater",))):
        for label, A in (("dense", A34), ("sparse", sps.csr_matrix(A34))):
            lin = api.LinearConstraint(A, kind, [False, False, False])
            lin.evaluate_and_initialize(np.array([1, 2, 3, 4]))
            out["linear_%s_%s" % (name, label)] = _canonical_record(api.to_canonical(lin), xs4,
                                                                    [], rng)

    mults = {
        "a": (("interval", [10, 20, 30], [10, INF, 70]),
              [([10], [5, 6, 3]), ([50], [4, -2, 30])]),
        "b": (("interval", [10, 20, 30], [20, 20, 70]),
              [([10], [5, 6, 3, 12]), ([50], [4, -2, 30, 2])]),
        "c": (("greater", [0, -INF, 5]), [([], [1, 2]), ([], [-3, 0.5])]),
        "d": (("equals", [1, 2, 3]), [([1, 2, 3], []), ([0, -1, 0], [])]),
        "e": (("less",), [([], [1, 2, 3])]),
    }
    for name, (kind, mv) in mults.items():
        for sparse in (False, True):
            nl = _nonlinear(api, kind, False, sparse, quad_hess_sparse if sparse else quad_hess)
            nl.evaluate_and_initialize(np.array([1, 2, 3, 4]))
            out["nonlinear_%s_%s" % (name, "sparse" if sparse else "dense")] = \
                _canonical_record(api.to_canonical(nl), xs4, mv, rng)

    # concatenation: linear + nonlinear + box + nonlinear under every Jacobian storage mix
    lin = api.LinearConstraint(A34, ("interval", [10, 20, 30], [10, INF, 70]), False)
    nl = _nonlinear(api, ("interval", [10, 20, 30], [10, INF, 70]))
    box = api.BoxConstraint(("interval", [10, 20, 30, -INF], [50, INF, 70, INF]), False)
    v_eq = [1, 2, 3]
    v_ineq = list(range(1, 15))
    for conf in ((None, None, None, None), (True, True, True, True), (False, False, False, False),
                 (False, False, True, False), (True, False, True, False),
                 (False, True, False, None)):
        parts = [deepcopy(lin), deepcopy(nl), deepcopy(box), deepcopy(nl)]
        xx = np.array([1, 2, 3, 4])
        for c, flag in zip(parts, conf):
            xx = c.evaluate_and_initialize(xx, flag)
        rec = _canonical_record(api.to_canonical(parts), xs4, [(v_eq, v_ineq)], rng)
        out["concat_" + "".join("N" if f is None else "TF"[not f] for f in conf)] = rec
    # a canonical constraint inside the list, and the errors of the concatenation
    parts = [deepcopy(lin), deepcopy(nl)]
    for c in parts:
        c.evaluate_and_initialize(np.array([1, 2, 3, 4]))
    inner = api.to_canonical(parts[0])
    out["concat_canonical_member"] = _canonical_record(api.to_canonical([inner, parts[1]]), xs4,
                                                       [([1, 2], [1, 2, 3, 4, 5, 6])], rng)
    out["errors"] = {
        "empty_list": _guard(lambda: api.to_canonical([])),
        "unknown_type": _guard(lambda: api.to_canonical([3.0])),
    }
    other = deepcopy(box)
    other.evaluate_and_initialize(np.array([1, 2, 3, 5]))
    out["errors"]["unmatching_x0"] = _guard(lambda: api.to_canonical([parts[0], other]).n_eq)
    small = api.BoxConstraint(("greater", 0), False)
    small.evaluate_and_initialize(np.array([1, 2, 3]))
    out["errors"]["unmatching_n"] = _guard(lambda: api.to_canonical([parts[0], small]).n_eq)
    return out


# ---------------------------------------------------------------- finite differences
def fd_scalar_scalar(x):
    return np.sinh(x)


def fd_scalar_vector(x):
    return np.array([x[0] ** 2, np.tan(x[0]), np.exp(x[0])])


def fd_vector_scalar(x):
    return np.sin(x[0] * x[1]) * np.log(x[0])


def fd_vector_vector(x):
    return np.array([x[0] * np.sin(x[1]), x[1] * np.cos(x[0]), x[0] ** 3 * x[1] ** -0.5])


FD = [("scalar_scalar", fd_scalar_scalar, 1.0), ("scalar_vector", fd_scalar_vector, 0.5),
      ("vector_scalar", fd_vector_scalar, np.array([100.0, -0.5])),
      ("vector_vector", fd_vector_vector, np.array([-100.0, 0.2]))]


def _fd_cases(api):
    out = {}
    rng = np.random.RandomState(1)
    for name, fun, x0 in FD:
        n = np.size(x0)
        ps = [rng.uniform(-10, 10, size=(n,)) for _ in range(6)] + [np.zeros(n)]
        for method in ("2-point", "3-point", "cs"):
            op = api.approx_derivative(fun, x0, method=method, as_linear_operator=True)
            out["%s_%s" % (name, method)] = {"shape": list(op.shape),
                                            "dot": [np.asarray(op.dot(p)).ravel() for p in ps]}
        op = api.approx_derivative(fun, x0, as_linear_operator=True)        # default method
        out[name + "_default"] = [np.asarray(op.dot(p)).ravel() for p in ps[:2]]
    out["bounds_refused"] = _guard(lambda: api.approx_derivative(
        fd_vector_vector, np.array([-100.0, 0.2]), method="2-point", bounds=(1, INF),
        as_linear_operator=True).shape)
    out["unknown_method"] = _guard(lambda: api.approx_derivative(
        fd_vector_vector, np.array([-100.0, 0.2]), method="5-point",
        as_linear_operator=True).shape)

    # the constraint classes' own use of it: hess='2-point' / '3-point' of a NonlinearConstraint
    def fun(x):
        return np.array([x[0] ** 2 + x[1] ** 3, 2 / x[0] + x[0] * x[1] ** 2])

    def jac(x):
        return np.array([[2 * x[0], 3 * x[1] ** 2],
                         [-2 / x[0] ** 2 + x[1] ** 2, 2 * x[0] * x[1]]])

    for method in ("2-point", "3-point", "cs"):
        nl = api.NonlinearConstraint(fun, ("equals"), jac, method)
        nl.evaluate_and_initialize([1, 2])
        rows = []
        for _ in range(4):
            v = rng.uniform(-5, 5, 2)
            x = rng.uniform(0.5, 5, 2)
            H = nl.hess(x, v)
            rows.append([np.asarray(H.dot(rng.uniform(-5, 5, 2))).ravel() for _ in range(3)])
        out["constraint_hess_" + method] = rows
    return out


def run(api):
    out = {}
    out["check_kind"] = [_guard(lambda: _checked(api, kind, m)) for m, kind in KINDS]
    out["check_enforce"] = [_guard(lambda: np.asarray(api.check_enforce_feasibility(f, m),
                                                      dtype=bool)) for f, m in ENFORCE]
    out["reinforce_box"] = [
        _guard(lambda: api.reinforce_box(_tuple_kind(api, kind, len(x0)),
                                         np.asarray(enforce, dtype=bool),
                                         np.asarray(x0, dtype=float)))
        for kind, enforce, x0 in REINFORCE]
    out["parse"] = []
    for kind in PARSE:
        eq, ineq, val_eq, val_ineq, sign, fun_len = api.parse_constraint(
            _tuple_kind(api, kind, len(np.atleast_1d(kind[1]))))
        out["parse"].append(_plain({"eq": np.asarray(eq, dtype=int), "ineq": np.asarray(ineq, dtype=int),
                                    "val_eq": np.asarray(val_eq, dtype=float),
                                    "val_ineq": np.asarray(val_ineq, dtype=float),
                                    "sign": np.asarray(sign, dtype=float), "fun_len": fun_len}))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out["conversions"] = _plain(_conversions(api))
        out["infeasible"] = _infeasible(api)
        out["canonical"] = _plain(_canonical_cases(api))
        out["fd"] = _plain(_fd_cases(api))
    return out


def _tuple_kind(api, kind, m):
    return tuple(api.check_kind(kind, m))
