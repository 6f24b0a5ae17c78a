"""User-facing constraint classes (host-side plumbing of the drop-in boundary).

Same constructors, ``kind`` grammar, feasibility rules and error messages as
the reference's ``ipsolver/_constraints.py`` (cited per function); nothing
here is on the GPU hot path -- these objects only wrap the user's callbacks
and hand their values to the canonical form (``canonical.py``).
"""
from warnings import warn

import numpy as np
import scipy.sparse as sps

from .fd import FiniteDifferenceOperator, FD_METHODS

__all__ = ['NonlinearConstraint', 'LinearConstraint', 'BoxConstraint']

_INFEASIBLE = ("Unfeasible initial point. Either set ``enforce_feasibility=False`` "
               "or choose a new feasible initial point ``x0``.")


def _is_operator(H):
    """LinearOperator-like: has .dot/.matvec but is neither ndarray nor sparse."""
    return (not sps.issparse(H) and not isinstance(H, (np.ndarray, list, tuple))
            and (hasattr(H, "matvec") or hasattr(H, "dot")) and hasattr(H, "shape"))


def wrap_hessian(hess, sample, nargs):
    """Normalise a user Hessian callback by the type of its first return value
    (reference _constraints.py:117-134, _minimize_constrained.py:395-412):
    sparse -> csr_matrix, operator -> as is, anything else -> 2-D ndarray."""
    if sps.issparse(sample):
        return lambda *a: sps.csr_matrix(hess(*a))
    if _is_operator(sample):
        return lambda *a: hess(*a)
    return lambda *a: np.atleast_2d(np.asarray(hess(*a)))


def check_kind(kind, m):
    """Validate / broadcast ``kind`` (reference _constraints.py:370-411)."""
    if not isinstance(kind, (tuple, list, str)):
        raise ValueError("The parameter `kind` should be a tuple,  a list, or a string.")
    if isinstance(kind, str):
        kind = (kind,)
    if len(kind) == 0:
        raise ValueError("The parameter `kind` should not be empty.")
    keyword, nargs = kind[0], len(kind)
    if keyword not in ("greater", "less", "equals", "interval"):
        raise ValueError("Keyword `%s` not available." % keyword)
    one_sided = keyword in ("greater", "less", "equals")
    if (nargs in (1, 2) and not one_sided) or (nargs == 3 and keyword != "interval"):
        raise ValueError("Invalid `kind` format.")
    if nargs == 1:
        kind = (keyword, 0)

    def bound(value, label):
        value = np.asarray(value, dtype=float)
        if np.size(value) not in (1, m):
            raise ValueError("`%s` has the wrong dimension." % label)
        # (np.resize of a scalar to 1e5 rows is a 5 ms concatenate loop)
        return np.full(m, float(value.reshape(-1)[0])) if np.size(value) == 1 \
            else np.array(value).reshape(m)

    if one_sided:
        label = {"greater": "lb", "less": "ub", "equals": "c"}[keyword]
        return (keyword, bound(kind[1], label))
    lb, ub = bound(kind[1], "lb"), bound(kind[2], "ub")
    if (lb > ub).any():
        raise ValueError("lb[i] > ub[i].")
    return (keyword, lb, ub)


def check_enforce_feasibility(flag, m):
    """Reference _constraints.py:414-426."""
    if isinstance(flag, bool):
        return np.full(m, flag, dtype=bool)
    flag = np.array(flag, dtype=bool)
    if flag.size != m:
        raise ValueError("The parameter 'enforce_feasibility' has the wrong "
                         "number of elements.")
    return flag


def kind_bounds(kind):
    """(lb, ub) arrays implied by a checked ``kind``."""
    keyword = kind[0]
    if keyword == "equals":
        return kind[1], kind[1]
    if keyword == "greater":
        return kind[1], np.full_like(kind[1], np.inf)
    if keyword == "less":
        return np.full_like(kind[1], -np.inf), kind[1]
    if keyword == "interval":
        return kind[1], kind[2]
    raise RuntimeError("Never be here.")


def is_feasible(kind, enforce, f0):
    """Only rows with enforce_feasibility are tested (reference :429-447)."""
    lb, ub = kind_bounds(kind)
    return bool((lb[enforce] <= f0[enforce]).all() and (f0[enforce] <= ub[enforce]).all())


def reinforce_box(kind, enforce, x0, rtol=0.01, atol=0.01):
    """Move enforced coordinates strictly inside their bounds (reference
    :450-477): margin min(atol, rtol*(ub-lb)) from each finite bound."""
    lb, ub = kind_bounds(kind)
    x = np.array(x0, dtype=float)
    for i in np.flatnonzero(enforce):
        if not np.isinf(lb[i]):
            x[i] = max(x[i], min(lb[i] + atol, lb[i] + rtol * (ub[i] - lb[i])))
        if not np.isinf(ub[i]):
            x[i] = min(x[i], max(ub[i] - atol, ub[i] - rtol * (ub[i] - lb[i])))
    return x


class _Initialised:
    """Fields every constraint exposes after ``evaluate_and_initialize``."""
    isinitialized = False

    def _finish(self, x0, f0):
        self.x0, self.f0 = x0, f0
        self.n, self.m = x0.size, f0.size
        self.kind = check_kind(self.kind, self.m)
        self.enforce_feasibility = check_enforce_feasibility(self.enforce_feasibility, self.m)


class NonlinearConstraint(_Initialised):
    """``lb <= fun(x) <= ub`` style constraint (reference _constraints.py:14-167).

    ``NonlinearConstraint(fun, kind, jac, hess='2-point', enforce_feasibility=False)``
    """

    def __init__(self, fun, kind, jac, hess='2-point', enforce_feasibility=False):
        self._fun, self._jac, self._hess = fun, jac, hess
        self.kind = kind
        self.enforce_feasibility = enforce_feasibility

    def evaluate_and_initialize(self, x0, sparse_jacobian=None):
        x0 = np.atleast_1d(x0).astype(float)
        f0 = np.atleast_1d(self._fun(x0))
        J0 = self._jac(x0)

        self.fun = lambda x: np.atleast_1d(self._fun(x))
        self.sparse_jacobian = bool(sparse_jacobian
                                    or (sparse_jacobian is None and sps.issparse(J0)))
        if self.sparse_jacobian:
            self.jac = lambda x: sps.csr_matrix(self._jac(x))
            self.J0 = sps.csr_matrix(J0)
        else:
            def dense_jac(x):
                J = self._jac(x)
                return J.toarray() if sps.issparse(J) else np.atleast_2d(J)
            self.jac = dense_jac
            self.J0 = J0.toarray() if sps.issparse(J0) else np.atleast_2d(J0)

        if callable(self._hess):
            self.hess = wrap_hessian(self._hess, self._hess(x0, np.zeros_like(f0)), 2)
        elif self._hess in FD_METHODS:
            method, jac = self._hess, self.jac

            def fd_hess(x, v):          # d/dx [J(x)' v] by differences (:136-146)
                return FiniteDifferenceOperator(lambda y: jac(y).T.dot(v), x, method)
            self.hess = fd_hess
        else:
            self.hess = self._hess
        self._finish(x0, f0)
        if not is_feasible(self.kind, self.enforce_feasibility, f0):
            raise ValueError(_INFEASIBLE)
        self.isinitialized = True
        return x0


class LinearConstraint(_Initialised):
    """``lb <= A x <= ub`` (reference _constraints.py:170-270)."""

    def __init__(self, A, kind, enforce_feasibility=False):
        self.A = A
        self.kind = kind
        self.enforce_feasibility = enforce_feasibility

    def evaluate_and_initialize(self, x0, sparse_jacobian=None):
        self.sparse_jacobian = bool(sparse_jacobian
                                    or (sparse_jacobian is None and sps.issparse(self.A)))
        if self.sparse_jacobian:
            self.A = sps.csr_matrix(self.A)
        else:
            self.A = self.A.toarray() if sps.issparse(self.A) else np.atleast_2d(self.A)
        x0 = np.atleast_1d(x0).astype(float)
        f0 = self.A.dot(x0)
        self.J0 = self.A
        self._finish(x0, f0)
        if not is_feasible(self.kind, self.enforce_feasibility, f0):
            raise ValueError(_INFEASIBLE)
        self.isinitialized = True
        return x0

    def to_nonlinear(self):
        if not self.isinitialized:
            raise RuntimeError("Trying to convert uninitialized constraint.")
        A = self.A
        nl = NonlinearConstraint(lambda x: A.dot(x), self.kind, lambda x: A, None,
                                 self.enforce_feasibility)
        nl.fun, nl.jac, nl.hess = nl._fun, nl._jac, None
        nl.isinitialized = True
        nl.constant_jac = True          # lets the solver keep one upload / factorization (N1)
        for name in ("m", "n", "sparse_jacobian", "x0", "f0", "J0"):
            setattr(nl, name, getattr(self, name))
        return nl


class BoxConstraint(_Initialised):
    """``lb <= x <= ub`` (reference _constraints.py:273-364)."""

    def __init__(self, kind, enforce_feasibility=False):
        self.kind = kind
        self.enforce_feasibility = enforce_feasibility

    def evaluate_and_initialize(self, x0, sparse_jacobian=None):
        x0 = np.atleast_1d(x0).astype(float)
        n = x0.size
        self.sparse_jacobian = bool(sparse_jacobian or sparse_jacobian is None)
        self.J0 = sps.eye(n).tocsr() if self.sparse_jacobian else np.eye(n)
        self._finish(x0, x0)
        self.isinitialized = True
        if not is_feasible(self.kind, self.enforce_feasibility, x0):
            warn("The initial point was changed in order to stay inside box constraints.")
            x0 = reinforce_box(self.kind, self.enforce_feasibility, x0)
            self.x0 = self.f0 = x0
        return x0

    def to_linear(self):
        if not self.isinitialized:
            raise RuntimeError("Trying to convert uninitialized constraint.")
        lin = LinearConstraint(self.J0, self.kind, self.enforce_feasibility)
        lin.isinitialized = True
        for name in ("m", "n", "sparse_jacobian", "x0", "f0", "J0"):
            setattr(lin, name, getattr(self, name))
        return lin

    def to_nonlinear(self):
        if not self.isinitialized:
            raise RuntimeError("Trying to convert uninitialized constraint.")
        return self.to_linear().to_nonlinear()
