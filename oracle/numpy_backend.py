"""Oracle (test infrastructure): numpy/scipy backend for the product's outer
loops (``ipsolver.sqp`` / ``ipsolver.barrier`` / ``ipsolver.minimize``).

Injected by tests only (``ipsolver.backend.use(...)``) so that the HOST logic
of the drop-in boundary -- canonicalisation, SQP and barrier control flow,
counters, stopping rules -- can be checked against the reference's golden
traces on a machine without a GPU.  The subproblem solvers it plugs in are
the oracle's own (``oracle.qp_subproblem`` / ``oracle.projections``).  Never
imported by the product.
"""
import numpy as np
import scipy.sparse as sps

from .projections import projections as _projections
from . import qp_subproblem as _qp

name = "numpy-oracle"


def asvec(a, space=None):
    return np.asarray(a, dtype=float)


def tohost(v):
    return np.asarray(v)


def zeros(n):
    return np.zeros(n)


def full(n, value, space=None):
    return np.full(n, float(value))


def copy(v):
    return np.copy(v)


def hstack(parts):
    return np.hstack(parts)


dot = np.dot
norm = np.linalg.norm


class pack:
    """Host twin of ipsolver.device.ScalarPack (values are immediate here)."""

    def __init__(self):
        self.vals = []

    def _add(self, v):
        self.vals.append(float(v))
        return len(self.vals) - 1

    def dot(self, a, b):
        return self._add(np.dot(a, b) if len(a) else 0.0)

    def norm(self, v):
        return self._add(np.linalg.norm(v) if len(v) else 0.0)

    def norm_inf(self, v):
        return self._add(np.linalg.norm(v, np.inf) if len(v) else 0.0)

    def read(self):
        return list(self.vals)


def norm_inf(v):
    return np.linalg.norm(v, np.inf)


def maximum(v, c):
    return np.maximum(v, c)


def where_positive(v, a, c):
    return np.where(v > 0, a, c)


def sum_log(s):
    return np.sum([np.log(t) if t > 0 else -np.inf for t in s]) if len(s) else 0.0


def assign_negated_where(s, mask, c):
    s[mask] = -c[mask]


def mark_constant(A):
    return A          # the host backend refactors every time, like the reference


def matrix(J, key=None):
    return J


class _Diag:
    def __init__(self, d):
        self.d = d

    def dot(self, x):
        return self.d * x


def diagonal_operator(d):
    return _Diag(d)


class _Hessian:
    """tr_interior_point.py:222-241 / _canonical_constraint.py:131-137."""

    def __init__(self, terms, n_vars, slack_block):
        self.terms, self.n_vars, self.slack = terms, n_vars, slack_block

    def dot(self, p):
        if self.slack is None:
            return self.terms.dot(p)
        return np.hstack((self.terms.dot(p[:self.n_vars]), self.slack * p[self.n_vars:]))


def hessian_operator(terms, n_vars, slack_block):
    return _Hessian(terms, n_vars, slack_block)


def augmented_jacobian(J_eq, J_ineq, s, n_vars, n_eq, n_ineq):
    """tr_interior_point.py:141-194."""
    if sps.issparse(J_eq) or sps.issparse(J_ineq):
        return sps.bmat([[sps.csr_matrix(J_eq), None],
                         [sps.csr_matrix(J_ineq), sps.diags(s)]], "csr")
    return np.asarray(np.bmat([[np.atleast_2d(J_eq).reshape(n_eq, n_vars),
                                np.zeros((n_eq, n_ineq))],
                               [np.atleast_2d(J_ineq).reshape(n_ineq, n_vars), np.diag(s)]]))


def projections(A, method=None):
    return _projections(A, method)


def _bounds(n, lb, ub):
    return (np.full(n, -np.inf) if lb is None else lb,
            np.full(n, np.inf) if ub is None else ub)


def modified_dogleg(A, Y, b, trust_radius, lb, ub, norm_out=None):
    lb, ub = _bounds(np.shape(A)[1], lb, ub)
    return _qp.modified_dogleg(A, Y, b, trust_radius, lb, ub)


def projected_cg(H, c, Z, Y, b, trust_radius, lb, ub):
    lb, ub = _bounds(len(c), lb, ub)
    if b is None:                      # the SQP's b_t = 0 (equality_constrained_sqp.py:126)
        b = np.zeros(Y.shape[1])
    return _qp.projected_cg(H, c, Z, Y, b, trust_radius, lb, ub)


box_intersections = _qp.box_intersections
