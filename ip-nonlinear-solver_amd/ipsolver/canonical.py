"""Canonical form ``c_ineq(x) <= 0, c_eq(x) = 0`` and the Lagrangian Hessian.

Host-side restatement of the reference's ``ipsolver/_canonical_constraint.py``
(same row ordering, sign conventions and multiplier re-signing -- they fix the
layout of every z-space vector the kernels see).  The
values produced here are numpy / scipy objects; the solver loops upload them
through their backend (``backend_hip``).  The Lagrangian Hessian is returned
as the *list of terms* of _canonical_constraint.py:119-139 so the device
backend can fuse them into one operator (``operators.DeviceHessian``).
"""
import numpy as np
import scipy.sparse as sps

from .constraints import NonlinearConstraint, LinearConstraint, BoxConstraint, kind_bounds

__all__ = ['CanonicalConstraint', 'to_canonical', 'lagrangian_hessian',
           'empty_canonical_constraint', 'HessianSum']

_EMPTY = np.empty(0)


class CanonicalConstraint:
    """Record of _canonical_constraint.py:14-46."""

    def __init__(self, n_vars, n_ineq, n_eq, constr, jac, hess, sparse_jacobian,
                 enforce_feasibility, x0, c_ineq0, c_eq0, J_ineq0, J_eq0, constant_jac=False):
        self.n_vars, self.n_ineq, self.n_eq = n_vars, n_ineq, n_eq
        self.constr, self.jac, self.hess = constr, jac, hess
        self.sparse_jacobian = sparse_jacobian
        self.enforce_feasibility = enforce_feasibility
        self.x0 = x0
        self.c_ineq0, self.c_eq0 = c_ineq0, c_eq0
        self.J_ineq0, self.J_eq0 = J_ineq0, J_eq0
        # True when jac(x) is the same pair of matrices for every x (linear and
        # box constraints): the solver then uploads / factors it once
        # (the reference recomputes, :81,225)
        self.constant_jac = constant_jac


class HessianSum:
    """``p -> sum_h h.dot(p)`` over ``terms`` in order (the matvec closures of
    _canonical_constraint.py:131-137 and :422-428); exposes the terms so a
    backend can fuse them."""

    def __init__(self, n, terms):
        self.shape = (n, n)
        self.terms = list(terms)

    def flat_terms(self):
        out = []
        for h in self.terms:
            out.extend(h.flat_terms() if isinstance(h, HessianSum) else [h])
        return out

    def dot(self, p):
        result = np.zeros_like(np.asarray(p, dtype=float))
        for h in self.terms:
            result += h.dot(p)
        return result

    matvec = dot


def lagrangian_hessian(constraint, hess):
    """Reference _canonical_constraint.py:119-139."""
    def lagr_hess(x, v_eq=_EMPTY, v_ineq=_EMPTY):
        terms = []
        if hess is not None:
            terms.append(hess(x))
        if constraint.hess is not None:
            terms.append(constraint.hess(x, v_eq, v_ineq))
        return HessianSum(len(x), terms)
    return lagr_hess


def empty_canonical_constraint(x0, n_vars, sparse_jacobian=None):
    """Reference _canonical_constraint.py:142-163."""
    if sparse_jacobian or sparse_jacobian is None:
        J = sps.csr_matrix(np.empty((0, n_vars)))
    else:
        J = np.empty((0, n_vars))
    return CanonicalConstraint(n_vars, 0, 0, lambda x: (_EMPTY, _EMPTY), lambda x: (J, J),
                               None, True, np.empty(0, dtype=bool), x0, _EMPTY, _EMPTY, J, J)


def parse_constraint(kind):
    """Index / sign tables of a checked ``kind`` (reference :283-360).

    Returns ``eq, ineq, val_eq, val_ineq, sign, fun_len``; for two-sided kinds
    every finite lower bound comes first (sign -1), then every finite upper
    bound (sign +1); ``lb == ub`` rows become equalities.
    """
    if kind[0] == "equals":
        c = np.asarray(kind[1], dtype=float)
        return (np.arange(len(c), dtype=int), np.empty(0, dtype=int), c, np.empty(0),
                np.empty(0), len(c))
    if kind[0] not in ("greater", "less", "interval"):
        raise RuntimeError("Never be here.")
    lb, ub = (np.asarray(b, dtype=float) for b in kind_bounds(kind))
    idx = np.arange(len(lb), dtype=int)
    has_lb, has_ub = ~np.isinf(lb), ~np.isinf(ub)
    is_eq = (lb == ub) & has_lb & has_ub
    lo, up = ~is_eq & has_lb, ~is_eq & has_ub
    ineq = np.hstack((idx[lo], idx[up]))
    val_ineq = np.hstack((lb[lo], ub[up]))
    sign = np.hstack((-np.ones(np.count_nonzero(lo)), np.ones(np.count_nonzero(up))))
    return idx[is_eq], ineq, lb[is_eq], val_ineq, sign, len(lb)


class _RowMap:
    """Row selection + sign flip of one constraint (reference :240-280)."""

    def __init__(self, kind, n_vars):
        (self.eq, self.ineq, self.val_eq, self.val_ineq,
         self.sign, self.fun_len) = parse_constraint(kind)
        self.n_eq, self.n_ineq, self.n_vars = len(self.eq), len(self.ineq), n_vars

    def values(self, c):
        c_eq = c[self.eq] - self.val_eq if self.n_eq > 0 else _EMPTY
        c_ineq = self.sign * (c[self.ineq] - self.val_ineq) if self.n_ineq > 0 else _EMPTY
        return c_ineq, c_eq

    def sparse_jac(self, J):
        empty = sps.csr_matrix(np.empty((0, self.n_vars)))
        J_eq = J[self.eq, :] if self.n_eq > 0 else empty
        J_ineq = sps.diags(self.sign).dot(J[self.ineq, :]).tocsr() if self.n_ineq > 0 else empty
        return J_ineq, J_eq

    def dense_jac(self, J):
        empty = np.empty((0, self.n_vars))
        J_eq = J[self.eq, :] if self.n_eq > 0 else empty
        J_ineq = J[self.ineq, :] * self.sign[:, None] if self.n_ineq > 0 else empty
        return J_ineq, J_eq

    def multipliers(self, v_eq, v_ineq):
        """Canonical multipliers back in the user's row order, re-signed
        (reference :210-218)."""
        v = np.zeros(self.fun_len)
        if len(v_eq) > 0:
            v[self.eq] += v_eq
        if len(v_ineq) > 0:
            up, lo = self.sign == 1, self.sign == -1
            v[self.ineq[up]] += v_ineq[up]
            v[self.ineq[lo]] -= v_ineq[lo]
        return v


def _nonlinear_to_canonical(nl):
    rows = _RowMap(nl.kind, nl.n)
    convert_jac = rows.sparse_jac if nl.sparse_jacobian else rows.dense_jac
    c_ineq0, c_eq0 = rows.values(nl.f0)
    J_ineq0, J_eq0 = convert_jac(nl.J0)
    if nl.hess is None:
        hess = None
    else:
        def hess(x, v_eq=_EMPTY, v_ineq=_EMPTY):
            return nl.hess(x, rows.multipliers(v_eq, v_ineq))
    enforce = nl.enforce_feasibility[rows.ineq] if rows.n_ineq else np.empty(0, dtype=bool)
    constant = bool(getattr(nl, "constant_jac", False))
    jac = (lambda x: (J_ineq0, J_eq0)) if constant else (lambda x: convert_jac(nl.jac(x)))
    return CanonicalConstraint(nl.n, rows.n_ineq, rows.n_eq,
                               lambda x: rows.values(nl.fun(x)), jac,
                               hess, nl.sparse_jacobian, enforce, nl.x0,
                               c_ineq0, c_eq0, J_ineq0, J_eq0, constant)


def _stack_values(pairs):
    return (np.hstack([p[0] for p in pairs]), np.hstack([p[1] for p in pairs]))


def _stack_sparse(pairs):
    return (sps.vstack([sps.csr_matrix(p[0]) for p in pairs], format="csr"),
            sps.vstack([sps.csr_matrix(p[1]) for p in pairs], format="csr"))


def _stack_dense(pairs):
    def dense(M):
        return M.toarray() if sps.issparse(M) else np.atleast_2d(M)
    return (np.vstack([dense(p[0]) for p in pairs]), np.vstack([dense(p[1]) for p in pairs]))


def _concatenate(parts):
    """Reference _canonical_constraint.py:363-438."""
    n_eq = sum(c.n_eq for c in parts)
    n_ineq = sum(c.n_ineq for c in parts)
    n_vars, x0 = parts[0].n_vars, parts[0].x0
    for c in parts:
        if c.n_vars != n_vars:
            raise RuntimeError("Unmatching constraint number of arguments.")
        if not np.array_equal(x0, c.x0):
            raise RuntimeError("Unmatching initial point.")
    use_sparse = bool(np.any([c.sparse_jacobian for c in parts]))
    stack_jac = _stack_sparse if use_sparse else _stack_dense

    def hess(x, v_eq=_EMPTY, v_ineq=_EMPTY):
        terms, i_eq, i_ineq = [], 0, 0
        for c in parts:
            if c.hess is not None:
                terms.append(c.hess(x, v_eq[i_eq:i_eq + c.n_eq],
                                    v_ineq[i_ineq:i_ineq + c.n_ineq]))
            i_eq += c.n_eq
            i_ineq += c.n_ineq
        return HessianSum(n_vars, terms)

    c_ineq0, c_eq0 = _stack_values([(c.c_ineq0, c.c_eq0) for c in parts])
    J_ineq0, J_eq0 = stack_jac([(c.J_ineq0, c.J_eq0) for c in parts])
    constant = all(c.constant_jac for c in parts)
    jac = (lambda x: (J_ineq0, J_eq0)) if constant \
        else (lambda x: stack_jac([c.jac(x) for c in parts]))
    return CanonicalConstraint(
        n_vars, n_ineq, n_eq,
        lambda x: _stack_values([c.constr(x) for c in parts]), jac,
        hess, use_sparse, np.hstack([c.enforce_feasibility for c in parts]),
        x0, c_ineq0, c_eq0, J_ineq0, J_eq0, constant)


def to_canonical(constraints):
    """Reference _canonical_constraint.py:49-81."""
    if isinstance(constraints, (NonlinearConstraint, LinearConstraint, BoxConstraint,
                                CanonicalConstraint)):
        constraints = [constraints]
    if not isinstance(constraints, (list, tuple)):
        raise ValueError("Unknown Constraint type.")
    parts = []
    for c in constraints:
        if isinstance(c, CanonicalConstraint):
            parts.append(c)
        elif isinstance(c, NonlinearConstraint):
            parts.append(_nonlinear_to_canonical(c))
        elif isinstance(c, LinearConstraint):
            parts.append(_nonlinear_to_canonical(c.to_nonlinear()))
        elif isinstance(c, BoxConstraint):
            parts.append(_nonlinear_to_canonical(c.to_linear().to_nonlinear()))
        else:
            raise ValueError("Unknown Constraint type.")
    if not parts:
        raise ValueError("Empty list.")
    return parts[0] if len(parts) == 1 else _concatenate(parts)
