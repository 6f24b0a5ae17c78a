"""Oracle (test infrastructure): the Z / LS / Y projection operators on the CPU.

Follows ``ipsolver/_large_scale_constrained/projections.py`` of the reference.
For a full-row-rank ``A`` (m x n):

    Z x  = x - A'(AA')^-1 A x     null-space projection
    LS x = (AA')^-1 A x           least-squares multipliers
    Y x  = A'(AA')^-1 x           minimum-norm solution of A y = x

The reference reaches these through SuperLU on the augmented system
(projections.py:93-172), pivoted QR (:175-233), SVD (:236-287) or CHOLMOD
normal equations (:58-90, needs scikit-sparse which is absent here).  The
first three are restated on the same scipy entry points; ``NormalEquation``
is restated with a sparse LU of ``A A'`` standing in for CHOLMOD (any exact
SPD solve gives the same operator) so that the formulation the GPU build
follows -- including its refinement loop :69-78 -- has a CPU twin.
"""
import warnings

import numpy as np
import scipy.linalg
import scipy.sparse as sps
import scipy.sparse.linalg as spla


class Operator:
    """Minimal stand-in for scipy's LinearOperator (projections.py:402-404):
    ``shape``, ``dot`` and ``matvec``; no dtype probing at construction."""

    def __init__(self, shape, fn):
        self.shape = shape
        self._fn = fn

    def dot(self, x):
        return self._fn(np.asarray(x, dtype=float))

    matvec = dot


def orthogonality(A, g):
    """||A g|| / (||A||_F ||g||), 0 when either norm vanishes.

    Reference: projections.py:23-55 (the Frobenius norm is recomputed on
    every call there; same value).
    """
    norm_g = np.linalg.norm(g)
    norm_A = spla.norm(A, ord='fro') if sps.issparse(A) \
        else np.linalg.norm(A, ord='fro')
    if norm_g == 0 or norm_A == 0:
        return 0
    return np.linalg.norm(A.dot(g)) / (norm_A * norm_g)


def _refine(A, z, project_once, orth_tol, max_refin):
    """Refinement loop shared by the normal-equation style methods
    (projections.py:69-78, :198-210, :255-267): re-project while the
    orthogonality measure exceeds orth_tol, at most max_refin times."""
    k = 0
    while orthogonality(A, z) > orth_tol:
        if k >= max_refin:
            break
        z = z - A.T.dot(project_once(z))
        k += 1
    return z


def _normal_equation(A, m, n, orth_tol, max_refin, tol):
    """projections.py:58-90 with splu(AA') in place of cholesky_AAt."""
    A = sps.csr_matrix(A)
    solve = spla.splu(sps.csc_matrix(A.dot(A.T))).solve if m > 0 \
        else (lambda w: w)

    def apply_inv(x):
        return solve(A.dot(x))

    def null_space(x):
        z = x - A.T.dot(apply_inv(x))
        return _refine(A, z, apply_inv, orth_tol, max_refin)

    def least_squares(x):
        return apply_inv(x)

    def row_space(x):
        return A.T.dot(solve(x))

    return null_space, least_squares, row_space


def _augmented_system(A, m, n, orth_tol, max_refin, tol):
    """projections.py:93-172: LU of K = [[I, A'], [A, 0]]."""
    K = sps.csc_matrix(sps.bmat([[sps.eye(n), A.T], [A, None]]))
    try:
        solve = spla.factorized(K)
    except RuntimeError:
        warnings.warn("Singular Jacobian matrix. Using dense SVD "
                      "decomposition to perform the factorizations.")
        return _svd(A.toarray(), m, n, orth_tol, max_refin, tol)

    def null_space(x):
        rhs = np.hstack([x, np.zeros(m)])
        sol = solve(rhs)
        z = sol[:n]
        k = 0
        while orthogonality(A, z) > orth_tol:       # :126-139
            if k >= max_refin:
                break
            sol = sol + solve(rhs - K.dot(sol))
            z = sol[:n]
            k += 1
        return z

    def least_squares(x):
        return solve(np.hstack([x, np.zeros(m)]))[n:n + m]

    def row_space(x):
        return solve(np.hstack([np.zeros(n), x]))[:n]

    return null_space, least_squares, row_space


def _qr(A, m, n, orth_tol, max_refin, tol):
    """projections.py:175-233: A' P = Q R (pivoted, economic)."""
    Q, R, P = scipy.linalg.qr(A.T, pivoting=True, mode='economic')
    if np.linalg.norm(R[-1, :], np.inf) < tol:
        warnings.warn("Singular Jacobian matrix. Using SVD decomposition "
                      "to perform the factorizations.")
        return _svd(A, m, n, orth_tol, max_refin, tol)

    def apply_inv(x):                                # v = P R^-1 Q' x
        v = np.zeros(m)
        v[P] = scipy.linalg.solve_triangular(R, Q.T.dot(x), lower=False)
        return v

    def null_space(x):
        z = x - A.T.dot(apply_inv(x))
        return _refine(A, z, apply_inv, orth_tol, max_refin)

    def row_space(x):                                # Q R^-T P' x
        return Q.dot(scipy.linalg.solve_triangular(R, x[P], lower=False,
                                                   trans='T'))

    return null_space, apply_inv, row_space


def _svd(A, m, n, orth_tol, max_refin, tol):
    """projections.py:236-287: thin SVD, singular values <= tol dropped."""
    U, s, Vt = scipy.linalg.svd(A, full_matrices=False)
    keep = s > tol
    U, Vt, s = U[:, keep], Vt[keep, :], s[keep]

    def apply_inv(x):
        return U.dot(1 / s * Vt.dot(x))

    def null_space(x):
        z = x - A.T.dot(apply_inv(x))
        return _refine(A, z, apply_inv, orth_tol, max_refin)

    def row_space(x):
        return Vt.T.dot(1 / s * U.T.dot(x))

    return null_space, apply_inv, row_space


_SPARSE = {'NormalEquation': _normal_equation,
           'AugmentedSystem': _augmented_system}
_DENSE = {'QRFactorization': _qr, 'SVDFactorization': _svd}


def projections(A, method=None, orth_tol=1e-12, max_refin=3, tol=1e-15):
    """Return the operators ``Z, LS, Y`` for ``A``.

    Reference: projections.py:290-406.  Sparse default AugmentedSystem,
    dense default QRFactorization (:372-387); an empty matrix is forced to
    the sparse path (:368-369).
    """
    m, n = np.shape(A)
    if m * n == 0:
        A = sps.csc_matrix(A)

    if sps.issparse(A):
        method = method or "AugmentedSystem"
        if method not in _SPARSE:
            raise ValueError("Method not allowed for sparse matrix.")
        build = _SPARSE[method]
    else:
        method = method or "QRFactorization"
        if method not in _DENSE:
            raise ValueError("Method not allowed for dense array.")
        build = _DENSE[method]

    null_space, least_squares, row_space = build(A, m, n, orth_tol,
                                                 max_refin, tol)
    return (Operator((n, n), null_space),
            Operator((m, n), least_squares),
            Operator((n, m), row_space))
