"""``ipsolver._numdiff.approx_derivative`` of the reference, in the one mode the solver path
reaches (``as_linear_operator=True``: ``hess='2-point'|'3-point'|'cs'``, reference
_numdiff.py:342-441).  The dense and sparse finite-difference Jacobians of the vendored scipy
routine are outside the hot path: asking for them raises."""
import numpy as np

from .fd import FiniteDifferenceOperator, FD_METHODS

__all__ = ['approx_derivative']


def approx_derivative(fun, x0, method='3-point', rel_step=None, f0=None,
                      bounds=(-np.inf, np.inf), sparsity=None, as_linear_operator=False,
                      args=(), kwargs={}):
    if method not in FD_METHODS:
        raise ValueError("Unknown method '%s'. " % method)
    x0 = np.atleast_1d(x0)
    if x0.ndim > 1:
        raise ValueError("`x0` must have at most 1 dimension.")
    lb, ub = (np.resize(np.asarray(b, dtype=float), x0.shape) for b in bounds)
    if as_linear_operator and not (np.all(np.isinf(lb)) and np.all(np.isinf(ub))):
        raise ValueError("Bounds not supported when `as_linear_operator` is True.")
    if not as_linear_operator:
        raise NotImplementedError(
            "approx_derivative: only as_linear_operator=True is provided (the mode the solver "
            "uses for hess='2-point'|'3-point'|'cs'); dense / sparse finite-difference "
            "Jacobians are scipy.optimize._numdiff.approx_derivative")
    f = (lambda x: fun(x, *args, **kwargs)) if (args or kwargs) else fun
    if f0 is not None and np.atleast_1d(f0).ndim > 1:
        raise ValueError("`f0` passed has more than 1 dimension.")
    op = FiniteDifferenceOperator(f, x0, method, rel_step, f0)
    if op.f0.ndim > 1:
        raise RuntimeError("`fun` return value has more than 1 dimension.")
    return op
