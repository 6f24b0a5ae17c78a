"""Finite-difference Hessian-vector products (host side, user callbacks).

The reference vendors scipy's ``approx_derivative`` and adds an
``as_linear_operator`` mode (ipsolver/_numdiff.py:403-441); only that mode is
reachable from the solver path (``hess='2-point'|'3-point'|'cs'``), so only it
is provided.  Each product is one or two evaluations of a *user* callback on
the host, which is why SURVEY.md section 2 leaves it outside the accelerated
path: vectors cross to the host for the call and come back.

``DeviceFiniteDifferenceOperator`` is the same rule for device-callback mode
(SURVEY.md section 8(f) N4): the callback takes and returns CUDA tensors, the
perturbed point and the difference quotient are ipx kernels, nothing crosses
PCIe.
"""
import numpy as np

FD_METHODS = ('2-point', '3-point', 'cs')
_EPS = np.finfo(np.float64).eps
_REL_STEP = {'2-point': _EPS ** 0.5, '3-point': _EPS ** (1 / 3), 'cs': _EPS ** 0.5}


class FiniteDifferenceOperator:
    """J(x0).p approximated by differences of ``fun`` (shape (m, n), ``dot``)."""

    def __init__(self, fun, x0, method, rel_step=None, f0=None):
        if method not in FD_METHODS:
            raise ValueError("Unknown method '%s'. " % method)
        self.fun = lambda x: np.atleast_1d(fun(x))
        self.x0 = np.atleast_1d(np.asarray(x0, dtype=float))
        self.f0 = self.fun(self.x0) if f0 is None else np.atleast_1d(f0)
        self.h = _REL_STEP[method] if rel_step is None else rel_step
        self.method = method
        self.shape = (self.f0.size, self.x0.size)
        self.host_only = True      # tells the device backend to round-trip vectors

    def dot(self, p):
        p = np.asarray(p, dtype=float)
        if not p.any():
            return np.zeros(self.shape[0])
        norm_p = np.linalg.norm(p)
        if self.method == '2-point':          # _numdiff.py:408-415
            dx = self.h / norm_p
            return (self.fun(self.x0 + dx * p) - self.f0) / dx
        if self.method == '3-point':          # :417-427
            dx = 2 * self.h / norm_p
            f1 = self.fun(self.x0 - (dx / 2) * p)
            f2 = self.fun(self.x0 + (dx / 2) * p)
            return (f2 - f1) / dx
        dx = self.h / norm_p                  # 'cs' :429-437
        return self.fun(self.x0 + dx * p * 1.j).imag / dx

    matvec = dot


class DeviceFiniteDifferenceOperator:
    """The rule above with ``fun``: CUDA tensor -> CUDA tensor (or DVec) and
    device vectors throughout.  'cs' (_numdiff.py:429-437) hands the user's callback a
    COMPLEX CUDA tensor ``x0 + i dx p`` -- the callback must be analytic in torch's complex
    arithmetic, exactly the reference's requirement on numpy callbacks -- and takes the
    imaginary part of what it returns; the perturbed point and the quotient are two torch
    elementwise operations on data that never leaves the device (the ipx kernels are real)."""
    device_operator = True       # backend_hip.hessian_operator keeps it on the device

    def __init__(self, fun, x0, method, f0=None):
        from .device import DVec
        if method not in FD_METHODS:
            raise ValueError("Unknown method '%s'. " % method)
        self._DVec = DVec
        self._raw = fun
        self.fun = lambda x: self._vec(fun(x.t))
        self.x0 = x0
        self.f0 = self._vec(f0) if f0 is not None else self.fun(x0)
        self.h = _REL_STEP[method]
        self.method = method
        self.shape = (len(self.f0), len(x0))

    def _vec(self, v):
        return v if isinstance(v, self._DVec) else self._DVec(v)

    def dot(self, p):
        from . import device as dv
        norm_p = dv.norm(p)
        if norm_p == 0:
            return self._DVec.zeros(self.shape[0])
        if self.method == '2-point':          # _numdiff.py:408-415
            dx = self.h / norm_p
            return (self.fun(self.x0.add_scaled(p, dx)) - self.f0) * (1.0 / dx)
        if self.method == 'cs':               # :429-437
            import torch
            dx = self.h / norm_p
            try:
                f1 = self._raw(torch.complex(self.x0.t, p.t * dx))
            except Exception as exc:          # (a callback built on real-only kernels)
                raise TypeError("hess='cs': the callback failed on a complex CUDA tensor (%r); "
                                "complex-step differences need a callback that is analytic in "
                                "torch's complex arithmetic -- use '2-point' or '3-point'"
                                % (exc,)) from exc
            f1 = f1.t if isinstance(f1, self._DVec) else f1
            if not torch.is_complex(f1):
                raise TypeError("hess='cs': the callback returned a real tensor for a complex "
                                "argument (it must be analytic in complex arithmetic)")
            return self._DVec((f1.imag * (1.0 / dx)).contiguous())
        dx = 2 * self.h / norm_p              # '3-point' :417-427
        f1 = self.fun(self.x0.add_scaled(p, -(dx / 2)))
        f2 = self.fun(self.x0.add_scaled(p, dx / 2))
        return (f2 - f1) * (1.0 / dx)

    matvec = dot
