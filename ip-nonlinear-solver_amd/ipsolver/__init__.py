"""MI355X-native drop-in for the trust-region subproblem path of
antonior92/ip-nonlinear-solver (``ipsolver``).

Public names mirror the reference (ipsolver/__init__.py:3-6).  Importing the
package needs no GPU; solving does -- the only arithmetic backend shipped is
``backend_hip`` (hand-written HIP kernels behind ``libipx.so``), and it fails
loudly when the library or a HIP device is missing.
"""
from .minimize import minimize_constrained
from .constraints import NonlinearConstraint, LinearConstraint, BoxConstraint

__all__ = ['minimize_constrained', 'NonlinearConstraint', 'LinearConstraint',
           'BoxConstraint']
