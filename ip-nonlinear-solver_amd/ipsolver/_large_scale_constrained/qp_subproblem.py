"""``ipsolver._large_scale_constrained.qp_subproblem`` of the reference ->
device implementations in ``ipsolver.qp``."""
from ..qp import *        # noqa: F401,F403
from ..qp import __all__  # noqa: F401
