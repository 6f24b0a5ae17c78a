"""Import-path compatibility with the reference's subpackage
(``ipsolver._large_scale_constrained``): thin re-exports of the device
implementations."""
from ..barrier import tr_interior_point
from ..sqp import equality_constrained_sqp

__all__ = ['tr_interior_point', 'equality_constrained_sqp']
