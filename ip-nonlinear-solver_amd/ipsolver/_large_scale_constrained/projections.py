"""``ipsolver._large_scale_constrained.projections`` of the reference ->
device implementations in ``ipsolver.projector``."""
from ..projector import projections, orthogonality  # noqa: F401

__all__ = ['projections', 'orthogonality']
