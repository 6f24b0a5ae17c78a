"""``ipsolver._canonical_constraint`` of the reference: the same names on the implementations
of ``ipsolver.canonical``."""
from .canonical import (CanonicalConstraint, to_canonical, lagrangian_hessian,  # noqa: F401
                        empty_canonical_constraint, parse_constraint as _parse_constraint)

__all__ = ['CanonicalConstraint', 'to_canonical', 'lagrangian_hessian',
           'empty_canonical_constraint']
