"""``ipsolver._constraints`` of the reference: the same names (private helpers included) on the
implementations of ``ipsolver.constraints``, so that code -- and tests -- written against the
reference's module path import unchanged."""
from .constraints import (NonlinearConstraint, LinearConstraint, BoxConstraint,  # noqa: F401
                          check_kind as _check_kind,
                          check_enforce_feasibility as _check_enforce_feasibility,
                          is_feasible as _is_feasible,
                          reinforce_box as _reinforce_box_constraint)

__all__ = ['NonlinearConstraint', 'LinearConstraint', 'BoxConstraint']
