"""CPU oracle for the trust-region subproblem path -- TEST INFRASTRUCTURE ONLY.

This package is a numpy/scipy restatement of the reference algorithm for the
hot path named in BASELINE.json (projected CG, modified dogleg, the
box/sphere intersection helpers, and the Z / LS / Y projection operators).
Every function cites the reference file:line it follows.

It is *not* part of the product.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the
checker / the reported CPU baseline -- never as the thing shipped.  The
product package (``ip-nonlinear-solver_amd/ipsolver``) never imports it and
fails loudly when its HIP library is missing.

Parity pin: the oracle is checked in ``tests/test_oracle_golden.py`` against
fixtures in ``tests/golden/`` that were produced by importing the reference
itself (``tests/golden/make_golden.py``), covering every known-answer case of
the reference's own hot-path tests plus seeded traces of the banded
benchmark problem.
"""

from .qp_subproblem import (sphere_intersections, box_intersections,  # noqa: F401
                            box_sphere_intersections, inside_box_boundaries,
                            reinforce_box_boundaries, modified_dogleg,
                            projected_cg, eqp_kktfact)
from .projections import projections, orthogonality  # noqa: F401
