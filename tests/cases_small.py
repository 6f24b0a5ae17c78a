"""Input tables for the small known-answer cases of the hot path.

These are the *inputs* of the reference's own unit tests for
``qp_subproblem.py`` / ``projections.py`` (test_qp_subproblem.py:24-649,
test_projections.py:25-217), written as plain data.  The expected outputs
live in ``tests/golden/qp_small.json`` and were produced by running the
reference on these inputs (``tests/golden/make_golden.py``).  Shared by the
generator, the oracle tests and the GPU parity tests so all three see the
same numbers.
"""
import numpy as np

inf = np.inf

# (z, d, radius) -- each evaluated with entire_line False and True
SPHERE = [
    ([0, 0], [1, 0], 0.5),
    ([2, 0], [0, 1], 1),
    ([2, 0], [1, 0], 1),
    ([2, 0], [-1, 0], 1.5),
    ([2, 0], [1, 0], 2),
    ([0, 0], [0, 0], 1.0),            # degenerate direction
    ([1, 2, 3], [0.5, -1, 2], inf),   # infinite radius shortcut
]

# (z, d, lb, ub) -- each evaluated with entire_line False and True
BOX = [
    ([2, 0], [0, 2], [1, 1], [3, 3]),
    ([2, 0], [0, 2], [1, -3], [3, -1]),
    ([2, 0], [0, 2], [-inf, 1], [inf, inf]),
    ([1, 0], [0, 1], [1, 1], [3, 3]),
    ([0, 0], [4, 4], [-2, -3], [3, 2]),
    ([2, 0], [0, 2], [-3, -3], [-1, -1]),
    ([2, 0], [0, 2], [-3, 3], [-1, 1]),
    ([2, 0], [0, 2], [-3, -inf], [-1, inf]),
    ([0, 0], [1, 100], [1, 1], [3, 3]),
    ([0.99, 0], [0, 2], [1, 1], [3, 3]),
    ([2, 2], [0, 1], [-2, -2], [2, 2]),
    ([1, 1, 0], [0, 0, 1], [1, 1, 1], [3, 3, 3]),
    ([1, 1, 0], [0, 0, -1], [1, 1, 1], [3, 3, 3]),
    ([2, 2, 2], [0, -1, 1], [1, 1, 1], [3, 3, 3]),
    ([0, 0], [0, 0], [-1, -1], [1, 1]),          # degenerate direction
]

# (z, d, lb, ub, radius) -- each evaluated with entire_line False and True
BOX_SPHERE = [
    ([1, 1], [-2, 2], [-1, -2], [1, 2], 2),
    ([1, 1], [-1, 1], [-1, -3], [1, 3], 10),
    ([1, 1], [-4, 4], [-1, -3], [1, 3], 10),
    ([1, 1], [-4, 4], [-1, -3], [1, 3], 2),
    ([2, 2], [-4, 4], [-1, -3], [1, 3], 2),
    ([1, 1], [-4, 4], [2, 4], [2, 4], 2),
]

# modified_dogleg: (A, b, radius, lb, ub)
_A1 = [[1, 8]]
_A2 = [[1, 8, 1], [4, 2, 2]]
DOGLEG = [
    (_A1, [-16], 2, [-inf, -inf], [inf, inf]),
    (_A1, [-16], 1, [-inf, -inf], [inf, inf]),
    (_A1, [-16], 2, [-inf, -inf], [0.1, inf]),
    (_A2, [-16, 2], 3, [-inf] * 3, [inf] * 3),
    (_A2, [-16, 2], 2, [-inf] * 3, [inf] * 3),
    (_A2, [-16, 2], 5, [-1, -inf, -inf], [inf] * 3),
    (_A2, [-16, 2], 1, [-inf] * 3, [inf] * 3),
    (_A2, [-16, 2], 2, [-inf] * 3, [inf, 1, inf]),
]

# projected_cg: dict(H, A, c, b, kwargs); "raises" marks the ValueError cases
_H3 = [[6, 2, 1], [2, 5, 2], [1, 2, 4]]
_A3 = [[1, 0, 1], [0, 1, 1]]
_H4 = [[6, 2, 1, 3], [2, 5, 2, 4], [1, 2, 4, 5], [3, 4, 5, 7]]
_A4 = [[1, 0, 1, 0], [0, 1, 1, 1]]
_Hneg = [[1, 2, 1, 3], [2, 0, 2, 4], [1, 2, 0, 2], [3, 4, 2, 0]]
_Aneg = [[1, 0, 1, 0], [0, 1, 0, 1]]
_c4 = [-2, -3, -3, 1]
_b = [-3, 0]
PCG = [
    dict(name="nocedal_16_2", H=_H3, A=_A3, c=[-8, -3, -3], b=_b, kw={}),
    dict(name="vs_kkt", H=_H4, A=_A4, c=_c4, b=_b, kw=dict(tol=0)),
    dict(name="tr_infeasible", H=_H4, A=_A4, c=_c4, b=_b,
         kw=dict(trust_radius=1), raises=True),
    dict(name="tr_barely_feasible", H=_H4, A=_A4, c=_c4, b=_b,
         kw=dict(tol=0, trust_radius=2.32379000772445021283),
         # ||x0|| equals the radius to the last bit: whether the loop runs 0 or
         # 1 iterations depends on the rounding of the norm; the reference's
         # test (test_qp_subproblem.py:474-491) pins only what is compared here
         knife_edge=True),
    dict(name="hits_boundary", H=_H4, A=_A4, c=_c4, b=_b,
         kw=dict(tol=0, trust_radius=3)),
    dict(name="negcurv_unconstrained", H=_Hneg, A=_Aneg, c=_c4, b=_b,
         kw=dict(tol=0), raises=True),
    dict(name="negcurv", H=_Hneg, A=_Aneg, c=_c4, b=_b,
         kw=dict(tol=0, trust_radius=1000)),
    dict(name="box_inactive", H=_H4, A=_A4, c=_c4, b=_b,
         kw=dict(tol=0, lb=[0.5, -inf, -inf, -inf])),
    dict(name="box_active_maxiter", H=_H4, A=_A4, c=_c4, b=_b,
         kw=dict(tol=0, lb=[0.8, -inf, -inf, -inf])),
    dict(name="box_active_boundary", H=_H4, A=_A4, c=_c4, b=_b,
         kw=dict(tol=0, ub=[inf, inf, 1.6, inf], trust_radius=3)),
    dict(name="box_active_boundary_infeasible_iter", H=_H4, A=_A4, c=_c4,
         b=_b, kw=dict(tol=0, ub=[inf, 0.1, inf, inf], trust_radius=4)),
    dict(name="box_active_negcurv", H=_Hneg, A=_Aneg, c=_c4, b=_b,
         kw=dict(tol=0, ub=[inf, inf, 100, inf], trust_radius=1000)),
]

# projections: the 3x8 matrix with its test points, and the two
# dense-vs-sparse comparison matrices
A38 = [[1, 2, 3, 4, 0, 5, 0, 7],
       [0, 8, 7, 0, 1, 5, 9, 0],
       [1, 0, 0, 0, 0, 1, 2, 3]]
A38_POINTS_N = [[1, 2, 3, 4, 5, 6, 7, 8],
                [1, 10, 3, 0, 1, 6, 7, 8],
                [1.12, 10, 0, 0, 100000, 6, 0.7, 8],
                [1, 0, 0, 0, 0, 1, 2, 3 + 1e-10]]
A38_POINTS_M = [[1, 2, 3], [1, 10, 3], [1.12, 10, 0]]
ORTH_VECTORS = [
    [-1.98931144, -1.56363389, -0.84115584, 2.2864762,
     5.599141, 0.09286976, 1.37040802, -0.28145812],
    [697.92794044, -4091.65114008, -3327.42316335, 836.86906951,
     99434.98929065, -1285.37653682, -4109.21503806, 2935.29289083],
]

# rank-deficient Jacobians (reference: SVD fallback with a warning,
# projections.py:101-108,181-187): a zero row; a row that is the sum of the others
RANK_DEFICIENT = {
    "zero_row": [[1, 2, 3, 4, 0, 5, 0, 7], [0, 8, 7, 0, 1, 5, 9, 0], [0] * 8],
    "sum_row": [[1, 2, 3, 4, 0, 5, 0, 7], [0, 8, 7, 0, 1, 5, 9, 0],
                [1, 10, 10, 4, 1, 10, 9, 7]],
}


def diag4_matrix():
    """[D D D D] with D = diag(1..100) (test_projections.py:110-111)."""
    D = np.diag(np.arange(1, 101, dtype=float))
    return np.hstack([D, D, D, D])


def diag3_matrix():
    """test_projections.py:125-128."""
    return np.hstack([np.diag([-1.7, 1, 0.5]), np.diag([1, -0.6, -0.3]),
                      np.diag([-0.3, -1.5, 2])])
