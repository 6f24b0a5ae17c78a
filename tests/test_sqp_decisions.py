"""The outer iteration's decision arithmetic (csrc/sqp.hip: ONE set of functions compiled for
the decide kernels and for the host) against a restatement of the reference's statements --
equality_constrained_sqp.py:135-153 (model, penalty, predicted reduction), :156-173 (actual /
predicted, second-order-correction test), :196-242 (trust-radius ladder, accept / reject) --
and the scalar tails of box_sphere_intersections (qp_subproblem.py:99-149, 194-234, 286-296)
against the oracle's routines on seeded vectors.  Host entry points only: no GPU."""
import ctypes

import numpy as np
import pytest

from ipsolver import _hip, sqp_chain as sc
import oracle.qp_subproblem as oqp


def _model_ref(hdd, cd, lin, norm_b, penalty, f):
    qm = 1 / 2 * hdd + cd                                         # :138
    vpred = max(1e-16, norm_b - lin)                              # :146
    prev = penalty
    if qm > 0:                                                    # :148-150
        penalty = max(penalty, qm / ((1 - 0.3) * vpred))
    return qm, vpred, prev, penalty, -qm + penalty * vpred, f + penalty * norm_b


def _ladder_ref(ratio, norm_d, radius, penalty, prev):
    if ratio >= 0.9:                                              # :196-212
        radius = max(7.0 * norm_d, radius)
    elif ratio >= 0.3:
        radius = max(2.0 * norm_d, radius)
    elif ratio < 1e-8:
        red = (1 - 1e-8) / (1 - ratio)
        new = red * norm_d
        if new >= 0.5 * radius:
            radius *= 0.5
        elif new >= 0.1 * radius:
            radius = new
        else:
            radius *= 0.1
    accept = ratio >= 1e-8                                        # :216, 241
    return radius, accept, penalty if accept else prev


def test_model_ratio_and_ladder_match_the_reference_statements():
    rng = np.random.default_rng(5)
    for _ in range(4000):
        q = sc.new_block()
        hdd, cd = rng.normal() * 10 ** rng.uniform(-6, 3), rng.normal() * 10 ** rng.uniform(-6, 3)
        norm_b = abs(rng.normal()) * 10 ** rng.uniform(-9, 2)
        lin = norm_b * rng.uniform(0, 1.2)
        penalty, f = 10 ** rng.uniform(0, 3), rng.normal() * 100
        q[sc.HDD], q[sc.CD], q[sc.LIN], q[sc.NORM_B] = hdd, cd, lin, norm_b
        q[sc.PENALTY], q[sc.F] = penalty, f
        sc.model_host(q)
        qm, vpred, prev, pen, pred, merit = _model_ref(hdd, cd, lin, norm_b, penalty, f)
        assert (q[sc.QMODEL], q[sc.VPRED], q[sc.PREV_PENALTY], q[sc.PENALTY], q[sc.PRED],
                q[sc.MERIT]) == (qm, vpred, prev, pen, pred, merit)
        f_next, nbn = f + rng.normal() * abs(pred), norm_b * rng.uniform(0, 1.5)
        norm_dn, norm_dt = abs(rng.normal()), abs(rng.normal())
        norm_d, radius = abs(rng.normal()) + 1e-3, 10 ** rng.uniform(-3, 3)
        q[sc.F_NEXT], q[sc.NORM_B_NEXT], q[sc.NORM_DN], q[sc.NORM_DT] = f_next, nbn, norm_dn, norm_dt
        q[sc.NORM_D], q[sc.RADIUS] = norm_d, radius
        sc.ratio_host(q)
        actual = merit - (f_next + pen * nbn)                     # :168-169
        ratio = actual / pred
        assert (q[sc.ACTUAL], q[sc.RATIO]) == (actual, ratio)
        assert bool(q[sc.SOC]) == (ratio < 1e-8 and norm_dn <= 0.1 * norm_dt)    # :172-173
        sc.radius_host(q)
        r, acc, p = _ladder_ref(ratio, norm_d, radius, pen, prev)
        assert (q[sc.RADIUS], bool(q[sc.ACCEPT]), q[sc.PENALTY]) == (r, acc, p)


def _sums7(z, d, lb, ub):
    """What ipx_box_sphere_reduce leaves (csrc/vec.hip RedBoxSphere)."""
    nz = d != 0
    with np.errstate(divide="ignore", invalid="ignore"):
        tl, tu = (lb - z) / d, (ub - z) / d
    lo = np.minimum(tl, tu)[nz]
    hi = np.maximum(tl, tu)[nz]
    return np.array([d.dot(d), z.dot(d), z.dot(z), lo.max() if nz.any() else -np.inf,
                     hi.min() if nz.any() else np.inf,
                     float(np.sum((~nz) & ((z < lb) | (z > ub)))), float(nz.sum())])


@pytest.mark.parametrize("entire_line", [False, True])
def test_box_sphere_tail_matches_the_oracle(entire_line):
    lib = _hip.load()
    rng = np.random.default_rng(11)
    out = (ctypes.c_double * 3)()
    checked = 0
    for trial in range(600):
        n = int(rng.integers(1, 9))
        z = rng.normal(size=n) * 10 ** rng.uniform(-2, 1)
        d = rng.normal(size=n)
        d[rng.random(n) < 0.2] = 0.0
        lb = np.where(rng.random(n) < 0.3, -np.inf, z - abs(rng.normal(size=n)) * rng.choice([1, -0.2]))
        ub = np.where(rng.random(n) < 0.3, np.inf, z + abs(rng.normal(size=n)) * rng.choice([1, -0.2]))
        radius = np.inf if trial % 7 == 0 else abs(rng.normal()) * 3 + 0.1
        if trial % 11 == 0:
            d[:] = 0.0
        s7 = _sums7(z, d, lb, ub)
        buf = (ctypes.c_double * 7)(*s7)
        lib.ipx_sqp_box_sphere_host(buf, ctypes.c_double(radius), int(entire_line), out)
        with np.errstate(all="ignore"):
            ta, tb, hit = oqp.box_sphere_intersections(z, d, lb, ub, radius, entire_line)
        assert bool(out[2]) == bool(hit), (trial, list(out), (ta, tb, hit))
        if hit:
            assert out[0] == pytest.approx(ta, rel=1e-15, abs=0) or out[0] == ta
            assert out[1] == pytest.approx(tb, rel=1e-15, abs=0) or out[1] == tb
            checked += 1
    assert checked > 100
