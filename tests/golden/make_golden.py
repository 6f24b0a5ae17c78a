#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run in the build container only (it needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported unmodified from /root/reference; two harness shims
live here: scipy >= 1.12 probes a dtype-less
LinearOperator with an int8 vector, which breaks the reference's
``np.zeros_like(p); result += ...`` accumulation, so ``dtype=float`` is
defaulted for operators built from a bare matvec.  Numerics are untouched.

Outputs (numbers only -- no reference source travels):
  qp_small.json     reference outputs for every case in tests/cases_small.py
  banded_*.npz      projected_cg / modified_dogleg / Z,LS,Y traces on the
                    seeded banded problem
  e2e.json          end-to-end minimize_constrained traces (tests/problems.py,
                    README example, banded NLPs)
  qp_extra.json     projection refinement cases (test_projections.py:48-65,141-156:
                    orth_tol=1e-18, max_refin 100 / 10) and rank-deficient Jacobians
                    (SVD fallback, projections.py:101-108,181-187,236-287)
  e2e_n20000.json   (``--n20000``) banded equality NLP at n=20000 / m=2000, both methods
  e2e_n100000.json  (``--n100000``) the same at n=100000 / m=10000 (world = 8 sharded tests)
  e2e_ineq_n12000.json  (``--ineq12000``: 4 minutes) box + inequality NLP at n=12000 / m=1200
  config3_n1e6.json (``--config3``: 7 minutes) BASELINE config 3 at its FULL size n=1e6 /
                    m=1e5 through the reference: all rows of the scalar trace, every 1000th
                    component of x, the one-ulp record
  e2e_ineq_n20000.json  (``--c5-n20000``: 25 minutes) box + inequality NLP at n=20000 / m=2000
  e2e_cs.json       (``--cs``) the two finite-difference test problems with hess='cs'
  e2e_sparse_barrier.json  (``--sparse-barrier``: a minute) box + linear inequalities with a
                    Jacobian of RANDOM sparsity, n=1200 / m=800 (tests/problems.py)
  e2e_dense_nl.json (``--dense-nl``) dense NONLINEAR equality constraints (a factorization per
                    accepted step): synthetic.CenteredDenseNLP at n = 300, m = 60
  config2.json      (``--big`` only: minutes) dense equality QP of BASELINE config 2 at
                    n=4000/m=800 and n=10000/m=2000: scalar traces + strided x
  api.json          (``--api``) the host-side API either side of the path -- kind grammar,
                    feasibility enforcement, conversions, canonical form (values, Jacobians,
                    re-signed Hessians, concatenation), operator-mode finite differences --
                    on the cases of tests/cases_api.py
  late_barrier_n400.npz, late_barrier_n12000.npz  (``--late-barrier``: 5 minutes) single
                    ``projected_cg`` calls of the reference's config-5 style runs, spread
                    over the run up to the last one (barrier parameter <= 1e-6, slacks of
                    active bounds ~1e-8): the inputs the reference passed (as the data the
                    subproblem is assembled from: x, s, the multipliers of the nonlinear
                    rows, the slack block of the Hessian, c_t, lb_t, the radius) and what it
                    returned (x, niter, stop_cond, hits_boundary, the first 20 iterates)
"""
import json
import os
import sys
import warnings

import numpy as np
import scipy.sparse as sps

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scipy.sparse.linalg._interface as _iface  # noqa: E402

_orig_init = _iface._CustomLinearOperator.__init__


def _init(self, shape, matvec, rmatvec=None, matmat=None, dtype=None,
          rmatmat=None):
    _orig_init(self, shape, matvec, rmatvec=rmatvec, matmat=matmat,
               dtype=float if dtype is None else dtype, rmatmat=rmatmat)


_iface._CustomLinearOperator.__init__ = _init

import ipsolver as ref  # noqa: E402  (the REFERENCE package)
from ipsolver._large_scale_constrained import qp_subproblem as rqp  # noqa: E402
from ipsolver._large_scale_constrained import projections as rproj  # noqa: E402
assert ref.__file__.startswith("/root/reference"), ref.__file__

import cases_small as cs  # noqa: E402
import problems  # noqa: E402
import banded_setup  # noqa: E402

synthetic = banded_setup.load_synthetic()


def jf(v):
    """JSON-safe float / list."""
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if np.isinf(v):
            return "inf" if v > 0 else "-inf"
        if np.isnan(v):
            return "nan"
        return v
    if isinstance(v, np.ndarray):
        return [jf(t) for t in v.tolist()]
    if isinstance(v, (list, tuple)):
        return [jf(t) for t in v]
    raise TypeError(type(v))


def small_cases():
    out = {"sphere": [], "box": [], "box_sphere": [], "dogleg": [], "pcg": []}
    for z, d, r in cs.SPHERE:
        for line in (False, True):
            out["sphere"].append(jf(rqp.sphere_intersections(z, d, r, line)))
    for z, d, lb, ub in cs.BOX:
        for line in (False, True):
            out["box"].append(jf(rqp.box_intersections(z, d, lb, ub, line)))
    for z, d, lb, ub, r in cs.BOX_SPHERE:
        for line in (False, True):
            out["box_sphere"].append(
                jf(rqp.box_sphere_intersections(z, d, lb, ub, r, line)))
    for A, b, r, lb, ub in cs.DOGLEG:
        A = np.array(A, dtype=float)
        _, _, Y = rproj.projections(A)
        out["dogleg"].append(jf(rqp.modified_dogleg(A, Y, np.array(b, float),
                                                    r, lb, ub)))
    for case in cs.PCG:
        H = sps.csc_matrix(np.array(case["H"], dtype=float))
        A = sps.csc_matrix(np.array(case["A"], dtype=float))
        c = np.array(case["c"], dtype=float)
        b = np.array(case["b"], dtype=float)
        Z, _, Y = rproj.projections(A)
        try:
            x, info = rqp.projected_cg(H, c, Z, Y, b, return_all=True,
                                       **case["kw"])
            rec = {"x": jf(x), "niter": info["niter"],
                   "stop_cond": info["stop_cond"],
                   "hits_boundary": bool(info["hits_boundary"]),
                   "allvecs": [jf(v) for v in info["allvecs"]]}
        except ValueError as e:
            rec = {"raises": str(e)}
        out["pcg"].append(rec)

    # projections on the 3x8 matrix, every method available here
    A38 = np.array(cs.A38, dtype=float)
    proj = {}
    for method, A in (("AugmentedSystem", sps.csc_matrix(A38)),
                      ("QRFactorization", A38), ("SVDFactorization", A38)):
        Z, LS, Y = rproj.projections(A, method)
        proj[method] = {
            "Z": [jf(Z.dot(np.array(p, float))) for p in cs.A38_POINTS_N],
            "LS": [jf(LS.dot(np.array(p, float))) for p in cs.A38_POINTS_N],
            "Y": [jf(Y.dot(np.array(p, float))) for p in cs.A38_POINTS_M]}
    out["proj38"] = proj
    out["orth"] = [jf(rproj.orthogonality(A38, np.array(v)))
                   for v in cs.ORTH_VECTORS]

    # dense-vs-sparse comparison matrices, seeded vectors
    for key, A in (("diag4", cs.diag4_matrix()), ("diag3", cs.diag3_matrix())):
        rng = np.random.RandomState(0)
        m, n = A.shape
        Zs, LSs, Ys = rproj.projections(sps.csc_matrix(A))
        Zd, LSd, Yd = rproj.projections(A)
        rec = {"Z_sparse": [], "LS_sparse": [], "Y_sparse": [],
               "Z_dense": [], "LS_dense": [], "Y_dense": []}
        for _ in range(3):
            z = rng.normal(size=n)
            x = rng.normal(size=m)
            rec["Z_sparse"].append(jf(Zs.dot(z)))
            rec["LS_sparse"].append(jf(LSs.dot(z)))
            rec["Y_sparse"].append(jf(Ys.dot(x)))
            rec["Z_dense"].append(jf(Zd.dot(z)))
            rec["LS_dense"].append(jf(LSd.dot(z)))
            rec["Y_dense"].append(jf(Yd.dot(x)))
        out[key] = rec
    return out


def banded_traces(n, m, full_vectors):
    """projected_cg / dogleg / projection traces on the Appendix C matrices."""
    inst = banded_setup.BandedInstance(n, m)
    A, H, c, bvec = inst.A, inst.H, inst.c, inst.b
    Z, LS, Y = rproj.projections(A)
    out = {}
    stride = inst.stride

    def keep(vec):
        vec = np.asarray(vec)
        return vec if full_vectors else vec[::stride]

    out["Z"] = np.array([keep(Z.dot(p)) for p in inst.probes_n])
    out["LS"] = np.array([keep(LS.dot(p)) for p in inst.probes_n])
    out["Y"] = np.array([keep(Y.dot(p)) for p in inst.probes_m])

    zero_b = np.zeros(m)
    gnorm = np.linalg.norm(Z.dot(c))
    out["gnorm"] = np.array([gnorm])
    for name, kw in inst.pcg_variants(gnorm).items():
        xs, info = rqp.projected_cg(H, c, Z, Y, zero_b, return_all=True, **kw)
        out["pcg_%s_x" % name] = keep(xs)
        out["pcg_%s_info" % name] = np.array(
            [info["niter"], info["stop_cond"], int(info["hits_boundary"])])
        out["pcg_%s_allvecs" % name] = np.array(
            [keep(t) for t in info["allvecs"]])
    # non-zero b (row-space start) with a ball
    y_b = Y.dot(bvec)
    xs, info = rqp.projected_cg(H, c, Z, Y, bvec, tol=0, max_iter=10,
                                trust_radius=10 * np.linalg.norm(y_b),
                                return_all=True)
    out["pcg_rowstart_x"] = keep(xs)
    out["pcg_rowstart_info"] = np.array(
        [info["niter"], info["stop_cond"], int(info["hits_boundary"])])
    out["y_b"] = keep(y_b)
    out["y_b_norm_max"] = np.array([np.linalg.norm(y_b), np.abs(y_b).max()])

    dl = []
    for radius, lo, hi in inst.dogleg_cfg(y_b):
        dl.append(keep(rqp.modified_dogleg(A, Y, bvec, radius,
                                           np.full(n, lo), np.full(n, hi))))
    out["dogleg"] = np.array(dl)
    out["stride"] = np.array([1 if full_vectors else stride])
    return out


def extra_cases():
    """Refinement and rank-deficient projection cases."""
    out = {}
    A38 = np.array(cs.A38, dtype=float)
    ref = {}
    for method, A, max_refin in (("AugmentedSystem", sps.csc_matrix(A38), 100),
                                 ("QRFactorization", A38, 10),
                                 ("SVDFactorization", A38, 10)):
        Z, LS, Y = rproj.projections(A, method, orth_tol=1e-18, max_refin=max_refin)
        zs = [Z.dot(np.array(p, float)) for p in cs.A38_POINTS_N]
        ref[method] = {"max_refin": max_refin, "Z": [jf(z) for z in zs],
                       "orth": [jf(rproj.orthogonality(A38, z)) for z in zs]}
    out["proj38_refine"] = ref

    rank = {}
    for name, rows in cs.RANK_DEFICIENT.items():
        A = np.array(rows, dtype=float)
        rec = {}
        for kind, M in (("sparse", sps.csc_matrix(A)), ("dense", A)):
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                Z, LS, Y = rproj.projections(M)
                rec[kind] = {
                    "warnings": [str(x.message) for x in w],
                    "Z": [jf(Z.dot(np.array(p, float))) for p in cs.A38_POINTS_N[:3]],
                    "LS": [jf(LS.dot(np.array(p, float))) for p in cs.A38_POINTS_N[:3]],
                    "Y": [jf(Y.dot(np.array(p, float))) for p in cs.A38_POINTS_M]}
        rec["singular_values"] = jf(np.linalg.svd(A, compute_uv=False))
        rank[name] = rec
    out["rank_deficient"] = rank
    return out


def banded_refine_traces(n, m):
    """projected_cg on the Appendix C matrices with projections that refine on
    every application (orth_tol far below the attainable orthogonality), so the
    refinement loop projections.py:126-139 runs inside the CG loop."""
    inst = banded_setup.BandedInstance(n, m)
    out = {}
    for max_refin in (1, 3):
        Z, LS, Y = rproj.projections(inst.A, orth_tol=1e-30, max_refin=max_refin)
        out["refine%d_Z" % max_refin] = np.array([Z.dot(p) for p in inst.probes_n])
        gnorm = np.linalg.norm(Z.dot(inst.c))
        for name, kw in inst.pcg_variants(gnorm).items():
            if name not in ("free", "box"):
                continue
            xs, info = rqp.projected_cg(inst.H, inst.c, Z, Y, np.zeros(m), **kw)
            out["refine%d_pcg_%s_x" % (max_refin, name)] = xs
            out["refine%d_pcg_%s_info" % (max_refin, name)] = np.array(
                [info["niter"], info["stop_cond"], int(info["hits_boundary"])])
    return out


def config2(n, m):
    """BASELINE config 2: dense random equality-constrained QP."""
    rng = np.random.default_rng(0)
    A = rng.standard_normal((m, n))
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    Hd = G.dot(G.T) + np.eye(n)
    c = rng.standard_normal(n)
    xf = rng.standard_normal(n)
    bq = A.dot(xf)
    rec = run_e2e("config2_n%d" % n, lambda x: 0.5 * x.dot(Hd.dot(x)) + c.dot(x),
                  np.zeros(n), lambda x: Hd.dot(x) + c, lambda x: Hd,
                  ref.LinearConstraint(A, ("equals", bq)),
                  method="equality_constrained_sqp")
    return rec


def _trace_of(fun, x0, grad, hess, constraints, kw):
    rows = []

    def cb(state):
        rows.append([int(state.niter), int(state.cg_niter),
                     float(state.trust_radius), float(state.penalty),
                     float(getattr(state, "barrier_parameter", np.nan)),
                     float(state.optimality),
                     float(state.constr_violation), int(state.nfev)])
        return False

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = ref.minimize_constrained(fun, x0, grad, hess, constraints,
                                       callback=cb, **kw)
    return res, rows


def one_ulp_sensitivity(rows, x, fun, x0, grad, hess, constraints, kw, seeds=(31, 32, 33)):
    """How well the reference's OWN trace is determined: the same run with every component of
    the objective gradient moved by one unit in the last place (seeded sign patterns).
    Returns ``stable_rows`` -- the leading rows whose integer columns (niter, cg_niter, nfev)
    are the same in every perturbed run -- and, per row of that prefix and float column, the
    largest absolute change; plus the largest relative change of the final x (None when a
    perturbed run ends elsewhere in the trace)."""
    ref_rows = np.array([[np.nan if v is None else v for v in r] for r in rows], dtype=float)
    stable = len(ref_rows)
    sens = np.zeros((len(ref_rows), 8))
    x_sens, same_end = 0.0, True
    for seed in seeds:
        rng = np.random.default_rng(seed)
        signs = rng.choice([-1.0, 1.0], size=np.size(x0))
        res_p, rows_p = _trace_of(fun, x0, lambda xx: np.asarray(grad(xx), dtype=float)
                                  * (1.0 + np.ldexp(1.0, -52) * signs), hess, constraints, kw)
        pr = np.array(rows_p, dtype=float)
        k = min(len(pr), len(ref_rows))
        ints = (0, 1, 7)
        agree = np.all(pr[:k][:, ints] == ref_rows[:k][:, ints], axis=1)
        first_bad = int(np.argmin(agree)) if not agree.all() else k
        stable = min(stable, first_bad)
        if len(pr) != len(ref_rows) or first_bad < k:
            same_end = False
        with np.errstate(invalid="ignore"):
            d = np.abs(pr[:k] - ref_rows[:k])
        d[~np.isfinite(d)] = 0.0
        sens[:k] = np.maximum(sens[:k], d)
        xp = np.asarray(res_p.x)
        x_sens = max(x_sens, float(np.max(np.abs(xp - x)) / max(np.max(np.abs(x)), 1e-300)))
    return {"stable_rows": int(stable), "rows": jf(sens[:stable]),
            "x": x_sens if same_end else None, "seeds": list(seeds)}


def run_e2e(name, fun, x0, grad, hess, constraints, x_stride=None, **kw):
    res, rows = _trace_of(fun, x0, grad, hess, constraints, kw)
    rec = {"x": jf(np.asarray(res.x)) if np.size(res.x) <= 64
           else jf(np.asarray(res.x)[::x_stride or max(1, np.size(res.x) // 50)]),
           "status": int(res.status), "niter": int(res.niter),
           "cg_niter": int(res.cg_niter), "nfev": int(res.nfev),
           "ngev": int(res.ngev), "nhev": int(res.nhev),
           "ncev": int(res.ncev), "njev": int(res.njev),
           "method": res.method, "optimality": jf(res.optimality),
           "constr_violation": jf(res.constr_violation),
           "trust_radius": jf(res.trust_radius), "penalty": jf(res.penalty),
           "fun": jf(float(res.fun)),
           "v": jf(np.asarray(res.v)) if np.size(res.v) <= 64 else None,
           "keys": sorted(res.keys()),
           "trace": jf(rows)}
    if "--no-sens" not in sys.argv:
        rec["one_ulp"] = one_ulp_sensitivity(rows, np.asarray(res.x), fun, x0, grad, hess,
                                             constraints, kw)
    if "s" in res:
        rec["s"] = jf(np.asarray(res.s)) if np.size(res.s) <= 64 else None
        rec["barrier_parameter"] = jf(res.barrier_parameter)
    print("  %-28s status %d niter %d cg %d   stable rows under one ulp: %s of %d, x moves %s"
          % (name, rec["status"], rec["niter"], rec["cg_niter"],
             rec.get("one_ulp", {}).get("stable_rows"), len(rows),
             rec.get("one_ulp", {}).get("x")))
    return rec


def e2e():
    out = {}
    for p in problems.exact_hessian_problems() + problems.fd_hessian_problems():
        out[p.name] = run_e2e(p.name, p.fun, p.x0, p.grad, p.hess_arg(),
                              p.constraints(ref))
    # Maratos via the SQP method explicitly; README example = hyperbolic_ineq
    p = problems.Maratos()
    out["maratos_sqp"] = run_e2e("maratos_sqp", p.fun, p.x0, p.grad, p.hess,
                                 p.constraints(ref),
                                 method="equality_constrained_sqp")
    # banded NLPs (Appendix C) at small n: equality (both methods) and
    # box + inequality (barrier)
    for n, m in ((2000, 200),):
        prob = synthetic.CenteredBandedNLP(n, m, eps=1e-3)
        for method in ("tr_interior_point", "equality_constrained_sqp"):
            key = "banded_eq_n%d_%s" % (n, method)
            out[key] = run_e2e(key, prob.fun, prob.x0, prob.grad, prob.hess,
                               prob.constraints(ref), method=method)
    prob = synthetic.CenteredBandedNLP(400, 40, eps=1.0)
    cons = (prob.constraints(ref, ("less", 0.0)),
            ref.BoxConstraint(("interval", -0.8, 0.8)))
    out["banded_ineq_n400"] = run_e2e("banded_ineq_n400", prob.fun, prob.x0,
                                      prob.grad, prob.hess, cons)
    # dense equality QP (config-2 style, small)
    rng = np.random.default_rng(0)
    n, m = 60, 12
    A = rng.standard_normal((m, n))
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    Hd = G.dot(G.T) + np.eye(n)
    c = rng.standard_normal(n)
    xf = rng.standard_normal(n)
    bq = A.dot(xf)
    out["dense_eq_qp_n60"] = run_e2e(
        "dense_eq_qp_n60", lambda x: 0.5 * x.dot(Hd.dot(x)) + c.dot(x),
        np.zeros(n), lambda x: Hd.dot(x) + c, lambda x: Hd,
        ref.LinearConstraint(A, ("equals", bq)),
        method="equality_constrained_sqp")
    return out


def late_barrier(n, m, ncalls):
    """Single projected_cg calls out of the reference's barrier run on the config-5 style
    problem (tr_interior_point.py:222-241 assembles H, :141-194 the augmented Jacobian;
    equality_constrained_sqp.py:125-132 the call).  The hooks only record."""
    # (the package re-exports functions under the modules' names: go through sys.modules)
    rsqp = sys.modules["ipsolver._large_scale_constrained.equality_constrained_sqp"]
    rtip = sys.modules["ipsolver._large_scale_constrained.tr_interior_point"]
    prob = synthetic.CenteredBandedNLP(n, m, eps=1.0)
    cons = (prob.constraints(ref, ("less", 0.0)), ref.BoxConstraint(("interval", -0.8, 0.8)))
    N, n_ineq = n + m + 2 * n, m + 2 * n
    stride = max(1, N // 400)
    last, calls = {}, []
    orig_pcg, orig_proj = rsqp.projected_cg, rsqp.projections
    orig_lh = rtip.BarrierSubproblem.lagrangian_hessian

    def proj(A, *a, **k):
        last["A"] = A
        return orig_proj(A, *a, **k)

    def lh(self, z, v):
        last["z"], last["v"] = np.array(z), np.array(v)
        last["Hs"] = np.array(self.lagrangian_hessian_s(z, v))
        last["mu"] = float(self.barrier_parameter)
        return orig_lh(self, z, v)

    def pcg(H, c, Z, Y, b, trust_radius, lb, ub, **kw):
        x, info = orig_pcg(H, c, Z, Y, b, trust_radius, lb, ub, return_all=True, **kw)
        z, v = last["z"], last["v"]
        xv, s = z[:n], z[n:]
        v_nl = v[:m]                              # ('less', 0): sign +1, nonlinear rows first
        # the subproblem re-assembled from the recorded data must BE the reference's
        J = prob.constr_jac(xv)
        eye = sps.identity(n, format="csr")
        A_re = sps.bmat([[sps.vstack([J, -eye, eye]), sps.diags(s)]], format="csr")
        dA = abs(A_re - sps.csr_matrix(last["A"]))
        assert dA.nnz == 0 or dA.max() == 0.0, dA.max()
        Hx, hd = prob.hess(xv), prob.kappa * prob.Wt.dot(v_nl)
        p = np.random.default_rng(len(calls)).standard_normal(N)
        mine = np.hstack((Hx.dot(p[:n]) + hd * p[:n], last["Hs"] * p[n:]))
        assert np.array_equal(mine, H.dot(p)), np.max(np.abs(mine - H.dot(p)))
        assert np.all(np.isinf(ub)) and np.all(b == 0)
        allv = info["allvecs"]
        calls.append(dict(
            x_vars=xv.copy(), s=s.copy(), v_nl=v_nl.copy(), Hs=last["Hs"].copy(),
            c=np.array(c), lb=np.array(lb), radius=float(trust_radius), mu=last["mu"],
            x=np.array(x), info=np.array([info["niter"], info["stop_cond"],
                                          int(info["hits_boundary"])]),
            allvecs=np.array([a[::stride] for a in allv[:20]]),
            _ops=(H, Z, Y, last["A"], np.array(b), np.array(ub))))
        return x, info

    def conditioning(rec):
        """How well the reference's OWN answer is determined (what parity can mean here):
        (a) its projection Z c against the exact one -- the augmented system solved to
        long-double accuracy by iterative refinement (late in the barrier run c lies almost
        entirely in the row space of A: Z c cancels ~8 digits of c); (b) its projected_cg
        result when every component of c moves by one unit in the last place."""
        H, Z, Y, A, b, ub = rec.pop("_ops")
        A = sps.csr_matrix(A)
        M, Nn = A.shape
        c = rec["c"]
        K = sps.bmat([[sps.identity(Nn), A.T], [A, None]], format="csc")
        lu = sps.linalg.splu(K)
        Kr = sps.csr_matrix(K)
        Kr.sort_indices()
        dat, idx = Kr.data.astype(np.longdouble), Kr.indices
        rows = np.repeat(np.arange(Nn + M), np.diff(Kr.indptr))

        def matvec_ld(v):
            out = np.zeros(Nn + M, dtype=np.longdouble)
            np.add.at(out, rows, dat * v[idx])
            return out
        rhs = np.concatenate((c, np.zeros(M))).astype(np.longdouble)
        sol = np.zeros(Nn + M, dtype=np.longdouble)
        for _ in range(6):
            sol = sol + lu.solve((rhs - matvec_ld(sol)).astype(np.float64)).astype(np.longdouble)
        z_true = sol[:Nn].astype(np.float64)
        z_ref = Z.dot(c)
        scale = np.max(np.abs(z_true))
        rec["z_true"] = z_true
        rec["proj"] = np.array([np.max(np.abs(z_ref - z_true)) / scale, np.max(np.abs(c)), scale])
        rows = []
        for seed in (99, 100, 101, 102):
            rng = np.random.default_rng(seed)
            c_p = c * (1.0 + np.ldexp(1.0, -52) * rng.choice([-1.0, 1.0], size=c.size))
            x_p, info_p = orig_pcg(H, c_p, Z, Y, b, rec["radius"], rec["lb"], ub, return_all=True)
            sx = np.max(np.abs(x_p - rec["x"])) / max(np.max(np.abs(rec["x"])), 1e-300)
            sv = [np.max(np.abs(a[::stride] - w)) / max(np.max(np.abs(w)), 1e-300)
                  for a, w in zip(info_p["allvecs"][:20], rec["allvecs"])]
            rows.append([info_p["niter"], info_p["stop_cond"], int(info_p["hits_boundary"]), sx,
                         max(sv) if sv else 0.0])
        rec["sens"] = np.array(rows)          # one row per perturbed run
        return rec

    rsqp.projected_cg, rsqp.projections = pcg, proj
    rtip.BarrierSubproblem.lagrangian_hessian = lh
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = ref.minimize_constrained(prob.fun, prob.x0, prob.grad, prob.hess, cons)
    finally:
        rsqp.projected_cg, rsqp.projections = orig_pcg, orig_proj
        rtip.BarrierSubproblem.lagrangian_hessian = orig_lh
    total = len(calls)
    picks = sorted(set(int(round(t)) for t in np.linspace(0, total - 1, ncalls)))
    for k in range(total):
        if k in picks:
            conditioning(calls[k])
        else:
            calls[k].pop("_ops")
    out = {"n": np.array([n, m]), "stride": np.array([stride]), "picks": np.array(picks),
           "total_calls": np.array([total]), "status": np.array([res.status, res.niter,
                                                                 res.cg_niter])}
    for j, k in enumerate(picks):
        for key, val in calls[k].items():
            out["c%d_%s" % (j, key)] = np.asarray(val)
    print("  late_barrier n=%d: %d projected_cg calls, picked %s, niter %s, mu %s"
          % (n, total, picks, [int(calls[k]["info"][0]) for k in picks],
             [calls[k]["mu"] for k in picks]))
    for k in picks:
        print("    call %d: reference projection error %.2e (|c| %.1e, |Zc| %.1e); one ulp in c: "
              "niter %d -> %s, x moves %.2e, first iterates %.2e"
              % (k, calls[k]["proj"][0], calls[k]["proj"][1], calls[k]["proj"][2],
                 calls[k]["info"][0], calls[k]["sens"][:, 0].astype(int).tolist(),
                 calls[k]["sens"][:, 3].max(),
                 calls[k]["sens"][:, 4].max()))
    return out


def main():
    if "--api" in sys.argv:
        # constraint classes, canonical form, operator-mode finite differences: the cases of
        # tests/cases_api.py evaluated by the reference's own classes
        import cases_api
        with open(os.path.join(HERE, "api.json"), "w") as f:
            json.dump(cases_api.run(cases_api.reference_api()), f)
        return
    if "--late-barrier" in sys.argv:
        for n, m, k in ((400, 40, 6), (12000, 1200, 4)):
            np.savez_compressed(os.path.join(HERE, "late_barrier_n%d.npz" % n),
                                **late_barrier(n, m, k))
        return
    if "--e2e" in sys.argv:
        with open(os.path.join(HERE, "e2e.json"), "w") as f:
            json.dump(e2e(), f)
        return
    if "--n20000" in sys.argv:
        # banded equality NLP at a size the row-sharded solver can split (8 blocks of 260 rows)
        out = {}
        prob = synthetic.CenteredBandedNLP(20000, 2000, eps=1e-3)
        for method in ("tr_interior_point", "equality_constrained_sqp"):
            key = "banded_eq_n20000_%s" % method
            out[key] = run_e2e(key, prob.fun, prob.x0, prob.grad, prob.hess,
                               prob.constraints(ref), method=method)
        with open(os.path.join(HERE, "e2e_n20000.json"), "w") as f:
            json.dump(out, f)
        return
    if "--n100000" in sys.argv:
        # the same at a size that splits over 8 ranks and more (38 blocks of 260 rows): the
        # world = 8 tests of the row-sharded solver; every 100th component of x
        out = {}
        prob = synthetic.CenteredBandedNLP(100000, 10000, eps=1e-3)
        for method in ("tr_interior_point", "equality_constrained_sqp"):
            key = "banded_eq_n100000_%s" % method
            out[key] = run_e2e(key, prob.fun, prob.x0, prob.grad, prob.hess,
                               prob.constraints(ref), x_stride=100, method=method)
            out[key]["x_stride"] = 100
        with open(os.path.join(HERE, "e2e_n100000.json"), "w") as f:
            json.dump(out, f)
        return
    if "--ineq12000" in sys.argv:
        # config-5 style (box + nonlinear inequalities) at a size the row-sharded solver can
        # split; 225 s with the reference
        prob = synthetic.CenteredBandedNLP(12000, 1200, eps=1.0)
        cons = (prob.constraints(ref, ("less", 0.0)), ref.BoxConstraint(("interval", -0.8, 0.8)))
        rec = run_e2e("banded_ineq_n12000", prob.fun, prob.x0, prob.grad, prob.hess, cons)
        with open(os.path.join(HERE, "e2e_ineq_n12000.json"), "w") as f:
            json.dump({"banded_ineq_n12000": rec}, f)
        return
    if "--config3" in sys.argv:
        # BASELINE config 3 at FULL size: every row of the scalar trace and
        # every 1000th component of x; ~100 s per run of the reference (x4 with the one-ulp
        # re-runs)
        prob = synthetic.CenteredBandedNLP(1000000, 100000, eps=1e-3)
        rec = run_e2e("config3_n1e6", prob.fun, prob.x0, prob.grad, prob.hess,
                      prob.constraints(ref), x_stride=1000, method="tr_interior_point")
        rec["x_stride"] = 1000
        with open(os.path.join(HERE, "config3_n1e6.json"), "w") as f:
            json.dump({"config3_n1e6": rec}, f)
        return
    if "--c5-n20000" in sys.argv:
        # config-5 style at the largest size the survey ran the reference on (327 s per run)
        prob = synthetic.CenteredBandedNLP(20000, 2000, eps=1.0)
        cons = (prob.constraints(ref, ("less", 0.0)), ref.BoxConstraint(("interval", -0.8, 0.8)))
        rec = run_e2e("banded_ineq_n20000", prob.fun, prob.x0, prob.grad, prob.hess, cons,
                      x_stride=100)
        rec["x_stride"] = 100
        with open(os.path.join(HERE, "e2e_ineq_n20000.json"), "w") as f:
            json.dump({"banded_ineq_n20000": rec}, f)
        return
    if "--cs" in sys.argv:
        # complex-step Hessian products (hess='cs', _numdiff.py:429-437) on the reference's own
        # finite-difference test problems
        out = {}
        for cls in (problems.Maratos, problems.HyperbolicIneq):
            p = cls()
            key = p.name + "_cs"
            out[key] = run_e2e(key, p.fun, p.x0, p.grad, "cs", p.constraints(ref))
        with open(os.path.join(HERE, "e2e_cs.json"), "w") as f:
            json.dump(out, f)
        return
    if "--sparse-barrier" in sys.argv:
        # barrier problem with a Jacobian of random sparsity (tests/problems.py SparseBarrierQP)
        p = problems.SparseBarrierQP(1200, 800)
        rec = run_e2e(p.name, p.fun, p.x0, p.grad, p.hess, p.constraints(ref), x_stride=10)
        rec["x_stride"] = 10
        with open(os.path.join(HERE, "e2e_sparse_barrier.json"), "w") as f:
            json.dump({p.name: rec}, f)
        return
    if "--dense-nl" in sys.argv:
        # dense NONLINEAR equality constraints (ipsolver/synthetic.py CenteredDenseNLP): the
        # Jacobian changes at every accepted step, so every one of them refactors
        out = {}
        for n, m in ((300, 60),):
            prob = synthetic.CenteredDenseNLP(n, m, eps=1e-3)
            key = "dense_nl_n%d" % n
            out[key] = run_e2e(key, prob.fun, prob.x0, prob.grad, prob.hess, prob.constraints(ref),
                               method="equality_constrained_sqp")
        with open(os.path.join(HERE, "e2e_dense_nl.json"), "w") as f:
            json.dump(out, f)
        return
    if "--big" in sys.argv:
        out = {}
        for n, m in ((4000, 800), (10000, 2000)):
            out["config2_n%d" % n] = config2(n, m)
        with open(os.path.join(HERE, "config2.json"), "w") as f:
            json.dump(out, f)
        return
    print("small cases ...")
    with open(os.path.join(HERE, "qp_small.json"), "w") as f:
        json.dump(small_cases(), f)
    print("banded traces ...")
    with open(os.path.join(HERE, "qp_extra.json"), "w") as f:
        json.dump(extra_cases(), f)
    np.savez_compressed(os.path.join(HERE, "banded_n2000.npz"),
                        **banded_traces(2000, 200, True))
    np.savez_compressed(os.path.join(HERE, "banded_refine_n2000.npz"),
                        **banded_refine_traces(2000, 200))
    np.savez_compressed(os.path.join(HERE, "banded_n20000.npz"),
                        **banded_traces(20000, 2000, False))
    print("end-to-end ...")
    with open(os.path.join(HERE, "e2e.json"), "w") as f:
        json.dump(e2e(), f)
    print("done")


if __name__ == "__main__":
    main()
