"""GPU parity of the hot path (projections, modified_dogleg, projected_cg and
the intersection helpers) through the C ABI, against

  * the golden fixtures produced by the reference itself, and
  * the CPU oracle on the same seeded inputs,

tolerance 1e-10 relative (BASELINE.json north_star), integer outputs
(niter / stop_cond / hits_boundary) exact.
"""
import numpy as np
import pytest
import scipy.sparse as sps

import cases_small as cs
from banded_setup import BandedInstance
from conftest import unjson, host, close_projection
from conftest import close as _close

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def ips():
    import ipsolver.qp as qp
    import ipsolver.projector as proj
    import ipsolver.device as dv

    class NS:
        pass
    ns = NS()
    ns.qp, ns.proj, ns.dv = qp, proj, dv
    return ns


def close(a, b, tol=TOL, zero_scale=None):
    _close(a, b, tol, zero_scale)


def check_interval(got, want):
    want = unjson(want)
    assert bool(got[2]) == bool(want[2])
    close(np.array(got[:2], dtype=float), want[:2])


def test_intersections_small(ips, qp_small):
    it = iter(qp_small["sphere"])
    for z, d, r in cs.SPHERE:
        for line in (False, True):
            check_interval(ips.qp.sphere_intersections(z, d, r, line), next(it))
    it = iter(qp_small["box"])
    for z, d, lb, ub in cs.BOX:
        for line in (False, True):
            check_interval(ips.qp.box_intersections(z, d, lb, ub, line), next(it))
    it = iter(qp_small["box_sphere"])
    for z, d, lb, ub, r in cs.BOX_SPHERE:
        for line in (False, True):
            check_interval(ips.qp.box_sphere_intersections(z, d, lb, ub, r, line), next(it))


@pytest.mark.parametrize("idx", range(len(cs.PCG)))
def test_projected_cg_small(ips, qp_small, idx):
    case, want = cs.PCG[idx], qp_small["pcg"][idx]
    H = ips.dv.DeviceCSR.from_scipy(sps.csr_matrix(np.array(case["H"], dtype=float)))
    A = sps.csc_matrix(np.array(case["A"], dtype=float))
    c = np.array(case["c"], dtype=float)
    b = np.array(case["b"], dtype=float)
    Z, _, Y = ips.proj.projections(A)
    if case.get("raises"):
        with pytest.raises(ValueError) as err:
            ips.qp.projected_cg(H, c, Z, Y, b, **case["kw"])
        assert str(err.value)[:20] == want["raises"][:20]
        return
    for return_all in (True, False):
        x, info = ips.qp.projected_cg(H, c, Z, Y, b, return_all=return_all, **case["kw"])
        assert info["stop_cond"] == want["stop_cond"]
        assert info["hits_boundary"] == want["hits_boundary"]
        if case.get("knife_edge"):
            # ||x0|| equals the radius to the last bit: the reference's own test
            # (test_qp_subproblem.py:474-491) pins stop_cond, hits_boundary and
            # ||x|| = radius only; 0 or 1 iterations depending on the rounding of the norm
            assert info["niter"] in (0, 1)
            assert abs(np.linalg.norm(host(x)) - case["kw"]["trust_radius"]) <= 1e-12
            close(x, unjson(want["x"]), 1e-6)
            continue
        close(x, unjson(want["x"]))
        assert info["niter"] == want["niter"]
        if return_all:
            assert len(info["allvecs"]) == len(want["allvecs"])
            for a, w in zip(info["allvecs"], want["allvecs"]):
                close(a, unjson(w))


def test_modified_dogleg_small(ips, qp_small):
    for (A, b, r, lb, ub), want in zip(cs.DOGLEG, qp_small["dogleg"]):
        A = np.array(A, dtype=float)
        Ad = ips.proj.as_device_matrix(A)
        _, _, Y = ips.proj.projections(Ad)
        close(ips.qp.modified_dogleg(Ad, Y, np.array(b, float), r, lb, ub), unjson(want))


@pytest.mark.parametrize("kind", ["sparse", "dense"])
def test_projections_3x8(ips, qp_small, kind):
    A38 = np.array(cs.A38, dtype=float)
    A = sps.csc_matrix(A38) if kind == "sparse" else A38
    Z, LS, Y = ips.proj.projections(A)
    want = qp_small["proj38"]["AugmentedSystem" if kind == "sparse" else "QRFactorization"]
    for p, wz, wl in zip(cs.A38_POINTS_N, want["Z"], want["LS"]):
        p = np.array(p, float)
        close_projection(Z.dot(p), unjson(wz), p, TOL)
        close(LS.matvec(p), unjson(wl))
        assert np.max(np.abs(A38.dot(host(Z.dot(p))))) < 1e-14 * max(1.0, np.max(np.abs(p)))
    for p, wy in zip(cs.A38_POINTS_M, want["Y"]):
        close(Y.dot(np.array(p, float)), unjson(wy), 1e-10)


@pytest.mark.parametrize("key", ["diag4", "diag3"])
def test_dense_vs_sparse(ips, qp_small, key):
    A = cs.diag4_matrix() if key == "diag4" else cs.diag3_matrix()
    m, n = A.shape
    rng = np.random.RandomState(0)
    Zs, LSs, Ys = ips.proj.projections(sps.csc_matrix(A))
    Zd, LSd, Yd = ips.proj.projections(A)
    want = qp_small[key]
    for k in range(3):
        z, x = rng.normal(size=n), rng.normal(size=m)
        close(Zs.dot(z), want["Z_sparse"][k], 1e-10)
        close(Zd.dot(z), want["Z_dense"][k], 1e-10)
        close(LSs.dot(z), want["LS_sparse"][k], 1e-10)
        close(LSd.dot(z), want["LS_dense"][k], 1e-10)
        close(Ys.dot(x), want["Y_sparse"][k], 1e-10)
        close(Yd.dot(x), want["Y_dense"][k], 1e-10)


def test_projection_errors(ips):
    A38 = np.array(cs.A38, dtype=float)
    with pytest.raises(ValueError):
        ips.proj.projections(A38, "AugmentedSystem")
    with pytest.raises(ValueError):
        ips.proj.projections(sps.csc_matrix(A38), "QRFactorization")
    Z, LS, Y = ips.proj.projections(np.empty((0, 5)))
    np.testing.assert_array_equal(host(Z.dot(np.arange(5.0))), np.arange(5.0))


@pytest.mark.parametrize("kind", ["sparse", "dense", "svd"])
def test_projections_refinement(ips, qp_extra, kind):
    """The reference's refinement cases (test_projections.py:48-65,141-156: orth_tol=1e-18,
    max_refin 100 / 10): the refinement loop projections.py:69-78 must actually run."""
    A38 = np.array(cs.A38, dtype=float)
    method = {"sparse": "AugmentedSystem", "dense": "QRFactorization",
              "svd": "SVDFactorization"}[kind]
    gold = qp_extra["proj38_refine"][method]
    Z, _, _ = ips.proj.projections(sps.csc_matrix(A38) if kind == "sparse" else A38, method,
                                   orth_tol=1e-18, max_refin=gold["max_refin"])
    assert type(Z.projector).__name__ == ("SVDProjector" if kind == "svd"
                                          else "NormalEquationProjector")
    for p, want in zip(cs.A38_POINTS_N, gold["Z"]):
        p = np.array(p, float)
        before = Z.projector.stats["refinements"]
        z = Z.matvec(p)
        assert Z.projector.stats["refinements"] > before
        close_projection(z, want, p, TOL)
        # the reference's own assertions (decimal=14 on A x, decimal=16 on the orthogonality)
        assert np.max(np.abs(A38.dot(host(z)))) < 1.5e-14 * max(1.0, np.max(np.abs(p)))
        assert ips.proj.orthogonality(A38, z) < 1.5e-16


@pytest.mark.parametrize("name,kind", [("zero_row", "sparse"), ("zero_row", "dense"),
                                       ("sum_row", "sparse")])
def test_rank_deficient_fallback(ips, qp_extra, name, kind):
    """Rank-deficient Jacobian: detected by the device factorization (pivot lost against its
    diagonal), then the reference's exit -- its warning text and the SVD projections
    (projections.py:101-108,181-187,236-287)."""
    A = np.array(cs.RANK_DEFICIENT[name], dtype=float)
    gold = qp_extra["rank_deficient"][name][kind]
    if name == "sum_row" and not np.linalg.svd(A, compute_uv=False)[-1] <= 1e-15:
        pytest.skip("this LAPACK leaves the rounding-level singular value above tol=1e-15")
    with pytest.warns(UserWarning) as rec:
        Z, LS, Y = ips.proj.projections(sps.csc_matrix(A) if kind == "sparse" else A)
    assert [str(w.message) for w in rec] == gold["warnings"]
    assert type(Z.projector).__name__ == "SVDProjector" and Z.projector.rank == 2
    for p, wz, wl in zip(cs.A38_POINTS_N[:3], gold["Z"], gold["LS"]):
        close(Z.dot(np.array(p, float)), wz)
        close(LS.dot(np.array(p, float)), wl, 1e-9)
    for p, wy in zip(cs.A38_POINTS_M, gold["Y"]):
        close(Y.dot(np.array(p, float)), wy)
    # and a projected-CG solve on top of the fallback operators (general driver)
    import oracle
    H = np.diag(np.arange(1.0, 9.0))
    c = np.arange(8.0) - 3.0
    b = A.dot(np.ones(8))
    with pytest.warns(UserWarning):
        Zo, _, Yo = oracle.projections(sps.csc_matrix(A) if kind == "sparse" else A)
    xo, io = oracle.projected_cg(sps.csr_matrix(H), c, Zo, Yo, b, tol=1e-20)
    x, info = ips.qp.projected_cg(ips.dv.DeviceCSR.from_scipy(sps.csr_matrix(H)), c, Z, Y, b,
                                  tol=1e-20)
    assert info["stop_cond"] == io["stop_cond"]
    close(x, xo, 1e-9)


def test_orthogonality_golden(ips, qp_small):
    """projector.orthogonality against the reference's values (projections.py:23-55)."""
    A38 = np.array(cs.A38, dtype=float)
    for v, want in zip(cs.ORTH_VECTORS, qp_small["orth"]):
        for A in (A38, sps.csc_matrix(A38)):
            assert abs(ips.proj.orthogonality(A, np.array(v)) - want) < 1e-15
    assert ips.proj.orthogonality(A38, np.zeros(8)) == 0
    # an O(1) value: 1e-10 relative
    rng = np.random.default_rng(0)
    inst = BandedInstance(2000, 200)
    g = rng.standard_normal(2000)
    want = np.linalg.norm(inst.A.dot(g)) / (sps.linalg.norm(inst.A) * np.linalg.norm(g))
    got = ips.proj.orthogonality(ips.dv.DeviceCSR.from_scipy(inst.A), g)
    assert abs(got - want) <= 1e-12 * want


@pytest.mark.parametrize("n,m", [(2000, 200), (20000, 2000), (300000, 30000)])
def test_orthogonality_from_the_normal_equation_residual(ips, n, m):
    """The fused loop takes ||A g||^2 (orthogonality, projections.py:52) from the residual of
    the normal equations, ||w - (A A') v||^2 with w = A r, g = r - A'v, instead of a second
    product by A.  The identity A g = w - S v holds for ANY v, so it is checked with the
    residual kernel fed a deliberately wrong solve: the factorization is that of A, the band
    the residual is taken against is that of a perturbed A2."""
    import ctypes
    import torch
    from ipsolver import _hip
    rng = np.random.default_rng(n)
    inst = BandedInstance(n, m)
    A = inst.A.tocsr()
    A2 = A.copy()
    A2.data = A.data * (1.0 + 0.05 * rng.standard_normal(A.nnz))
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    solver = ips.proj.BandedNormalSolver(Ad)
    A2d = ips.dv.DeviceCSR(Ad.pattern, torch.from_numpy(A2.data).cuda())
    band2 = torch.empty_like(solver.band)
    pat = Ad.pattern
    _hip.call("ipx_aat_band_w", m, solver.k, ips.dv._p(pat.indptr), ips.dv._p(pat.indices),
              ips.dv._p(A2d.val), None, None, ips.dv._p(band2), ips.dv.stream_ptr())
    solver.band.copy_(band2)                      # residual against S2 = A2 A2'
    # (the chunk-recurrence form of the solve: the cyclic-reduction form has no factorization
    # to go stale -- its residual is checked in test_banded_solve_by_parallel_cyclic_reduction)
    _hip.load().ipx_banded_set_decoupling(ctypes.c_void_p(solver.handle), 2)
    r = rng.standard_normal(n)
    w = A2.dot(r)
    wd = ips.dv.DVec.from_host(w)
    v = torch.empty(m, dtype=torch.float64, device="cuda")
    part = torch.zeros((m + 255) // 256 + 1, dtype=torch.float64, device="cuda")
    npart = ctypes.c_int32(0)
    _hip.call("ipx_banded_solve_resid", ctypes.c_void_p(solver.handle), ips.dv._p(wd.t),
              ips.dv._p(v), ips.dv._p(part), ctypes.byref(npart), None, ips.dv.stream_ptr())
    got = float(np.sum(part.cpu().numpy()[:npart.value]))
    vh = v.cpu().numpy()
    g = r - A2.T.dot(vh)
    want = float(np.sum(A2.dot(g) ** 2))          # ||A2 g||^2, the reference's form
    assert want > 1e-6 * np.sum(w ** 2)           # a real residual, not rounding noise
    assert abs(got - want) <= 1e-10 * want


@pytest.mark.parametrize("m,k,chunk", [(1, 1, 64), (5, 1, 64), (70, 1, 8), (1000, 1, 16),
                                       (1000, 2, 16), (777, 3, 12), (4000, 4, 32),
                                       (100000, 1, 64), (30000, 2, 64)])
def test_banded_solver(ips, m, k, chunk):
    """(A A')^-1 against scipy's sparse LU on random SPD band matrices built
    as A A' from a random banded A."""
    import ctypes
    import torch
    from ipsolver import _hip
    rng = np.random.default_rng(m + k)
    n = 3 * m + 3 * k + 1
    # row i touches columns [3i, 3i + 3k] -> rows i, i+k share a column: half bandwidth k
    rows, cols, vals = [], [], []
    for d in range(3 * k + 1):
        rows.append(np.arange(m))
        cols.append(3 * np.arange(m) + d)
        vals.append(rng.standard_normal(m))
    A = sps.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                       shape=(m, n))
    A.sort_indices()
    S = sps.csc_matrix(A.dot(A.T))
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    solver = ips.proj.BandedNormalSolver(Ad, chunk=chunk)
    assert solver.k == k
    w = rng.standard_normal(m)
    got = solver.solve(ips.dv.DVec.from_host(w)).to_host()
    want = sps.linalg.splu(S).solve(w) if m > 1 else w / S[0, 0]
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err < 1e-11, err
    # bitwise reproducible
    again = solver.solve(ips.dv.DVec.from_host(w)).to_host()
    assert np.array_equal(got, again)
    # the LDS-resident three-launch path against the one-kernel-per-level sweep
    W = ips.dv.DVec.from_host(w)
    out = torch.empty(m, dtype=torch.float64, device="cuda")
    _hip.call("ipx_banded_solve_multilaunch", ctypes.c_void_p(solver.handle),
              ips.dv._p(W.t), ips.dv._p(out), ips.dv.stream_ptr())
    ref = out.cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref))


def test_banded_solver_with_reordering(ips):
    """Rows given in an order that makes A A' wide; RCM (symbolic, host) brings
    it back to a band and the permutation is folded into the solve."""
    rng = np.random.default_rng(5)
    m = 3000
    inst = BandedInstance(30000, m)
    shuffle = rng.permutation(m)
    A = sps.csr_matrix(inst.A[shuffle])
    A.sort_indices()
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    solver = ips.proj.BandedNormalSolver(Ad)
    assert solver.perm is not None and solver.k <= 3
    w = rng.standard_normal(m)
    got = solver.solve(ips.dv.DVec.from_host(w)).to_host()
    want = sps.linalg.splu(sps.csc_matrix(A.dot(A.T))).solve(w)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 1e-11


@pytest.mark.parametrize("size", ["n2000", "n20000"])
def test_banded_traces(ips, size, banded2000, banded20000):
    import oracle
    gold = banded2000 if size == "n2000" else banded20000
    n, m = (2000, 200) if size == "n2000" else (20000, 2000)
    inst = BandedInstance(n, m)
    s = int(gold["stride"][0])
    A = ips.dv.DeviceCSR.from_scipy(inst.A)
    H = ips.dv.DeviceCSR.from_scipy(inst.H)
    Z, LS, Y = ips.proj.projections(A)
    for p, w in zip(inst.probes_n, gold["Z"]):
        close(host(Z.dot(p))[::s], w)
    for p, w in zip(inst.probes_n, gold["LS"]):
        close(host(LS.dot(p))[::s], w)
    for p, w in zip(inst.probes_m, gold["Y"]):
        close(host(Y.dot(p))[::s], w)

    gnorm = float(gold["gnorm"][0])
    Zo, _, Yo = oracle.projections(inst.A)
    for name, kw in inst.pcg_variants(gnorm).items():
        want = gold["pcg_%s_info" % name]
        for return_all in (True, False):
            x, info = ips.qp.projected_cg(H, inst.c, Z, Y, np.zeros(m),
                                          return_all=return_all, **kw)
            assert [info["niter"], info["stop_cond"], int(info["hits_boundary"])] == list(want), name
            close(host(x)[::s], gold["pcg_%s_x" % name])
            if return_all:
                for a, w in zip(info["allvecs"], gold["pcg_%s_allvecs" % name]):
                    close(host(a)[::s], w)
        # full-vector check against the oracle on the same inputs
        xo, _ = oracle.projected_cg(inst.H, inst.c, Zo, Yo, np.zeros(m), **kw)
        close(x, xo)

    y_b = host(Y.dot(inst.b))
    x, info = ips.qp.projected_cg(H, inst.c, Z, Y, inst.b, tol=0, max_iter=10,
                                  trust_radius=10 * np.linalg.norm(y_b))
    assert [info["niter"], info["stop_cond"],
            int(info["hits_boundary"])] == list(gold["pcg_rowstart_info"])
    close(host(x)[::s], gold["pcg_rowstart_x"])

    for (radius, lo, hi), w in zip(inst.dogleg_cfg(y_b), gold["dogleg"]):
        got = ips.qp.modified_dogleg(A, Y, inst.b, radius, np.full(n, lo), np.full(n, hi))
        close(host(got)[::s], w)


def _sharded_problem(world, rank, n, m):
    from ipsolver import sharded
    inst = BandedInstance(n, m)
    A = inst.A.tocsr()
    lay = sharded.ShardLayout(A.indptr, A.indices, A.shape, world, rank)
    sh = sharded.Sharding(lay, sharded.ShardComm(), sharded.HipOps())
    return inst, sh, sharded.ShardCSR.from_global(sh, inst.A), \
        sharded.ShardHessian.from_global(sh, inst.H)


def test_sharded_fused_loop_single_rank(ips):
    """The partitioned sharded solver with the HIP kernels at world size 1 (same kernels, no
    neighbours, own-range partial sums = everything) against the single-GPU device loop:
    every exit of the loop, refinement included."""
    from ipsolver import sharded
    n, m = 20000, 2000
    inst, sh, A, H = _sharded_problem(1, 0, n, m)
    Z, LS, Y = sharded.projections(A)
    A1 = ips.dv.DeviceCSR.from_scipy(inst.A)
    H1 = ips.dv.DeviceCSR.from_scipy(inst.H)
    Z1, _, Y1 = ips.proj.projections(A1)
    c = sh.from_global(inst.c, "col")
    gnorm = ips.dv.norm(Z1.dot(inst.c))
    for name, kw in inst.pcg_variants(gnorm).items():
        kws = dict(kw)
        for key in ("lb", "ub"):
            if key in kws:
                kws[key] = sh.from_global(kws[key], "col")
        calls = sharded.STATS["fused_calls"]
        x, info = ips.qp.projected_cg(H, c, Z, Y, sh.zeros("row"), **kws)
        assert sharded.STATS["fused_calls"] == calls + 1, "device-resident sharded loop not taken"
        x1, info1 = ips.qp.projected_cg(H1, inst.c, Z1, Y1, np.zeros(m), **kw)
        assert info == info1, (name, info, info1)
        close(x.to_host(), host(x1), 1e-12)
    Zr, _, Yr = sharded.projections(A, orth_tol=1e-30, max_refin=2)
    ev = sharded.STATS["refine_events"]
    x, info = ips.qp.projected_cg(H, c, Zr, Yr, sh.zeros("row"), tol=0, max_iter=12)
    assert sharded.STATS["refine_events"] - ev >= 11
    Z2, _, Y2 = ips.proj.projections(A1, orth_tol=1e-30, max_refin=2)
    x1, info1 = ips.qp.projected_cg(H1, inst.c, Z2, Y2, np.zeros(m), tol=0, max_iter=12)
    assert info == info1
    close(x.to_host(), host(x1), 1e-12)


def _multi_rank_worker(rank, world, port, out_path, transport="dist", n=20000, m=2000):
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ip-nonlinear-solver_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["IPX_SHARD_TRANSPORT"] = transport.split("-")[0]
    forms = {"ipc-pack": "pack-comm,no-resident", "ipc": "no-resident"}.get(transport)
    if forms:
        os.environ["IPX_DEBUG_FORMS"] = forms
    else:
        os.environ.pop("IPX_DEBUG_FORMS", None)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ipsolver import sharded, qp
        torch.set_num_threads(1)
        inst, sh, A, H = _sharded_problem(world, rank, n, m)
        # no torch.distributed call may happen between the boundaries of a batch of the
        # device loop when it runs on the peer mailboxes
        leaks = []
        plain_iterate = sharded.FusedShardedCG.iterate

        def watched(self, it_begin, it_end):
            keys = ("all_reduce", "exchange")
            before = [self.sh.comm.stats[k] for k in keys]
            plain_iterate(self, it_begin, it_end)
            after = [self.sh.comm.stats[k] for k in keys]
            if self.mailbox is not None and after != before:
                leaks.append((before, after))
        sharded.FusedShardedCG.iterate = watched
        Z, LS, Y = sharded.projections(A)
        c = sh.from_global(inst.c, "col")
        out = {}
        gnorm = float(np.sqrt(Z.dot(c).sumsq_amax()[0]))
        for name, kw in inst.pcg_variants(gnorm).items():
            kws = dict(kw)
            for key in ("lb", "ub"):
                if key in kws:
                    kws[key] = sh.from_global(kws[key], "col")
            x, info = qp.projected_cg(H, c, Z, Y, sh.zeros("row"), **kws)
            out["pcg_%s_x" % name] = x.to_host()
            out["pcg_%s_info" % name] = np.array([info["niter"], info["stop_cond"],
                                                  int(info["hits_boundary"])])
        Zr, _, Yr = sharded.projections(A, orth_tol=1e-30, max_refin=2)
        x, info = qp.projected_cg(H, c, Zr, Yr, sh.zeros("row"), tol=0, max_iter=15)
        out["refine_x"] = x.to_host()
        out["stats"] = np.array([sharded.STATS[k] for k in ("fused_calls", "box_events",
                                                            "refine_events")]
                                + [sh.comm.stats["exchange"]])
        out["ipc"] = np.array([float(sh.transport == "ipc"), sh.comm.stats["ipc_batches"],
                               sh.comm.stats["ipc_iterations"], len(leaks)]
                              + list(sh.mailbox().sequence() if sh.mailbox() else (0, 0))
                              + [sh.mailbox().fused_launches() if sh.mailbox() else 0])
        out["resident"] = np.array([sharded.STATS["resident_batches"],
                                    sharded.STATS["resident_halo_syncs"],
                                    sh.mailbox().resident_launches() if sh.mailbox() else 0])
        flags = torch.tensor([float(sharded.STATS["fused_calls"])])
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)          # engaged on every rank?
        out["fused_min"] = flags.numpy()
        if rank == 0:
            np.savez(out_path, **out)
    finally:
        dist.destroy_process_group()


def _wide_band_worker(rank, world, port, out_path):
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ip-nonlinear-solver_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ipsolver import sharded, qp
        import ipsolver.device as dv
        import ipsolver.projector as proj
        from ipsolver.operators import DeviceHessian
        from ipsolver.synthetic import CenteredBandedNLP
        n, m = 35184, 4398                       # 8 columns per row step: A A' has half bandwidth 2
        prob = CenteredBandedNLP(n, m, seed=2)
        x = prob.x0
        v = 0.1 * np.random.default_rng(2).standard_normal(m)
        A_h, H_h = prob.constr_jac(x).tocsr(), prob.hess(x)
        hd = prob.kappa * prob.Wt.dot(v)
        c_h = prob.grad(x)
        lay = sharded.ShardLayout(A_h.indptr, A_h.indices, A_h.shape, world, rank)
        sh = sharded.Sharding(lay, sharded.ShardComm(), sharded.HipOps())
        A = sharded.ShardCSR.from_global(sh, A_h)
        H = sharded.ShardHessian.from_global(sh, H_h, hd)
        Z, LS, Y = sharded.projections(A)
        before = sharded.STATS["fused_calls"]
        c = sh.from_global(c_h, "col")
        out = {"k": np.array([sharded._banded_of(Z.projector).k]),
               "loop": np.array([float(sharded.fused_supports(H, Z, Y))])}
        A1 = dv.DeviceCSR.from_scipy(A_h)
        H1 = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(H_h), diag=dv.DVec.from_host(hd))
        Z1, _, Y1 = proj.projections(A1)
        cases = {"free": dict(tol=0, max_iter=20),
                 "box": dict(tol=0, max_iter=20, lb=np.full(n, -0.02), ub=np.full(n, 0.03))}
        for name, kw in cases.items():
            ks = {a: (sh.from_global(b, "col") if a in ("lb", "ub") else b) for a, b in kw.items()}
            xs, info = qp.projected_cg(H, c, Z, Y, sh.zeros("row"), **ks)
            x1, info1 = qp.projected_cg(H1, c_h, Z1, Y1, np.zeros(m), **kw)
            out[name] = np.concatenate(([info["niter"], info["stop_cond"], info1["niter"],
                                         info1["stop_cond"]], xs.to_host() - x1.to_host(),
                                        [np.max(np.abs(x1.to_host()))]))
        out["fused_calls"] = np.array([sharded.STATS["fused_calls"] - before])
        if rank == 0:
            np.savez(out_path, **out)
    finally:
        dist.destroy_process_group()


def test_sharded_wider_band_takes_the_general_driver(tmp_path, ips):
    """A Jacobian whose A A' has half bandwidth 2: the banded factorization then works in chunks
    of 67 rows, its workgroups (268 rows) do not follow the layout's 260-row blocks, and the
    device-resident sharded loop -- which sums per-workgroup partials over a rank's own rows --
    does not apply.  ``qp.projected_cg`` then runs the general driver on the same distributed
    vectors (it used to raise NotImplementedError): same iterates as one GPU."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "wide.npz")
    mp.spawn(_wide_band_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert int(got["k"][0]) == 2 and got["loop"][0] == 0.0 and int(got["fused_calls"][0]) == 0
    for name in ("free", "box"):
        r = got[name]
        assert list(r[:2]) == list(r[2:4]), name
        assert np.max(np.abs(r[4:-1])) <= 1e-11 * r[-1], name


def _desync_worker(rank, world, port, out_path, mode="lone", n=20000, m=2000):
    import os
    import sys
    import warnings
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ip-nonlinear-solver_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["IPX_SHARD_TRANSPORT"] = "ipc"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ipsolver import sharded, qp, _hip
        torch.set_num_threads(1)
        inst, sh, A, H = _sharded_problem(world, rank, n, m)
        Z, LS, Y = sharded.projections(A)
        c = sh.from_global(inst.c, "col")
        x, info = qp.projected_cg(H, c, Z, Y, sh.zeros("row"), tol=0, max_iter=10)
        ok_before = sh.transport == "ipc"
        sh.mailbox().set_timeout(0.5)       # (default 10 s: keep the test short)
        lone = 0
        if mode == "lone" and rank == 0:    # one rank falls out of step: an all-reduce alone
            try:
                sh.mailbox().allreduce([1.0])
            except _hip.IpxError:
                lone = 1
        if mode == "resident-lone" and rank == 0:
            # the resident form: ONE rank enqueues a batch twice.  The first launch pairs with
            # the peer's; the second finds no partner, its hop 0 times out (stop code 7, nothing
            # written back), and this rank's tags run ahead of the peer's from then on
            plain = sharded.FusedShardedCG.iterate
            fired = []

            def iterate(self, it_begin, it_end):
                plain(self, it_begin, it_end)
                if self.resident and not fired:
                    fired.append(1)
                    plain(self, it_begin, it_end)
            sharded.FusedShardedCG.iterate = iterate
            lone = 1
        if mode == "resident-lone" and rank != 0:
            lone = 0
        if mode == "one-sided" and rank == 0:
            # the LAST communicating launch before a host read times out on rank 0
            # only -- rank 1 completes it and sees an ordinary end of the subproblem.  Emulated
            # by writing the stop code into rank 0's state block after the batch.
            plain = sharded.FusedShardedCG.iterate

            def iterate(self, it_begin, it_end):
                plain(self, it_begin, it_end)
                if self.mailbox is not None and it_end == 10:
                    self.L.state[sharded.ST_STOP] = 7.0
            sharded.FusedShardedCG.iterate = iterate
            lone = 1
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            x2, info2 = qp.projected_cg(H, c, Z, Y, sh.zeros("row"), tol=0, max_iter=10)
        warned = any("falling back to the torch.distributed transport" in str(w.message)
                     for w in caught)
        x3, info3 = qp.projected_cg(H, c, Z, Y, sh.zeros("row"), tol=0, max_iter=10)
        flags = torch.tensor([float(ok_before), float(warned), float(sh.transport == "dist")])
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        if rank == 0:
            np.savez(out_path, x=x.to_host(), x2=x2.to_host(), x3=x3.to_host(),
                     flags=flags.numpy(), lone=np.array([lone]),
                     niter=np.array([info["niter"], info2["niter"], info3["niter"]]))
        else:
            x.to_host(), x2.to_host(), x3.to_host()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["lone", "one-sided", "resident-lone"])
def test_sharded_loop_survives_a_rank_out_of_step(mode, tmp_path, ips):
    """A peer that falls out of step (``lone``: one rank issues a mailbox all-reduce on its own,
    so its sequence numbers run one ahead) makes the waits of the device loop time out -- after
    the mailbox's deadline, on every rank, with stop code 7 instead of a hung GPU.  The group then gives the mailbox
    transport up together, warns, and solves the SAME subproblem again through torch.distributed:
    same iterates as before the incident.  ``one-sided``: only ONE rank sees the timeout (in the
    last launch before a host read, which the other rank completes normally); the ranks agree on
    it before either acts on its state block, and fall back together all the same.
    ``resident-lone``: the same for the resident form of the loop (one launch per rank and batch,
    csrc/resident.hip): one rank's tags run ahead, the workgroups' waits time out on both ranks,
    nothing is written back, the group falls back to the separate launches over
    torch.distributed."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "desync.npz")
    mp.spawn(_desync_worker, args=(2, port, out, mode), nprocs=2, join=True)
    got = np.load(out)
    assert list(got["flags"]) == [1.0, 1.0, 1.0] and int(got["lone"][0]) == 1
    assert list(got["niter"]) == [10, 10, 10]
    close(got["x2"], got["x"], 1e-13)
    close(got["x3"], got["x"], 1e-13)


@pytest.mark.parametrize("transport", ["ipc-resident", "ipc", "ipc-pack", "dist"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fused_loop_multi_rank(world, transport, tmp_path, banded20000, ips):
    """The HIP kernels under the row partition: `world` processes share cuda:0.  The
    device-resident loop -- two all-reduces and one halo exchange of g per iteration --
    against the REFERENCE's golden traces (every exit of the loop: tolerance, trust region,
    box events), with refining projections against the oracle, and against the single-GPU
    device loop to 1e-12.

    transport "ipc": the scalars and the halo travel through the peer mailboxes (hipIpc-mapped
    device memory, csrc/peer.hip) inside the loop's own launches -- a batch of iterations is
    ONE C call and no torch.distributed call happens between its boundaries (asserted on
    ``ShardComm.stats``); the collectives are done in the prologues of the kernels that consume
    them (3 launches per iteration; "ipc-pack": in pack kernels of their own, 5 launches -- the
    form the problems with a box always take).  "ipc-resident" (what a group takes by itself when
    the problem qualifies: no box, tridiagonal A A'): a batch is ONE resident launch per rank
    (csrc/resident.hip, PEER form) -- the workgroups of all ranks hand their scalars and halos to
    each other directly, two hops per iteration; the host synchronises the halos of x, p, r, Hp
    when it leaves the loop.  transport "dist": three torch.distributed calls per iteration (over
    gloo here, staged through the host: RCCL refuses two ranks on one device)."""
    import socket
    import torch.multiprocessing as mp
    import oracle
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "x.npz")
    mp.spawn(_multi_rank_worker, args=(world, port, out, transport), nprocs=world, join=True)
    got, gold = np.load(out), banded20000
    st = int(gold["stride"][0])
    inst = BandedInstance(20000, 2000)
    for name in inst.pcg_variants(1.0):
        assert list(got["pcg_%s_info" % name]) == list(gold["pcg_%s_info" % name]), name
        close(got["pcg_%s_x" % name][::st], gold["pcg_%s_x" % name])
    Zo, _, Yo = oracle.projections(inst.A, "NormalEquation", orth_tol=1e-30, max_refin=2)
    xo, _ = oracle.projected_cg(inst.H, inst.c, Zo, Yo, np.zeros(2000), tol=0, max_iter=15)
    close(got["refine_x"], xo)
    fused_calls, box_events, refine_events, exchanges = got["stats"]
    assert got["fused_min"][0] >= 6 and box_events > 0 and refine_events >= 14
    is_ipc, ipc_batches, ipc_iterations, leaks, seq, hseq, fused = got["ipc"]
    if transport.startswith("ipc"):
        assert is_ipc == 1 and leaks == 0 and ipc_batches > 10 and ipc_iterations > 100
        if transport != "ipc-resident":      # (resident launches count their own tags)
            assert seq >= 2 * ipc_iterations and hseq >= ipc_iterations
        # the problems without a box ran with the collectives in the prologues of the loop's
        # own kernels (3 launches per iteration), the box ones on the pack kernels (5)
        res_batches, res_syncs, res_launches = got["resident"]
        if transport == "ipc":
            assert 100 <= fused < 2 * ipc_iterations and fused % 2 == 0
        elif transport == "ipc-resident":
            # the problems without a box ran as resident launches (one per batch), the box ones
            # on the pack kernels
            assert res_batches >= 6 and res_launches == res_batches and 0 < res_syncs <= res_batches
            assert fused == 0
        else:
            assert fused == 0
        if transport != "ipc-resident":
            assert res_batches == 0 and res_launches == 0
    else:
        assert is_ipc == 0 and ipc_batches == 0 and exchanges > 100
    # the same subproblems on the single-GPU device loop
    A1 = ips.dv.DeviceCSR.from_scipy(inst.A)
    H1 = ips.dv.DeviceCSR.from_scipy(inst.H)
    Z1, _, Y1 = ips.proj.projections(A1)
    gnorm = ips.dv.norm(Z1.dot(inst.c))
    for name, kw in inst.pcg_variants(gnorm).items():
        x1, info1 = ips.qp.projected_cg(H1, inst.c, Z1, Y1, np.zeros(2000), **kw)
        assert list(got["pcg_%s_info" % name]) == [info1["niter"], info1["stop_cond"],
                                                    int(info1["hits_boundary"])]
        close(got["pcg_%s_x" % name], host(x1), 1e-12)


@pytest.mark.parametrize("transport", ["ipc-resident", "ipc"])
def test_sharded_fused_loop_eight_ranks(transport, tmp_path, ips):
    """The rank count of the target node: EIGHT processes share cuda:0 (n = 100000, m = 10000:
    39 blocks of 260 rows, 4 or 5 per rank, six interior ranks with both neighbours; every
    workgroup of every rank co-resident for the resident PEER form -- one per compute unit).
    Every exit of the loop (tolerance, trust region, box events) and the refining projections
    against the single-GPU device loop to 1e-12 and against the oracle; the mailbox tables for
    eight peers, the rank-ordered folds over eight contributions, seven neighbour pairs of halo
    slots."""
    import socket
    import torch.multiprocessing as mp
    import oracle
    n, m, world = 100000, 10000, 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "x8.npz")
    mp.spawn(_multi_rank_worker, args=(world, port, out, transport, n, m), nprocs=world, join=True)
    got = np.load(out)
    inst = BandedInstance(n, m)
    A1 = ips.dv.DeviceCSR.from_scipy(inst.A)
    H1 = ips.dv.DeviceCSR.from_scipy(inst.H)
    Z1, _, Y1 = ips.proj.projections(A1)
    gnorm = ips.dv.norm(Z1.dot(inst.c))
    for name, kw in inst.pcg_variants(gnorm).items():
        x1, info1 = ips.qp.projected_cg(H1, inst.c, Z1, Y1, np.zeros(m), **kw)
        assert list(got["pcg_%s_info" % name]) == [info1["niter"], info1["stop_cond"],
                                                    int(info1["hits_boundary"])], name
        close(got["pcg_%s_x" % name], host(x1), 1e-12)
    Zo, _, Yo = oracle.projections(inst.A, "NormalEquation", orth_tol=1e-30, max_refin=2)
    xo, _ = oracle.projected_cg(inst.H, inst.c, Zo, Yo, np.zeros(m), tol=0, max_iter=15)
    close(got["refine_x"], xo)
    fused_calls, box_events, refine_events, exchanges = got["stats"]
    assert got["fused_min"][0] >= 6 and box_events > 0 and refine_events >= 14
    is_ipc, ipc_batches, ipc_iterations, leaks, seq, hseq, fused = got["ipc"]
    assert is_ipc == 1 and leaks == 0 and ipc_batches > 10 and ipc_iterations > 100
    res_batches, res_syncs, res_launches = got["resident"]
    if transport == "ipc-resident":
        assert res_batches >= 6 and res_launches == res_batches and fused == 0
    else:
        assert res_batches == 0 and 100 <= fused < 2 * ipc_iterations


def test_eight_ranks_survive_a_rank_out_of_step(tmp_path, ips):
    """``resident-lone`` of the test above on EIGHT ranks: one rank enqueues a resident batch
    twice, its tags run ahead, the waits of all eight ranks' workgroups time out (stop code 7,
    nothing written back), the group agrees and solves the subproblem again through
    torch.distributed: the iterates of before the incident."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "desync8.npz")
    mp.spawn(_desync_worker, args=(8, port, out, "resident-lone", 100000, 10000), nprocs=8,
             join=True)
    got = np.load(out)
    assert list(got["flags"]) == [1.0, 1.0, 1.0] and int(got["lone"][0]) == 1
    assert list(got["niter"]) == [10, 10, 10]
    close(got["x2"], got["x"], 1e-13)
    close(got["x3"], got["x"], 1e-13)


def _refusal_worker(rank, world, port, out_path):
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "ip-nonlinear-solver_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ipsolver import sharded
        from test_sharded_gloo import _slow_decay_jacobian
        out = {}
        for name, delta in (("fast", 0.5), ("slow", 0.01)):
            A = _slow_decay_jacobian(1300, delta)
            lay = sharded.ShardLayout(A.indptr, A.indices, A.shape, world, rank)
            sh = sharded.Sharding(lay, sharded.ShardComm(), sharded.HipOps())
            try:
                Z, LS, Y = sharded.projections(sharded.ShardCSR.from_global(sh, A))
                w = sh.from_global(np.ones(1300), "row")
                out[name] = np.concatenate(([0.0], Y.dot(w).to_host()))
            except NotImplementedError:
                out[name] = np.array([1.0])
        if rank == 0:
            np.savez(out_path, **out)
    finally:
        dist.destroy_process_group()


def test_sharded_projections_refuse_a_slowly_decaying_inverse(tmp_path, ips):
    """Two processes on the HIP kernels: a Jacobian whose (A A')^-1 decays across the 260-row
    halo is solved (row-space operator = the global one); one whose inverse does not
    (0.07 across the halo) is refused by the projector itself -- from the decoupling the
    device factorization measures -- on every rank, whoever the caller is."""
    import torch.multiprocessing as mp
    import scipy.sparse.linalg as spla
    from test_sharded_gloo import _slow_decay_jacobian, _free_port
    out = str(tmp_path / "refuse.npz")
    mp.spawn(_refusal_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert got["slow"][0] == 1.0 and got["fast"][0] == 0.0
    A = _slow_decay_jacobian(1300, 0.5)
    want = A.T @ spla.splu(sps.csc_matrix(A @ A.T)).solve(np.ones(1300))       # Y = A'(AA')^-1
    close(got["fast"][1:], want, 1e-12)


@pytest.mark.parametrize("n,m", [(400, 40), (6000, 600)])
def test_box_schur_solver(ips, n, m):
    """(A A')^-1 with the box rows eliminated analytically (csrc/boxschur.hip)
    against a direct sparse solve, on the barrier problem's augmented Jacobian
    [[J, diag(s)], [-I, diag(s_l)], [+I, diag(s_u)]] (BASELINE config 5 shape)."""
    from ipsolver.boxschur import BoxSchurNormalSolver, analysis_for
    rng = np.random.default_rng(n)
    inst = BandedInstance(n, m)
    J = inst.A
    s = rng.uniform(0.05, 2.0, m + 2 * n)
    I = sps.eye(n, format="csr")
    A = sps.bmat([[J, sps.diags(s[:m]), None, None],
                  [-I, None, sps.diags(s[m:m + n]), None],
                  [I, None, None, sps.diags(s[m + n:])]], format="csr")
    A.sort_indices()
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    an = analysis_for(Ad.pattern)
    assert an.n_simple == 2 * n and len(an.general) == m and np.all(an.rowq >= 0)
    solver = BoxSchurNormalSolver(Ad)
    assert solver.inner.k == 1                      # Schur complement stays tridiagonal
    G = sps.csc_matrix(A.dot(A.T))
    lu = sps.linalg.splu(G)
    for _ in range(3):
        w = rng.standard_normal(m + 2 * n)
        got = solver.solve(ips.dv.DVec.from_host(w)).to_host()
        want = lu.solve(w)
        assert np.max(np.abs(got - want)) <= 1e-11 * np.max(np.abs(want))
    # and through the public seam: projections() picks this solver
    Z, LS, Y = ips.proj.projections(Ad)
    assert type(Z.projector.solver).__name__ == "BoxSchurNormalSolver"
    x = rng.standard_normal(A.shape[1])
    z = host(Z.dot(x))
    assert np.max(np.abs(A.dot(z))) <= 1e-9 * np.max(np.abs(x))


def test_unbounded_trust_region_skips_the_norm(ips, monkeypatch):
    """trust_radius=inf without a box: ``norm(x_next) >= trust_radius`` (qp_subproblem.py:583)
    cannot be True, the device loop does not form the norm (csrc/cg.hip no_xn2).  Same
    iterates, bit for bit, as with the norm formed; and a finite radius still stops."""
    n, m = 20000, 2000
    inst = BandedInstance(n, m)
    A = ips.dv.DeviceCSR.from_scipy(inst.A)
    H = ips.dv.DeviceCSR.from_scipy(inst.H)
    Z, LS, Y = ips.proj.projections(A)
    b = np.zeros(m)
    runs = []
    for keep in ("", "1"):
        if keep:
            monkeypatch.setenv("IPX_DEBUG_FORMS", "keep-xn2")
        else:
            monkeypatch.delenv("IPX_DEBUG_FORMS", raising=False)
        x, info = ips.qp.projected_cg(H, inst.c, Z, Y, b, tol=1e-12)
        runs.append((host(x), info))
    assert runs[0][1] == runs[1][1] and np.array_equal(runs[0][0], runs[1][0])
    monkeypatch.delenv("IPX_DEBUG_FORMS", raising=False)
    radius = 0.5 * float(np.linalg.norm(runs[0][0]))
    x, info = ips.qp.projected_cg(H, inst.c, Z, Y, b, trust_radius=radius, tol=1e-12)
    assert info["stop_cond"] == 2 and info["hits_boundary"]
    assert abs(np.linalg.norm(host(x)) - radius) <= 1e-12 * radius


@pytest.mark.parametrize("bounds", ["all", "ragged", "scaled", "all-large"])
def test_box_schur_projection_without_matrix_rows(ips, bounds, monkeypatch):
    """The CG loop's projection for barrier problems (csrc/boxschur.hip ipx_boxschur_project,
    csrc/cg.hip k_cg_step1_box): g = r - A'(A A')^-1 A r with the box rows handled per group
    on r itself -- never multiplied as rows of A or A'.  Against the operator built from the
    SpMVs and ipx_boxschur_solve (same formulas: 1e-13), against a direct sparse solve, with
    every variable bounded on both sides and with a ragged mix (lower only, upper only, both,
    none: single-row groups and x-columns outside every group), and with bound rows whose
    entries are not +-1 ("scaled": the compact group table must be refused); the compact
    tables (ipx_boxschur_args.grp2, yell_*) against the full ones, bit for bit; then a whole
    projected-CG run with slack bounds through the device loop against the oracle.
    "all-large": enough general rows for the cyclic-reduction solve, which then forms the Schur
    right-hand side A_R u itself (one launch less; the SpMV's scratch stays untouched)."""
    import ctypes
    import torch
    import oracle
    import ipsolver.cg_fused as cg_fused
    from ipsolver import _hip
    from ipsolver.boxschur import BoxSchurNormalSolver
    n, m = (30000, 3000) if bounds == "all-large" else (6000, 600)
    large, bounds = bounds == "all-large", bounds.split("-")[0]
    rng = np.random.default_rng(7)
    inst = BandedInstance(n, m)
    J = inst.A
    if bounds in ("all", "scaled"):
        L = U = np.arange(n)
    else:
        kind = rng.integers(0, 4, n)                    # 0 none, 1 lower, 2 upper, 3 both
        L, U = np.flatnonzero(kind & 1), np.flatnonzero(kind & 2)
    nl, nu = len(L), len(U)
    I = sps.eye(n, format="csr")
    s = rng.uniform(1e-6, 2.0, m + nl + nu)             # some slacks of (nearly) active bounds
    fl, fu = (2.0, 3.0) if bounds == "scaled" else (1.0, 1.0)
    A = sps.bmat([[J, sps.diags(s[:m]), None, None],
                  [-fl * I[L], None, sps.diags(s[m:m + nl]), None],
                  [fu * I[U], None, None, sps.diags(s[m + nl:])]], format="csr")
    A.sort_indices()
    N, M = A.shape[1], A.shape[0]
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    Z, LS, Y = ips.proj.projections(Ad)
    solver = Z.projector.solver
    assert isinstance(solver, BoxSchurNormalSolver)
    args = solver.c_args()
    assert args is not None and args.gcol and args.ngen == N - (n if bounds != "ragged" else
                                                                 len(np.union1d(L, U))) - nl - nu
    assert bool(args.grp2) == (bounds != "scaled") and args.yell_col and args.yell_val
    assert bool(args.gaffine) == (bounds == "all")      # computed columns: no gcol reads
    r = rng.standard_normal(N)
    rd = ips.dv.DVec.from_host(r)
    lib = _hip.load()
    nblk = lib.ipx_boxschur_project_count(ctypes.byref(args))
    g = torch.empty(N, dtype=torch.float64, device="cuda")
    pg = torch.zeros(2 * nblk, dtype=torch.float64, device="cuda")
    pres = torch.zeros(M // 256 + 2, dtype=torch.float64, device="cuda")
    n3, n4 = ctypes.c_int32(0), ctypes.c_int32(0)
    assert args.AR_rowlen == 16                   # 15 entries of the band + the row's slack
    solver._scratch[1].fill_(float("nan"))        # the scratch an SpMV A_R u would write
    _hip.call("ipx_boxschur_project", ctypes.byref(args), ips.dv._p(rd.t), ips.dv._p(g),
              ips.dv._p(pg), ctypes.byref(n3), ips.dv._p(pres), ctypes.byref(n4), None,
              ips.dv.stream_ptr())
    got = g.cpu().numpy()
    if large:      # the right-hand side was formed inside the solve kernel
        assert bool(torch.isnan(solver._scratch[1]).all())
    # the SpMV form with the same solver
    v = solver.solve(Ad.dot(rd))
    ref_dev = host(Ad.rmatvec_sub(v, rd))
    err = np.max(np.abs(got - ref_dev)) / np.max(np.abs(r))
    # (entries 2 and 3 next to slacks of 1e-6: B^-1 is ~10x larger, 3.4e-12 measured)
    # (and 4e-13 at five times the size: slacks of 1e-6 on 30000 variables)
    assert err <= (1e-11 if bounds == "scaled" else 1e-12 if large else 1e-13), err
    # a direct sparse solve
    lu = sps.linalg.splu(sps.csc_matrix(A @ A.T))
    want = r - A.T @ lu.solve(A @ r)
    assert np.max(np.abs(got - want)) <= 1e-10 * np.max(np.abs(r))
    assert n3.value == nblk
    assert abs(float(pg[:nblk].sum()) - float(got @ got)) <= 1e-12 * float(got @ got)
    # in place
    g2 = rd.t.clone()
    _hip.call("ipx_boxschur_project", ctypes.byref(args), ips.dv._p(g2), ips.dv._p(g2),
              ips.dv._p(pg), ctypes.byref(n3), ips.dv._p(pres), ctypes.byref(n4), None,
              ips.dv.stream_ptr())
    assert np.array_equal(g2.cpu().numpy(), got)
    # round 5: with the compact tables and the cyclic-reduction solve the per-item back
    # substitution is the TAIL of the solve's kernel (3 launches instead of 4).  Against the
    # separate launch: g bit for bit (same expressions), ||g||^2 partials per workgroup of the
    # solve instead of per block of items (their sums agree to rounding)
    assert bool(args.post_own_g) == (bounds == "all")      # (used when the solve is the cyclic reduction)
    pg_unfused = pg
    if args.post_own_g:
        monkeypatch.setenv("IPX_DEBUG_FORMS", "no-post-tail")
        Zn, _, _ = ips.proj.projections(ips.dv.DeviceCSR.from_scipy(A))
        nargs = Zn.projector.solver.c_args()
        assert not nargs.post_own_g
        g5, pg5 = torch.empty_like(g), torch.zeros_like(pg)
        _hip.call("ipx_boxschur_project", ctypes.byref(nargs), ips.dv._p(rd.t), ips.dv._p(g5),
                  ips.dv._p(pg5), ctypes.byref(n3), ips.dv._p(pres), ctypes.byref(n4), None,
                  ips.dv.stream_ptr())
        bad = np.flatnonzero(g5.cpu().numpy() != got)
        assert bad.size == 0, (bad[:10], bad.size, np.max(np.abs(g5.cpu().numpy() - got)))
        assert n3.value == nblk
        assert abs(float(pg5[:nblk].sum()) - float(pg[:nblk].sum())) <= 1e-14 * float(pg5[:nblk].sum())
        assert float(pg[nblk:].abs().sum()) == 0.0
        pg_unfused = pg5
        monkeypatch.delenv("IPX_DEBUG_FORMS")
    # the full tables (4 doubles per group, the columns of A_R through their row pointers)
    monkeypatch.setenv("IPX_DEBUG_FORMS", "no-compact-groups")
    Zf, _, Yf = ips.proj.projections(ips.dv.DeviceCSR.from_scipy(A))
    fargs = Zf.projector.solver.c_args()
    assert not fargs.grp2 and not fargs.yell_col and not fargs.yell_val
    g3, pg3 = torch.empty_like(g), torch.zeros_like(pg)
    _hip.call("ipx_boxschur_project", ctypes.byref(fargs), ips.dv._p(rd.t), ips.dv._p(g3),
              ips.dv._p(pg3), ctypes.byref(n3), ips.dv._p(pres), ctypes.byref(n4), None,
              ips.dv.stream_ptr())
    assert np.array_equal(g3.cpu().numpy(), got) and torch.equal(pg3, pg_unfused)
    # ... and the compact coefficients with the column table READ instead of computed
    monkeypatch.setenv("IPX_DEBUG_FORMS", "no-affine-groups")
    Za, _, _ = ips.proj.projections(ips.dv.DeviceCSR.from_scipy(A))
    aargs = Za.projector.solver.c_args()
    assert not aargs.gaffine
    g4, pg4 = torch.empty_like(g), torch.zeros_like(pg)
    _hip.call("ipx_boxschur_project", ctypes.byref(aargs), ips.dv._p(rd.t), ips.dv._p(g4),
              ips.dv._p(pg4), ctypes.byref(n3), ips.dv._p(pres), ctypes.byref(n4), None,
              ips.dv.stream_ptr())
    assert np.array_equal(g4.cpu().numpy(), got) and torch.equal(pg4, pg_unfused)
    monkeypatch.delenv("IPX_DEBUG_FORMS")
    # the device loop on the barrier-shaped subproblem: bounds on the slacks only
    Hz = sps.block_diag([inst.H, sps.diags(rng.uniform(0.5, 2.0, N - n))], format="csr")
    Hd = ips.dv.DeviceCSR.from_scipy(Hz)
    c = rng.standard_normal(N)
    b = np.zeros(M)
    lb = np.concatenate((np.full(n, -np.inf), np.full(N - n, -0.995)))
    assert cg_fused.supports(Hd, Z, Y)
    before = dict(cg_fused.STATS)
    x, info = cg_fused.projected_cg(Hd, ips.dv.DVec.from_host(c), Z, Y, ips.dv.DVec.from_host(b),
                                    trust_radius=5.0, lb=ips.dv.DVec.from_host(lb), tol=1e-10)
    assert cg_fused.STATS["calls"] == before["calls"] + 1
    Zo, _, Yo = oracle.projections(A)
    xo, io = oracle.projected_cg(Hz, c, Zo, Yo, b, trust_radius=5.0, lb=lb,
                                 ub=np.full(N, np.inf), tol=1e-10)
    assert (info["stop_cond"], info["hits_boundary"]) == (io["stop_cond"], io["hits_boundary"])
    assert info["niter"] == io["niter"]
    close(x, xo, 1e-10)
    # ... and with the full tables: the same iterates bit for bit
    xf, inf_ = cg_fused.projected_cg(Hd, ips.dv.DVec.from_host(c), Zf, Yf,
                                     ips.dv.DVec.from_host(b), trust_radius=5.0,
                                     lb=ips.dv.DVec.from_host(lb), tol=1e-10)
    assert inf_ == info
    if args.post_own_g:        # (||g||^2 summed per workgroup of the solve: beta to rounding)
        close(host(xf), host(x), 1e-12)
    else:
        assert np.array_equal(host(xf), host(x))


@pytest.mark.parametrize("variant", ["plain", "sphere", "box"])
def test_step2_fused_into_hp_is_bit_identical(ips, variant, monkeypatch):
    """For banded Hessians step2 runs inside the H.p SpMV (k_cg_step2_hp); it uses the
    same expressions in the same order as the two separate kernels, so whole CG
    runs must agree bit for bit, through every kind of exit."""
    import ipsolver.cg_fused as cg_fused
    n, m = 20000, 2000
    inst = BandedInstance(n, m)
    A = ips.dv.DeviceCSR.from_scipy(inst.A)
    H = ips.dv.DeviceCSR.from_scipy(inst.H)
    assert cg_fused.fuse_halo(H.pattern) == 1            # tridiagonal
    assert cg_fused.fuse_own(A.pattern) is not None      # banded Jacobian: step1 rides in A.r
    Z, LS, Y = ips.proj.projections(A)
    b = np.zeros(m)
    x_free, _ = ips.qp.projected_cg(H, inst.c, Z, Y, b, tol=1e-12)
    kw = {"plain": dict(tol=1e-12),
          "sphere": dict(trust_radius=0.5 * ips.dv.norm(x_free)),
          "box": dict(lb=np.full(n, -0.3 * np.max(np.abs(host(x_free)))),
                      ub=np.full(n, 0.3 * np.max(np.abs(host(x_free)))))}[variant]
    runs = []
    for no_fuse in ("", "1"):
        if no_fuse:
            monkeypatch.setenv("IPX_DEBUG_FORMS", "no-fuse")
        else:
            monkeypatch.delenv("IPX_DEBUG_FORMS", raising=False)
        x, info = ips.qp.projected_cg(H, inst.c, Z, Y, b, **kw)
        runs.append((host(x), info))
    (x1, i1), (x2, i2) = runs
    assert (i1["niter"], i1["stop_cond"], i1["hits_boundary"]) == \
        (i2["niter"], i2["stop_cond"], i2["hits_boundary"])
    if variant == "box":
        assert np.array_equal(x1, x2)           # step1 stays a separate launch with a box
    else:
        # the fused step1 sums ||x + alpha p||^2 per row tile of A instead of per vector
        # chunk: the iterates themselves are computed by identical expressions
        assert np.max(np.abs(x1 - x2)) <= 1e-13 * np.max(np.abs(x2))


@pytest.mark.parametrize("form", ["no-fuse"])
def test_debug_forms_of_the_three_launch_loop(ips, form, monkeypatch):
    """Every form the one debug switch (IPX_DEBUG_FORMS, ipsolver/_hip.py) can turn off has a
    parity test; here the one of the separate-launch loop at a size that is not resident
    (n = 1e6 rows are the bench's; 6e5 is the smallest): no fused kernels at all (1e-13:
    ||x + alpha p||^2 per vector chunk instead of per row tile).  (The 32-bit column indices
    and the unmerged diagonal term of rounds 3-4 are gone with their switches.)"""
    import ipsolver.cg_fused as cg_fused
    from ipsolver.operators import DeviceHessian
    n, m = 600000, 60000
    inst = BandedInstance(n, m)
    A = ips.dv.DeviceCSR.from_scipy(inst.A)
    hd = np.random.default_rng(3).uniform(0.0, 0.5, n)
    b = np.zeros(m)
    runs = []
    for flag in ("", form):
        if flag:
            monkeypatch.setenv("IPX_DEBUG_FORMS", flag)
        else:
            monkeypatch.delenv("IPX_DEBUG_FORMS", raising=False)
        H = DeviceHessian(n, csr=ips.dv.DeviceCSR.from_scipy(inst.H), diag=ips.dv.DVec.from_host(hd))
        assert H.diag is None                                       # merged into the CSR values
        Z, LS, Y = ips.proj.projections(A)
        before = cg_fused.STATS["resident_calls"]
        x, info = ips.qp.projected_cg(H, inst.c, Z, Y, b, tol=0, max_iter=25, trust_radius=1e300)
        assert cg_fused.STATS["resident_calls"] == before        # the separate launches
        runs.append((host(x), info))
    (x1, i1), (x2, i2) = runs
    assert i1 == i2
    assert np.max(np.abs(x1 - x2)) <= 1e-13 * np.max(np.abs(x1))


@pytest.mark.parametrize("rl,shift,m", [(4, 3, 50001), (9, 5, 40000), (16, 9, 30000),
                                        (3, 2, 30001)])
def test_solve_tail_for_rows_of_every_length(ips, rl, shift, m, monkeypatch):
    """The tail of the cyclic-reduction solve (g = r - A'v from the ELL(2) form of A' with one
    16-bit row offset per variable, csrc/banded.hip k_solve_pcr / ipsolver/cg_fused.py
    ell_rows) on Jacobians of other row lengths and overlaps than the benchmark's (3 .. 16
    entries, every variable-per-lane budget of the kernel), with an odd row count (the last
    workgroup's partial block) and the entries of A' stored in descending row order: the
    iterates of the loop without fused kernels to 1e-13, and the projected-CG trace of the
    host oracle (qp_subproblem.py:332-637) to 1e-10."""
    import scipy.sparse as sp
    import ipsolver.cg_fused as cg_fused
    from ipsolver.operators import DeviceHessian
    rng = np.random.default_rng(rl * 1000 + shift)
    n = (m - 1) * shift + rl
    n += n % 2                               # (the pair loads of the tail want an even n)
    rows = np.repeat(np.arange(m), rl)
    cols = (np.arange(m)[:, None] * shift + np.arange(rl)[None, :]).ravel()
    Ah = sp.csr_matrix((rng.uniform(0.5, 1.5, m * rl) * rng.choice([-1.0, 1.0], m * rl),
                        (rows, cols)), shape=(m, n))
    off = rng.uniform(-0.4, 0.4, n - 1)
    Hh = sp.diags([off, rng.uniform(1.5, 2.5, n), off], [-1, 0, 1], format="csr")
    c = rng.standard_normal(n)
    b = np.zeros(m)
    runs = []
    for flag in ("", "no-fuse"):
        if flag:
            monkeypatch.setenv("IPX_DEBUG_FORMS", flag)
        else:
            monkeypatch.delenv("IPX_DEBUG_FORMS", raising=False)
        A = ips.dv.DeviceCSR.from_scipy(Ah)
        if rl == 9:
            # A' with each variable's two entries in descending row order (unsorted CSR rows)
            At = Ah.T.tocsr()
            At.sort_indices()
            ip = At.indptr
            two = np.flatnonzero(np.diff(ip) == 2)
            for arr in (At.indices, At.data):
                lo = arr[ip[two]].copy()
                arr[ip[two]] = arr[ip[two] + 1]
                arr[ip[two] + 1] = lo
            At.has_sorted_indices = False
            A._T = ips.dv.DeviceCSR.from_scipy(At)
            A._T._T = A
        H = DeviceHessian(n, csr=ips.dv.DeviceCSR.from_scipy(Hh))
        Z, LS, Y = ips.proj.projections(A)
        L = cg_fused._Loop(H, Z.projector, None, None, resident=False)
        assert bool(L.args.At_ell_val) == (flag == "") and bool(L.args.At_ell_row) == (flag == "")
        x, info = ips.qp.projected_cg(H, c, Z, Y, b, tol=0, max_iter=12, trust_radius=1e300)
        runs.append((host(x), info))
    (x1, i1), (x2, i2) = runs
    assert i1 == i2
    assert np.max(np.abs(x1 - x2)) <= 1e-13 * np.max(np.abs(x1))
    import oracle
    Zo, LSo, Yo = oracle.projections(Ah)
    xo, io = oracle.projected_cg(Hh, c, Zo, Yo, b, tol=0, max_iter=12, trust_radius=1e300)
    assert io["niter"] == i1["niter"] and io["stop_cond"] == i1["stop_cond"]
    assert np.max(np.abs(x1 - xo)) <= 1e-10 * np.max(np.abs(xo))


@pytest.mark.parametrize("n,m", [(40000, 4000), (700000, 70000)])
def test_priming_forms_agree(ips, n, m, monkeypatch):
    """The three ways a call is primed (qp_subproblem.py:502-530: x0 = Y(-b), r0 = Z(H x0 + c),
    g0 = Z r0, rt_g, the default tolerance, the distance to the trust-region boundary): ONE C
    call with the first batch behind it (csrc/cg.hip ipx_cg_prime: norms out of the products'
    epilogues, one fold, -A'(A A')^-1 b and -g0 by the products' scalars), launch by launch
    from the host with the state block written on the device (the path of problems whose
    partials do not fit the reduction workspace), and the reference's order with every scalar
    read back.  The two device forms give the same iterates bit for bit -- the shortcuts are
    exact --, the host form to an ulp; for b != 0 and b = 0, at a resident and at a
    three-launch size."""
    import ipsolver.cg_fused as cg_fused
    from ipsolver import _hip
    inst = BandedInstance(n, m)
    A = ips.dv.DeviceCSR.from_scipy(inst.A)
    H = ips.dv.DeviceCSR.from_scipy(inst.H)
    Z, LS, Y = ips.proj.projections(A)
    c = ips.dv.DVec.from_host(inst.c)
    lib = _hip.load()
    for b_h in (inst.b, None):
        b = ips.dv.DVec.from_host(b_h) if b_h is not None else ips.dv.DVec.zeros(m)
        radius = 3.0 * float(np.linalg.norm(inst.b)) + 1.0
        kw = dict(trust_radius=radius, lb=None, ub=None, tol=None, max_iter=14,
                  max_infeasible_iter=None, batch=None, stats=None, b_zero=b_h is None)
        outs = []
        for form in ("one_call", "by_launch", "host"):
            before = dict(cg_fused.STATS)
            with monkeypatch.context() as mp:
                if form == "by_launch":
                    mp.setattr(lib, "ipx_cg_prime_ws_doubles", lambda *a: 1 << 40)
                x, info = cg_fused._projected_cg(H, c, Z, Y, b, fast=form != "host", **kw)
            primed = cg_fused.STATS["primed_on_device"] - before["primed_on_device"]
            assert primed == (0 if form == "host" else 1)
            assert cg_fused.STATS["prime_retries"] == before["prime_retries"]
            outs.append((host(x), info))
        assert outs[1][1] == outs[0][1] and outs[2][1] == outs[0][1]
        # (round 6: the one-call form takes its projections through the loop's own solve + tail
        # launch where the factorization has it -- ||g0||^2 summed per workgroup of the solve
        # instead of per row tile of the product: an ulp of rt_g against the launch-by-launch form)
        assert np.max(np.abs(outs[1][0] - outs[0][0])) <= 1e-15 * np.max(np.abs(outs[0][0]))
        # (the host sums ||g0||^2 with the norm kernel, the device paths take it from the
        # product's epilogue: another order, an ulp of rt_g)
        assert np.max(np.abs(outs[2][0] - outs[0][0])) <= 1e-15 * np.max(np.abs(outs[0][0]))
        assert outs[0][1]["niter"] >= 3


def test_fused_loop_on_random_band_shapes(ips):
    """tests/fuzz_fused_loop.py: 24 random Jacobians with rows of 2..16 entries, tridiagonal or
    diagonal A A', row counts around the solve's workgroup size (1, 2, 259..261, 519..521, ...),
    unconstrained variables, one case in five with its rows shuffled (banded only after the
    projector's row permutation), b = 0 and b != 0, three kinds of trust radius -- the loop with its
    fused kernels (resident where it fits) against the loop without them and against the three
    launches to 1e-11 (observed: 3e-16), small cases against the host oracle's projected CG
    (qp_subproblem.py:332-637) to 1e-9."""
    import fuzz_fused_loop
    assert fuzz_fused_loop.run(24, 4, verbose=False) <= 1e-11
    # found by a 300-case run: ONE 260-row block that owns 2898 variables -- 16 per tail lane of
    # the solve, a count the cyclic-reduction kernel was not compiled for (IPX_EINVAL out of the
    # loop's first launch)
    assert fuzz_fused_loop.run(3, 11, verbose=False, fixed=(12, 11, 260, 37)) <= 1e-11


def test_box_schur_loop_on_random_shapes(ips):
    """tests/fuzz_box_schur.py: 16 random barrier-shaped subproblems (rows of J with 3..16
    entries, bounds on every variable / lower only / a ragged mix, slacks down to 1e-6, three
    trust radii, a bound on the slacks' step) through the device loop with the box rows
    eliminated per group (csrc/boxschur.hip): compact against general group tables bit for bit,
    small cases against the host oracle's projected CG (qp_subproblem.py:332-637) -- same
    exits, iterates to 1e-9 (observed: 2e-15)."""
    import fuzz_box_schur
    assert fuzz_box_schur.run(16, 3, verbose=False) <= 1e-12


def test_projections_on_random_structures(ips):
    """tests/fuzz_projections.py: Z, LS, Y (projections.py:14-290) applied to random vectors --
    and the modified dogleg step (qp_subproblem.py:320-413: Newton point accepted, cut by the
    sphere, cut by a box) -- for
    20 random Jacobians -- random sparsity of three densities, a band with shuffled rows, block
    diagonal, a band with two dense rows, dense storage; 1 .. 1500 rows; every third case without
    the dense Cholesky -- so that the banded solver with and without reordering, the dense
    Cholesky and the preconditioned CG on A A' each take their turn, against
    the host oracle's projections: 1e-9 (observed: 2e-13)."""
    import fuzz_projections
    assert fuzz_projections.run(20, 2, verbose=False, max_m=1500) <= 1e-9


def test_device_loop_with_an_operator_hessian(ips):
    """A Hessian that is only an operator (``dot`` over device vectors -- what the reference's
    LinearOperator terms are: finite differences, user callbacks, _canonical_constraint.py:
    119-139) keeps the device-resident loop: the operator is applied between two iterations,
    the scalar-gated branches (qp_subproblem.py:551,558,583) stay on the device and the host
    reads the state once per batch instead of ~8 scalars per iteration.  Same exits and
    iterates as the CSR Hessian inside the loop (p'Hp is summed in another order: 1e-12)."""
    import ipsolver.cg_fused as cg_fused
    inst = BandedInstance(20000, 2000)
    A = ips.dv.DeviceCSR.from_scipy(inst.A)
    H = ips.dv.DeviceCSR.from_scipy(inst.H)

    class Operator:
        shape = H.shape
        calls = 0

        def dot(self, p):
            Operator.calls += 1
            return H.dot(p)
    Z, LS, Y = ips.proj.projections(A)
    b = np.zeros(2000)
    gnorm = ips.dv.norm(Z.dot(inst.c))
    assert cg_fused.supports(Operator(), Z, Y)
    for name, kw in inst.pcg_variants(gnorm).items():
        before, batches = cg_fused.STATS["operator_calls"], cg_fused.STATS["batches"]
        Operator.calls = 0
        x, info = ips.qp.projected_cg(Operator(), inst.c, Z, Y, b, **kw)
        assert cg_fused.STATS["operator_calls"] == before + 1, "device loop not taken"
        reads = cg_fused.STATS["batches"] - batches
        x1, info1 = ips.qp.projected_cg(H, inst.c, Z, Y, b, **kw)
        assert info == info1, (name, info, info1)
        close(x, host(x1), 1e-12)
        if name in ("free", "ball") and info["niter"] >= 20:
            # several iterations per state read (box events hand every iteration to the host)
            assert reads <= info["niter"] // 4 + 4, (name, reads, info["niter"])


@pytest.mark.parametrize("n,m", [(20000, 2000), (125000, 12500), (5210, 521), (500000, 50000)])
def test_resident_loop_matches_the_separate_launches(ips, n, m, monkeypatch):
    """A whole batch of iterations as ONE resident launch (csrc/resident.hip: one workgroup per
    260 rows of A A' for the whole iteration, matrix entries in registers, vectors in LDS, the
    two reductions and the halos through tagged 8-byte words) against the three-launch form:
    element by element the same expressions, ||g||^2 and the residual summed per workgroup in
    the same order; p'Hp and ||x + alpha p||^2 are summed per workgroup of the resident
    decomposition instead of per row tile, so alpha may differ in the last bit -- iterates to
    1e-13, counts and exits identical.  All exits: iteration limit, tolerance, trust region,
    and refinement events (the host finishes the iteration on the buffer that holds g and
    resumes on the separate launches' step2).  n = 5e5 is the largest size that is resident
    (193 workgroups)."""
    import ipsolver.cg_fused as cg_fused
    inst = BandedInstance(n, m)
    A = ips.dv.DeviceCSR.from_scipy(inst.A)
    H = ips.dv.DeviceCSR.from_scipy(inst.H)
    b = np.zeros(m)
    runs = {}
    for flag in ("resident", "separate"):
        if flag == "resident":
            monkeypatch.delenv("IPX_DEBUG_FORMS", raising=False)
        else:
            monkeypatch.setenv("IPX_DEBUG_FORMS", "no-resident")
        Z, LS, Y = ips.proj.projections(A)
        x_free, _ = ips.qp.projected_cg(H, inst.c, Z, Y, b, tol=1e-12)
        out = []
        before = cg_fused.STATS["resident_calls"]
        for kw in (dict(tol=0, max_iter=41), dict(tol=1e-12),
                   dict(trust_radius=0.5 * ips.dv.norm(x_free)),
                   dict(tol=0, max_iter=30, trust_radius=1e300), dict(tol=0, max_iter=1)):
            x, info = ips.qp.projected_cg(H, inst.c, Z, Y, b, **kw)
            out.append((host(x), info))
        Zr, _, Yr = ips.proj.projections(A, orth_tol=1e-30, max_refin=2)   # refines every time
        x, info = ips.qp.projected_cg(H, inst.c, Zr, Yr, b, tol=0, max_iter=9)
        out.append((host(x), info))
        engaged = cg_fused.STATS["resident_calls"] - before
        assert engaged == (6 if flag == "resident" else 0), engaged
        assert cg_fused.STATS["resident_fallbacks"] == 0
        runs[flag] = out
    for k, ((x0, i0), (x1, i1)) in enumerate(zip(runs["resident"], runs["separate"])):
        assert i0 == i1, (k, i0, i1)
        assert np.max(np.abs(x0 - x1)) <= 1e-13 * np.max(np.abs(x1)), (k, np.max(np.abs(x0 - x1)))


def test_resident_timeout_restarts_the_call(ips, monkeypatch):
    """A resident launch in which a hand-off timed out (stop code 8) may have been committed by
    some of its workgroups and not by others -- one that arrives past the others' deadline and
    then finds every record commits alone (the commit hop is not transactional).
    The loop's vectors are therefore not replayed from: the CALL starts over from its priming on
    the separate launches, and the pattern stays on them.  Injected: after a resident batch,
    stop code 8 with half of x advanced.  Same result as an undisturbed call, through the public
    call and through the outer iteration's chain."""
    import ipsolver.cg_fused as cg_fused
    monkeypatch.delenv("IPX_DEBUG_FORMS", raising=False)
    inst = BandedInstance(20000, 2000)
    A = ips.dv.DeviceCSR.from_scipy(inst.A)
    H = ips.dv.DeviceCSR.from_scipy(inst.H)
    Z, LS, Y = ips.proj.projections(A)
    b = np.zeros(2000)
    cg_fused._NO_RESIDENT.clear()
    x0, i0 = ips.qp.projected_cg(H, inst.c, Z, Y, b, tol=0, max_iter=25)
    before = dict(cg_fused.STATS)
    assert before["resident_calls"] > 0
    cg_fused._INJECT_RESIDENT_TIMEOUT.append(1)
    x1, i1 = ips.qp.projected_cg(H, inst.c, Z, Y, b, tol=0, max_iter=25)
    assert not cg_fused._INJECT_RESIDENT_TIMEOUT
    assert cg_fused.STATS["resident_fallbacks"] == before["resident_fallbacks"] + 1
    assert i1 == i0
    assert np.max(np.abs(host(x1) - host(x0))) <= 1e-13 * np.max(np.abs(host(x0)))
    # the pattern stays on the separate launches
    calls = cg_fused.STATS["resident_calls"]
    x2, i2 = ips.qp.projected_cg(H, inst.c, Z, Y, b, tol=0, max_iter=25)
    assert cg_fused.STATS["resident_calls"] == calls and i2 == i0
    assert np.array_equal(host(x2), host(x1))
    cg_fused._NO_RESIDENT.clear()
    cg_fused._POOL.clear()


@pytest.mark.parametrize("n,m,hbw,abw,seed", [(5000, 400, 2, 9, 0), (12345, 1500, 3, 6, 1),
                                               (3001, 299, 1, 21, 2), (40000, 2500, 5, 30, 3),
                                               (30000, 3000, 2, 30, 4), (30000, 3000, 1, 30, 5),
                                               (30000, 3000, 1, 22, 5)])
def test_fused_kernels_on_other_band_shapes(ips, n, m, hbw, abw, seed, monkeypatch):
    """The two fused kernels (step1 in A.r, step2 in H.p) against the separate launches on
    banded problems of other shapes: Hessian half bandwidth 1..5, Jacobian rows of ragged
    length with irregular starts, sizes that are no multiple of anything."""
    import scipy.sparse as sps
    import ipsolver.cg_fused as cg_fused
    rng = np.random.default_rng(seed)
    # SPD banded Hessian: diagonally dominant
    offs = list(range(-hbw, hbw + 1))
    bands = [rng.uniform(-1, 1, n - abs(o)) for o in offs]
    Hm = sps.diags(bands, offs, format="csr")
    Hm = sps.csr_matrix(0.5 * (Hm + Hm.T) + sps.diags(np.full(n, 2.0 * hbw + 1.0)))
    # Jacobian: row i has 3..abw contiguous entries starting near i*n/m (monotone starts)
    starts = np.minimum((np.arange(m) * (n // m) + rng.integers(0, 3, m)), n - abw)
    starts = np.maximum.accumulate(starts)
    lens = rng.integers(3, abw + 1, m)
    rows = np.repeat(np.arange(m), lens)
    cols = np.concatenate([s + np.arange(k) for s, k in zip(starts, lens)])
    Am = sps.csr_matrix((rng.standard_normal(len(cols)), (rows, cols)), shape=(m, n))
    A = ips.dv.DeviceCSR.from_scipy(Am)
    H = ips.dv.DeviceCSR.from_scipy(Hm)
    Z, LS, Y = ips.proj.projections(A)
    c = rng.standard_normal(n)
    assert cg_fused.supports(H, Z, Y)
    assert cg_fused.fuse_halo(H.pattern) == hbw
    # (whether step1 also fuses depends on how many columns a row tile of A spans: <= 2048)
    # half bandwidth 2..4 of A A' with decoupled separators: g = r - A'v rides in the banded
    # solve's launch (up to k + 1 constraints per variable, read in the kernel's tail)
    kS = Z.projector.solver.k
    if seed >= 4:
        import ctypes
        from ipsolver import _hip
        assert 2 <= kS <= 4
        assert _hip.load().ipx_banded_decoupled(ctypes.c_void_p(Z.projector.solver.handle)) == 1
        assert cg_fused._Loop(H, Z.projector, None, None).args.At_qv > 0
    runs = []
    for no_fuse in ("", "1"):
        if no_fuse:
            monkeypatch.setenv("IPX_DEBUG_FORMS", "no-fuse")
        else:
            monkeypatch.delenv("IPX_DEBUG_FORMS", raising=False)
        x, info = ips.qp.projected_cg(H, c, Z, Y, np.zeros(m), tol=1e-14, max_iter=60)
        runs.append((host(x), info))
    (x1, i1), (x2, i2) = runs
    assert (i1["niter"], i1["stop_cond"]) == (i2["niter"], i2["stop_cond"])
    assert np.max(np.abs(x1 - x2)) <= 1e-12 * np.max(np.abs(x2))
    # and against the CPU oracle
    import oracle
    Zo, _, Yo = oracle.projections(Am)
    xo, io = oracle.projected_cg(Hm, c, Zo, Yo, np.zeros(m), tol=1e-14, max_iter=60)
    assert io["niter"] == i1["niter"]
    assert np.max(np.abs(x1 - xo)) <= 1e-9 * np.max(np.abs(xo))


def test_projections_accept_unsorted_device_csr(ips):
    """A DeviceCSR assembled by hand (device-callback mode) may list a row's columns in any
    order; the projections must not depend on it."""
    import torch
    from ipsolver.device import CSRPattern, DeviceCSR
    inst = BandedInstance(2000, 200)
    A = inst.A.tocsr()
    rng = np.random.default_rng(0)
    indices, data = A.indices.copy(), A.data.copy()
    for i in range(A.shape[0]):                      # shuffle every row
        a, b = A.indptr[i], A.indptr[i + 1]
        perm = rng.permutation(b - a)
        indices[a:b], data[a:b] = indices[a:b][perm], data[a:b][perm]
    Au = DeviceCSR(CSRPattern(A.indptr, indices, A.shape), torch.from_numpy(data).cuda())
    Zs, LSs, Ys = ips.proj.projections(ips.dv.DeviceCSR.from_scipy(A))
    Zu, LSu, Yu = ips.proj.projections(Au)
    x, b = rng.standard_normal(2000), rng.standard_normal(200)
    close(Zu.dot(x), host(Zs.dot(x)), 1e-12)
    close(LSu.dot(x), host(LSs.dot(x)), 1e-12)
    close(Yu.dot(b), host(Ys.dot(b)), 1e-12)


@pytest.mark.parametrize("n,m", [(2, 1), (3, 1), (5, 2), (9, 3), (17, 4), (64, 8), (257, 30)])
def test_device_loop_at_tiny_sizes(ips, n, m):
    """The device-resident loop and its fused kernels at sizes of a few rows (one tile,
    halo clipped at both ends, fewer chunks than a workgroup solves) against the oracle:
    same iteration count, same exit, same point."""
    import scipy.sparse as sps
    import ipsolver.cg_fused as cg_fused
    import oracle
    rng = np.random.default_rng(n)
    H = sps.diags([rng.uniform(-1, 1, n - 1), np.full(n, 3.0), rng.uniform(-1, 1, n - 1)],
                  [-1, 0, 1], format="csr")
    H = sps.csr_matrix(0.5 * (H + H.T))
    stride = max(n // m, 1)
    rows = np.repeat(np.arange(m), 2)
    cols = np.minimum(np.concatenate([[i * stride, i * stride + 1] for i in range(m)]), n - 1)
    A = sps.csr_matrix((rng.standard_normal(2 * m), (rows, cols)), shape=(m, n))
    A.sum_duplicates()
    c = rng.standard_normal(n)
    Ad, Hd = ips.dv.DeviceCSR.from_scipy(A), ips.dv.DeviceCSR.from_scipy(H)
    Z, LS, Y = ips.proj.projections(Ad)
    assert cg_fused.supports(Hd, Z, Y)
    x, info = ips.qp.projected_cg(Hd, c, Z, Y, np.zeros(m), tol=1e-14)
    Zo, _, Yo = oracle.projections(A)
    xo, io = oracle.projected_cg(H, c, Zo, Yo, np.zeros(m), tol=1e-14)
    assert (info["niter"], info["stop_cond"]) == (io["niter"], io["stop_cond"])
    close(x, xo, 1e-12)


@pytest.mark.parametrize("k", [2, 3, 4, 5, 8])
def test_single_launch_banded_solve_with_block_separators(ips, k):
    """Half bandwidth k > 1: the separator system is block tridiagonal (k x k blocks); when
    the blocks decouple numerically the solve is still one launch, with the separators'
    diagonal blocks inverted at factor time (k_decoupling_check_block).  Against a direct
    sparse solve and the general three-launch / multi-launch sweeps."""
    import ctypes
    import scipy.sparse.linalg as spla
    from ipsolver import _hip
    from ipsolver.projector import BandedNormalSolver
    rng = np.random.default_rng(k)
    m, n = 9000, 9000 * 4 + 3 * k
    # row i touches 4k contiguous columns starting at 4 i: rows i, i+d overlap for d < k... <= k
    starts = 4 * np.arange(m)
    cols = (starts[:, None] + np.arange(4 * k)[None, :]).ravel()
    rows = np.repeat(np.arange(m), 4 * k)
    A = sps.csr_matrix((rng.standard_normal(len(cols)), (rows, cols)), shape=(m, n + 4 * k))
    S = (A @ A.T).tocsc()
    coo = S.tocoo()
    assert np.max(np.abs(coo.row - coo.col)) == k - 1 or np.max(np.abs(coo.row - coo.col)) <= k
    solver = BandedNormalSolver(ips.dv.DeviceCSR.from_scipy(A))
    lib = _hip.load()
    # whether the separator blocks decouple is a property of the numbers (checked at every
    # factorization); for these matrices they do at k = 2, 3.  Half bandwidths beyond 4 have
    # no compiled separator level (2k-1 > 8): decoupled, or defect correction on the
    # single-launch solve (test_banded_defect_correction).  Either way the result must be right.
    if solver.k == 2:
        assert lib.ipx_banded_decoupled(ctypes.c_void_p(solver.handle)) == 1
    w = rng.standard_normal(m)
    wd = ips.dv.DVec.from_host(w)
    v = host(solver.solve(wd))
    vref = spla.spsolve(S, w)
    assert np.max(np.abs(v - vref)) <= 1e-10 * np.max(np.abs(vref))
    if solver.k >= 5:
        return                      # no level-by-level sweep to compare with
    v2 = ips.dv.DVec.zeros(m)
    _hip.call("ipx_banded_solve_multilaunch", ctypes.c_void_p(solver.handle), ips.dv._p(wd.t),
              ips.dv._p(v2.t), ips.dv.stream_ptr())
    assert np.max(np.abs(v - host(v2))) <= 1e-12 * np.max(np.abs(v))


def test_general_sparse_jacobian_beyond_the_factorizations(ips, monkeypatch):
    """A sparse Jacobian with scattered columns and more rows than the dense device Cholesky
    takes: neither banded nor small.  The projections then run on the matrix-free solver
    (Jacobi-preconditioned CG on A A', projector.IterativeNormalSolver) and must still be
    the reference's operators (projections.py:290-406), here against the CPU oracle."""
    import oracle
    from ipsolver.dense import DenseNormalSolver
    from ipsolver.projector import IterativeNormalSolver
    # (the dense limit is lowered so that a size the CPU oracle factors in a second takes
    # the same route as a 20000-row Jacobian would)
    monkeypatch.setattr(DenseNormalSolver, "MAX_ROWS_FROM_SPARSE", 100)
    rng = np.random.default_rng(0)
    m, n = 3000, 9000
    rows = np.repeat(np.arange(m), 4)
    cols = rng.integers(0, n, 4 * m)
    A = sps.csr_matrix((rng.standard_normal(4 * m), (rows, cols)), shape=(m, n))
    A.sum_duplicates()
    Z, LS, Y = ips.proj.projections(ips.dv.DeviceCSR.from_scipy(A))
    assert isinstance(Z.projector.solver, IterativeNormalSolver)
    Zo, LSo, Yo = oracle.projections(A)
    x, b = rng.standard_normal(n), rng.standard_normal(m)
    close(Z.dot(x), Zo.dot(x), 1e-9)
    close(LS.dot(x), LSo.dot(x), 1e-9)
    close(Y.dot(b), Yo.dot(b), 1e-9)
    zx = Z.dot(x)
    norm_A = float(np.sqrt((A.data ** 2).sum()))
    assert ips.dv.norm(ips.dv.DeviceCSR.from_scipy(A).dot(zx)) <= 1e-12 * norm_A * ips.dv.norm(zx)
    # and a projected-CG solve on top of it (general driver)
    H = sps.diags(rng.uniform(1, 2, n), format="csr")
    c = rng.standard_normal(n)
    xg, info = ips.qp.projected_cg(ips.dv.DeviceCSR.from_scipy(H), c, Z, Y, np.zeros(m), max_iter=30)
    xo, io = oracle.projected_cg(H, c, Zo, Yo, np.zeros(m), max_iter=30)
    assert info["niter"] == io["niter"] and info["stop_cond"] == io["stop_cond"]
    close(xg, xo, 1e-8)


@pytest.mark.parametrize("max_refin", [1, 3])
def test_banded_traces_with_refinement(ips, banded_refine2000, max_refin):
    """projected_cg with projections that refine on every application (orth_tol far below
    the attainable orthogonality) against traces of the reference run the same way: the
    device loop must raise stop code 6 in every iteration and hand the refinement
    (projections.py:69-78) to the host, the general driver must run the loop in
    ``NormalEquationProjector.null_space``."""
    import ipsolver.cg_fused as cg_fused
    gold = banded_refine2000
    n, m = 2000, 200
    inst = BandedInstance(n, m)
    A = ips.dv.DeviceCSR.from_scipy(inst.A)
    H = ips.dv.DeviceCSR.from_scipy(inst.H)
    Z, LS, Y = ips.proj.projections(A, orth_tol=1e-30, max_refin=max_refin)
    P = Z.projector
    for p, w in zip(inst.probes_n, gold["refine%d_Z" % max_refin]):
        before = P.stats["refinements"]
        close(Z.dot(p), w)
        assert P.stats["refinements"] - before == max_refin
    assert cg_fused.supports(H, Z, Y)
    for name, kw in inst.pcg_variants(1.0).items():
        if name not in ("free", "box"):
            continue
        want = list(gold["refine%d_pcg_%s_info" % (max_refin, name)])
        for return_all in (False, True):
            ev0, rf0 = cg_fused.STATS["refine_events"], P.stats["refinements"]
            x, info = ips.qp.projected_cg(H, inst.c, Z, Y, np.zeros(m), return_all=return_all,
                                          **kw)
            assert [info["niter"], info["stop_cond"], int(info["hits_boundary"])] == want
            close(x, gold["refine%d_pcg_%s_x" % (max_refin, name)])
            assert P.stats["refinements"] - rf0 >= max_refin * info["niter"]
            if not return_all:          # the device-resident loop: one hand-back per iteration
                assert cg_fused.STATS["refine_events"] - ev0 >= info["niter"] - 1


@pytest.mark.parametrize("cond", [1e3, 1e5, 1e6])
def test_dense_ill_conditioned(ips, cond):
    """Dense Jacobians far from the benchmark's conditioning (cond(A) = 2.6 there).

    (1) Applying the explicit inverse of G = A A' loses what two triangular solves with its
    Cholesky factor lose: both are limited by the normal equations (cond(A)^2 eps), so the
    one-matvec form is no liability.  (2) What the normal equations lose against the
    reference's pivoted QR (projections.py:175-233, error ~cond(A) eps) is recovered by the
    refinement steps the solver enables from its measured pivot loss: Z, LS and Y agree with
    the QR operators to 1e-10 or QR's own accuracy."""
    import scipy.linalg
    import oracle
    rng = np.random.default_rng(int(np.log10(cond)))
    m, n = 200, 1000
    U, _ = np.linalg.qr(rng.standard_normal((m, m)))
    V, _ = np.linalg.qr(rng.standard_normal((n, m)))
    A = (U * np.logspace(0, -np.log10(cond), m)) @ V.T
    x, b = rng.standard_normal(n), rng.standard_normal(m)
    Zo, LSo, Yo = oracle.projections(A)              # pivoted QR, as the reference
    Z, LS, Y = ips.proj.projections(A)
    solver = Z.projector.solver
    assert solver.pivot_ratio < 1e-3 and solver.refine_steps >= 1

    def rel(a, b):
        return np.max(np.abs(host(a) - b)) / np.max(np.abs(b))
    # (1) one application of (A A')^-1: explicit inverse vs two triangular solves
    w = A @ x
    v_true = LSo.dot(x)
    v_inv = solver.solve(ips.dv.DVec.from_host(w))
    v_trsv = scipy.linalg.cho_solve(scipy.linalg.cho_factor(A @ A.T), w)
    assert rel(v_inv, v_true) <= 5 * rel(v_trsv, v_true)
    # (2) the operators, refined
    tol = max(1e-10, 50 * cond * np.finfo(float).eps)
    assert rel(LS.dot(x), v_true) <= tol
    assert rel(Y.dot(b), Yo.dot(b)) <= tol
    assert rel(Z.dot(x), Zo.dot(x)) <= tol
    assert ips.proj.orthogonality(A, Z.dot(x)) <= 1e-12


@pytest.mark.parametrize("n,m", [(2000, 200), (3000, 300), (20000, 2000), (1000000, 100000)])
def test_banded_solve_by_parallel_cyclic_reduction(ips, n, m):
    """Tridiagonal A A' (the banded benchmark): the single-launch solve runs as parallel
    cyclic reduction over windows of rows (csrc/banded.hip k_solve_pcr), the level at which
    the reduction has decoupled being measured at every factorization.  Against a direct
    sparse solve, against the chunk-recurrence form of the same launch, and the fused
    residual ||w - S v||^2 against numpy."""
    import ctypes
    import torch
    from ipsolver import _hip
    lib = _hip.load()
    inst = BandedInstance(n, m)
    A = inst.A.tocsr()
    solver = ips.proj.BandedNormalSolver(ips.dv.DeviceCSR.from_scipy(A))
    h = ctypes.c_void_p(solver.handle)
    if m <= 260:
        return        # one chunk-workgroup at most: nothing to decouple from
    L = lib.ipx_banded_pcr_level(h)
    assert lib.ipx_banded_decoupled(h) == 1 and 1 <= L <= 7
    rng = np.random.default_rng(m)
    w = rng.standard_normal(m)
    wd = ips.dv.DVec.from_host(w)
    S = sps.csc_matrix(A @ A.T)
    want = sps.linalg.splu(S).solve(w)
    v = torch.empty(m, dtype=torch.float64, device="cuda")
    part = torch.zeros((m + 255) // 256 + 1, dtype=torch.float64, device="cuda")
    npart = ctypes.c_int32(0)
    _hip.call("ipx_banded_solve_resid", h, ips.dv._p(wd.t), ips.dv._p(v), ips.dv._p(part),
              ctypes.byref(npart), None, ips.dv.stream_ptr())
    got = v.cpu().numpy()
    assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))
    res = float(np.sum(part.cpu().numpy()[:npart.value]))
    res_np = float(np.sum((w - S @ got) ** 2))
    assert 0.25 * res_np <= res <= 4 * res_np + 1e-300       # both are rounding noise
    assert res <= 1e-28 * np.sum(w ** 2)
    again = torch.empty_like(v)
    _hip.call("ipx_banded_solve", h, ips.dv._p(wd.t), ips.dv._p(again), ips.dv.stream_ptr())
    assert torch.equal(again, v)                            # bitwise reproducible
    # a reduction stopped two levels early is an inexact solve: the fused residual must be
    # the residual of what the kernel returned (the orthogonality measure, projections.py:52)
    if L > 2:
        lib.ipx_banded_set_decoupling(h, 16 + L - 2)
        _hip.call("ipx_banded_solve_resid", h, ips.dv._p(wd.t), ips.dv._p(again), ips.dv._p(part),
                  ctypes.byref(npart), None, ips.dv.stream_ptr())
        rough = again.cpu().numpy()
        res = float(np.sum(part.cpu().numpy()[:npart.value]))
        res_np = float(np.sum((w - S @ rough) ** 2))
        assert res_np > 1e-20 * np.sum(w ** 2) and abs(res - res_np) <= 1e-10 * res_np
    lib.ipx_banded_set_decoupling(h, 2)                     # the chunk-recurrence form
    assert lib.ipx_banded_pcr_level(h) == 0
    v2 = torch.empty_like(v)
    _hip.call("ipx_banded_solve", h, ips.dv._p(wd.t), ips.dv._p(v2), ips.dv.stream_ptr())
    assert np.max(np.abs(v2.cpu().numpy() - got)) <= 1e-13 * np.max(np.abs(got))


def test_iterative_normal_solver_on_the_device(ips):
    """m = 20000 rows (more than the dense device Cholesky takes) with A A' of half bandwidth 11
    (more than the banded solver takes): the projections run on the device-resident
    preconditioned CG (csrc/pcg.hip) -- one C call per batch of inner iterations, convergence
    decided on the device -- and must be the reference's operators, here against the oracle."""
    import oracle
    from ipsolver.projector import IterativeNormalSolver
    rng = np.random.default_rng(1)
    m = 20000
    n = 2 * m + 24
    cols = (2 * np.arange(m)[:, None] + np.arange(24)[None, :]).ravel()
    A = sps.csr_matrix((rng.standard_normal(24 * m), (np.repeat(np.arange(m), 24), cols)),
                       shape=(m, n))
    Z, LS, Y = ips.proj.projections(ips.dv.DeviceCSR.from_scipy(A))
    solver = Z.projector.solver
    assert isinstance(solver, IterativeNormalSolver)
    Zo, LSo, Yo = oracle.projections(A)
    x, b = rng.standard_normal(n), rng.standard_normal(m)
    close(Z.dot(x), Zo.dot(x), 1e-9)
    close(LS.dot(x), LSo.dot(x), 1e-9)
    close(Y.dot(b), Yo.dot(b), 1e-9)
    # several inner iterations per host read-back: the loop is device resident
    assert solver.stats["solves"] >= 3
    assert solver.stats["iterations"] >= 4 * solver.stats["batches"]
    zx = Z.dot(x)
    norm_A = float(np.sqrt((A.data ** 2).sum()))
    assert ips.dv.norm(ips.dv.DeviceCSR.from_scipy(A).dot(zx)) <= 1e-12 * norm_A * ips.dv.norm(zx)
    # a zero row is a rank-deficient Jacobian: reported, not iterated on
    A0 = A.tolil()
    A0[5, :] = 0
    with pytest.raises(np.linalg.LinAlgError):
        IterativeNormalSolver(ips.dv.DeviceCSR.from_scipy(sps.csr_matrix(A0)))


def test_ill_conditioned_full_rank_jacobian_too_large_for_the_svd_exit(ips):
    """A pivot of the device factorization that has lost 43 bits marks the Jacobian numerically
    rank deficient.  Small matrices then take the reference's SVD exit (goldens:
    test_rank_deficient_fallback); a large sparse one (here 40000 x 80001: a dense SVD is out of
    reach) used to abort -- the reference's sparse LU only bails on EXACT singularity
    (projections.py:101-108).  Now the pivots being positive, the factorization is kept (a
    warning says so) and the orthogonality-driven refinement of the null-space operator
    (projections.py:69-78) recovers what the conditioning loses."""
    rng = np.random.default_rng(2)
    m, delta = 40000, 2e-7
    # pairs of nearly parallel rows: (1, 1, 0) and (1, 1, delta) on three private columns,
    # tied (by delta too) to the next pair: the second pivot of every pair is ~ delta^2 / 2 of its
    # diagonal entry, i.e. 2e-14 < 2^-43
    j = np.arange(m // 2)
    rows = np.concatenate((2 * j, 2 * j, 2 * j + 1, 2 * j + 1, 2 * j + 1, 2 * j[:-1] + 1))
    cols = np.concatenate((3 * j, 3 * j + 1, 3 * j, 3 * j + 1, 3 * j + 2, 3 * j[:-1] + 3))
    vals = np.concatenate((np.ones(2 * m), np.full(m // 2, delta), np.full(m // 2 - 1, delta)))
    A = sps.csr_matrix((vals, (rows, cols)), shape=(m, 3 * (m // 2)))
    assert A.shape[0] * A.shape[1] > 2 ** 25
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    with pytest.warns(UserWarning, match="Ill-conditioned Jacobian"):
        Z, LS, Y = ips.proj.projections(Ad)
    assert type(Z.projector.solver).__name__ == "BandedNormalSolver"
    x = rng.standard_normal(A.shape[1])
    z = Z.dot(x)
    assert ips.proj.orthogonality(Ad, z) <= 1e-9
    # an exactly dependent pair of rows is still refused
    vals0 = vals.copy()
    vals0[2 * m + 3] = 0.0                         # delta of pair 3 -> rows 6 and 7 identical
    vals0[2 * m + m // 2 + 3] = 0.0                # (and no link out of row 7)
    B = sps.csr_matrix((vals0, (rows, cols)), shape=A.shape)
    B.eliminate_zeros()
    with pytest.raises(np.linalg.LinAlgError):
        ips.proj.projections(ips.dv.DeviceCSR.from_scipy(B))


@pytest.mark.parametrize("eps", [3.0, 1.0, 0.3])
def test_block_jacobi_preconditioner_over_condition_numbers(ips, eps):
    """The inner solve of ``IterativeNormalSolver`` with the block-Jacobi preconditioner
    (32 x 32 diagonal blocks of A A' in the symbolic analysis' row order, built and inverted on
    the device: csrc/pcg.hip) against the diagonal one of round 2, on Jacobians whose
    neighbouring rows are nearly parallel -- moving averages over 12 columns + ``eps`` on a
    private column, cond(A A') ~ (12 / eps)^2: 16 ... 1600 -- at m = 20000 (beyond the dense
    Cholesky; half bandwidth 11 is beyond the banded solver).  Same answer as a sparse LU of
    A A' (1e-9 of it), fewer inner iterations at every conditioning, the more the worse it is."""
    import scipy.sparse.linalg as spla
    from ipsolver.projector import IterativeNormalSolver
    rng = np.random.default_rng(5)
    m = 20000
    A = _moving_average_rows(m, 11, eps, rng)
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    w = rng.standard_normal(m)
    want = spla.splu(sps.csc_matrix(A @ A.T)).solve(w)
    its = {}
    for precond in ("jacobi", "block"):
        solver = IterativeNormalSolver(Ad, precond=precond)
        v = host(solver.solve(ips.dv.DVec.from_host(w)))
        its[precond] = solver.stats["iterations"]
        assert np.max(np.abs(v - want)) <= 1e-9 * np.max(np.abs(want)), (precond, eps)
    print("eps %.1f: inner iterations jacobi %d, block Jacobi %d" % (eps, its["jacobi"], its["block"]))
    assert its["block"] < its["jacobi"]
    if eps <= 1.0:
        assert its["block"] <= 0.6 * its["jacobi"]
    # through the public seam the block preconditioner is the default
    Z, LS, Y = ips.proj.projections(Ad)
    assert isinstance(Z.projector.solver, IterativeNormalSolver) \
        and Z.projector.solver.precond == "block"


@pytest.mark.parametrize("kA", [4, 5, 6, 7, 8, 9])
def test_banded_defect_correction(ips, kA):
    """Half bandwidths 3..8 at m = 1e5.  The separator system of the
    partitioned factorization has half bandwidth 2k-1: a serial sweep over 4200..11000 rows
    for k = 3, 4 (1.7 / 3.0 ms per solve in round 1) and no compiled kernel at all for
    k >= 5 (the whole band was one serially swept chunk: 80-100 ms).  Now the separator
    matrix is only formed, the contraction of block Jacobi on it is measured at the
    factorization (eta), and a solve is N(eta) steps of defect correction on the
    single-launch solve (csrc/banded.hip iter_solve): 34-180 us.  Against a sparse LU, the
    fused residual against numpy, bitwise reproducible, through ``projections``."""
    import ctypes
    import torch
    import scipy.sparse.linalg as spla
    from ipsolver import _hip
    from ipsolver.projector import BandedNormalSolver
    rng = np.random.default_rng(kA)
    m = 100000
    cols = (4 * np.arange(m)[:, None] + np.arange(4 * kA)[None, :]).ravel()
    A = sps.csr_matrix((rng.standard_normal(len(cols)), (np.repeat(np.arange(m), 4 * kA), cols)),
                       shape=(m, 4 * m + 4 * kA))
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    Z, LS, Y = ips.proj.projections(Ad)
    solver = Z.projector.solver
    assert isinstance(solver, BandedNormalSolver) and solver.k == kA - 1
    lib = _hip.load()
    h = ctypes.c_void_p(solver.handle)
    eta = ctypes.c_double(0.0)
    steps = lib.ipx_banded_refine_steps(h, ctypes.byref(eta))
    assert lib.ipx_banded_decoupled(h) == 1 or (1 <= steps <= 8 and 0 < eta.value < 1e-3)
    S = (A @ A.T).tocsc()
    w = rng.standard_normal(m)
    wd = ips.dv.DVec.from_host(w)
    v = host(solver.solve(wd))
    vref = spla.splu(S).solve(w)
    assert np.max(np.abs(v - vref)) <= 1e-12 * np.max(np.abs(vref))
    assert np.array_equal(v, host(solver.solve(wd)))
    # in place (x may alias w: the contract of ipx_banded_solve), whichever form the solve takes
    inplace = wd.t.clone()
    _hip.call("ipx_banded_solve", h, ips.dv._p(inplace), ips.dv._p(inplace), ips.dv.stream_ptr())
    assert np.array_equal(inplace.cpu().numpy(), v)
    # the solve + residual form the CG loop uses
    out = torch.empty(m, dtype=torch.float64, device="cuda")
    part = torch.zeros((m + 255) // 256 + 1, dtype=torch.float64, device="cuda")
    npart = ctypes.c_int32(0)
    _hip.call("ipx_banded_solve_resid", h, ips.dv._p(wd.t), ips.dv._p(out), ips.dv._p(part),
              ctypes.byref(npart), None, ips.dv.stream_ptr())
    assert np.array_equal(out.cpu().numpy(), v)
    res2 = float(part[:npart.value].sum().item())
    assert res2 <= 1e-24 * float(np.sum(w ** 2)) * S.shape[0]
    # the operators
    x = rng.standard_normal(A.shape[1])
    z = host(Z.dot(x))
    assert np.max(np.abs(A @ z)) <= 1e-11 * np.max(np.abs(x)) * np.sqrt(4 * kA)


def _moving_average_rows(m, k, eps, rng):
    """Row i = k+1 nearly equal weights on columns i..i+k plus ``eps`` on a private column:
    S = A A' is the triangle-kernel Toeplitz matrix + eps^2 I, half bandwidth k, whose
    inverse decays the more slowly the smaller eps is (cond ~ (k+1)^2 / eps^2)."""
    cols = (np.arange(m)[:, None] + np.arange(k + 1)[None, :]).ravel()
    rows = np.repeat(np.arange(m), k + 1)
    vals = (1.0 + 0.01 * rng.standard_normal((m, k + 1))).ravel()
    return sps.csr_matrix((np.concatenate((vals, np.full(m, eps))),
                           (np.concatenate((rows, np.arange(m))),
                            np.concatenate((cols, m + k + np.arange(m))))), shape=(m, 2 * m + k))


def test_banded_defect_correction_many_steps(ips):
    """A slowly decaying inverse (half bandwidth 6, contraction bound 0.26): 27 correction
    steps still give the direct solve to 1e-13."""
    import ctypes
    import scipy.sparse.linalg as spla
    from ipsolver import _hip
    from ipsolver.projector import BandedNormalSolver
    rng = np.random.default_rng(6)
    m = 20000
    A = _moving_average_rows(m, 6, 0.3, rng)
    solver = BandedNormalSolver(ips.dv.DeviceCSR.from_scipy(A))
    eta = ctypes.c_double(0.0)
    steps = _hip.load().ipx_banded_refine_steps(ctypes.c_void_p(solver.handle), ctypes.byref(eta))
    assert solver.k == 6 and 10 <= steps <= 54 and 0.05 < eta.value < 0.5
    w = rng.standard_normal(m)
    v = host(solver.solve(ips.dv.DVec.from_host(w)))
    vref = spla.splu((A @ A.T).tocsc()).solve(w)
    assert np.max(np.abs(v - vref)) <= 1e-12 * np.max(np.abs(vref))


def test_coupled_wide_band_takes_the_iterative_solver(ips):
    """Half bandwidth 6 with chunks that do NOT decouple (a slowly decaying inverse): block
    Jacobi on the separators is not known to contract fast enough (bound >= 0.5), the banded
    solver reports IPX_EUNSUPPORTED and ``projections`` takes the device-resident
    preconditioned CG.  Same operators: against the oracle."""
    import oracle
    from ipsolver.projector import (BandedNormalSolver, BandedNotDecoupled, IterativeNormalSolver,
                                    _symbolic_for)
    rng = np.random.default_rng(6)
    m = 20000
    A = _moving_average_rows(m, 6, 0.1, rng)
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    assert _symbolic_for(Ad.pattern).k == 6
    with pytest.raises(BandedNotDecoupled):
        BandedNormalSolver(Ad)
    Z, LS, Y = ips.proj.projections(Ad)
    assert isinstance(Z.projector.solver, IterativeNormalSolver)
    Zo, LSo, Yo = oracle.projections(A)
    x = rng.standard_normal(A.shape[1])
    close(Z.dot(x), Zo.dot(x), 1e-8)
    close(Y.dot(A @ x), Yo.dot(A @ x), 1e-8)


@pytest.mark.parametrize("hessian", ["dense", "csr"])
def test_dense_jacobian_device_loop(ips, hessian):
    """Dense Jacobians (BASELINE config 2) on the device-resident loop (csrc/cg.hip
    cg_iterate_dense): same iterates, counts and exits as the statement-by-statement driver
    and as the oracle (pivoted QR projections, like the reference) -- tolerance exit,
    trust-region exit, and a box."""
    import oracle
    import ipsolver.cg_fused as cg_fused
    from ipsolver.dense import DeviceDense
    rng = np.random.default_rng(3)
    m, n = 60, 400
    A = rng.standard_normal((m, n))
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    Hh = G @ G.T + np.eye(n)
    if hessian == "csr":
        Hh = np.triu(np.tril(Hh, 2), -2)
        H = ips.dv.DeviceCSR.from_scipy(sps.csr_matrix(Hh))
    else:
        H = DeviceDense.from_host(Hh)
    c = rng.standard_normal(n)
    b = A @ rng.standard_normal(n) * 0.01
    Z, LS, Y = ips.proj.projections(DeviceDense.from_host(A))
    assert cg_fused.supports(H, Z, Y)
    Zo, _, Yo = oracle.projections(A)
    x_free, _ = oracle.projected_cg(sps.csr_matrix(Hh), c, Zo, Yo, b, tol=1e-14)
    for kw in (dict(tol=1e-14), dict(), dict(trust_radius=0.6 * np.linalg.norm(x_free)),
               dict(lb=np.full(n, -0.5 * np.abs(x_free).max()),
                    ub=np.full(n, 0.5 * np.abs(x_free).max()), max_iter=25)):
        calls = cg_fused.STATS["calls"]
        x, info = ips.qp.projected_cg(H, c, Z, Y, b, **kw)
        assert cg_fused.STATS["calls"] == calls + 1          # the device-resident loop ran
        xg, info_g = ips.qp.projected_cg(H, c, Z, Y, b, return_all=True, **kw)
        xo, info_o = oracle.projected_cg(sps.csr_matrix(Hh), c, Zo, Yo, b, **kw)
        assert info == {k: info_g[k] for k in info} == {k: info_o[k] for k in info}, kw
        close(x, xo, 1e-9)
        close(x, host(xg), 1e-11)


def test_box_schur_tail_is_refused_when_its_workgroups_outnumber_the_partials(ips):
    """the per-item back substitution as the tail of the Schur solve's kernel writes
    one ||g||^2 partial per WORKGROUP OF THE SOLVE into an array the consumer folds
    ceil(items / 1024) entries of.  General rows of three variables (two variables per row:
    n = 2 m + 1) give the solve more workgroups (m / 260) than that count (3 m / 1024): the tail
    must be refused (the separate k_pairs_post launch runs) and the projection, its ||g||^2 and
    a whole CG run must be right -- before the check they silently were not."""
    import ctypes
    import torch
    import oracle
    import ipsolver.cg_fused as cg_fused
    from ipsolver import _hip
    from ipsolver.boxschur import BoxSchurNormalSolver
    m = 6000
    n = 2 * m + 1
    rng = np.random.default_rng(3)
    rows = np.repeat(np.arange(m), 3)
    cols = (2 * np.arange(m)[:, None] + np.arange(3)[None, :]).ravel()
    J = sps.csr_matrix((rng.standard_normal(3 * m), (rows, cols)), shape=(m, n))
    I = sps.eye(n, format="csr")
    s = rng.uniform(1e-4, 2.0, m + 2 * n)
    A = sps.bmat([[J, sps.diags(s[:m]), None, None],
                  [-I, None, sps.diags(s[m:m + n]), None],
                  [I, None, None, sps.diags(s[m + n:])]], format="csr")
    A.sort_indices()
    N, M = A.shape[1], A.shape[0]
    Ad = ips.dv.DeviceCSR.from_scipy(A)
    Z, LS, Y = ips.proj.projections(Ad)
    solver = Z.projector.solver
    assert isinstance(solver, BoxSchurNormalSolver)
    args = solver.c_args()
    assert args.AR_rowlen == 4 and args.gaffine and args.grp2       # the tail's other conditions hold
    lib = _hip.load()
    nblk = lib.ipx_boxschur_project_count(ctypes.byref(args))
    geo = (ctypes.c_int32 * 2)()
    assert lib.ipx_banded_decoupled_geometry(ctypes.c_void_p(solver.inner.handle), geo)
    assert lib.ipx_banded_pcr_level(ctypes.c_void_p(solver.inner.handle)) > 0
    assert geo[1] > nblk, (geo[1], nblk)            # more workgroups than partial entries
    assert not args.post_own_g                      # ... so the tail is not offered
    r = rng.standard_normal(N)
    rd = ips.dv.DVec.from_host(r)
    g = torch.empty(N, dtype=torch.float64, device="cuda")
    pg = torch.full((2 * nblk + 64,), float("nan"), dtype=torch.float64, device="cuda")
    pres = torch.zeros(M // 256 + 2, dtype=torch.float64, device="cuda")
    n3, n4 = ctypes.c_int32(0), ctypes.c_int32(0)
    _hip.call("ipx_boxschur_project", ctypes.byref(args), ips.dv._p(rd.t), ips.dv._p(g),
              ips.dv._p(pg), ctypes.byref(n3), ips.dv._p(pres), ctypes.byref(n4), None,
              ips.dv.stream_ptr())
    got = g.cpu().numpy()
    lu = sps.linalg.splu(sps.csc_matrix(A @ A.T))
    want = r - A.T @ lu.solve(A @ r)
    assert np.max(np.abs(got - want)) <= 1e-10 * np.max(np.abs(r))
    assert n3.value == nblk
    assert bool(torch.isnan(pg[2 * nblk:]).all())   # nothing written past the two halves
    assert abs(float(pg[:nblk].sum()) - float(got @ got)) <= 1e-12 * float(got @ got)
    assert float(pg[nblk:2 * nblk].abs().sum()) == 0.0
    # a C caller that hands the tail's tables over anyway is turned down by the library itself
    # (the launch falls back to the separate back substitution: same g)
    from ipsolver.boxschur import _i32
    own = _i32(np.minimum(np.arange(geo[1] + 1) * 1000, n))
    forced = type(args).from_buffer_copy(args)
    forced.post_own_g, forced.post_own_e = own.data_ptr(), own.data_ptr()
    forced.post_rows_wg, forced.post_reach = geo[0], 1
    g2 = torch.empty_like(g)
    pg.fill_(float("nan"))
    _hip.call("ipx_boxschur_project", ctypes.byref(forced), ips.dv._p(rd.t), ips.dv._p(g2),
              ips.dv._p(pg), ctypes.byref(n3), ips.dv._p(pres), ctypes.byref(n4), None,
              ips.dv.stream_ptr())
    assert np.array_equal(g2.cpu().numpy(), got)
    assert bool(torch.isnan(pg[2 * nblk:]).all())
    # the device loop on the barrier-shaped subproblem against the oracle
    Hz = sps.diags(rng.uniform(0.5, 2.0, N), format="csr")
    Hd = ips.dv.DeviceCSR.from_scipy(Hz)
    c = rng.standard_normal(N)
    b = np.zeros(M)
    lb = np.concatenate((np.full(n, -np.inf), np.full(N - n, -0.995)))
    x, info = cg_fused.projected_cg(Hd, ips.dv.DVec.from_host(c), Z, Y, ips.dv.DVec.from_host(b),
                                    trust_radius=5.0, lb=ips.dv.DVec.from_host(lb), tol=1e-10)
    Zo, _, Yo = oracle.projections(A)
    xo, io = oracle.projected_cg(Hz, c, Zo, Yo, b, trust_radius=5.0, lb=lb,
                                 ub=np.full(N, np.inf), tol=1e-10)
    assert (info["stop_cond"], info["hits_boundary"], info["niter"]) == \
        (io["stop_cond"], io["hits_boundary"], io["niter"])
    close(x, xo, 1e-10)


def test_banded_refresh_in_three_launches_and_the_chunk_factorization_on_demand():
    """``ipx_banded_refactor``: on a handle whose last verdict was clean and that runs the
    cyclic-reduction solve, a numeric refresh is the band, the reduction's own check and the
    verdict kernel; the chunked LDL' is left out until a solve needs it.  Solves that read the
    band only (out of place) and solves that need the chunk factors (in place) both equal the
    solves of a full factorization of the same matrix bit for bit; a matrix whose verdict
    differs (a row of zeros: a reduced diagonal entry that is not positive) raises the verdict
    word."""
    import ctypes
    import torch
    from banded_setup import load_synthetic
    from ipsolver import _hip, device as dv, projector
    from ipsolver.device import DVec, _p, stream_ptr
    lib = _hip.load()
    rng = np.random.default_rng(5)
    prob = load_synthetic().CenteredBandedNLP(40000, 4000, eps=1e-3)
    J = prob.constr_jac(prob.x0).tocsr()
    A = dv.DeviceCSR.from_scipy(J)
    full = projector.BandedNormalSolver(A)                 # blocking factorization: clean verdict
    assert lib.ipx_banded_pcr_level(ctypes.c_void_p(full.handle)) > 0
    launches = lambda: int(lib.ipx_launch_count())

    class Deferred:
        verdict = torch.zeros(2, dtype=torch.float64, device="cuda")
    # new values on the same pattern, through the pooled handle
    J2 = J.copy()
    J2.data = J.data * (1.0 + 0.1 * rng.standard_normal(J.nnz))
    A2 = dv.DeviceCSR(A.pattern, torch.from_numpy(J2.data).cuda())
    ref = projector.BandedNormalSolver(A2)                 # (a second handle: the reference solves)
    del full                                               # -> the pool: recycled below
    n0 = launches()
    lazy = projector.BandedNormalSolver(A2, deferred=Deferred)
    assert lazy.pending and launches() - n0 == 3
    assert dv.read_doubles(Deferred.verdict, 1)[0] == 0.0
    w = DVec(torch.from_numpy(rng.standard_normal(4000)).cuda())
    out_lazy, out_ref = lazy.solve(w), ref.solve(w)
    assert torch.equal(out_lazy.t, out_ref.t)
    # in place: the chunk factors, made now (with the blocking verdict)
    a, b = w.t.clone(), w.t.clone()
    n0 = launches()
    _hip.call("ipx_banded_solve", ctypes.c_void_p(lazy.handle), _p(a), _p(a), stream_ptr())
    assert launches() - n0 > 3                             # (factor + checks + the solve)
    _hip.call("ipx_banded_solve", ctypes.c_void_p(ref.handle), _p(b), _p(b), stream_ptr())
    assert torch.equal(a, b)
    S = (J2 @ J2.T).tocsc()
    import scipy.sparse.linalg as spla
    want = spla.spsolve(S, w.t.cpu().numpy())
    assert np.abs(a.cpu().numpy() - want).max() <= 1e-10 * np.abs(want).max()
    # a matrix the assumed verdict does not hold for
    J3 = J2.copy()
    J3.data[J3.indptr[1000]:J3.indptr[1001]] = 0.0
    A3 = dv.DeviceCSR(A.pattern, torch.from_numpy(J3.data).cuda())
    del lazy
    bad = projector.BandedNormalSolver(A3, deferred=Deferred)
    assert bad.pending and dv.read_doubles(Deferred.verdict, 1)[0] != 0.0
