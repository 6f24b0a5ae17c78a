"""Randomised cross-check of the row-sharded device loop (ipsolver/sharded.py: halo partition,
the loop's collectives inside its kernels over the peer mailboxes) on `world` ranks sharing
cuda:0: random band shapes -- rows of 3..16 entries, tridiagonal A A', row counts that do and do
not divide into the 260-row blocks, with and without a box, three trust radii -- each solved by
the sharded loop and, inside every rank, by the single-GPU loop on the whole problem
(qp_subproblem.py:332-637 both).  A shape the halo partition refuses must be refused by every
rank alike (NotImplementedError before any collective).

    python tests/fuzz_sharded.py [world] [cases] [seed] [solves|mixes]    (tests/test_gpu_e2e.py: 2 ranks)"""
import os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np


def worker(rank, world, port, out_path, cases, seed):
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
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
        rng = np.random.default_rng(seed)
        lines, worst = [], 0.0
        for case in range(cases):
            rl = int(rng.integers(3, 17))
            shift = int(rng.integers((rl + 1) // 2, rl + 1))
            m = int(rng.choice([1300, 2600, 5000, 5201, 26001]))
            n = (m - 1) * shift + rl + int(rng.integers(0, 30))
            n += n % 2
            rows = np.repeat(np.arange(m), rl)
            cols = (np.arange(m)[:, None] * shift + np.arange(rl)[None, :]).ravel()
            A_h = sp.csr_matrix((rng.uniform(0.5, 1.5, m * rl) * rng.choice([-1.0, 1.0], m * rl),
                                 (rows, cols)), shape=(m, n))
            off = rng.uniform(-0.4, 0.4, n - 1)
            H_h = sp.diags([off, rng.uniform(1.5, 2.5, n), off], [-1, 0, 1], format="csr")
            c_h = rng.standard_normal(n)
            box = bool(rng.random() < 0.4)
            kw = dict(tol=0, max_iter=14,
                      trust_radius=float(rng.choice([1e300, np.inf, 0.02 * np.sqrt(n)])))
            if box:
                kw.update(lb=np.full(n, -0.05), ub=np.full(n, 0.08))
            tag = "case %2d rl=%2d shift=%2d m=%6d n=%7d box=%d radius=%-6g" % (
                case, rl, shift, m, n, box, kw["trust_radius"])
            try:
                lay = sharded.ShardLayout(A_h.indptr, A_h.indices, A_h.shape, world, rank)
            except NotImplementedError as exc:
                lines.append(tag + "  refused by the layout: %s" % str(exc)[:60])
                continue
            sh = sharded.Sharding(lay, sharded.ShardComm(), sharded.HipOps())
            A = sharded.ShardCSR.from_global(sh, A_h)
            H = sharded.ShardHessian.from_global(sh, H_h)
            Z, LS, Y = sharded.projections(A)
            c = sh.from_global(c_h, "col")
            before = sharded.STATS["fused_calls"]
            ks = {a: (sh.from_global(b, "col") if a in ("lb", "ub") else b) for a, b in kw.items()}
            xs, info = qp.projected_cg(H, c, Z, Y, sh.zeros("row"), **ks)
            fused = sharded.STATS["fused_calls"] - before
            A1 = dv.DeviceCSR.from_scipy(A_h)
            H1 = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(H_h))
            Z1, _, Y1 = proj.projections(A1)
            x1, info1 = qp.projected_cg(H1, c_h, Z1, Y1, np.zeros(m), **kw)
            x1 = x1.to_host()
            d = float(np.max(np.abs(xs.to_host() - x1)) / max(np.max(np.abs(x1)), 1e-300))
            lines.append(tag + "  loop=%d transport=%s  |dx| %.1e  %s" % (fused, sh.transport, d, info))
            assert info == info1, (tag, info, info1)
            assert d <= 1e-10, lines[-1]
            worst = max(worst, d)
        if rank == 0:
            np.savez(out_path, worst=worst, lines=np.array(lines))
    finally:
        dist.destroy_process_group()


def run(world, cases, seed, out_path, verbose=True):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(world, port, out_path, cases, seed), nprocs=world, join=True)
    got = np.load(out_path)
    if verbose:
        print("\n".join(got["lines"]))
    return float(got["worst"]), [str(l) for l in got["lines"]]




# ---- whole solves through the public API ------------------------------------------------------
def solve_worker(rank, world, port, out_path, cases, seed):
    """``minimize_constrained(..., options={'shard': True})`` (numpy callbacks, the reference's
    constraint classes) on random banded NLP shapes -- equality rows by both methods, inequality
    rows + a box by the barrier method -- against the same call without sharding, inside every
    rank."""
    import warnings
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ipsolver
        from ipsolver import sharded
        from ipsolver.synthetic import CenteredBandedNLP
        rng = np.random.default_rng(seed)
        lines, worst = [], 0.0
        for case in range(cases):
            bw = int(rng.integers(3, 17))
            stride = int(rng.integers((bw + 1) // 2 + 1, bw + 3))
            kind = str(rng.choice(["eq-barrier", "eq-sqp", "ineq+box"]))
            m = int(rng.choice([600, 1300, 2700])) if kind != "ineq+box" else 600
            n = m * stride + int(rng.integers(0, stride))
            prob = CenteredBandedNLP(n, m, bw=bw, seed=int(rng.integers(1 << 20)),
                                     eps=1.0 if kind == "ineq+box" else 1e-3)
            if kind == "ineq+box":
                cons = (prob.constraints(ipsolver, ("less", 0.0)),
                        ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
                prob.x0 = np.clip(prob.x0, -0.75, 0.75)
                method, max_iter = "tr_interior_point", 12
            else:
                cons = prob.constraints(ipsolver)
                method = "tr_interior_point" if kind == "eq-barrier" else "equality_constrained_sqp"
                max_iter = 1000
            outs = []
            for shard in (True, False):
                rows = []

                def record(state):
                    rows.append([int(state.niter), int(state.cg_niter), float(state.optimality),
                                 float(state.constr_violation)])
                    return False
                before = sharded.STATS["fused_calls"]
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    res = ipsolver.minimize_constrained(
                        prob.fun, prob.x0, prob.grad, prob.hess, cons, method=method,
                        callback=record, max_iter=max_iter, options={"shard": True} if shard else {})
                outs.append((res, np.array(rows), sharded.STATS["fused_calls"] - before))
            (got, rows, fused), (want, wrows, _) = outs
            dx = float(np.max(np.abs(got.x - want.x)) / max(1.0, np.max(np.abs(want.x))))
            lines.append("case %2d %-10s bw=%2d stride=%2d m=%5d n=%6d  %d/%d outer %d/%d CG  sharded loops %d  |dx| %.1e" % (
                case, kind, bw, stride, m, n, got.niter, want.niter, got.cg_niter, want.cg_niter, fused, dx))
            k = min(10, len(rows), len(wrows))
            assert np.array_equal(rows[:k, :2], wrows[:k, :2]), lines[-1]
            assert np.allclose(rows[:k, 2:], wrows[:k, 2:], rtol=1e-6, atol=1e-10), lines[-1]
            if kind != "ineq+box":
                assert (got.status, got.niter, got.cg_niter) == (want.status, want.niter, want.cg_niter), lines[-1]
                assert dx <= 1e-9, lines[-1]
            else:
                assert dx <= 1e-6, lines[-1]
            worst = max(worst, dx)
        if rank == 0:
            np.savez(out_path, worst=worst, lines=np.array(lines))
    finally:
        dist.destroy_process_group()


def run_solves(world, cases, seed, out_path, verbose=True):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(solve_worker, args=(world, port, out_path, cases, seed), nprocs=world, join=True)
    got = np.load(out_path)
    if verbose:
        print("\n".join(got["lines"]))
    return float(got["worst"]), [str(l) for l in got["lines"]]


# ---- random mixes of the three constraint classes on the plain block partition ----------------
def mixes_worker(rank, world, port, out_path, cases, seed):
    """tests/fuzz_minimize.py's random problems (dense / sparse equalities, linear and nonlinear
    inequalities, ragged boxes) through ``minimize_constrained(..., options={'shard': True})``:
    no banded shape, so the plain block partition with its general driver
    -- against the same call without sharding, inside every rank."""
    import warnings
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ipsolver
        import fuzz_minimize as fm
        rng = np.random.default_rng(seed)
        lines, worst = [], 0.0
        for case in range(cases):
            P = fm.problem(rng)
            if any(t.startswith("fd-") for t in P["tags"]):
                continue
            method = P["methods"][-1]
            outs = []
            for shard in (True, False):
                rows = []

                def record(state):
                    rows.append([int(state.niter), int(state.cg_niter), float(state.optimality),
                                 float(state.constr_violation)])
                    return False
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    try:
                        res = ipsolver.minimize_constrained(
                            P["fun"], P["x0"], P["grad"], P["hess"], P["cons"], method=method,
                            sparse_jacobian=True, callback=record, max_iter=40,
                            options={"shard": True} if shard else {})
                    except NotImplementedError as exc:
                        res = str(exc)[:70]
                outs.append((res, np.array(rows)))
            (got, rows), (want, wrows) = outs
            tag = "case %2d n=%3d %-24s %-30s" % (case, P["n"], method, "+".join(P["tags"]) or "-")
            if isinstance(got, str):
                lines.append(tag + "  refused: " + got)
                continue
            k = min(5, len(rows), len(wrows))
            long_cg = np.flatnonzero(np.diff(wrows[:k, 1]) > 12)
            if len(long_cg):
                k = int(long_cg[0]) + 1
            dx = float(np.max(np.abs(got.x - want.x)) / max(1.0, np.max(np.abs(want.x))))
            lines.append(tag + "  %d/%d outer  rows compared %d  |dx| %.1e" % (got.niter, want.niter, k, dx))
            assert np.array_equal(rows[:k, :2], wrows[:k, :2]), lines[-1]
            assert np.allclose(rows[:k, 2:], wrows[:k, 2:], rtol=1e-6, atol=1e-10), lines[-1]
            if got.status in (1, 2) and want.status in (1, 2):
                assert dx <= 1e-4, lines[-1]
                worst = max(worst, dx)
        if rank == 0:
            np.savez(out_path, worst=worst, lines=np.array(lines))
    finally:
        dist.destroy_process_group()


def run_mixes(world, cases, seed, out_path, verbose=True):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(mixes_worker, args=(world, port, out_path, cases, seed), nprocs=world, join=True)
    got = np.load(out_path)
    if verbose:
        print("\n".join(got["lines"]))
    return float(got["worst"]), [str(l) for l in got["lines"]]


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if len(sys.argv) > 4 and sys.argv[4] == "mixes":
        w, _ = run_mixes(world, ncase, seed, "/tmp/fuzz_sharded_mixes.npz")
        print("ok, worst end-point deviation of the sharded mixed problems %.1e" % w)
    elif len(sys.argv) > 4 and sys.argv[4] == "solves":
        w, _ = run_solves(world, ncase, seed, "/tmp/fuzz_sharded_solves.npz")
        print("ok, worst end-point deviation of the sharded solves %.1e" % w)
    else:
        w, _ = run(world, ncase, seed, "/tmp/fuzz_sharded.npz")
        print("ok, worst deviation of the sharded loop from the single-GPU loop %.1e" % w)
