"""Random banded problems on W ranks sharing one GPU (peer-mailbox transport, collectives in the
loop kernels' prologues): the sharded device loop against the single-GPU loop, for every exit of
projected CG -- tolerance, trust region, box events, refinement on every application (dev tool).

    python scripts/stress_sharded.py [world] [seeds] [first seed]
"""
import os
import sys
import traceback

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, seeds, first=0):
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("STRESS_DUMP_AFTER", "600")), exit=True)
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["IPX_SHARD_TRANSPORT"] = "ipc"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bad = 0
    try:
        from ipsolver import sharded, qp
        import ipsolver.device as dv
        import ipsolver.projector as proj
        from ipsolver.operators import DeviceHessian
        from ipsolver.synthetic import CenteredBandedNLP
        for seed in range(first, first + seeds):
            rng = np.random.default_rng(seed)
            m = int(rng.integers(world * 2 * 260 + 50, world * 2 * 260 + 4000))
            n = m * int(rng.integers(7, 13))
            prob = CenteredBandedNLP(n, m, seed=seed)
            x = prob.x0
            v = 0.1 * rng.standard_normal(m)
            A_h, H_h = prob.constr_jac(x).tocsr(), prob.hess(x)
            hd = prob.kappa * prob.Wt.dot(v)
            c_h = prob.grad(x)
            lay = sharded.ShardLayout(A_h.indptr, A_h.indices, A_h.shape, world, rank)
            sh = sharded.Sharding(lay, sharded.ShardComm(), sharded.HipOps())
            A = sharded.ShardCSR.from_global(sh, A_h)
            H = sharded.ShardHessian.from_global(sh, H_h, hd)
            c = sh.from_global(c_h, "col")
            A1 = dv.DeviceCSR.from_scipy(A_h)
            H1 = DeviceHessian(n, csr=dv.DeviceCSR.from_scipy(H_h), diag=dv.DVec.from_host(hd))
            if os.environ.get("STRESS_GEOMETRY"):
                import ctypes
                from ipsolver import _hip
                Pz = sharded.projections(A)[0].projector
                bd = sharded._banded_of(Pz)
                geo = (ctypes.c_int32 * 2)()
                dec = _hip.load().ipx_banded_decoupled_geometry(ctypes.c_void_p(bd.handle), geo) \
                    if bd is not None else -1
                print("  rank %d seed %d: banded %s k=%s decoupled=%s geo=%s row geom %s loop ok %s"
                      % (rank, seed, bd is not None, getattr(bd, "k", None), dec, list(geo),
                         sh.lay.geom("row"), sharded._loop_geometry_ok(Pz)), flush=True)
            for refine in (False, True):
                kw = dict(orth_tol=1e-30, max_refin=2) if refine else {}
                Z, LS, Y = sharded.projections(A, **kw)
                Z1, _, Y1 = proj.projections(A1, **kw)
                g = dv.norm(Z1.dot(c_h))
                cases = {"free": dict(tol=0, max_iter=25), "tol": dict(),
                         "ball": dict(tol=0, max_iter=25, trust_radius=0.6 * g),
                         "box": dict(tol=0, max_iter=25, lb=np.full(n, -0.02), ub=np.full(n, 0.03))}
                for name, k in cases.items():
                    ks = {a: (sh.from_global(b, "col") if a in ("lb", "ub") else b)
                          for a, b in k.items()}
                    xs, info = qp.projected_cg(H, c, Z, Y, sh.zeros("row"), **ks)
                    x1, info1 = qp.projected_cg(H1, c_h, Z1, Y1, np.zeros(m), **k)
                    xs_h, x1_h = xs.to_host(), x1.to_host()
                    err = np.max(np.abs(xs_h - x1_h)) / max(np.max(np.abs(x1_h)), 1e-300)
                    same = (info["niter"], info["stop_cond"], info["hits_boundary"]) == \
                        (info1["niter"], info1["stop_cond"], info1["hits_boundary"])
                    ok = same and err <= 1e-11
                    bad += not ok
                    if rank == 0 and not ok:
                        print("seed %d n=%d m=%d refine=%d %-5s: sharded %s single %s err %.1e  <-- CHECK"
                              % (seed, n, m, refine, name, info, info1, err), flush=True)
            mb = sh.mailbox()              # (collective when first used: every rank asks)
            if rank == 0:
                print("seed %d n=%d m=%d world=%d transport=%s prologue launches %d: ok so far, bad=%d"
                      % (seed, n, m, world, "ipc" if mb is not None else "dist",
                         mb.fused_launches() if mb is not None else 0, bad), flush=True)
    except Exception:
        print("RANK", rank, "FAILED:\n", traceback.format_exc(), flush=True)
        os._exit(3)
    dist.destroy_process_group()
    if rank == 0:
        print("bad:", bad)


if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    f = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    mp.spawn(worker, args=(w, 29620 + w, s, f), nprocs=w, join=True)
