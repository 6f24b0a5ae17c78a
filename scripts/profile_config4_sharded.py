"""Where a sharded config-4 solve spends its wall clock (dev tool): WORLD ranks sharing one GPU
over gloo, device-resident sharded callbacks; cProfile of rank 0 + the collectives' counters.
    python scripts/profile_config4_sharded.py [world] [n]"""
import cProfile, io, os, pstats, socket, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))


def worker(rank, world, port, n):
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    torch.set_num_threads(1)       # (as torch.distributed.run sets OMP_NUM_THREADS=1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ipsolver import sharded
    from ipsolver.synthetic import CenteredBandedNLP, ShardedCallbacks
    warnings.simplefilter("ignore")
    prob = CenteredBandedNLP(n, n // 10, eps=1e-3)
    A = prob.A0.tocsr()
    lay = sharded.ShardLayout(A.indptr, A.indices, A.shape, world, rank)
    sh = sharded.Sharding(lay, sharded.ShardComm(), sharded.HipOps())
    cb = ShardedCallbacks(prob, sh)

    def solve():
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        res = sharded.minimize_equality_constrained(sh, cb.fun, cb.grad, cb.lagr_hess, cb.constr_fun,
                                                    cb.constr_jac, cb.x0, method="tr_interior_point")
        torch.cuda.synchronize()
        return res, time.perf_counter() - t0
    for k in range(3):
        before = dict(sh.comm.stats)
        res, dt = solve()
        if rank == 0:
            print("run %d: %.1f ms status %d %d/%d" % (k, 1e3 * dt, res.status, res.niter, res.cg_niter),
                  {k: sh.comm.stats[k] - before[k] for k in before})
    pr = cProfile.Profile()
    if rank == 0:
        pr.enable()
    res, dt = solve()
    if rank == 0:
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
        print(s.getvalue()[:9000])
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(20)
        print(s.getvalue()[:4000])
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(world, port, n), nprocs=world, join=True)
