"""Dev tool: the refinement scenario of test_sharded_fused_loop_multi_rank on 3 ranks with every
rank's exception printed."""
import os
import sys
import traceback

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port):
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["IPX_SHARD_TRANSPORT"] = "ipc"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ipsolver import sharded, qp
        from test_gpu_qp import _sharded_problem
        inst, sh, A, H = _sharded_problem(world, rank, 20000, 2000)
        c = sh.from_global(inst.c, "col")
        Z, LS, Y = sharded.projections(A)
        x, info = qp.projected_cg(H, c, Z, Y, sh.zeros("row"), tol=0, max_iter=15)
        print(rank, "free ok", info["niter"], flush=True)
        Zr, _, Yr = sharded.projections(A, orth_tol=1e-30, max_refin=2)
        x, info = qp.projected_cg(H, c, Zr, Yr, sh.zeros("row"), tol=0, max_iter=15)
        print(rank, "refine ok", info["niter"], flush=True)
    except Exception:
        print("RANK", rank, "FAILED:\n", traceback.format_exc(), flush=True)
        os._exit(3)
    dist.destroy_process_group()


if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mp.spawn(worker, args=(w, 29611), nprocs=w, join=True)
