"""BASELINE config 5 on the sharded backend (barrier method, box + nonlinear inequalities).
    python -m torch.distributed.run --nproc-per-node N scripts/config5_sharded.py [n] [m] [max_iter]
IPX_BENCH_BACKEND=gloo lets several ranks share one GPU (collectives staged through the host)."""
import json, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch, torch.distributed as dist
from ipsolver import sharded
from ipsolver.synthetic import CenteredBandedNLP, ShardedCallbacks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
m = int(sys.argv[2]) if len(sys.argv) > 2 else n // 10
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
backend = os.environ.get("IPX_BENCH_BACKEND", "nccl")
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local if backend == "nccl" else 0)
dist.init_process_group(backend)
warnings.simplefilter("ignore")
prob = CenteredBandedNLP(n, m, eps=1.0)
A = prob.A0.tocsr()
lay = sharded.ShardLayout(A.indptr, A.indices, A.shape, dist.get_world_size(), dist.get_rank())
sh = sharded.Sharding(lay, sharded.ShardComm(), sharded.HipOps())
cb = ShardedCallbacks(prob, sh)
t0 = time.time()
res = sharded.minimize_box_inequality(sh, cb.fun, cb.grad, cb.lagr_hess, cb.constr_fun, cb.constr_jac,
                                      cb.x0, sh.full("col", -0.8), sh.full("col", 0.8), max_iter=max_iter)
torch.cuda.synchronize()
wall = time.time() - t0
x = res.x.to_host()
if dist.get_rank() == 0:
    print(json.dumps({"n": n, "m": m, "world": dist.get_world_size(), "status": int(res.status),
                      "niter": int(res.niter), "cg_niter": int(res.cg_niter), "wall_s": wall,
                      "cg_it_per_s": res.cg_niter / wall, "optimality": float(res.optimality),
                      "constr_violation": float(res.constr_violation), "fun": float(res.fun),
                      "active_bounds": int(np.sum(np.abs(np.abs(x) - 0.8) < 1e-6)),
                      "collectives": sh.comm.stats, "fused": sharded.STATS,
                      "transport": sh.transport,
                      "prologue_collective_launches": sh.mailbox().fused_launches() if sh.mailbox()
                      else 0}))
dist.destroy_process_group()
