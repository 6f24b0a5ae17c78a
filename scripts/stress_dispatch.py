"""Random problems of every shape through ``minimize_constrained(..., options={'shard': ...})`` on
W gloo ranks with the numpy twin of the local kernels, against the same call on the
single-process oracle backend (dev tool; CPU only).

    python scripts/stress_dispatch.py [world] [seeds]
"""
import os
import sys
import traceback
import warnings

import numpy as np
import scipy.sparse as sps
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def problem(seed):
    rng = np.random.default_rng(seed)
    shape = ["banded_eq", "banded_boxed", "random_eq", "mixed", "bounds"][seed % 5]
    if shape in ("banded_eq", "banded_boxed"):
        from banded_setup import load_synthetic
        m = int(rng.integers(540, 900))
        n = m * int(rng.integers(8, 12))
        prob = load_synthetic().CenteredBandedNLP(n, m, eps=1e-3 if shape == "banded_eq" else 1.0,
                                                  seed=seed)
        return shape, prob, None
    n, m_eq, m_in = int(rng.integers(60, 140)), int(rng.integers(8, 20)), int(rng.integers(10, 30))
    A = sps.random(m_eq, n, density=0.1, random_state=np.random.RandomState(seed), format="csr")
    A = sps.csr_matrix(A + sps.csr_matrix((np.ones(m_eq), (np.arange(m_eq), np.arange(m_eq))),
                                          shape=(m_eq, n)))
    B = sps.random(m_in, n, density=0.1, random_state=np.random.RandomState(seed + 1), format="csr")
    B = sps.csr_matrix(B + sps.csr_matrix((np.ones(m_in), (np.arange(m_in), m_eq + np.arange(m_in))),
                                          shape=(m_in, n)))
    x0 = 0.3 * rng.standard_normal(n)
    d = dict(n=n, A=A, B=B, x0=x0, q=rng.uniform(0.5, 2.0, n), c=rng.standard_normal(n),
             ub_in=B.dot(x0) + rng.uniform(0.2, 1.0, m_in))
    return shape, None, d


def solve(ipsolver, seed, **options):
    shape, prob, d = problem(seed)
    rows = []

    def record(state):
        rows.append([int(state.niter), int(state.cg_niter)])
        return False
    kw = dict(callback=record, options=options, max_iter=40)
    if shape == "banded_eq":
        res = ipsolver.minimize_constrained(prob.fun, prob.x0, prob.grad, prob.hess,
                                            prob.constraints(ipsolver), **kw)
    elif shape == "banded_boxed":
        cons = (prob.constraints(ipsolver, ("less", 0.0)),
                ipsolver.BoxConstraint(("interval", -0.8, 0.8)))
        res = ipsolver.minimize_constrained(prob.fun, prob.x0, prob.grad, prob.hess, cons, **kw)
    else:
        f = lambda x: 0.5 * x.dot(d["q"] * x) + d["c"].dot(x)
        g = lambda x: d["q"] * x + d["c"]
        h = lambda x: sps.diags(d["q"]).tocsr()
        eq = ipsolver.LinearConstraint(d["A"], ("equals", d["A"].dot(d["x0"])))
        if shape == "random_eq":
            cons = [eq]
        elif shape == "mixed":
            cons = [eq, ipsolver.LinearConstraint(d["B"], ("less", d["ub_in"]))]
        else:
            cons = [eq, ipsolver.BoxConstraint(("greater", d["x0"] - 0.6))]
        res = ipsolver.minimize_constrained(f, d["x0"], g, h, cons, sparse_jacobian=True, **kw)
    return shape, res, np.array(rows)


def worker(rank, world, port, seeds, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ipsolver
        from oracle.numpy_local import NumpyOps
        for seed in range(seeds):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                shape, res, rows = solve(ipsolver, seed, shard=NumpyOps())
            if rank == 0:
                np.savez(os.path.join(out_dir, "s%d.npz" % seed), x=res.x, rows=rows,
                         status=res.status)
    except Exception:
        print("RANK", rank, "FAILED:\n", traceback.format_exc(), flush=True)
        os._exit(3)
    dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile
    import ipsolver
    import oracle.numpy_backend as nb
    from ipsolver import backend
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    out = tempfile.mkdtemp()
    mp.spawn(worker, args=(w, 29700 + w, s, out), nprocs=w, join=True)
    bad = 0
    for seed in range(s):
        got = np.load(os.path.join(out, "s%d.npz" % seed))
        with backend.use(nb), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            shape, res, rows = solve(ipsolver, seed, shard=False)
        k = min(8, len(rows), len(got["rows"]))
        same = np.array_equal(rows[:k], got["rows"][:k])
        dx = np.max(np.abs(got["x"] - res.x)) / max(np.max(np.abs(res.x)), 1e-300)
        ok = same and int(got["status"]) == res.status and dx <= 1e-4
        bad += not ok
        print("seed %d %-12s status %d/%d rows %d/%d first %d equal %s dx %.1e %s"
              % (seed, shape, int(got["status"]), res.status, len(got["rows"]), len(rows), k, same,
                 dx, "" if ok else "<-- CHECK"))
    print("flagged:", bad)
