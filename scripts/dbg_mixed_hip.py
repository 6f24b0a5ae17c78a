"""dev tool: the mixed-constraints problem of tests/test_sharded_gloo.py on 2 ranks with the HIP
kernels (plain partition), its trace tail next to the single-process oracle's."""
import os, sys, socket, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ip-nonlinear-solver_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch.multiprocessing as mp

if __name__ == "__main__":
    from test_sharded_gloo import _mixed_worker, _mixed_problem, _mixed_solve
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    path = "/tmp/mixed_dbg.npz"
    mp.spawn(_mixed_worker, args=(2, port, path, 1000, True), nprocs=2, join=True)
    got = np.load(path)
    import ipsolver, oracle.numpy_backend as nb
    from ipsolver import backend
    with backend.use(nb), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res, want = _mixed_solve(ipsolver, _mixed_problem(), 1000, shard=False)
    have = got["rows"]
    np.set_printoptions(linewidth=200, precision=4)
    print("product rows", len(have), "oracle rows", len(want))
    print("product tail:\n", have[24:])
    print("oracle tail:\n", want[24:])
    print("dx", np.max(np.abs(got["x"] - res.x)) / np.max(np.abs(res.x)), "dfun", abs(float(got["fun"]) - res.fun) / abs(res.fun))
