"""Experiment: cost of the partial folds / grid size in the CG vector kernels (dev tool)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import numpy as np, torch
from ipsolver import _hip, device as dv
lib = _hip.load(); st = dv.stream_ptr()
n = 1000000
dev = "cuda"
x, p, r, Hp, g = (torch.randn(n, dtype=torch.float64, device=dev) * 1e-3 for _ in range(5))
state = torch.zeros(16, dtype=torch.float64, device=dev)
init = np.zeros(16); init[0] = init[1] = 1.0; init[3] = np.inf; init[4] = 1e-6
big = torch.full((8192,), 1e-3, dtype=torch.float64, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())

def timeit(name, fn, N=400):
    state.copy_(torch.from_numpy(init))
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); torch.cuda.synchronize()
    bad = float(state[5].item())
    print("%-44s %7.2f us  (stop=%g)" % (name, e0.elapsed_time(e1) / N * 1e3, bad))

for grid in (384, 512, 768, 1024):
    part2 = torch.zeros(2 * grid, dtype=torch.float64, device=dev)
    for np1 in (1, 1465):
        timeit("step1 grid=%d np1=%d" % (grid, np1),
               lambda: lib.ipx_cg_step1(n, P(state), 0, P(big), np1, P(x), P(p), P(r), P(Hp), None, None,
                                        P(part2), grid, st))
    for (a, b, c) in ((1, 1, 1), (grid, 977, 391), (1, 977, 1), (grid, 1, 1), (1, 1, 391)):
        timeit("step2 grid=%d np2=%d np3=%d np4=%d" % (grid, a, b, c),
               lambda: lib.ipx_cg_step2(n, P(state), 0, 0, P(part2), min(a, grid), P(big), b, P(big), c,
                                        P(x), P(p), P(g), grid, st))
