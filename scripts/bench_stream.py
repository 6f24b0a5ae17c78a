"""Streaming micro-benchmark: fixed cost and bandwidth of a simple ipx kernel (dev tool)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import torch
from ipsolver import device as dv, _hip
lib = _hip.load()
st = dv.stream_ptr()
for n in (1 << 10, 1 << 16, 1 << 18, 1000000, 4000000, 16000000, 64000000):
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    y = torch.randn(n, dtype=torch.float64, device="cuda")
    o = torch.empty_like(x)
    for name, fn, nbytes in (
        ("axpby", lambda: lib.ipx_axpby(n, 1.0, dv._p(x), 2.0, dv._p(y), dv._p(o), st), 24 * n),
        ("torch_add", lambda: torch.add(x, y, out=o), 24 * n),
        ("torch_copy", lambda: o.copy_(x), 16 * n)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        N = 200
        t0 = time.perf_counter()
        for _ in range(N):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        print("n=%9d %-10s %8.2f us  %7.1f GB/s" % (n, name, dt * 1e6, nbytes / dt / 1e9))
