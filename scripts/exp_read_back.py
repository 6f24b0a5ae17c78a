"""Dev experiment: cost of one blocking scalar read-back behind a small kernel -- torch's
``tensor[:k].tolist()`` (pageable destination) against ipx_read_doubles (pinned staging)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ip-nonlinear-solver_amd"))
import torch
from ipsolver import _hip, device as dv
lib = _hip.load()
c = dv.ctx()
st = dv.stream_ptr()
x = dv.DVec.full(1000000, 1.0)
buf = (ctypes.c_double * 512)()
def kernel():
    _hip.call("ipx_fill", 1000000, 2.0, dv._p(x.t), st)      # ~3 us of GPU work
for name, rd in (("tolist", lambda k: c.out[:k].tolist()),
                 ("ipx_read_doubles", lambda k: (_hip.call("ipx_read_doubles", dv._p(c.out), k, buf, st), buf[:k])[1])):
    for k in (4, 16):
        for _ in range(50):
            kernel(); rd(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 2000
        for _ in range(R):
            kernel(); rd(k)
        dt = (time.perf_counter() - t0) / R
        t0 = time.perf_counter()
        for _ in range(R):
            rd(k)
        dt2 = (time.perf_counter() - t0) / R
        print("%-18s k=%2d: kernel + read %.1f us, read alone (idle GPU) %.1f us" % (name, k, 1e6 * dt, 1e6 * dt2))
