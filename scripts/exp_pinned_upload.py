import time, numpy as np, torch
a = np.random.default_rng(0).standard_normal((10000, 10000))
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
def up(t, nb=False):
    torch.cuda.synchronize(); t0 = time.time()
    d = t.to(dev, non_blocking=nb); torch.cuda.synchronize()
    return time.time() - t0, d
t = torch.from_numpy(a)
for i in range(3):
    print("pageable", round(up(t)[0]*1e3, 1), "ms")
rt = torch.cuda.cudart()
t0 = time.time()
rc = rt.cudaHostRegister(t.data_ptr(), t.numel() * 8, 0)
print("register rc", rc, round((time.time()-t0)*1e3, 1), "ms", "is_pinned", t.is_pinned())
for i in range(3):
    dt, d = up(t, True)
    print("registered", round(dt*1e3, 1), "ms", 0.8/dt, "GB/s")
print("equal", bool(torch.equal(d.cpu(), t)))
t0 = time.time(); rc = rt.cudaHostUnregister(t.data_ptr()); print("unregister rc", rc, round((time.time()-t0)*1e3,1), "ms")
p = torch.empty_like(t).pin_memory()
t0=time.time(); p.copy_(t); print("host copy to pinned", round((time.time()-t0)*1e3,1), "ms")
for i in range(2):
    dt, d = up(p, True); print("pinned buffer", round(dt*1e3,1), "ms")
