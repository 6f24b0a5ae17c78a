"""Row-sharded projected CG across the GPUs of one node (BASELINE config 4).

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI).
Partition:

* z-space vectors x, p, r, H p are split into contiguous variable blocks, one per
  rank; rank g owns the columns ``A[:, n0:n1]`` of the Jacobian, the matching
  rows of ``A'`` and the rows ``H[n0:n1, :]`` of the Hessian (its few
  off-block columns are served from halo copies of the neighbours' boundary
  entries of p);
* constraint-space vectors (length m = n/10) and the banded ``(A A')^-1``
  factorization are replicated: ``w = A r`` is formed as per-rank partial
  products summed by ONE all-reduce (0.8 MB at m = 1e5), after which every rank
  solves the same banded system and applies its own rows of ``A'``.  The
  orthogonality test uses the constraint-space residual ``||w - (AA')v||``
  (DESIGN.md section 4), which is replicated as well and needs no collective.

Collectives per CG iteration: THREE all-reduces and nothing else --
``p'Hp`` (2 doubles); the partial ``A r`` vector; one packed buffer with
``||x+ap||^2, #violations, ||g||^2`` and every rank's boundary entries of g.
The halo copies of p are then advanced locally with the same ``beta p - g``
the owner applies (bit-identical), so there is no neighbour exchange.  The
reduced scalars are bit-identical on every rank, so the device-side branches of
``csrc/cg.hip`` take the same way everywhere and no rank needs the host.

The orchestration below is engine-agnostic: ``HipEngine`` runs the ipx
kernels; the test-suite runs the same code over gloo with the oracle's numpy
engine (tests/test_sharded_gloo.py).
"""
import ctypes

import numpy as np
import scipy.sparse as sps
import torch
import torch.distributed as dist

ST_RTG0, ST_RTG1, ST_TOL, ST_RADIUS, ST_ALPHA, ST_STOP, ST_NITER, ST_BETA = range(8)
ST_PTHP, ST_ORTH_RHS, ST_XNORM2, ST_VIOL, ST_ORTH, ST_IT_DONE = 8, 9, 10, 11, 12, 13
STATE_SIZE = 16


class ShardExt(ctypes.Structure):
    """Mirror of ipx_shard_ext (include/ipx.h)."""
    _fields_ = [("p_ext", ctypes.c_void_p)] + \
               [(k, ctypes.c_int64) for k in ("hl", "hr", "h", "rank", "world")] + \
               [("s1", ctypes.c_void_p), ("pack", ctypes.c_void_p), ("np4", ctypes.c_int64)]


def block_range(n, world, rank):
    return (rank * n) // world, ((rank + 1) * n) // world


def half_bandwidth(M):
    coo = sps.coo_matrix(M)
    return int(np.max(np.abs(coo.row - coo.col))) if coo.nnz else 0


class ShardedProjectedCG:
    """Projected CG for ``min 1/2 x'Hx + c'x  s.t.  A x = 0, ||x|| <= radius`` with
    the variables sharded over ``dist``'s ranks.  ``H`` (scipy CSR, banded),
    ``hdiag`` (optional diagonal term) and ``A`` (scipy CSR) are given in full
    on every rank; each rank keeps its block."""

    def __init__(self, engine, A, H, hdiag=None, group=None):
        self.eng = eng = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        A = sps.csr_matrix(A)
        H = sps.csr_matrix(H)
        self.m, self.n = A.shape
        self.n0, self.n1 = block_range(self.n, self.world, self.rank)
        n0, n1 = self.n0, self.n1
        self.nloc = n1 - n0
        # halo of p needed by the rows of H owned here
        self.h = half_bandwidth(H)
        if self.world > 1 and self.h > min(block_range(self.n, self.world, r)[1]
                                           - block_range(self.n, self.world, r)[0]
                                           for r in range(self.world)):
            raise NotImplementedError("Hessian bandwidth exceeds a rank's block")
        self.hl = self.h if self.rank > 0 else 0
        self.hr = self.h if self.rank < self.world - 1 else 0
        A_cols = sps.csr_matrix(A[:, n0:n1])
        self.A_cols = eng.csr(A_cols)
        self.At_rows = eng.csr(sps.csr_matrix(A_cols.T))
        self.H_rows = eng.csr(sps.csr_matrix(H[n0:n1, n0 - self.hl:n1 + self.hr]))
        self.hdiag = eng.upload(np.asarray(hdiag)[n0:n1]) if hdiag is not None else None
        self.solver = eng.banded(A)                 # replicated (A A')^-1
        self.norm_A = eng.frobenius(A)
        # buffers
        self.x, self.r, self.Hp = (eng.zeros(self.nloc) for _ in range(3))
        self.p_ext = eng.zeros(self.hl + self.nloc + self.hr)
        self.p = eng.view(self.p_ext, self.hl, self.hl + self.nloc)
        self.w, self.v, self.t = (eng.zeros(self.m) for _ in range(3))
        self.state = eng.zeros(STATE_SIZE)
        self.s1, self.s4 = eng.zeros(2), eng.zeros(2)
        self.part1 = eng.zeros(2 * eng.ntiles(self.H_rows))
        self.part3 = eng.zeros(2 * max(eng.ntiles(self.At_rows), (self.m + 255) // 256 + 1))
        self.part4 = eng.zeros((self.m + 255) // 256 + 1)
        self.grid = eng.vec_grid(self.nloc)
        self.part2 = eng.zeros(2 * max(self.grid, eng.ntiles(self.A_cols)))
        # packed all-reduce buffer: 4 scalars + [world][2h] boundary entries
        self.pack = eng.zeros(4 + 2 * self.h * self.world)
        h, r = self.h, self.rank
        self.g_left = eng.view(self.pack, 4 + (2 * (r - 1) + 1) * h, 4 + 2 * r * h) \
            if self.hl else None                       # right boundary of rank-1
        self.g_right = eng.view(self.pack, 4 + 2 * (r + 1) * h, 4 + (2 * (r + 1) + 1) * h) \
            if self.hr else None                       # left boundary of rank+1
        self.p_left = eng.view(self.p_ext, 0, self.hl) if self.hl else None
        self.p_right = eng.view(self.p_ext, self.hl + self.nloc,
                                self.hl + self.nloc + self.hr) if self.hr else None
        self.np4 = 1

    # ---- collectives ---------------------------------------------------------
    def _allreduce(self, buf):
        if self.world == 1:
            return
        t = self.eng.tensor(buf)
        if t.is_cuda and dist.get_backend(self.group) != "nccl":
            h = t.cpu()          # gloo (test-only combination): stage through the host
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _seed_p_halo(self):
        """p halos <- neighbours' boundary entries of p (priming only; inside
        the loop the halos are advanced locally)."""
        if self.world == 1 or self.h == 0:
            return
        eng = self.eng
        eng.halo_pack(self.p, self.h, self.rank, self.world, eng.view(self.pack, 4, None))
        self._allreduce(self.pack)
        if self.hl:
            eng.axpby(1.0, self.g_left, 0.0, None, self.p_left)
        if self.hr:
            eng.axpby(1.0, self.g_right, 0.0, None, self.p_right)

    # ---- pieces ---------------------------------------------------------------
    def _project(self, y, out):
        """out = Z y = y - A'(AA')^-1 A y on the local block (one all-reduce)."""
        eng = self.eng
        eng.spmv(self.A_cols, y, self.w)
        self._allreduce(self.w)
        eng.solve(self.solver, self.w, self.v)
        eng.spmv(self.At_rows, self.v, out, alpha=-1.0, beta=1.0, yin=y)

    def _hp(self, guard=True):
        """Hp = H p on the local rows, p'Hp partials in part1."""
        self.eng.spmv(self.H_rows, self.p_ext, self.Hp, diag=self.hdiag, xrow=self.p,
                      partial=self.part1, guard=self.state if guard else None)

    def prime(self, c, tol, trust_radius, orth_tol=1e-12):
        """Initial point of qp_subproblem.py:502-512 for b = 0: x = 0,
        r = Z c, g = Z r, p = -g, Hp = H p."""
        eng = self.eng
        c_loc = eng.upload(np.asarray(c)[self.n0:self.n1])
        eng.fill(self.x, 0.0)
        self._project(c_loc, self.r)
        g = eng.zeros(self.nloc)
        self._project(self.r, g)
        eng.axpby(-1.0, g, 0.0, None, self.p)
        eng.sumsq(g, self.s4)
        self._allreduce(self.s4)
        rt_g = float(eng.download(self.s4)[0])
        if tol is None:
            tol = max(min(0.01 * np.sqrt(rt_g), 0.1 * rt_g), 1e-25)
        init = np.zeros(STATE_SIZE)
        init[ST_RTG0], init[ST_TOL], init[ST_RADIUS] = rt_g, tol, trust_radius
        init[ST_ORTH_RHS] = orth_tol * self.norm_A
        eng.assign(self.state, init)
        self._seed_p_halo()
        self._hp(guard=False)
        eng.fold2(self.part1, eng.ntiles(self.H_rows), self.s1)
        return rt_g

    def iterate(self, it_begin, it_end):
        """Enqueue iterations [it_begin, it_end); no host synchronisation.  Each
        iteration is three local segments separated by the three all-reduces
        (s1 holds this rank's folded p'Hp partials on entry)."""
        eng = self.eng
        for it in range(it_begin, it_end):
            self._allreduce(self.s1)                                   # p'Hp
            eng.segment(self, 0, it)
            self._allreduce(self.w)                                    # partial A r
            eng.segment(self, 1, it)
            self._allreduce(self.pack)     # ||x+ap||^2, #viol, ||g||^2 + boundary g of all ranks
            eng.segment(self, 2, it)

    def read_state(self):
        return self.eng.download(self.state)

    def gather_x(self):
        """Full solution vector on every rank (host).  Blocks may differ by one
        row, so every rank contributes a block padded to the longest."""
        x_loc = np.ascontiguousarray(self.eng.download(self.x))
        if self.world == 1:
            return x_loc
        sizes = [block_range(self.n, self.world, r)[1] - block_range(self.n, self.world, r)[0]
                 for r in range(self.world)]
        pad = np.zeros(max(sizes))
        pad[:len(x_loc)] = x_loc
        mine = torch.from_numpy(pad)
        on_gpu = dist.get_backend(self.group) == "nccl"
        if on_gpu:
            mine = mine.to(torch.device("cuda", torch.cuda.current_device()))
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        return np.concatenate([p.cpu().numpy()[:k] for p, k in zip(parts, sizes)])

    def solve(self, c, tol=None, trust_radius=np.inf, max_iter=None, batch=8):
        """Run to a stop condition; returns (x_full, info) like projected_cg
        (stop codes 1 iteration limit, 2 boundary, 4 tolerance)."""
        self.prime(c, tol, trust_radius)
        if max_iter is None:
            max_iter = self.n - self.m
        max_iter = min(max_iter, self.n - self.m)
        it, stop_cond, hits_boundary = 0, 1, False
        while it < max_iter:
            end = min(max_iter, it + batch)
            self.iterate(it, end)
            s = self.read_state()
            stop = int(s[ST_STOP])
            if stop == 0:
                it = end
                continue
            if stop == 4:
                stop_cond = 4
                break
            if stop in (2, 3):
                # trust-region boundary (qp_subproblem.py:583-596) or negative curvature
                # (:558-576): move to the sphere along p.  The three inner products are
                # summed over the ranks; every rank then takes the same step on its block.
                if stop == 3 and np.isinf(trust_radius):
                    raise ValueError("Negative curvature not allowed for unrestricted "
                                     "problems.")
                self._to_boundary(float(s[ST_ALPHA]), trust_radius, entire_line=(stop == 3))
                stop_cond, hits_boundary = stop, True
                break
            raise NotImplementedError(
                "sharded projected CG: stop code %d (box or refinement events) is handled by "
                "the single-GPU path only" % stop)
        s = self.read_state()
        return self.gather_x(), {'niter': int(s[ST_NITER]), 'stop_cond': stop_cond,
                                 'hits_boundary': hits_boundary}

    def _to_boundary(self, alpha, trust_radius, entire_line):
        """x <- x + theta d with d = p (negative curvature: the whole line, the positive
        root) or d = alpha p (step leaving the region: the segment), theta from
        sphere_intersections on the all-reduced d.d, x.d, x.x."""
        from .qp import _sphere_from_scalars
        eng = self.eng
        scale = 1.0 if entire_line else alpha
        dots = eng.zeros(4)
        eng.dots3(self.x, self.p, dots)               # x.x, x.p, p.p on this block
        self._allreduce(dots)
        xx, xp, pp = (float(v) for v in eng.download(dots)[:3])
        ta, tb, intersect = _sphere_from_scalars(scale * scale * pp, scale * xp, xx, trust_radius,
                                                 entire_line)
        if intersect:
            eng.axpby(1.0, self.x, tb * scale, self.p, self.x)



class SegmentsByKernel:
    """The three local segments of an iteration, kernel by kernel (what
    ipx_cg_shard_segment does in one call).  The numpy engine of the test-suite
    inherits this; HipEngine overrides it with the single C call."""

    def segment(self, cg, phase, it):
        st = cg.state
        if phase == 0:
            self.step1(st, it, cg.s1, 1, cg.x, cg.p, cg.r, cg.Hp, cg.part2, cg.grid)
            self.spmv(cg.A_cols, cg.r, cg.w, guard=st)                 # partial A r
        elif phase == 1:
            cg.np4 = self.solve_resid(cg.solver, cg.w, cg.v, cg.part4, guard=st)
            self.spmv(cg.At_rows, cg.v, cg.r, alpha=-1.0, beta=1.0, yin=cg.r,
                      partial=cg.part3, guard=st)                       # g = r - A'v
            self.shard_pack(cg.part2, cg.grid, cg.part3, self.ntiles(cg.At_rows), cg.r, cg.h,
                            cg.rank, cg.world, cg.pack)
        else:
            self.step2(st, it, 0, self.view(cg.pack, 0, 2), 1, self.view(cg.pack, 2, 4), 1,
                       cg.part4, cg.np4, cg.x, cg.p, cg.r, cg.grid)
            self.halo_apply(st, cg.g_left, cg.g_right, cg.p_left, cg.p_right)
            cg._hp()
            self.fold2(cg.part1, self.ntiles(cg.H_rows), cg.s1)


class HipEngine(SegmentsByKernel):
    """Local compute of the sharded loop on one GPU: ipx kernels."""

    def segment(self, cg, phase, it):
        args = getattr(cg, "_c_args", None)
        if args is None:
            args = cg._c_args = self._build_args(cg)
        self._hip.call("ipx_cg_shard_segment", ctypes.byref(args[0]), ctypes.byref(args[1]),
                       int(phase), int(it), self._st())

    def _build_args(self, cg):
        from .cg_fused import CgArgs
        if cg.solver.perm is not None:
            raise NotImplementedError("sharded CG needs A A' banded in its natural row order")
        a = CgArgs()
        a.n, a.m = cg.nloc, cg.m
        for pre, M in (("A", cg.A_cols), ("At", cg.At_rows), ("H", cg.H_rows)):
            pat = M.pattern
            setattr(a, pre + "_rowptr", pat.indptr.data_ptr())
            setattr(a, pre + "_colidx", pat.indices.data_ptr())
            setattr(a, pre + "_val", M.val.data_ptr() if M.val.numel() else None)
            setattr(a, pre + "_tiles", pat.tiles.data_ptr())
            setattr(a, pre + "_ntiles", pat.ntiles)
        a.H_diag = cg.hdiag.data_ptr() if cg.hdiag is not None else None
        a.banded = cg.solver.handle
        a.x, a.p, a.r, a.Hp = (t.data_ptr() for t in (cg.x, cg.p, cg.r, cg.Hp))
        a.w, a.v, a.t = cg.w.data_ptr(), cg.v.data_ptr(), cg.t.data_ptr()
        a.state = cg.state.data_ptr()
        a.part1, a.part2, a.part3, a.part4 = (t.data_ptr() for t in (cg.part1, cg.part2,
                                                                       cg.part3, cg.part4))
        a.vec_grid, a.solver_kind = cg.grid, 0
        # the single-GPU loop's fusions, rank-local (csrc/cg.hip): step1 inside the partial
        # A.r SpMV, g = r - A'v as the tail of the replicated banded solve
        import os
        from . import cg_fused
        if not os.environ.get("IPX_NO_FUSE"):
            own = cg_fused.fuse_own(cg.A_cols.pattern)
            if own is not None and own[3] <= cg.part2.numel() // 2:
                cg.r_next = self.zeros(cg.nloc)
                cg.own = own[0]
                a.r_next, a.A_own, a.A_span = cg.r_next.data_ptr(), cg.own.data_ptr(), own[1]
            geo = (ctypes.c_int32 * 2)()
            if self.lib.ipx_banded_decoupled_geometry(ctypes.c_void_p(cg.solver.handle), geo) \
                    and geo[1] <= 512 and geo[1] <= cg.part3.numel() // 2:
                vown = cg_fused.fuse_vown(cg.At_rows.pattern, geo[0], geo[1])
                if vown is not None:
                    cg.vown = vown[0]
                    a.At_vown, a.At_qv = cg.vown.data_ptr(), vown[1]
        e = ShardExt()
        e.p_ext = cg.p_ext.data_ptr()
        e.hl, e.hr, e.h, e.rank, e.world = cg.hl, cg.hr, cg.h, cg.rank, cg.world
        e.s1, e.pack, e.np4 = cg.s1.data_ptr(), cg.pack.data_ptr(), 1
        return a, e

    def __init__(self):
        from . import _hip, device
        self._hip, self.dv = _hip, device
        self.lib = _hip.load()
        self.ctx = device.ctx()

    def _st(self):
        return self.dv.stream_ptr()

    @staticmethod
    def _ptr(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None

    def zeros(self, n):
        return torch.zeros(int(n), dtype=torch.float64, device=self.ctx.device)

    def upload(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.ctx.device)

    def download(self, buf):
        return buf.cpu().numpy()

    def assign(self, buf, host):
        buf.copy_(torch.from_numpy(np.ascontiguousarray(host, dtype=np.float64)))

    def tensor(self, buf):
        return buf

    def view(self, buf, a, b):
        return buf[a:b]

    def fill(self, buf, value):
        self._hip.call("ipx_fill", buf.numel(), float(value), self._ptr(buf), self._st())

    def axpby(self, a, x, b, y, out):
        self._hip.call("ipx_axpby", x.numel(), float(a), self._ptr(x), float(b), self._ptr(y),
                       self._ptr(out), self._st())

    def csr(self, M):
        return self.dv.DeviceCSR.from_scipy(M)

    def ntiles(self, M):
        return M.pattern.ntiles

    def vec_grid(self, n):
        return int(self.lib.ipx_cg_vec_grid(max(int(n), 1)))

    def banded(self, A):
        from . import projector
        self._A_full = self.dv.DeviceCSR.from_scipy(A)
        return projector.BandedNormalSolver(self._A_full)

    def frobenius(self, A):
        """||A||_F on the device (of the matrix just handed to banded())."""
        return self._A_full.frobenius_norm()

    def solve(self, solver, w, v, guard=None):
        if solver.perm is not None:
            raise NotImplementedError("sharded CG needs A A' banded in its natural row order")
        if guard is None:
            self._hip.call("ipx_banded_solve", ctypes.c_void_p(solver.handle), self._ptr(w),
                           self._ptr(v), self._st())
        else:
            self._hip.call("ipx_banded_solve_guarded_c", ctypes.c_void_p(solver.handle),
                           self._ptr(w), self._ptr(v), self._ptr(guard[ST_STOP:]), self._st())

    def solve_resid(self, solver, w, v, partial, guard=None):
        """v = (A A')^-1 w plus partial sums of ||w - (A A')v||^2; returns their count."""
        if solver.perm is not None:
            raise NotImplementedError("sharded CG needs A A' banded in its natural row order")
        npart = ctypes.c_int32(0)
        self._hip.call("ipx_banded_solve_resid", ctypes.c_void_p(solver.handle), self._ptr(w),
                       self._ptr(v), self._ptr(partial), ctypes.byref(npart),
                       self._ptr(guard[ST_STOP:]) if guard is not None else None, self._st())
        return int(npart.value)

    def halo_pack(self, g, h, rank, world, out):
        self._hip.call("ipx_cg_halo_pack", g.numel(), int(h), int(rank), int(world), self._ptr(g),
                       self._ptr(out), self._st())

    def shard_pack(self, part2, np2, part3, np3, g, h, rank, world, out):
        self._hip.call("ipx_cg_shard_pack", self._ptr(part2), int(np2), self._ptr(part3), int(np3),
                       g.numel(), int(h), int(rank), int(world), self._ptr(g), self._ptr(out),
                       self._st())

    def halo_apply(self, state, g_left, g_right, p_left, p_right):
        hl = p_left.numel() if p_left is not None else 0
        hr = p_right.numel() if p_right is not None else 0
        if hl or hr:
            self._hip.call("ipx_cg_halo_apply", self._ptr(state), hl, hr, self._ptr(g_left),
                           self._ptr(g_right), self._ptr(p_left), self._ptr(p_right), self._st())

    def dots3(self, x, p, out):
        """out[0..3) = x.x, x.p, p.p"""
        c = self.ctx
        for k, (a, b) in enumerate(((x, x), (x, p), (p, p))):
            self._hip.call("ipx_dot", a.numel(), self._ptr(a), self._ptr(b), self._ptr(c.out),
                           self._ptr(c.ws), self._st())
            out[k:k + 1].copy_(c.out[:1])

    def spmv(self, M, x, out, alpha=1.0, diag=None, beta=0.0, yin=None, xrow=None, partial=None,
             guard=None):
        p = M.pattern
        self._hip.call("ipx_csr_spmv_ex", p.shape[0], p.shape[1], self._ptr(p.indptr),
                       self._ptr(p.indices), self._ptr(M.val), self._ptr(p.tiles), p.ntiles,
                       self._ptr(x), float(alpha), self._ptr(diag), float(beta), self._ptr(yin),
                       self._ptr(out), self._ptr(xrow), self._ptr(partial),
                       self._ptr(guard[ST_STOP:]) if guard is not None else None, self._st())

    def fold2(self, partial, count, out2):
        self._hip.call("ipx_fold2", self._ptr(partial), int(count), self._ptr(out2), None,
                       self._st())

    def sumsq(self, x, out2):
        self._hip.call("ipx_norms", x.numel(), self._ptr(x), self._ptr(out2), self._ptr(self.ctx.ws),
                       self._st())

    def step1(self, state, it, p1, np1, x, p, r, Hp, part2, grid):
        self._hip.call("ipx_cg_step1", x.numel(), self._ptr(state), int(it), self._ptr(p1),
                       int(np1), self._ptr(x), self._ptr(p), self._ptr(r), self._ptr(Hp), None,
                       None, self._ptr(part2), int(grid), self._st())

    def step2(self, state, it, mode, p2, np2, p3, np3, p4, np4, x, p, g, grid):
        self._hip.call("ipx_cg_step2", x.numel(), self._ptr(state), int(it), int(mode),
                       self._ptr(p2), int(np2), self._ptr(p3), int(np3), self._ptr(p4), int(np4),
                       self._ptr(x), self._ptr(p), self._ptr(g), int(grid), self._st())
