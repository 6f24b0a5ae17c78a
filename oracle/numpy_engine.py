"""Oracle (test infrastructure): numpy twin of ``ipsolver.sharded.HipEngine``.

Lets tests/test_sharded_gloo.py run the product's sharded projected-CG
orchestration (partitioning, halo exchange, all-reduces, device-style state
machine) over gloo on CPUs.  Each method restates the kernel of the same name
in csrc/cg.hip / csrc/spmv.hip with numpy.
"""
import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla
import torch

ST_RTG0, ST_RTG1, ST_TOL, ST_RADIUS, ST_ALPHA, ST_STOP, ST_NITER, ST_BETA = range(8)
ST_PTHP, ST_ORTH_RHS, ST_XNORM2, ST_VIOL, ST_ORTH, ST_IT_DONE = 8, 9, 10, 11, 12, 13


def _segments_base():
    from ipsolver.sharded import SegmentsByKernel
    return SegmentsByKernel


class NumpyEngine(_segments_base()):
    def zeros(self, n):
        return np.zeros(int(n))

    def upload(self, a):
        return np.array(a, dtype=float)

    def download(self, buf):
        return np.array(buf)

    def assign(self, buf, host):
        buf[:] = host

    def tensor(self, buf):
        return torch.from_numpy(buf)

    def view(self, buf, a, b):
        return buf[a:b]

    def fill(self, buf, value):
        buf[:] = value

    def axpby(self, a, x, b, y, out):
        out[:] = a * x + (b * y if y is not None else 0.0)

    def csr(self, M):
        return sps.csr_matrix(M)

    def ntiles(self, M):
        return 1

    def vec_grid(self, n):
        return 1

    def frobenius(self, A):
        return float(np.sqrt((sps.csr_matrix(A).data ** 2).sum()))

    def banded(self, A):
        A = sps.csr_matrix(A)
        S = sps.csc_matrix(A.dot(A.T))
        lu = spla.splu(S)
        lu_S = type("Factor", (), {})()
        lu_S.solve, lu_S.S = lu.solve, sps.csr_matrix(S)
        return lu_S

    @staticmethod
    def _stopped(guard):
        return guard is not None and guard[ST_STOP] != 0

    def solve(self, solver, w, v, guard=None):
        if not self._stopped(guard):
            v[:] = solver.solve(w)

    def solve_resid(self, solver, w, v, partial, guard=None):      # k_correct_oop<RESID>
        if not self._stopped(guard):
            v[:] = solver.solve(w)
            res = w - solver.S.dot(v)
            partial[0] = res.dot(res)
        return 1

    def halo_pack(self, g, h, rank, world, out):                    # k_cg_halo_pack
        out[:2 * h * world] = 0.0
        out[2 * h * rank:2 * h * rank + h] = g[:h]
        out[2 * h * rank + h:2 * h * (rank + 1)] = g[len(g) - h:]

    def shard_pack(self, part2, np2, part3, np3, g, h, rank, world, out):   # k_cg_shard_pack
        out[0], out[1] = part2[:np2].sum(), part2[np2:2 * np2].sum()
        out[2], out[3] = part3[:np3].sum(), part3[np3:2 * np3].sum()
        self.halo_pack(g, h, rank, world, out[4:])

    def halo_apply(self, st, g_left, g_right, p_left, p_right):     # k_cg_halo_apply
        if st[ST_STOP] != 0:
            return
        if p_left is not None:
            p_left[:] = st[ST_BETA] * p_left - g_left
        if p_right is not None:
            p_right[:] = st[ST_BETA] * p_right - g_right

    def dots3(self, x, p, out):
        out[0], out[1], out[2] = x.dot(x), x.dot(p), p.dot(p)

    def spmv(self, M, x, out, alpha=1.0, diag=None, beta=0.0, yin=None, xrow=None, partial=None,
             guard=None):
        if self._stopped(guard):
            return
        y = alpha * M.dot(x)
        if xrow is None and M.shape[0] == M.shape[1]:
            xrow = x
        if diag is not None:
            y = y + diag * xrow
        if yin is not None:
            y = y + beta * yin
        out[:] = y
        if partial is not None:
            partial[0] = y.dot(y)
            partial[1] = xrow.dot(y) if xrow is not None else 0.0

    def fold2(self, partial, count, out2):
        out2[0] = partial[:count].sum()
        out2[1] = partial[count:2 * count].sum()

    def sumsq(self, x, out2):
        out2[0] = x.dot(x)
        out2[1] = np.abs(x).max() if len(x) else 0.0

    def step1(self, st, it, p1, np1, x, p, r, Hp, part2, grid):       # k_cg_step1
        if st[ST_STOP] != 0:
            return
        ptHp = p1[np1:2 * np1].sum()
        rtg = st[ST_RTG1 if it & 1 else ST_RTG0]
        if rtg < st[ST_TOL]:
            st[ST_STOP] = 4
            return
        if ptHp <= 0:
            st[ST_NITER] += 1
            st[ST_PTHP] = ptHp
            st[ST_STOP] = 3
            return
        alpha = rtg / ptHp
        st[ST_NITER] += 1
        st[ST_PTHP], st[ST_ALPHA] = ptHp, alpha
        xn = x + alpha * p
        part2[0], part2[1] = xn.dot(xn), 0.0
        r += alpha * Hp

    def step2(self, st, it, mode, p2, np2, p3, np3, p4, np4, x, p, g, grid):   # k_cg_step2
        if st[ST_STOP] != 0:
            return
        if not mode & 1:
            xn2, viol = p2[:np2].sum(), p2[np2:2 * np2].sum()
            if np.sqrt(xn2) >= st[ST_RADIUS]:
                st[ST_XNORM2], st[ST_STOP] = xn2, 2
                return
            if viol > 0:
                st[ST_VIOL], st[ST_STOP] = viol, 5
                return
        gg = p3[:np3].sum()
        if not mode & 2:
            tt = p4[:np4].sum()
            rhs = st[ST_ORTH_RHS]
            if rhs > 0 and gg > 0 and np.sqrt(tt) > rhs * np.sqrt(gg):
                st[ST_ORTH], st[ST_STOP] = np.sqrt(tt) / np.sqrt(gg), 6
                return
        par = it & 1
        rtg = st[ST_RTG1 if par else ST_RTG0]
        beta = gg / rtg
        alpha = st[ST_ALPHA]
        st[ST_RTG0 if par else ST_RTG1] = gg
        st[ST_BETA] = beta
        st[ST_IT_DONE] += 1
        x += alpha * p
        p[:] = beta * p - g
