// Matrix-free (A A')^-1 w for sparse Jacobians whose A A' is neither banded nor small enough
// for the dense Cholesky: preconditioned conjugate gradients on A (A' v) = w, entirely
// on the device (ipsolver/projector.py IterativeNormalSolver; the reference factors any sparse
// A with SuperLU, projections.py:93-172).  One call enqueues a batch of iterations:
//
//   t  = A' p                                   (CSR SpMV)
//   Sp = A t, partials of p'Sp                  (CSR SpMV, fused epilogue)
//   k_pcg_update     alpha = rz / p'Sp;  v += alpha p;  r -= alpha Sp;  partials of ||r||^2, r'z
//   [k_pcg_block     z = M^-1 r, partials of r'z        (block-Jacobi preconditioner only)]
//   k_pcg_direction  convergence / stall tests;  beta = rz_next / rz;  p = M^-1 r + beta p
//
// Preconditioner M: the diagonal of A A' (Jacobi), or -- round 3 -- its diagonal BLOCKS of 32
// rows taken in a bandwidth-reducing order of the rows (block Jacobi: k_blockjacobi_build forms
// every 32 x 32 block of A A' by merge joins of A's rows, factors it by Cholesky in LDS and
// keeps the explicit inverse; an application is one 32 x 32 matvec per block).  Block Jacobi of
// an SPD matrix is SPD, and captures the couplings of neighbouring rows that make A A' of
// chain-like Jacobians ill conditioned.
//
// The scalars never leave the device: every kernel's prologue folds its predecessor's partial
// sums in a fixed order (all workgroups derive the same bits), workgroup 0 records the
// decision in the state block, and a non-zero `done` turns the rest of the batch into no-ops;
// the host reads the state once per batch.  State words that one kernel both reads and
// updates are double buffered by iteration parity.
#include "ipx_common.h"

enum {
  PS_RZ0 = 0, PS_RZ1 = 1,        // r'z by parity
  PS_BEST0 = 2, PS_BEST1 = 3,    // smallest ||r|| so far
  PS_STALL0 = 4, PS_STALL1 = 5,  // consecutive iterations without improvement
  PS_DONE = 6,                   // 0 running, 1 converged, 2 no progress any more (the floor of fp64,
                                 // or a stall further up), 3 not positive definite
  PS_ITERS = 7, PS_NORM_W = 8, PS_RTOL = 9, PS_NORM_R = 10,
  PS_STALL_FAR = 11,             // iterations without progress that end a solve far above the floor
  PS_SIZE = 16
};

namespace {

constexpr int PB = IPX_BLOCK;
constexpr int PU = 4;
constexpr int BJ = 32;           // block size of the block-Jacobi preconditioner

// One workgroup of BJ x BJ lanes per block: S_b = A_b A_b' (rows order[b*BJ + s], s < BJ;
// order < 0: an identity row), Cholesky, explicit inverse -> binv[b] (BJ x BJ, symmetric).
// flag != 0 afterwards: a pivot was not positive (rows of the block linearly dependent).
__global__ void __launch_bounds__(BJ *BJ)
k_blockjacobi_build(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                    const double *__restrict__ val, const int32_t *__restrict__ order,
                    double *__restrict__ binv, int *__restrict__ flag) {
  __shared__ double T[BJ][BJ + 1];
  __shared__ double Li[BJ][BJ + 1];
  const int r = threadIdx.y, c = threadIdx.x;
  const int b = blockIdx.x;
  const int ri = order[b * BJ + r], ci = order[b * BJ + c];
  double s = 0.0;
  if (c <= r) {
    if (ri >= 0 && ci >= 0) {
      int p = rowptr[ri], pe = rowptr[ri + 1];
      int u = rowptr[ci], ue = rowptr[ci + 1];
      while (p < pe && u < ue) {
        const int cp = colidx[p], cu = colidx[u];
        if (cp == cu) { s += val[p] * val[u]; ++p; ++u; }
        else if (cp < cu) ++p;
        else ++u;
      }
    } else {
      s = (r == c) ? 1.0 : 0.0;
    }
  }
  T[r][c] = s;
  __syncthreads();
  for (int j = 0; j < BJ; ++j) {                       // Cholesky, lower triangle
    if (r == j && c == j) {
      const double d = T[j][j];
      if (!(d > 0.0)) atomicOr(flag, 1);
      T[j][j] = sqrt(d > 0.0 ? d : 1.0);
    }
    __syncthreads();
    if (c == j && r > j) T[r][j] /= T[j][j];
    __syncthreads();
    if (c > j && r >= c) T[r][c] -= T[r][j] * T[c][j];
    __syncthreads();
  }
  // Li = L^-1 (lower): column c by forward substitution, one lane per column
  if (r == 0) {
    for (int i = 0; i < BJ; ++i) {
      double acc = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) acc -= T[i][k] * Li[k][c];
      Li[i][c] = (i >= c) ? acc / T[i][i] : 0.0;
    }
  }
  __syncthreads();
  double x = 0.0;                                      // (L L')^-1 = L^-T L^-1
  for (int k = max(r, c); k < BJ; ++k) x += Li[k][r] * Li[k][c];
  binv[((int64_t)b * BJ + r) * BJ + c] = x;
}

// z = M^-1 r for the block-Jacobi M: eight blocks per workgroup, lane (s, blk) forms entry s
// of its block (the inverse is symmetric: column reads are coalesced); partials of r'z.
__global__ void __launch_bounds__(PB)
k_pcg_block(int64_t m, const double *st, const double *__restrict__ r,
            const int32_t *__restrict__ order, const double *__restrict__ binv,
            double *__restrict__ z, double *__restrict__ p3, int nblk) {
  __shared__ double rb[PB];
  __shared__ double lds[PB / IPX_WAVE];
  if (st[PS_DONE] != 0.0) return;
  const int tid = threadIdx.x, sl = tid & (BJ - 1), lb = tid / BJ;
  const int b = blockIdx.x * (PB / BJ) + lb;
  const int row = b < nblk ? order[b * BJ + sl] : -1;
  const double rv = row >= 0 ? r[row] : 0.0;
  rb[tid] = rv;
  __syncthreads();
  double acc = 0.0;
  if (row >= 0) {
    const double *X = binv + (int64_t)b * BJ * BJ + sl;        // column sl = row sl
#pragma unroll 8
    for (int k = 0; k < BJ; ++k) acc += X[k * BJ] * rb[lb * BJ + k];
    z[row] = acc;
  }
  const double tot = ipx_block_reduce<IPX_SUM>(row >= 0 ? rv * acc : 0.0, lds);
  if (tid == 0) p3[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(PB)
k_pcg_update(int64_t m, double *st, int parity, const double *__restrict__ p1, int np1,
             double *v, double *r, const double *__restrict__ p, const double *__restrict__ Sp,
             const double *__restrict__ dinv, double *__restrict__ p2, int nchunks) {
  __shared__ double lds[2 * (PB / IPX_WAVE)];
  const int c = ipx_xcd_item(blockIdx.x, nchunks);
  if (c < 0) return;
  if (st[PS_DONE] != 0.0) return;
  const double rz = st[parity ? PS_RZ1 : PS_RZ0];
  const double pSp = ipx_sum_partials<IPX_SUM>(p1 + np1, np1, lds);     // second half: x.y sums
  const bool lead = c == 0 && threadIdx.x == 0;
  if (!(pSp > 0.0)) {                       // A A' not positive definite: rank-deficient A
    if (lead) st[PS_DONE] = 3.0;
    return;
  }
  const double alpha = rz / pSp;
  const int64_t len = (m + nchunks - 1) / nchunks;
  const int64_t lo = (int64_t)c * len, hi = min(m, lo + len);
  double srr = 0.0, srz = 0.0;
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)PU * PB) {
    double vv[PU], rv[PU], pv[PU], sv[PU], dv[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int64_t i = min(i0 + (int64_t)u * PB, m - 1);
      vv[u] = v[i]; rv[u] = r[i]; pv[u] = p[i]; sv[u] = Sp[i]; dv[u] = dinv[i];
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int64_t i = i0 + (int64_t)u * PB;
      if (i < hi) {
        const double rn = rv[u] - alpha * sv[u];
        v[i] = vv[u] + alpha * pv[u];
        r[i] = rn;
        srr += rn * rn;
        srz += rn * (dv[u] * rn);            // (block Jacobi: dinv = 0, r'z comes from k_pcg_block)
      }
    }
  }
  double red[2] = {srr, srz}, out[2];
  ipx_block_sum_multi<2>(red, lds, out);
  if (threadIdx.x == 0) { p2[c] = out[0]; p2[nchunks + c] = out[1]; }
}

__global__ void __launch_bounds__(PB)
k_pcg_direction(int64_t m, double *st, int parity, const double *__restrict__ p2, int np2,
                const double *__restrict__ r, double *p, const double *__restrict__ dinv,
                int nchunks, const double *__restrict__ z, const double *__restrict__ p3,
                int np3) {
  __shared__ double lds[2 * (PB / IPX_WAVE)];
  const int c = ipx_xcd_item(blockIdx.x, nchunks);
  if (c < 0) return;
  if (st[PS_DONE] != 0.0) return;
  const double rz = st[parity ? PS_RZ1 : PS_RZ0];
  const double best = st[parity ? PS_BEST1 : PS_BEST0];
  const double stall = st[parity ? PS_STALL1 : PS_STALL0];
  const double norm_w = st[PS_NORM_W], rtol = st[PS_RTOL];
  const double *const parts[2] = {p2, p2 + np2};
  const int counts[2] = {np2, np2};
  double red[2];
  ipx_sum_partials_multi<2>(parts, counts, lds, red);
  if (z) red[1] = ipx_sum_partials<IPX_SUM>(p3, np3, lds);      // block Jacobi: r'z of k_pcg_block
  const double nr = sqrt(red[0]), rz_next = red[1];
  const bool lead = c == 0 && threadIdx.x == 0;
  const double stall_next = nr >= best ? stall + 1.0 : 0.0;
  int done = 0;
  if (nr <= rtol * norm_w) done = 1;
  // no new smallest residual for 5 iterations once the residual is down at 1e-9 ||w||: the floor
  // of fp64.  Further up CG's residuals plateau and oscillate on ill-conditioned systems
  // (a stop after 5 there returned a solve 2e-3 off: tests/fuzz_projections.py): the caller's
  // count, PS_STALL_FAR
  else if (stall_next >= (best <= 1e-9 * norm_w ? 5.0 : fmax(st[PS_STALL_FAR], 5.0))) done = 2;
  if (lead) {
    st[PS_ITERS] += 1.0;
    st[PS_NORM_R] = nr;
    st[parity ? PS_RZ0 : PS_RZ1] = rz_next;
    st[parity ? PS_BEST0 : PS_BEST1] = fmin(best, nr);
    st[parity ? PS_STALL0 : PS_STALL1] = stall_next;
    if (done) st[PS_DONE] = (double)done;
  }
  if (done) return;
  const double beta = rz_next / rz;
  const int64_t len = (m + nchunks - 1) / nchunks;
  const int64_t lo = (int64_t)c * len, hi = min(m, lo + len);
  if (z) {
    for (int64_t i = lo + threadIdx.x; i < hi; i += PB) p[i] = z[i] + beta * p[i];
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += PB) p[i] = dinv[i] * r[i] + beta * p[i];
  }
}

}  // namespace

extern "C" {

int ipx_pcg_state_size(void) { return PS_SIZE; }

// Enqueue iterations [it_begin, it_end) of the preconditioned CG on A A' v = w.  The caller
// has set v = 0, r = w, p = D^-1 r and the state block (PS_RZ0 = r'D^-1 r, PS_BEST0 = inf,
// PS_NORM_W, PS_RTOL).  part1 needs 2 * A_ntiles doubles, part2 2 * grid (grid =
// ipx_cg_vec_grid(m)).  Never synchronises.
int ipx_pcg_iterate(const ipx_pcg_args *a, int32_t it_begin, int32_t it_end, void *stream) {
  if (!a || it_end < it_begin || !a->state || !a->part1 || !a->part2 || a->grid < 1)
    return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const double *guard = a->state + PS_DONE;
  ipx_csr_view A{(int)a->m, (int)a->n, a->A_rowptr, a->A_colidx, a->A_val, a->A_tiles,
                 (int)a->A_ntiles};
  ipx_csr_view At{(int)a->n, (int)a->m, a->At_rowptr, a->At_colidx, a->At_val, a->At_tiles,
                  (int)a->At_ntiles};
  const int grid = (int)a->grid;
  for (int it = it_begin; it < it_end; ++it) {
    int rc = ipx_spmv_launch(At, a->p, 1.0, nullptr, 0.0, nullptr, a->t, nullptr, guard, st);
    if (rc) return rc;
    rc = ipx_spmv_launch(A, a->t, 1.0, nullptr, 0.0, nullptr, a->Sp, a->part1, guard, st, a->p);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pcg_update, dim3(ipx_xcd_grid(grid)), dim3(PB), 0, st, a->m, a->state,
                       it & 1, a->part1, (int)a->A_ntiles, a->v, a->r, a->p, a->Sp, a->dinv,
                       a->part2, grid);
    IPX_CHECK_LAUNCH();
    const bool block = a->binv != nullptr;
    const int nwg3 = block ? (int)((a->nblk + PB / BJ - 1) / (PB / BJ)) : 0;
    if (block) {
      hipLaunchKernelGGL(k_pcg_block, dim3(nwg3), dim3(PB), 0, st, a->m, a->state, a->r, a->border,
                         a->binv, a->z, a->part3, (int)a->nblk);
      IPX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_pcg_direction, dim3(ipx_xcd_grid(grid)), dim3(PB), 0, st, a->m, a->state,
                       it & 1, a->part2, grid, a->r, a->p, a->dinv, grid,
                       block ? a->z : (const double *)nullptr, a->part3, nwg3);
    IPX_CHECK_LAUNCH();
  }
  return IPX_OK;
}

// Block-Jacobi preconditioner of A A': nblk blocks of 32 rows, block b = rows
// order[32 b .. 32 b + 31] (-1: padding); binv receives nblk x 32 x 32 doubles, *flag (device
// int, zeroed by the caller) is set when a block is not positive definite.
int ipx_blockjacobi_build(int64_t nblk, const int32_t *rowptr, const int32_t *colidx,
                          const double *val, const int32_t *order, double *binv, int *flag,
                          void *stream) {
  if (nblk < 1 || !rowptr || !order || !binv || !flag) return IPX_EINVAL;
  hipLaunchKernelGGL(k_blockjacobi_build, dim3((unsigned)nblk), dim3(BJ, BJ), 0,
                     (hipStream_t)stream, rowptr, colidx, val, order, binv, flag);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// z = M^-1 r on its own (the first direction of a solve), *rz <- r'z (ws: >= nblk/8 + 1 doubles)
int ipx_blockjacobi_apply(int64_t m, int64_t nblk, const int32_t *order, const double *binv,
                          const double *r, double *z, double *ws, const double *state,
                          void *stream) {
  if (nblk < 1 || !order || !binv || !r || !z || !ws || !state) return IPX_EINVAL;
  const int nwg = (int)((nblk + PB / BJ - 1) / (PB / BJ));
  hipLaunchKernelGGL(k_pcg_block, dim3(nwg), dim3(PB), 0, (hipStream_t)stream, m, state, r, order,
                     binv, z, ws, (int)nblk);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

}  // extern "C"
