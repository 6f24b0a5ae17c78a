// fp64 CSR SpMV for gfx950 with LDS-staged row tiles ("CSR-stream").
//
// A workgroup owns a tile of consecutive rows whose nonzeros form ONE
// contiguous range of val/colidx.  Phase 1 streams that range from HBM with
// fully coalesced loads (lane i -> element i), multiplies by the gathered
// x[col] (served by L2 / Infinity Cache: the x vector of a banded Jacobian is
// reused by neighbouring rows) and parks the products in LDS.  Phase 2 gives
// each row to one lane, which adds its products left to right -- the same
// order as scipy's csr_matvec, so row sums are bit-identical to the
// reference's `A.dot(x)` (compiled with -ffp-contract=off).
//
// Fused epilogue (what the projected-CG loop needs, qp_subproblem.py:556,624;
// projections.py:52,67): y = alpha*Ax [+ diag*x] [+ beta*yin], and per-tile
// partials of sum(y^2) and sum(x_row*y) written in tile order.
//
// Algorithmic HBM bytes per launch: 12*nnz + 4*(rows+1) + 8*rows + 8*cols
// (+8*rows for each of diag / yin).
#include "ipx_common.h"

namespace {

IPX_STAMP_DECL(ipx_dbg_spmv);
#define SPMV_STAMP(k) IPX_STAMP_TO(ipx_dbg_spmv, k)

constexpr int TILE_NNZ = IPX_SPMV_TILE_NNZ;
constexpr int TILE_ROWS = IPX_SPMV_TILE_ROWS;   // max rows per tile (ipx_csr_tiles_host max_rows)

template <bool HAS_DIAG, bool HAS_YIN, bool REDUCE>
__global__ void __launch_bounds__(IPX_BLOCK)
k_csr_spmv(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
           const double *__restrict__ val, const int32_t *__restrict__ tiles,
           const double *__restrict__ x, double alpha, const double *__restrict__ diag,
           double beta, const double *yin, double *yout, const double *__restrict__ xrow,
           double *__restrict__ partial, int ntiles, const double *__restrict__ guard) {
  __shared__ double prod[TILE_NNZ];
  __shared__ int rp[TILE_ROWS + 1];
  __shared__ double red_lds[IPX_BLOCK / IPX_WAVE];
  SPMV_STAMP(0);
  const int tile = ipx_xcd_item(blockIdx.x, ntiles);
  if (tile < 0) return;
  // One round trip for everything the tile's addresses depend on: the stop
  // flag, the row range and the nonzero range (second half of the tile table)
  // are requested together; the stream loads below depend on nothing else.
  const double stop = guard ? *guard : 0.0;
  const int r0 = tiles[tile], r1 = tiles[tile + 1];
  const int s = tiles[ntiles + 1 + tile], e = tiles[ntiles + 2 + tile];
  if (stop != 0.0) return;              // device-side stop flag of the fused CG loop
  SPMV_STAMP(1);
  double acc_yy = 0.0, acc_xy = 0.0;

  const int nrows = r1 - r0;
  if (e - s <= TILE_NNZ && nrows <= TILE_ROWS) {
    // Phase 1: coalesced stream of the tile's nonzeros.  All loads of a lane
    // are issued before the first use (a rolled loop would pay one full
    // memory latency per trip: the kernel is latency-, not bandwidth-limited
    // otherwise), then the gathers, then the LDS stores.
    constexpr int U = TILE_NNZ / IPX_BLOCK;
    constexpr int Q = TILE_ROWS / IPX_BLOCK;
    const int tid = threadIdx.x;
    int rpv[Q + 1];
#pragma unroll
    for (int q = 0; q <= Q; ++q) {
      const int i = tid + q * IPX_BLOCK;
      rpv[q] = rowptr[r0 + min(i, nrows)] - s;     // unconditional: no branch, no wait per load
    }
    // epilogue operands of this lane's rows: requested with the first batch
    // (before the dependent gathers), consumed after the barrier
    double dg[Q], xr[Q], yi[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int r = r0 + min(tid + q * IPX_BLOCK, nrows - 1);
      dg[q] = HAS_DIAG ? diag[r] : 0.0;
      xr[q] = xrow ? xrow[r] : 0.0;
      yi[q] = HAS_YIN ? yin[r] : 0.0;
    }
    if (e > s) {
      int c[U];
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = min(s + tid + u * IPX_BLOCK, e - 1);
        c[u] = colidx[jj];
        v[u] = val[jj];
      }
      SPMV_STAMP(2);
      double xg[U];
#pragma unroll
#ifdef IPX_EXP_NOGATHER              // experiment: what do the gathers cost?
      for (int u = 0; u < U; ++u) xg[u] = (double)c[u];
#else
      for (int u = 0; u < U; ++u) xg[u] = x[c[u]];
#endif
      SPMV_STAMP(3);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = s + tid + u * IPX_BLOCK;
        if (jj < e) prod[jj - s] = v[u] * xg[u];
      }
    }
#pragma unroll
    for (int q = 0; q <= Q; ++q) {
      const int i = tid + q * IPX_BLOCK;
      if (i <= nrows) rp[i] = rpv[q];
    }
    SPMV_STAMP(4);
    __syncthreads();
    SPMV_STAMP(5);
    // Phase 2: one lane per row, left-to-right row sums out of LDS.  The y
    // values stay in registers until the block reduction is done: the stores
    // are the last thing the workgroup issues (nothing waits on them).
    double yq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int i = tid + q * IPX_BLOCK;
      yq[q] = 0.0;
      if (i < nrows) {
        const int a = rp[i], b = rp[i + 1];
        double sum = 0.0;
#ifdef IPX_EXP_NOROWS                // experiment: what does the LDS row-sum phase cost?
        sum = prod[a] + (double)b;
#else
        for (int k = a; k < b; ++k) sum += prod[k];
#endif
        double y = alpha * sum;
        if (HAS_DIAG) y += dg[q] * xr[q];
        if (HAS_YIN) y += beta * yi[q];
        yq[q] = y;
        if (REDUCE) {
          acc_yy += y * y;
          if (xrow) acc_xy += xr[q] * y;
        }
      }
    }
    if (REDUCE) {
      const double a = ipx_block_reduce<IPX_SUM>(acc_yy, red_lds);
      const double b = ipx_block_reduce<IPX_SUM>(acc_xy, red_lds);
      if (threadIdx.x == 0) { partial[tile] = a; partial[ntiles + tile] = b; }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int i = tid + q * IPX_BLOCK;
      if (i < nrows) yout[r0 + i] = yq[q];
    }
    SPMV_STAMP(7);
    return;
  } else if (e - s <= TILE_NNZ) {
    // (tiles with more rows than TILE_ROWS: generic loops)
    for (int j = s + (int)threadIdx.x; j < e; j += IPX_BLOCK)
      prod[j - s] = val[j] * x[colidx[j]];
    __syncthreads();
    for (int r = r0 + (int)threadIdx.x; r < r1; r += IPX_BLOCK) {
      const int a = rowptr[r] - s, b = rowptr[r + 1] - s;
      double sum = 0.0;
      for (int k = a; k < b; ++k) sum += prod[k];
      double y = alpha * sum;
      if (HAS_DIAG) y += diag[r] * xrow[r];
      if (HAS_YIN) y += beta * yin[r];
      yout[r] = y;
      if (REDUCE) {
        acc_yy += y * y;
        if (xrow) acc_xy += xrow[r] * y;
      }
    }
  } else {
    // A tile is over-long only when it is a single very long row: the whole
    // workgroup strides it and the lane sums are folded in a fixed order.
    for (int r = r0; r < r1; ++r) {
      const int a = rowptr[r], b = rowptr[r + 1];
      double part = 0.0;
      for (int j = a + (int)threadIdx.x; j < b; j += IPX_BLOCK) part += val[j] * x[colidx[j]];
      double sum = ipx_block_reduce<IPX_SUM>(part, red_lds);
      if (threadIdx.x == 0) {
        double y = alpha * sum;
        if (HAS_DIAG) y += diag[r] * xrow[r];
        if (HAS_YIN) y += beta * yin[r];
        yout[r] = y;
        if (REDUCE) {
          acc_yy += y * y;
          if (xrow) acc_xy += xrow[r] * y;
        }
      }
    }
  }
  SPMV_STAMP(6);
  if (REDUCE) {
    double a = ipx_block_reduce<IPX_SUM>(acc_yy, red_lds);
    double b = ipx_block_reduce<IPX_SUM>(acc_xy, red_lds);
    if (threadIdx.x == 0) { partial[tile] = a; partial[ntiles + tile] = b; }
  }
  SPMV_STAMP(7);
}

// Fold per-tile partials (any count) into red[0..1] in tile order.
__global__ void __launch_bounds__(IPX_BLOCK)
k_spmv_fold(const double *partial, int ntiles, double *red) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  double a = ipx_sum_partials<IPX_SUM>(partial, ntiles, lds);
  double b = ipx_sum_partials<IPX_SUM>(partial + ntiles, ntiles, lds);
  if (threadIdx.x == 0) { red[0] = a; red[1] = b; }
}

template <bool D, bool Y, bool R>
void launch(int ntiles, hipStream_t st, const int32_t *rowptr, const int32_t *colidx,
            const double *val, const int32_t *tiles, const double *x, double alpha,
            const double *diag, double beta, const double *yin, double *yout,
            const double *xrow, double *partial, const double *guard) {
  hipLaunchKernelGGL((k_csr_spmv<D, Y, R>), dim3(ipx_xcd_grid(ntiles)), dim3(IPX_BLOCK), 0, st, rowptr, colidx,
                     val, tiles, x, alpha, diag, beta, yin, yout, xrow, partial, ntiles, guard);
}

}  // namespace

// Internal launcher shared with cg.hip: per-tile partials go to `partial`
// (2*ntiles doubles: sum y^2 then sum x*y) and are NOT folded; `guard` is an
// optional device stop flag.
// `xrow` = the vector whose row entries pair with the output rows (diag*xrow,
// sum xrow*y): x itself for a square matrix, NULL for a rectangular one, or an
// explicit pointer when x carries halo entries (row-sharded H, sharded.py).
int ipx_spmv_launch(const ipx_csr_view &A, const double *x, double alpha, const double *diag,
                    double beta, const double *yin, double *yout, double *partial,
                    const double *guard, hipStream_t st, const double *xrow_override) {
  if (A.nrows == 0 || A.ntiles == 0) return IPX_OK;
  if (beta == 0.0) yin = nullptr;
  const bool D = diag != nullptr, Y = yin != nullptr, R = partial != nullptr;
  const double *xrow = xrow_override ? xrow_override : (A.nrows == A.ncols ? x : nullptr);
  if (D && !xrow) return IPX_EINVAL;
#define GO(d, y, r)                                                                       \
  launch<d, y, r>(A.ntiles, st, A.rowptr, A.colidx, A.val, A.tiles, x, alpha, diag, beta, \
                  yin, yout, xrow, partial, guard)
  if (D) { if (Y) { if (R) GO(true, true, true); else GO(true, true, false); }
           else   { if (R) GO(true, false, true); else GO(true, false, false); } }
  else   { if (Y) { if (R) GO(false, true, true); else GO(false, true, false); }
           else   { if (R) GO(false, false, true); else GO(false, false, false); } }
#undef GO
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

IPX_STAMP_EXPORT(ipx_debug_stamps_spmv, ipx_dbg_spmv)

extern "C" {

// Host-side symbolic step: cut rows into tiles of <= tile_nnz nonzeros and
// <= max_rows rows (a longer single row gets a tile of its own).  The table
// has 2*(nt+1) entries: the nt+1 row boundaries, then rowptr at those rows (so
// a workgroup learns its row range and its nonzero range in one round trip).
int ipx_csr_tiles_host(int64_t nrows, const int32_t *rowptr, int32_t tile_nnz,
                       int32_t max_rows, int32_t *tiles_out, int64_t cap) {
  if (nrows < 0 || !rowptr || !tiles_out || tile_nnz < 1 || max_rows < 1) return IPX_EINVAL;
  int64_t nt = 0;
  int64_t r = 0;
  if (cap < 2) return IPX_EINVAL;
  tiles_out[0] = 0;
  while (r < nrows) {
    int64_t r_end = r + 1;   // always take at least one row
    while (r_end < nrows && r_end - r < max_rows &&
           rowptr[r_end + 1] - rowptr[r] <= tile_nnz)
      ++r_end;
    ++nt;
    if (2 * (nt + 1) > cap) return IPX_EINVAL;
    tiles_out[nt] = (int32_t)r_end;
    r = r_end;
  }
  for (int64_t t = 0; t <= nt; ++t) tiles_out[nt + 1 + t] = rowptr[tiles_out[t]];
  return (int)nt;
}

int ipx_csr_spmv(int64_t nrows, int64_t ncols, const int32_t *rowptr, const int32_t *colidx,
                 const double *val, const int32_t *tiles, int32_t ntiles, const double *x,
                 double alpha, const double *diag, double beta, const double *yin, double *yout,
                 int square, double *red, double *ws, void *stream) {
  if (nrows < 0 || ncols < 0 || ntiles < 0) return IPX_EINVAL;
  if (nrows > 0 && (!rowptr || !tiles || !yout || (!x && ncols > 0))) return IPX_EINVAL;
  if (nrows == 0 || ntiles == 0) {
    if (red) (void)hipMemsetAsync(red, 0, 2 * sizeof(double), (hipStream_t)stream);
    return IPX_OK;
  }
  if (red && (!ws || 2 * (int64_t)ntiles > IPX_WS_DOUBLES)) return IPX_EINVAL;
  if (beta == 0.0) yin = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const bool D = diag != nullptr, Y = yin != nullptr, R = red != nullptr;
  const double *xrow = square ? x : nullptr;
  if (D && !xrow) return IPX_EINVAL;
#define GO(d, y, r)                                                                          \
  launch<d, y, r>(ntiles, st, rowptr, colidx, val, tiles, x, alpha, diag, beta, yin, yout, \
                  xrow, ws, nullptr)
  if (D) { if (Y) { if (R) GO(true, true, true); else GO(true, true, false); }
           else   { if (R) GO(true, false, true); else GO(true, false, false); } }
  else   { if (Y) { if (R) GO(false, true, true); else GO(false, true, false); }
           else   { if (R) GO(false, false, true); else GO(false, false, false); } }
#undef GO
  IPX_CHECK_LAUNCH();
  if (R) {
    hipLaunchKernelGGL(k_spmv_fold, dim3(1), dim3(IPX_BLOCK), 0, st, ws, ntiles, red);
    IPX_CHECK_LAUNCH();
  }
  return IPX_OK;
}

// Extended form used by the row-sharded CG (ipsolver/sharded.py): explicit row
// vector `xrow` (may be NULL), device stop flag `guard` (may be NULL), per-tile
// partials written to `partial` (2*ntiles doubles, NOT folded; may be NULL).
int ipx_csr_spmv_ex(int64_t nrows, int64_t ncols, const int32_t *rowptr, const int32_t *colidx,
                    const double *val, const int32_t *tiles, int32_t ntiles, const double *x,
                    double alpha, const double *diag, double beta, const double *yin,
                    double *yout, const double *xrow, double *partial, const double *guard,
                    void *stream) {
  if (nrows < 0 || ncols < 0 || ntiles < 0) return IPX_EINVAL;
  if (nrows > 0 && (!rowptr || !tiles || !yout || (!x && ncols > 0))) return IPX_EINVAL;
  ipx_csr_view A{(int)nrows, (int)ncols, rowptr, colidx, val, tiles, ntiles};
  // rows pair with `xrow`; a square matrix without one pairs them with x itself
  return ipx_spmv_launch(A, x, alpha, diag, beta, yin, yout, partial, guard,
                         (hipStream_t)stream, xrow);
}

// red[0] = sum partial[0..count), red[1] = sum partial[count..2count), fixed order.
int ipx_fold2(const double *partial, int32_t count, double *red, const double *guard,
              void *stream) {
  if (!partial || !red || count < 0) return IPX_EINVAL;
  hipLaunchKernelGGL(k_spmv_fold, dim3(1), dim3(IPX_BLOCK), 0, (hipStream_t)stream, partial, count,
                     red);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

}  // extern "C"
