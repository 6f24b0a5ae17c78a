// Dense Jacobian path (BASELINE config 2; reference projections.py:175-233
// uses LAPACK pivoted QR).  On MI355X:
//   * y = A x             : one wave per row, coalesced along the row (HBM-bound)
//   * G = A A'            : fp64 MFMA (v_mfma_f64_16x16x4_f64), the only
//                           matmul-shaped op of the path
//   * G = L L', G^-1      : blocked right-looking Cholesky and a blocked
//                           two-sweep inverse, 32x32 tiles through LDS
// G^-1 is then applied as one dense matvec per projection, so the per-CG-
// iteration cost of the (AA')^-1 step is a 8*m^2-byte stream instead of two
// latency-bound triangular sweeps.
#include "ipx_common.h"

namespace {


// ---------------------------------------------------------------- gemv
// y_r = alpha * sum_c A[r][c] x[c]  [+ diag_r x_r] [+ beta yin_r]; one wave per row.
// Partial sums per lane over a strided column set, then a wave butterfly:
// fixed order, independent of scheduling.
template <bool REDUCE>
__global__ void __launch_bounds__(IPX_BLOCK)
k_dense_gemv(int m, int n, const double *__restrict__ A, int64_t lda,
             const double *__restrict__ x, double alpha, const double *__restrict__ diag,
             double beta, const double *yin, double *yout, int square,
             double *__restrict__ partial, const double *__restrict__ guard) {
  __shared__ double red_lds[IPX_BLOCK / IPX_WAVE];
  if (guard && *guard != 0.0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wpb = IPX_BLOCK / IPX_WAVE;
  double acc_yy = 0.0, acc_xy = 0.0;
  for (int r = blockIdx.x * wpb + wave; r < m; r += gridDim.x * wpb) {
    const double *row = A + (int64_t)r * lda;
    double s = 0.0;
    {
      // lane l sums elements l, l+64, l+128, ... in that order (what the results of the
      // dense path are pinned to); eight loads of each operand are issued before the first
      // product so that a wave keeps 8 KB in flight instead of 1
      int c = lane;
      for (; c + 7 * IPX_WAVE < n; c += 8 * IPX_WAVE) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = row[c + u * IPX_WAVE]; b[u] = x[c + u * IPX_WAVE]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u] * b[u];
      }
      for (; c < n; c += IPX_WAVE) s += row[c] * x[c];
    }
    s = ipx_wave_sum(s);
    if (lane == 0) {
      double y = alpha * s;
      if (diag) y += diag[r] * x[r];
      if (yin) y += beta * yin[r];
      yout[r] = y;
      if (REDUCE) {
        acc_yy += y * y;
        if (square) acc_xy += x[r] * y;
      }
    }
  }
  if (REDUCE) {
    double a = ipx_block_reduce<IPX_SUM>(acc_yy, red_lds);
    double b = ipx_block_reduce<IPX_SUM>(acc_xy, red_lds);
    if (threadIdx.x == 0) { partial[blockIdx.x] = a; partial[gridDim.x + blockIdx.x] = b; }
  }
}

__global__ void __launch_bounds__(IPX_BLOCK)
k_fold2(const double *partial, int count, double *red) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  double a = ipx_sum_partials<IPX_SUM>(partial, count, lds);
  double b = ipx_sum_partials<IPX_SUM>(partial + count, count, lds);
  if (threadIdx.x == 0) { red[0] = a; red[1] = b; }
}

// ---------------------------------------------------------------- Gram
// G[i][j] = sum_k A[i][k] A[j][k] for j <= i, mirrored; G is M x M (M = m rounded up to 32)
// with an identity tail so it stays SPD.  The one matmul-shaped operation of the path, on
// the fp64 matrix cores (v_mfma_f64_16x16x4_f64: lane l supplies A-operand element
// (row l&15, k l>>4) and B-operand element (k l>>4, col l&15); D layout col = l&15,
// row = (l>>4) + 4*reg; cdna_hip_programming.md section 3).
//
// A workgroup (4 waves) owns a 64 x 64 tile of the lower triangle of G; wave w the 32 x 32
// quadrant (w>>1, w&1) = 2 x 2 MFMA tiles, 4 accumulators.  K is walked in chunks of 32
// columns: the two 64-row panels of A (one when the tile sits on the diagonal) are staged
// in LDS by 16-byte loads -- 16 lanes cover the 256 contiguous bytes of a row's chunk -- and
// the loads of chunk c+1 are issued into registers before the MFMAs of chunk c, so global
// latency hides behind 32 MFMAs per wave.
//
// LDS layout (round 3).  Round 2 kept a panel as [row][k] with a pitch of 34 doubles and read
// one double per lane and MFMA; the compiler paired the reads of consecutive k-steps into
// ds_read2_b64, whose banking is per 16 CONTIGUOUS lanes modulo 32 dwords: rows r and r + 8
// collided, 4 conflict cycles per MFMA (SQ_LDS_BANK_CONFLICT 84.6 M for 21.15 M MFMAs,
// profiles/r02_dense_gram_pmc.json).  Now a lane fetches the operands of TWO MFMAs with one
// ds_read_b128: the sum over k may visit k in any order as long as both operands use the same
// one, so lane group lk (= lane >> 4) is given the k-values 8 j + 2 lk + {0, 1} of step j --
// adjacent in memory.  The panel is four sub-panels, one per lk, of 64 rows pitched 10
// doubles: within the lane groups of ds_read_b128 ({0-3,12-15,20-27}, ... :
// MI355X_MICROARCH.md, LDS) the 16-byte slot of a lane is (5 row + j) mod 16 -- 5 row is a
// bijection on 16 rows and the sub-panels are a multiple of 16 slots apart -- so every group
// covers the 16 slots once: conflict-free, half the LDS instructions.
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int GT = 64;          // tile edge of G per workgroup
constexpr int GK = 32;          // K chunk
constexpr int GSP = 10;         // sub-panel row pitch (doubles): 8 k-values of one lane group + 2
constexpr int GSUB = GT * GSP;  // sub-panel (one lane group lk): 640 doubles = 320 slots
constexpr int GPANEL = 4 * GSUB;

// Which double2 of the 64 x 32 panel thread `tid` moves in its u-th load / LDS store: row r,
// column pair t (columns 2t, 2t+1).  A wave instruction covers four whole rows' 256-byte
// chunks (coalesced), and every 8 CONTIGUOUS lanes -- the banking group of ds_write_b128,
// modulo 32 dwords -- write one lane group's (lk) pairs j = 0..3 of rows b and b + 4: dword
// offsets 20 row + 4 j = const + {0, 4, ..., 28}, every bank once.  (idx -> (idx >> 4,
// idx & 15) put the four lk of one row and j on the same banks: 4-way conflicts.)
__device__ __forceinline__ void gram_slot(int tid, int u, int &r, int &t) {
  const int lane = tid & 63, wave = tid >> 6;
  const int c = wave * 4 + u;                    // 16 (wave, u) combinations x 4 rows
  const int b = 8 * (c >> 1) + 2 * (c & 1);      // rows b, b+1, b+4, b+5
  const int g = lane >> 3, i = lane & 7;
  const int j = i & 3, rsel = i >> 2, lk = g & 3, rpair = g >> 2;
  r = b + 4 * rsel + rpair;
  t = 4 * j + lk;
}

template <bool VEC>
__device__ __forceinline__ void gram_fetch(const double *__restrict__ A, int64_t lda, int m,
                                           int n, int row0, int k0, int tid, v2d (&reg)[4]) {
  // panel of 64 rows x 32 columns = 1024 double2, four per thread (gram_slot: which)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int r, t;
    gram_slot(tid, u, r, t);
    const int c = 2 * t;
    const int row = row0 + r, k = k0 + c;
    v2d v = {0.0, 0.0};
    if (row < m) {
      const double *src = A + (int64_t)row * lda + k;
      if (VEC && k + 1 < n) v = *reinterpret_cast<const v2d *>(src);
      else {
        if (k < n) v.x = src[0];
        if (k + 1 < n) v.y = src[1];
      }
    }
    reg[u] = v;
  }
}

__device__ __forceinline__ void gram_stash(double *panel, int tid, const v2d (&reg)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int r, t;
    gram_slot(tid, u, r, t);                      // columns 2t, 2t + 1 = 8 j + 2 lk + {0, 1}
    const int j = t >> 2, lk = t & 3;
    *reinterpret_cast<v2d *>(panel + lk * GSUB + r * GSP + 2 * j) = reg[u];
  }
}

template <bool VEC>
__global__ void __launch_bounds__(IPX_BLOCK)
k_gram_mfma(int m, int n, const double *__restrict__ A, int64_t lda, double *__restrict__ G,
            int M, int ntile, int splits, double *__restrict__ ws) {
  __shared__ __attribute__((aligned(16))) double sA[GPANEL];
  __shared__ __attribute__((aligned(16))) double sB[GPANEL];
  // splits > 1: this workgroup sums over the K-chunks [kc0, kc1) only and leaves its 64 x 64
  // partial tile in ws; k_gram_reduce adds the splits in a fixed order.  (528 tiles on 256
  // CUs leave 16 CUs with one workgroup more than the rest: 1.30 ms against 1.01 ms for
  // 496 tiles; finer units balance.)
  const int ntri = ntile * (ntile + 1) / 2;
  const int tile = (int)blockIdx.x % ntri, split = (int)blockIdx.x / ntri;
  // linear tile index -> (ti, tj) with tj <= ti
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  while (ti * (ti + 1) / 2 > tile) --ti;
  const int tj = tile - ti * (ti + 1) / 2;
  if (ti >= ntile) return;
  const int nchunk = (n + GK - 1) / GK;
  const int kbeg = (int)((int64_t)split * nchunk / splits) * GK;
  const int kend = min(n, (int)((int64_t)(split + 1) * nchunk / splits) * GK);
  const bool diag = ti == tj;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int lr = lane & 15, lk = lane >> 4;
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
  v2d ra[4], rb[4];
  gram_fetch<VEC>(A, lda, m, n, ti * GT, kbeg, tid, ra);
  if (!diag) gram_fetch<VEC>(A, lda, m, n, tj * GT, kbeg, tid, rb);
  const double *pB = diag ? sA : sB;
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    gram_stash(sA, tid, ra);
    if (!diag) gram_stash(sB, tid, rb);
    __syncthreads();
    if (k0 + GK < kend) {                    // next chunk's loads fly during the MFMAs
      gram_fetch<VEC>(A, lda, m, n, ti * GT, k0 + GK, tid, ra);
      if (!diag) gram_fetch<VEC>(A, lda, m, n, tj * GT, k0 + GK, tid, rb);
    }
    const double *qa = sA + lk * GSUB + (wr + lr) * GSP;
    const double *qb = pB + lk * GSUB + (wc + lr) * GSP;
#pragma unroll
    for (int j = 0; j < GK / 8; ++j) {
      const v2d a0 = *reinterpret_cast<const v2d *>(qa + 2 * j);
      const v2d a1 = *reinterpret_cast<const v2d *>(qa + 16 * GSP + 2 * j);
      const v2d b0 = *reinterpret_cast<const v2d *>(qb + 2 * j);
      const v2d b1 = *reinterpret_cast<const v2d *>(qb + 16 * GSP + 2 * j);
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b1.x, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b0.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.x, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.y, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1.y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b0.y, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  if (splits > 1) {
    double *out = ws + ((int64_t)split * ntri + tile) * (GT * GT);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          out[(wr + 16 * a + lk + 4 * reg) * GT + wc + 16 * b + lr] = acc[a][b][reg];
    return;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = ti * GT + wr + 16 * a + lk + 4 * reg;
        const int col = tj * GT + wc + 16 * b + lr;
        if (row >= M || col >= M || col > row) continue;      // lower triangle (+ mirror)
        double v = acc[a][b][reg];
        if (row >= m || col >= m) v = (row == col) ? 1.0 : 0.0;
        G[(int64_t)row * M + col] = v;
        G[(int64_t)col * M + row] = v;
      }
    }
  }
}

// Sum of the K-splits' partial tiles (split order: deterministic), lower triangle + mirror.
__global__ void __launch_bounds__(IPX_BLOCK)
k_gram_reduce(int m, int M, int ntile, int splits, const double *__restrict__ ws,
              double *__restrict__ G) {
  const int ntri = ntile * (ntile + 1) / 2;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)ntri * GT * GT) return;
  const int tile = (int)(e / (GT * GT)), in = (int)(e % (GT * GT));
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  while (ti * (ti + 1) / 2 > tile) --ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int row = ti * GT + in / GT, col = tj * GT + in % GT;
  if (row >= M || col >= M || col > row) return;
  double v = 0.0;
  for (int s = 0; s < splits; ++s) v += ws[((int64_t)s * ntri + tile) * (GT * GT) + in];
  if (row >= m || col >= m) v = (row == col) ? 1.0 : 0.0;
  G[(int64_t)row * M + col] = v;
  G[(int64_t)col * M + row] = v;
}

// G = A A' for a CSR A whose A A' is not narrow-banded: one lane per (i, j <= i),
// merge join of the two sorted rows.  Same padded layout as k_gram_mfma.
__global__ void __launch_bounds__(IPX_BLOCK)
k_aat_dense(int m, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
            const double *__restrict__ val, double *__restrict__ G, int M) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * M) return;
  const int i = (int)(idx / M), j = (int)(idx % M);
  if (j > i) return;
  double s = 0.0;
  if (i < m && j < m) {
    int p = rowptr[i], pe = rowptr[i + 1];
    int u = rowptr[j], ue = rowptr[j + 1];
    while (p < pe && u < ue) {
      const int cp = colidx[p], cu = colidx[u];
      if (cp == cu) { s += val[p] * val[u]; ++p; ++u; }
      else if (cp < cu) ++p;
      else ++u;
    }
  } else {
    s = (i == j) ? 1.0 : 0.0;
  }
  G[(int64_t)i * M + j] = s;
  G[(int64_t)j * M + i] = s;
}

// ------------------------------------------------------- blocked Cholesky + inverse (round 5)
// G = L L' and G^-1 per accepted step of a dense NONLINEAR constraint (the reference spends
// 76 % of such a run in its pivoted QR, projections.py:179).  Rounds 1-4: 32 x 32 tiles of scalar
// FMAs, 441 launches, 2.7 + 3.6 ms at M = 2016.  Now 64 x 64 tiles with every matrix-matrix
// product on the fp64 matrix cores (the Gram kernel's tile: 4 waves x 2 x 2
// v_mfma_f64_16x16x4_f64 out of K-permuted LDS panels):
//   factor   right-looking, per tile column k: the diagonal tile in one workgroup (LDS), the
//            panel below it by forward substitution (a wave per 64 rows), the trailing
//            update G_ij -= L_ik L_jk' as MFMA tiles                         (3 launches x M/64)
//   inverse  X = L^-1 IN PLACE by recursive doubling: all diagonal tiles at once, then per level
//            s = 64, 128, ...: X21 = -X22 (L21 X11) for every pair of neighbouring s-blocks
//            -- two batched MFMA launches per level, log2(M/64) levels instead of M/32
//            dependent sweeps -- with the intermediate L21 X11 parked in the (otherwise unused)
//            upper triangle; then G^-1 = X'X as MFMA tiles over the rows where X is non-zero.
constexpr int FB = 64;           // tile edge of the factorization

// element (i, k) of a 64 x K operand panel at p[i * rs + k * ks]
struct Opnd {
  const double *p;
  int64_t rs, ks;
};

__device__ __forceinline__ void opnd_fetch(const Opnd &O, int k0, int tid, v2d (&reg)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int r, t;
    gram_slot(tid, u, r, t);
    const double *src = O.p + (int64_t)r * O.rs + (int64_t)(k0 + 2 * t) * O.ks;
    reg[u] = (v2d){src[0], src[O.ks]};
  }
}

// acc (the 64 x 64 tile C = P Q', summed over k in [0, K), K a multiple of 32) by the four
// waves of the workgroup: element (row, col) = (wr + 16 a + lk + 4 reg, wc + 16 b + lr) in
// acc[a][b][reg] (k_gram_mfma's layout).  sA / sB: GPANEL doubles each.
__device__ __forceinline__ void tile_pqt(const Opnd &P, const Opnd &Q, int K, v4d (&acc)[2][2],
                                         double *sA, double *sB) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
  if (K <= 0) return;
  v2d ra[4], rb[4];
  opnd_fetch(P, 0, tid, ra);
  opnd_fetch(Q, 0, tid, rb);
  for (int k0 = 0; k0 < K; k0 += GK) {
    gram_stash(sA, tid, ra);
    gram_stash(sB, tid, rb);
    __syncthreads();
    if (k0 + GK < K) {
      opnd_fetch(P, k0 + GK, tid, ra);
      opnd_fetch(Q, k0 + GK, tid, rb);
    }
    const double *qa = sA + lk * GSUB + (wr + lr) * GSP;
    const double *qb = sB + lk * GSUB + (wc + lr) * GSP;
#pragma unroll
    for (int j = 0; j < GK / 8; ++j) {
      const v2d a0 = *reinterpret_cast<const v2d *>(qa + 2 * j);
      const v2d a1 = *reinterpret_cast<const v2d *>(qa + 16 * GSP + 2 * j);
      const v2d b0 = *reinterpret_cast<const v2d *>(qb + 2 * j);
      const v2d b1 = *reinterpret_cast<const v2d *>(qb + 16 * GSP + 2 * j);
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b1.x, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b0.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.x, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.y, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1.y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b0.y, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
}

// for (row, col, value) of the tile held in acc
#define IPX_TILE_FOREACH(acc, row, col, val, ...)                                    \
  do {                                                                                \
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;                     \
    const int wr_ = (wave_ >> 1) * 32, wc_ = (wave_ & 1) * 32;                        \
    const int lr_ = lane_ & 15, lk_ = lane_ >> 4;                                     \
    _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                                  \
    _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                  \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                \
      const int row = wr_ + 16 * a_ + lk_ + 4 * g_, col = wc_ + 16 * b_ + lr_;        \
      const double val = acc[a_][b_][g_];                                             \
      __VA_ARGS__;                                                                    \
    }                                                                                 \
  } while (0)

// work[0..M) = the diagonal of G before the factorization; work[M] = running minimum of
// pivot / original diagonal entry (how many digits the factorization lost: ~1/cond(G)).
__global__ void __launch_bounds__(IPX_BLOCK) k_save_diag(const double *G, int M, double *work) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) work[i] = G[(int64_t)i * M + i];
  if (i == 0) work[M] = 1.0;
}

// value of lane `lane` (a constant after unrolling) of the wave, in every lane
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

constexpr int PB = 16;           // column block inside a tile

// 1 / sqrt(d) for a positive, normal d: the hardware estimate (~26 bits) and two Newton steps
// (the library routine's scaling and special cases are a third of the dependent chain a column
// of the diagonal block waits for)
__device__ __forceinline__ double potrf_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
  }
  return y;
}

// Diagonal tile kb: G_kk = L_kk L_kk' by one workgroup, the tile in LDS, in four column blocks
// of 16:
//   (a) the 16 x 16 diagonal block by ONE wave without a barrier: lane r holds row r in
//       registers, per column the pivot and the column's entries travel by v_readlane
//       (static lanes: both loops unrolled), L[r][j] = x / sqrt(d) as x * rsqrt(d);
//   (b) the rows below it: a lane per row, forward substitution out of registers, the block's
//       entries LDS broadcasts, right-looking (the dependent chain of a step is one multiply
//       and one multiply-add);
//   (c) the rank-16 update of what is left, an element per lane and pass.
// Three barriers per block, twelve per tile (the first version took one per COLUMN and formed
// every column's entries in all 256 threads: 30 us per tile, the largest item of the
// Cholesky's 2.0 ms at M = 2048).  The strict upper triangle of the tile is zeroed.
__global__ void __launch_bounds__(IPX_BLOCK)
k_potrf64(double *G, int M, int kb, int *flag, double *work) {
  __shared__ double T[FB][FB + 1];
  __shared__ double rinv[FB];
  __shared__ double diag0[FB];
  const int tid = threadIdx.x;
  double *g = G + ((int64_t)kb * FB) * M + (int64_t)kb * FB;
  {
    double v[FB * FB / IPX_BLOCK];                    // (all sixteen loads in flight)
#pragma unroll
    for (int u = 0; u < FB * FB / IPX_BLOCK; ++u) {
      const int e = tid + u * IPX_BLOCK;
      v[u] = g[(int64_t)(e >> 6) * M + (e & 63)];
    }
#pragma unroll
    for (int u = 0; u < FB * FB / IPX_BLOCK; ++u) {
      const int e = tid + u * IPX_BLOCK;
      T[e >> 6][e & 63] = v[u];
    }
  }
  if (tid < FB) diag0[tid] = work[kb * FB + tid];
  __syncthreads();
  int bits = 0;
  double ratio = 1.0;
#pragma unroll
  for (int b = 0; b < FB / PB; ++b) {
    const int c0 = PB * b;
    if (tid < IPX_WAVE) {                             // (a) -- the whole first wave, lanes >= 16 idle along
      const int r = tid & (PB - 1);
      double x[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) x[c] = T[c0 + r][c0 + c];
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const double d = lane_bcast(x[j], j);
        if (tid == 0) {
          // bit 1: the pivot lost 43 bits against its diagonal entry (numerically rank deficient,
          // the factorization goes on); bit 4: it is not positive (no factorization)
          const double d0 = diag0[c0 + j];
          if (!(d > IPX_PIVOT_RTOL * d0)) bits |= (d > 0.0) ? 1 : 5;
          if (d > 0.0) ratio = fmin(ratio, d / d0);
        }
        const double dd = d > 0.0 ? d : 1.0;
        const double rs = potrf_rsqrt(dd);
        const double l = x[j] * rs;                   // (lane j: d / sqrt(d) = L[j][j] to an ulp)
        x[j] = l;
        if (tid == j) rinv[c0 + j] = rs;
#pragma unroll
        for (int c = j + 1; c < PB; ++c) x[c] = __builtin_fma(-l, lane_bcast(l, c), x[c]);
      }
      if (tid < PB) {
#pragma unroll
        for (int c = 0; c < PB; ++c) T[c0 + r][c0 + c] = c <= r ? x[c] : 0.0;
      }
    }
    __syncthreads();
    const int below = FB - c0 - PB;                   // rows under the block
    if (tid < below) {                                // (b)
      const int R = c0 + PB + tid;
      double x[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) x[c] = T[R][c0 + c];
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const double xj = x[j] * rinv[c0 + j];
        x[j] = xj;
#pragma unroll
        for (int c = j + 1; c < PB; ++c) x[c] = __builtin_fma(-xj, T[c0 + c][c0 + j], x[c]);
      }
#pragma unroll
      for (int c = 0; c < PB; ++c) T[R][c0 + c] = x[c];
    }
    __syncthreads();
    for (int e = tid; e < below * below; e += IPX_BLOCK) {       // (c)
      const int i = e / below, j = e - i * below;
      if (j <= i) {
        const int ri = c0 + PB + i, rj = c0 + PB + j;
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < PB; ++k) sum = __builtin_fma(T[ri][c0 + k], T[rj][c0 + k], sum);
        T[ri][rj] -= sum;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < FB * FB / IPX_BLOCK; ++u) {
    const int e = tid + u * IPX_BLOCK, r = e >> 6, c = e & 63;
    g[(int64_t)r * M + c] = (c <= r) ? T[r][c] : 0.0;
  }
  if (tid == 0) {
    if (bits) atomicOr(flag, bits);
    work[M] = fmin(work[M], ratio);             // (one workgroup at a time: no race)
  }
}

// Panel below the diagonal tile: G_ik <- G_ik L_kk^-T for the tile rows i > kb, 64 rows per
// workgroup, both tiles in LDS, in four column blocks of 16: (1) what the earlier blocks
// contribute, X[:, :c0] L[c0:c0+16, :c0]', as a small product over all 256 threads (four
// outputs each); (2) the 16 x 16 triangular solve, a lane per row out of registers,
// right-looking (one multiply + one multiply-add on the dependent chain per column).  The
// first version walked all 64 columns with four lanes per row and a DPP combine per column:
// 22 us per tile column of the panel.
__global__ void __launch_bounds__(IPX_BLOCK)
k_trsm64(double *G, int M, int kb) {
  __shared__ double Lk[FB][FB + 1];
  __shared__ double P[FB][FB + 1];
  __shared__ double rinv[FB];
  const int tid = threadIdx.x;
  const int ib = kb + 1 + blockIdx.x;
  const double *lk = G + ((int64_t)kb * FB) * M + (int64_t)kb * FB;
  double *p = G + ((int64_t)ib * FB) * M + (int64_t)kb * FB;
  {
    double va[FB * FB / IPX_BLOCK], vb[FB * FB / IPX_BLOCK];
#pragma unroll
    for (int u = 0; u < FB * FB / IPX_BLOCK; ++u) {
      const int e = tid + u * IPX_BLOCK;
      va[u] = lk[(int64_t)(e >> 6) * M + (e & 63)];
      vb[u] = p[(int64_t)(e >> 6) * M + (e & 63)];
    }
#pragma unroll
    for (int u = 0; u < FB * FB / IPX_BLOCK; ++u) {
      const int e = tid + u * IPX_BLOCK;
      Lk[e >> 6][e & 63] = va[u];
      P[e >> 6][e & 63] = vb[u];
    }
  }
  __syncthreads();
  if (tid < FB) rinv[tid] = 1.0 / Lk[tid][tid];
  const int row = tid >> 2, q = tid & 3;
#pragma unroll
  for (int b = 0; b < FB / PB; ++b) {
    const int c0 = PB * b;
    if (b > 0) {                                      // (1) columns c0 + 4 q .. + 4 of this row
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
      for (int k = 0; k < c0; ++k) {
        const double xv = P[row][k];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_fma(xv, Lk[c0 + 4 * q + u][k], acc[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) P[row][c0 + 4 * q + u] -= acc[u];
    }
    __syncthreads();                                  // (also: rinv, first trip)
    if (tid < FB) {                                   // (2) row tid
      double x[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) x[c] = P[tid][c0 + c];
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const double xj = x[j] * rinv[c0 + j];
        x[j] = xj;
#pragma unroll
        for (int c = j + 1; c < PB; ++c) x[c] = __builtin_fma(-xj, Lk[c0 + c][c0 + j], x[c]);
      }
#pragma unroll
      for (int c = 0; c < PB; ++c) P[tid][c0 + c] = x[c];
    }
    __syncthreads();
  }
  for (int e = tid; e < FB * FB; e += IPX_BLOCK) {
    const int r = e >> 6, c = e & 63;
    p[(int64_t)r * M + c] = P[r][c];
  }
}

// Trailing update G_ij -= L_ik L_jk' for kb < j <= i (a workgroup per tile of the lower
// triangle; a diagonal tile is updated whole).
__global__ void __launch_bounds__(IPX_BLOCK)
k_syrk64(double *G, int M, int kb, int rest) {
  __shared__ __attribute__((aligned(16))) double sA[GPANEL];
  __shared__ __attribute__((aligned(16))) double sB[GPANEL];
  const int tile = blockIdx.x;
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  while (ti * (ti + 1) / 2 > tile) --ti;
  const int tj = tile - ti * (ti + 1) / 2;
  if (ti >= rest) return;
  const int ib = kb + 1 + ti, jb = kb + 1 + tj;
  const Opnd Pn{G + ((int64_t)ib * FB) * M + (int64_t)kb * FB, M, 1};
  const Opnd Qn{G + ((int64_t)jb * FB) * M + (int64_t)kb * FB, M, 1};
  v4d acc[2][2];
  tile_pqt(Pn, Qn, FB, acc, sA, sB);
  double *c = G + ((int64_t)ib * FB) * M + (int64_t)jb * FB;
  IPX_TILE_FOREACH(acc, row, col, val, c[(int64_t)row * M + col] -= val);
}

// Every diagonal tile of L inverted in place (lower triangular; the strict upper part of the
// tile zero): column c of the inverse by forward substitution, a lane per column, the column in
// registers (fully unrolled), L[i][t] an LDS broadcast.
__global__ void __launch_bounds__(IPX_WAVE)
k_trtri_diag64(double *G, int M) {
  __shared__ double Lk[FB][FB + 1];
  const int lane = threadIdx.x, kb = blockIdx.x;
  double *g = G + ((int64_t)kb * FB) * M + (int64_t)kb * FB;
  for (int r = 0; r < FB; ++r) Lk[r][lane] = g[(int64_t)r * M + lane];
  __syncthreads();
  // x_i = (e_c[i] - sum_{t<i} L[i][t] x_t) / L[i][i]   (x_t = 0 for t < c: exact zeros)
  double x[FB];
#pragma unroll
  for (int i = 0; i < FB; ++i) {
    double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int t = 0; t < i; ++t) s = __builtin_fma(-Lk[i][t], x[t], s);
    x[i] = (i >= lane) ? s / Lk[i][i] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < FB; ++i) Lk[i][lane] = x[i];          // (row i, column lane of the inverse)
  __syncthreads();
  for (int r = 0; r < FB; ++r) g[(int64_t)r * M + lane] = Lk[r][lane];
}

// One level of the in-place triangular inverse: pair p = blockIdx.z of neighbouring blocks of
// `sb` tiles -- block 1 = tiles [2 p sb, (2 p + 1) sb), block 2 = the tiles after it (up to sb,
// fewer at the end of the matrix).  STAGE 1: T = L21 X11, stored TRANSPOSED where X12 would be
// (the upper triangle is free); STAGE 2: X21 = -X22 T into the place of L21.  The K ranges stop
// where the triangular operands do (tile aligned), so nothing parked in the upper triangle is
// ever read as data.
template <int STAGE>
__global__ void __launch_bounds__(IPX_BLOCK)
k_trtri_level(double *G, int M, int nb, int sb) {
  __shared__ __attribute__((aligned(16))) double sA[GPANEL];
  __shared__ __attribute__((aligned(16))) double sB[GPANEL];
  const int p = blockIdx.z, tx = blockIdx.x, ty = blockIdx.y;
  const int b1 = 2 * p * sb, b2 = b1 + sb;              // first tiles of the two blocks
  const int n2 = min(sb, nb - b2);                      // tiles of block 2
  if (n2 <= 0 || ty >= n2) return;
  const int64_t c1 = (int64_t)b1 * FB, r2 = (int64_t)b2 * FB;
  v4d acc[2][2];
  if (STAGE == 1) {
    // T[ty][tx] = sum_k L21[ty][k] X11[k][tx],  k over the tiles tx .. sb-1 of block 1
    const int k0 = tx * FB, K = sb * FB - k0;
    const Opnd Pn{G + (r2 + (int64_t)ty * FB) * M + c1 + k0, M, 1};
    const Opnd Qn{G + (c1 + k0) * M + c1 + (int64_t)tx * FB, 1, M};
    tile_pqt(Pn, Qn, K, acc, sA, sB);
    double *t = G + (c1 + (int64_t)tx * FB) * M + r2 + (int64_t)ty * FB;     // (transposed)
    IPX_TILE_FOREACH(acc, row, col, val, t[(int64_t)col * M + row] = val);
  } else {
    // X21[ty][tx] = -sum_k X22[ty][k] T[k][tx],  k over the tiles 0 .. ty of block 2
    const int K = (ty + 1) * FB;
    const Opnd Pn{G + (r2 + (int64_t)ty * FB) * M + r2, M, 1};
    const Opnd Qn{G + (c1 + (int64_t)tx * FB) * M + r2, M, 1};
    tile_pqt(Pn, Qn, K, acc, sA, sB);
    double *x = G + (r2 + (int64_t)ty * FB) * M + c1 + (int64_t)tx * FB;
    IPX_TILE_FOREACH(acc, row, col, val, x[(int64_t)row * M + col] = -val);
  }
}

// Y = X'X for the lower triangular X in the lower triangle of G: tile (ti, tj), tj <= ti, sums
// over the rows from tile ti on (X is zero above); written with its mirror.
__global__ void __launch_bounds__(IPX_BLOCK)
k_xtx64(const double *G, int M, int nb, double *Y) {
  __shared__ __attribute__((aligned(16))) double sA[GPANEL];
  __shared__ __attribute__((aligned(16))) double sB[GPANEL];
  const int tile = blockIdx.x;
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  while (ti * (ti + 1) / 2 > tile) --ti;
  const int tj = tile - ti * (ti + 1) / 2;
  if (ti >= nb) return;
  const int64_t k0 = (int64_t)ti * FB;
  const Opnd Pn{G + k0 * M + (int64_t)ti * FB, 1, M};
  const Opnd Qn{G + k0 * M + (int64_t)tj * FB, 1, M};
  v4d acc[2][2];
  tile_pqt(Pn, Qn, M - (int)k0, acc, sA, sB);
  IPX_TILE_FOREACH(acc, row, col, val, {
    const int64_t gr = (int64_t)ti * FB + row, gc = (int64_t)tj * FB + col;
    if (gc <= gr) { Y[gr * M + gc] = val; Y[gc * M + gr] = val; }
  });
}

}  // namespace

int ipx_dense_gemv_launch(int m, int n, const double *A, int64_t lda, const double *x,
                          double alpha, const double *diag, double beta, const double *yin,
                          double *yout, double *partial, int *npartial, const double *guard,
                          hipStream_t st) {
  if (m == 0) { if (npartial) *npartial = 0; return IPX_OK; }
  if (beta == 0.0) yin = nullptr;
  const int wpb = IPX_BLOCK / IPX_WAVE;
  int grid = (m + wpb - 1) / wpb;
  if (grid > 2048) grid = 2048;
  if (npartial) *npartial = grid;
  const int square = m == n;
  if (partial)
    hipLaunchKernelGGL(k_dense_gemv<true>, dim3(grid), dim3(IPX_BLOCK), 0, st, m, n, A, lda, x,
                       alpha, diag, beta, yin, yout, square, partial, guard);
  else
    hipLaunchKernelGGL(k_dense_gemv<false>, dim3(grid), dim3(IPX_BLOCK), 0, st, m, n, A, lda, x,
                       alpha, diag, beta, yin, yout, square, partial, guard);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

extern "C" {

// yout = alpha*A x [+ diag*x] [+ beta*yin]; red (optional) = {sum y^2, sum x*y}.
int ipx_dense_gemv(int64_t m, int64_t n, const double *A, int64_t lda, const double *x,
                   double alpha, const double *diag, double beta, const double *yin,
                   double *yout, double *red, double *ws, void *stream) {
  if (m < 0 || n < 0 || !A || !x || !yout || lda < n) return IPX_EINVAL;
  if (m == 0) {
    if (red) (void)hipMemsetAsync(red, 0, 2 * sizeof(double), (hipStream_t)stream);
    return IPX_OK;
  }
  if (red && !ws) return IPX_EINVAL;
  int np = 0;
  int rc = ipx_dense_gemv_launch((int)m, (int)n, A, lda, x, alpha, diag, beta, yin, yout,
                                 red ? ws : nullptr, &np, nullptr, (hipStream_t)stream);
  if (rc != IPX_OK) return rc;
  if (red) {
    hipLaunchKernelGGL(k_fold2, dim3(1), dim3(IPX_BLOCK), 0, (hipStream_t)stream, ws, np, red);
    IPX_CHECK_LAUNCH();
  }
  return IPX_OK;
}

int64_t ipx_dense_padded(int64_t m) { return ((m + FB - 1) / FB) * FB; }

// G (M x M, M = ipx_dense_padded(m)) = A A' via fp64 MFMA.
// K-splits that even out the load: with one split the tiles of the lower triangle are dealt
// to the CUs whole (528 tiles at m = 2000: one CU in sixteen gets a third workgroup and the
// launch lasts until it is done); S splits make S times as many, shorter units.  The
// smallest S <= 8 whose busiest CU carries <= 6 % more than the average, 1 if the matrix is
// too short to split.
int ipx_gram_splits(int64_t m, int64_t n) {
  const int M = (int)ipx_dense_padded(m);
  const int nt = (M + GT - 1) / GT;
  const int64_t ntri = (int64_t)nt * (nt + 1) / 2;
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess &&
        p.multiProcessorCount > 0)
      cus = p.multiProcessorCount;
  }
  const int64_t chunks = (n + GK - 1) / GK;
  for (int s = 1; s <= 8; ++s) {
    if (chunks / s < 16) break;
    const int64_t units = ntri * s;
    const int64_t busiest = (units + cus - 1) / cus;
    if ((double)busiest * cus <= 1.06 * (double)units) return s;
  }
  return 1;
}

int64_t ipx_gram_ws_doubles(int64_t m, int32_t splits) {
  if (splits <= 1) return 0;
  const int M = (int)ipx_dense_padded(m);
  const int64_t nt = (M + GT - 1) / GT;
  return (int64_t)splits * (nt * (nt + 1) / 2) * GT * GT;
}

// ws: ipx_gram_ws_doubles(m, splits) doubles of scratch (NULL with splits <= 1).
int ipx_gram_f64_mfma_split(int64_t m, int64_t n, const double *A, int64_t lda, double *G,
                            double *ws, int32_t splits, void *stream) {
  if (m < 1 || n < 0 || !A || !G || lda < n || splits < 1 || (splits > 1 && !ws)) return IPX_EINVAL;
  const int M = (int)ipx_dense_padded(m);
  const int nt = (M + GT - 1) / GT;
  const int ntri = nt * (nt + 1) / 2;
  const dim3 grid(ntri * splits), block(IPX_BLOCK);
  // 16-byte loads need every row start 16-byte aligned
  const bool vec = (lda % 2 == 0) && (((uintptr_t)A) % 16 == 0);
  if (vec)
    hipLaunchKernelGGL(k_gram_mfma<true>, grid, block, 0, (hipStream_t)stream, (int)m, (int)n, A,
                       lda, G, M, nt, (int)splits, ws);
  else
    hipLaunchKernelGGL(k_gram_mfma<false>, grid, block, 0, (hipStream_t)stream, (int)m, (int)n, A,
                       lda, G, M, nt, (int)splits, ws);
  IPX_CHECK_LAUNCH();
  if (splits > 1) {
    const int64_t tot = (int64_t)ntri * GT * GT;
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((tot + IPX_BLOCK - 1) / IPX_BLOCK)),
                       dim3(IPX_BLOCK), 0, (hipStream_t)stream, (int)m, M, nt, (int)splits, ws, G);
    IPX_CHECK_LAUNCH();
  }
  return IPX_OK;
}

int ipx_gram_f64_mfma(int64_t m, int64_t n, const double *A, int64_t lda, double *G,
                      void *stream) {
  return ipx_gram_f64_mfma_split(m, n, A, lda, G, nullptr, 1, stream);
}

// G (M x M padded) = A A' for CSR A (dense fallback of the sparse path).
int ipx_aat_dense(int64_t m, const int32_t *rowptr, const int32_t *colidx, const double *val,
                  double *G, void *stream) {
  if (m < 1 || !rowptr || !G) return IPX_EINVAL;
  const int M = (int)ipx_dense_padded(m);
  const int64_t tot = (int64_t)M * M;
  hipLaunchKernelGGL(k_aat_dense, dim3((unsigned)((tot + IPX_BLOCK - 1) / IPX_BLOCK)),
                     dim3(IPX_BLOCK), 0, (hipStream_t)stream, (int)m, rowptr, colidx, val, G, M);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// In place: lower triangle of G <- L with G = L L' (the strict upper triangle of the diagonal
// tiles is zeroed, the rest of the upper triangle keeps G).  flag (device int) != 0 afterwards
// when a pivot fell below IPX_PIVOT_RTOL x its original diagonal entry (numerically rank
// deficient Jacobian); work (M + 1 doubles): work[M] = min pivot / diagonal, an estimate of
// 1/cond(G).
int ipx_chol_factor(int64_t M, double *G, int *flag, double *work, void *stream) {
  if (M < FB || M % FB || !G || !flag || !work) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)(M / FB);
  if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) return IPX_ELAUNCH;
  hipLaunchKernelGGL(k_save_diag, dim3(((int)M + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0,
                     st, G, (int)M, work);
  IPX_CHECK_LAUNCH();
  for (int k = 0; k < nb; ++k) {
    hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(IPX_BLOCK), 0, st, G, (int)M, k, flag, work);
    IPX_CHECK_LAUNCH();
    const int rest = nb - k - 1;
    if (rest > 0) {
      hipLaunchKernelGGL(k_trsm64, dim3(rest), dim3(IPX_BLOCK), 0, st, G, (int)M, k);
      IPX_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_syrk64, dim3(rest * (rest + 1) / 2), dim3(IPX_BLOCK), 0, st, G, (int)M,
                         k, rest);
      IPX_CHECK_LAUNCH();
    }
  }
  return IPX_OK;
}

// X (M x M) <- (L L')^-1 from the factor in the lower triangle of G.  G is used as the
// workspace of the triangular inverse: on return its lower triangle holds L^-1.
int ipx_chol_inverse(int64_t M, double *G, double *X, void *stream) {
  if (M < FB || M % FB || !G || !X) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)(M / FB);
  hipLaunchKernelGGL(k_trtri_diag64, dim3(nb), dim3(IPX_WAVE), 0, st, G, (int)M);
  IPX_CHECK_LAUNCH();
  for (int sb = 1; sb < nb; sb *= 2) {
    const int pairs = (nb + 2 * sb - 1) / (2 * sb);
    const dim3 grid(sb, sb, pairs), block(IPX_BLOCK);
    hipLaunchKernelGGL(k_trtri_level<1>, grid, block, 0, st, G, (int)M, nb, sb);
    IPX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_trtri_level<2>, grid, block, 0, st, G, (int)M, nb, sb);
    IPX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_xtx64, dim3(nb * (nb + 1) / 2), dim3(IPX_BLOCK), 0, st, G, (int)M, nb, X);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

}  // extern "C"
