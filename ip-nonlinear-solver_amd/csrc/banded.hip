// Banded SPD solver for S = A A' (the normal-equation matrix of the projection
// operators, reference projections.py:58-90) on gfx950.
//
// A sequential banded Cholesky has m dependent steps; at m = 1e5 that would
// dominate a CG iteration.  Instead the matrix is cut into P chunks of c
// interior rows separated by k separator rows (k = half bandwidth).  Interior
// blocks are independent SPD band matrices: one lane per chunk factors it
// (LDL') and solves with it, all chunks in parallel, with factor data laid out
// [step][chunk] so lane-adjacent chunks read adjacent addresses.  Eliminating
// the interiors leaves an SPD band matrix of half bandwidth 2k-1 on the
// (P-1)k separator unknowns (a Schur complement), which is handled by the
// same code recursively until one chunk holds everything.  Per solve:
//
//   down:  y_I = B^-1 w_I per chunk;  reduced rhs g = w_S - F'y - E'y
//   top :  one chunk
//   up  :  x_I = y_I - V x_S(left) - W x_S(right)      (V, W = B^-1 E, B^-1 F
//                                                       "spikes", kept from the
//                                                       factorization)
//
// All arithmetic is fp64 with a fixed operation order (no atomics), so a
// solve is bitwise reproducible.  A non-positive pivot sets an error flag
// that ipx_banded_status() reports (rank-deficient Jacobian).
#include "ipx_common.h"
#include <vector>

namespace {

constexpr int KMAX = 8;       // largest half bandwidth with a compiled kernel
constexpr int MAX_LEVELS = 8;
constexpr int WIDE_MIN_ROWS = 2048;   // below: one serially swept chunk is fast enough
constexpr int ITER_TOP_MIN_ROWS = 1024;   // a serial top level longer than this is worth avoiding
constexpr double ITER_ETA_MAX = 0.5;      // contraction bound up to which defect correction is used

struct Level {
  int m, k, c, P, q;          // size, half bandwidth, interior size, chunks, q = c + k
  int mR;                     // (P-1)*k  (0 at the top level)
  double *band;               // (k+1) x m lower band: band[d*m+i] = T[i][i-d]   (level 0: caller's)
  double *Dinv;               // [step][chunk]   1/d_j
  double *L;                  // [(step*k + (d-1))][chunk]   l_{j,j-d}
  double *V, *W;              // m x k spikes (rows of separators unused)
  double *rhs;                // m   (levels > 0: reduced rhs written by the level below)
  double *sol;                // m   (levels > 0)
};

struct Banded {
  int nlev;
  Level lev[MAX_LEVELS];
  int *flag;                  // device: != 0 after a non-positive pivot
  double *gL, *gR;            // halves of the level-1 right-hand side (fast path)
  double *ybuf;               // level-0 down-sweep result (fast path is out of place)
  double *rinv;               // 1 / diag of the level-1 matrix (decoupled path)
  bool decoupled;             // level-1 system numerically diagonal: skip the middle kernel
  bool upper_done;            // levels >= 1 factored (deferred while a decoupled solve may do)
  bool fast;                  // three-launch path usable (LDS budget)
  bool fast_plan;             // ... as planned at creation (a factorization may step down)
  bool wide;                  // half bandwidth 5..8 with chunks: the separator system (half
                              // bandwidth 2k-1 > KMAX) is only FORMED, to test that its blocks
                              // decouple; solves run the single-launch decoupled kernel or not
                              // at all (ipx_banded_status: IPX_EUNSUPPORTED)
  int down_T;                 // chunks per workgroup in k_down0
  size_t lds_down;
  int mid_buf;                // doubles of LDS in k_middle for the level vectors
  double *slab;               // factor data of levels >= 1, contiguous
  size_t nslab;               // its length in doubles
  int nslab_lds;              // = nslab when the slab is staged in LDS, else 0
  // parallel cyclic reduction (k = 1): scratch for the factor-time level check, per-level
  // flags (device), and the level at which the reduced system is numerically diagonal
  int *pcr_flags;             // 2 x (PCR_LMAX + 1) ints: [s] != 0 = level s still coupled,
                              // [PCR_LMAX + 1 + s] != 0 = a reduced diagonal entry not positive
  int pcr_L;                  // 0: PCR solve not usable
  // defect correction on the single-launch solve (separator blocks coupled, top level too
  // long for a serial sweep or not compiled): see iter_solve
  bool mid_fits;              // k_middle's level vectors fit in LDS
  bool iter_cand;             // geometry qualifies; iter_buf / eta allocated
  int iter_N;                 // correction steps per solve (0: mode off), set by ipx_banded_status
  double *iter_buf;           // 4 m doubles: x (two copies), residual, correction
  double *eta;                // device: max_t || D_t^-1 [E_t,t-1  E_t,t+1] ||_inf  (block Jacobi)
  double eta_host;
  bool chunk_pending;         // ipx_banded_refactor left the chunk factorization out (solves on
                              // the cyclic reduction do not read it): ensure_chunks runs it
  int last_L;                 // pcr_L of the last factorization that ipx_banded_status found clean
                              // (no flag bit, decoupled, cyclic reduction usable), else 0: what
                              // ipx_banded_status_deferred assumes for the next one
  std::vector<void *> allocs;
};

constexpr int PCR_LMAX = 7;     // reduction distance up to 2^7 = 128 rows

__device__ __forceinline__ int chunk_rows(int m, int q, int c, int P, int t) {
  // interior rows of chunk t: all chunks have c rows except the last, which
  // takes what is left (1 .. c+k rows).
  return (t == P - 1) ? (m - t * q) : c;
}

// ------------------------------------------------------------------ factor
// One lane per chunk: LDL' of the interior block, then the spikes.
template <int K>
__global__ void __launch_bounds__(IPX_WAVE)
k_factor_chunks(int m, int c, int P, const double *__restrict__ band, double *__restrict__ Dinv,
                double *__restrict__ L, double *__restrict__ V, double *__restrict__ W,
                int *flag) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P) return;
  const int q = c + K;
  const int base = t * q;
  const int ct = chunk_rows(m, q, c, P, t);

  // ---- LDL': window of the last K rows' multipliers and pivots
  double lw[K][K];   // lw[r][d-1] = l_{j-1-r, j-1-r-d}
  double dw[K];      // dw[r] = d_{j-1-r}
#pragma unroll
  for (int r = 0; r < K; ++r) {
    dw[r] = 1.0;
#pragma unroll
    for (int d = 0; d < K; ++d) lw[r][d] = 0.0;
  }
  for (int j = 0; j < ct; ++j) {
    const int i = base + j;
    double lrow[K];   // lrow[d-1] = l_{j,j-d}
    // columns from far (d = K) to near (d = 1)
#pragma unroll
    for (int d = K; d >= 1; --d) {
      double s = 0.0;
      if (d <= j) {
        s = band[(int64_t)d * m + i];
        // subtract sum over r = j-e (e > d, e <= K): l_{j,j-e} d_{j-e} l_{j-d,j-e}
#pragma unroll
        for (int e = K; e > d; --e) {
          if (e <= j) s -= lrow[e - 1] * dw[e - 1] * lw[d - 1][e - d - 1];
        }
        s = s / dw[d - 1];
      }
      lrow[d - 1] = s;
    }
    double dj = band[i];
#pragma unroll
    for (int e = K; e >= 1; --e)
      if (e <= j) dj -= lrow[e - 1] * lrow[e - 1] * dw[e - 1];
    if (!(dj > IPX_PIVOT_RTOL * band[i])) atomicOr(flag, (dj > 0.0) ? 1 : 5);   // 4: not positive
    Dinv[(int64_t)j * P + t] = 1.0 / dj;
#pragma unroll
    for (int d = 1; d <= K; ++d) L[((int64_t)j * K + (d - 1)) * P + t] = lrow[d - 1];
    // shift windows
#pragma unroll
    for (int r = K - 1; r > 0; --r) {
      dw[r] = dw[r - 1];
#pragma unroll
      for (int d = 0; d < K; ++d) lw[r][d] = lw[r - 1][d];
    }
    dw[0] = dj;
#pragma unroll
    for (int d = 0; d < K; ++d) lw[0][d] = lrow[d];
  }

  // ---- spikes: solve B X = E (left coupling) and B X = F (right coupling)
  for (int side = 0; side < 2; ++side) {
    if (side == 0 && t == 0) continue;
    if (side == 1 && t == P - 1) continue;
    double *out = side == 0 ? V : W;
    for (int a = 0; a < K; ++a) {
      // forward: z_j = rhs_j - sum_d l_{j,j-d} z_{j-d};  stored scaled later
      double zw[K];
#pragma unroll
      for (int r = 0; r < K; ++r) zw[r] = 0.0;
      for (int j = 0; j < ct; ++j) {
        double rhs = 0.0;
        if (side == 0) {          // E[j][a] = T[base+j][base-K+a], d = j+K-a
          if (j <= a) rhs = band[(int64_t)(j + K - a) * m + base + j];
        } else {                  // F[j][a] = T[base+ct+a][base+j], d = ct+a-j
          const int d = ct + a - j;
          if (d <= K) rhs = band[(int64_t)d * m + base + ct + a];
        }
        double z = rhs;
#pragma unroll
        for (int d = 1; d <= K; ++d)
          if (d <= j) z -= L[((int64_t)j * K + (d - 1)) * P + t] * zw[d - 1];
#pragma unroll
        for (int r = K - 1; r > 0; --r) zw[r] = zw[r - 1];
        zw[0] = z;
        out[(int64_t)(base + j) * K + a] = z;
      }
      // diagonal scale + backward: y_j = z_j/d_j - sum_d l_{j+d,j} y_{j+d}
      double yw[K];
#pragma unroll
      for (int r = 0; r < K; ++r) yw[r] = 0.0;
      for (int j = ct - 1; j >= 0; --j) {
        double y = out[(int64_t)(base + j) * K + a] * Dinv[(int64_t)j * P + t];
#pragma unroll
        for (int d = 1; d <= K; ++d)
          if (j + d < ct) y -= L[((int64_t)(j + d) * K + (d - 1)) * P + t] * yw[d - 1];
#pragma unroll
        for (int r = K - 1; r > 0; --r) yw[r] = yw[r - 1];
        yw[0] = y;
        out[(int64_t)(base + j) * K + a] = y;
      }
    }
  }
}

// Coupling entries read straight from the band storage.
template <int K>
__device__ __forceinline__ double coupF(const double *band, int m, int base, int ct, int j, int a) {
  const int d = ct + a - j;        // F[j][a], j in [0,ct)
  return (d >= 1 && d <= K) ? band[(int64_t)d * m + base + ct + a] : 0.0;
}
template <int K>
__device__ __forceinline__ double coupE(const double *band, int m, int base, int j, int a) {
  const int d = j + K - a;         // E[j][a], j in [0,ct)
  return (d >= 1 && d <= K) ? band[(int64_t)d * m + base + j] : 0.0;
}

// Reduced (Schur complement) matrix in band storage, half bandwidth 2K-1.
// One lane per (separator row q, offset d).
template <int K>
__global__ void __launch_bounds__(IPX_BLOCK)
k_reduced_matrix(int m, int c, int P, const double *__restrict__ band,
                 const double *__restrict__ V, const double *__restrict__ W,
                 double *__restrict__ Rband) {
  constexpr int KR = 2 * K - 1;
  const int mR = (P - 1) * K;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)mR * (KR + 1)) return;
  const int d = (int)(idx / mR), qrow = (int)(idx % mR);
  const int qcol = qrow - d;
  double r = 0.0;
  if (qcol >= 0) {
    const int q = c + K;
    const int t = qrow / K, a = qrow % K;
    const int t2 = qcol / K, b = qcol % K;
    const int base = t * q;                 // chunk t (left of separator t)
    const int ct = c;                        // chunk t < P-1 always has c rows
    const int sa = base + ct + a;            // global row of separator (t,a)
    if (t2 == t) {
      // T[sa][sb] - sum_j F_t[j][a] W_t[j][b] - sum_j E_{t+1}[j][a] V_{t+1}[j][b]
      r = band[(int64_t)(a - b) * m + sa];
      for (int j = ct - K; j < ct; ++j)
        if (j >= 0) r -= coupF<K>(band, m, base, ct, j, a) * W[(int64_t)(base + j) * K + b];
      const int base2 = (t + 1) * q;
      const int ct2 = chunk_rows(m, q, c, P, t + 1);
      for (int j = 0; j < K && j < ct2; ++j)
        r -= coupE<K>(band, m, base2, j, a) * V[(int64_t)(base2 + j) * K + b];
    } else if (t2 == t - 1) {
      // coupling through chunk t: - sum_j E_t[j][a']... rows: separator t (right
      // of chunk t) with separator t-1 (left of chunk t):  - F_t' V_t
      for (int j = ct - K; j < ct; ++j)
        if (j >= 0) r -= coupF<K>(band, m, base, ct, j, a) * V[(int64_t)(base + j) * K + b];
    }
  }
  Rband[(int64_t)d * mR + qrow] = r;
}

// ------------------------------------------------------------------- solve
// Down sweep: one lane per chunk solves B y = w on its interior rows.
template <int K>
__global__ void __launch_bounds__(IPX_WAVE)
k_solve_chunks(int m, int c, int P, const double *__restrict__ Dinv, const double *__restrict__ L,
               const double *w, double *y, const double *__restrict__ guard) {
  if (guard && *guard != 0.0) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P) return;
  const int q = c + K, base = t * q;
  const int ct = chunk_rows(m, q, c, P, t);
  double zw[K];
#pragma unroll
  for (int r = 0; r < K; ++r) zw[r] = 0.0;
  for (int j = 0; j < ct; ++j) {
    double z = w[base + j];
#pragma unroll
    for (int d = 1; d <= K; ++d)
      if (d <= j) z -= L[((int64_t)j * K + (d - 1)) * P + t] * zw[d - 1];
#pragma unroll
    for (int r = K - 1; r > 0; --r) zw[r] = zw[r - 1];
    zw[0] = z;
    y[base + j] = z;
  }
  double yw[K];
#pragma unroll
  for (int r = 0; r < K; ++r) yw[r] = 0.0;
  for (int j = ct - 1; j >= 0; --j) {
    double v = y[base + j] * Dinv[(int64_t)j * P + t];
#pragma unroll
    for (int d = 1; d <= K; ++d)
      if (j + d < ct) v -= L[((int64_t)(j + d) * K + (d - 1)) * P + t] * yw[d - 1];
#pragma unroll
    for (int r = K - 1; r > 0; --r) yw[r] = yw[r - 1];
    yw[0] = v;
    y[base + j] = v;
  }
}

// Reduced right-hand side: g[(t,a)] = w[s] - F_t' y_I(t) - E_{t+1}' y_I(t+1).
template <int K>
__global__ void __launch_bounds__(IPX_BLOCK)
k_reduced_rhs(int m, int c, int P, const double *__restrict__ band, const double *__restrict__ w,
              const double *__restrict__ y, double *__restrict__ g,
              const double *__restrict__ guard) {
  if (guard && *guard != 0.0) return;
  const int mR = (P - 1) * K;
  const int qrow = blockIdx.x * blockDim.x + threadIdx.x;
  if (qrow >= mR) return;
  const int q = c + K, t = qrow / K, a = qrow % K;
  const int base = t * q, ct = c;
  double r = w[base + ct + a];
  for (int j = ct - K; j < ct; ++j)
    if (j >= 0) r -= coupF<K>(band, m, base, ct, j, a) * y[base + j];
  const int base2 = (t + 1) * q;
  const int ct2 = chunk_rows(m, q, c, P, t + 1);
  for (int j = 0; j < K && j < ct2; ++j) r -= coupE<K>(band, m, base2, j, a) * y[base2 + j];
  g[qrow] = r;
}

// Up sweep: x_I = y_I - V xs(left) - W xs(right); separators take xs.
template <int K>
__global__ void __launch_bounds__(IPX_BLOCK)
k_correct(int m, int c, int P, const double *__restrict__ V, const double *__restrict__ W,
          const double *__restrict__ xs, double *__restrict__ x,
          const double *__restrict__ guard) {
  if (guard && *guard != 0.0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int q = c + K;
  int t = i / q;
  if (t > P - 1) t = P - 1;      // the last chunk may be longer than q
  const int j = i - t * q;
  const int ct = chunk_rows(m, q, c, P, t);
  if (j >= ct) {                 // separator row (t, j-ct)
    x[i] = xs[t * K + (j - ct)];
    return;
  }
  double v = x[i];
  if (t > 0) {
#pragma unroll
    for (int a = 0; a < K; ++a) v -= V[(int64_t)i * K + a] * xs[(t - 1) * K + a];
  }
  if (t < P - 1) {
#pragma unroll
    for (int a = 0; a < K; ++a) v -= W[(int64_t)i * K + a] * xs[t * K + a];
  }
  x[i] = v;
}

// Level-0 up sweep of the three-launch path, out of place (y -> x), optionally
// fused with the normal-equation residual ||w - S x||^2 (k_band_residual): the
// corrected values of the 2K neighbours a row needs are recomputed from y, so
// no second pass over x (and no extra kernel boundary) is needed.
// Separator solution.  Normally the middle kernel's output `xs`.  When the
// separator (Schur complement) system is diagonal to working precision --
// S^-1 of a well-conditioned band matrix decays geometrically, so with chunks
// of 64 rows the coupling between neighbouring separators is often below
// 2^-56 of the diagonal -- it is evaluated in place as (gL + gR) / R_tt and the
// middle kernel is skipped (checked numerically at every factorization,
// k = 1 only; see k_decoupling_check).
// max into a non-negative device double (its bit pattern orders like the value)
__device__ __forceinline__ void eta_max(double *eta, double v) {
  if (!(v >= 0.0)) v = 1.0e300;                                   // NaN: no contraction claimed
  atomicMax((unsigned long long *)eta, (unsigned long long)__double_as_longlong(v));
}

struct SepValues {
  const double *xs, *gL, *gR, *rinv;
  __device__ __forceinline__ double operator()(int q) const {
    return xs ? xs[q] : (gL[q] + gR[q]) * rinv[q];
  }
};

__global__ void __launch_bounds__(IPX_BLOCK)
k_decoupling_check(int mR, const double *__restrict__ Rband, double *__restrict__ rinv,
                   int *flag, double *eta) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= mR) return;
  const double d = Rband[t];
  const double lo = Rband[(int64_t)mR + t];                       // R[t][t-1]
  const double up = (t + 1 < mR) ? Rband[(int64_t)mR + t + 1] : 0.0;   // R[t+1][t]
  rinv[t] = 1.0 / d;
  const double tiny = 1.3877787807814457e-17;                     // 2^-56
  if (!(fmax(fabs(lo), fabs(up)) <= tiny * fabs(d))) {
    atomicOr(flag, 2);
    if (eta) eta_max(eta, (fabs(lo) + fabs(up)) / fabs(d));
  }
}

// Block form for half bandwidth K > 1: the separator system is block tridiagonal with
// K x K blocks.  One lane per separator t: inverse of its diagonal block (Gauss-Jordan on
// the SPD block, K <= 8) into rinv[t][K][K], and the coupling block to separator t-1
// tested entry by entry against 2^-56 sqrt(d_a d_b).  The diagonal blocks of a Schur
// complement of an SPD matrix are SPD; a non-positive pivot raises the factor flag.
template <int K>
__global__ void __launch_bounds__(IPX_BLOCK)
k_decoupling_check_block(int nsep, const double *__restrict__ Rband, double *__restrict__ rinv,
                         int *flag, double *eta) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nsep) return;
  const int mR = nsep * K;
  // R[(t,a)][(t,b)], a >= b: offset a - b, row t*K + a
  double D[K][K], Inv[K][K];
#pragma unroll
  for (int a = 0; a < K; ++a) {
#pragma unroll
    for (int b = 0; b < K; ++b) {
      const int hi = a > b ? a : b, lo = a > b ? b : a;
      D[a][b] = Rband[(int64_t)(hi - lo) * mR + (int64_t)t * K + hi];
      Inv[a][b] = a == b ? 1.0 : 0.0;
    }
  }
  const double tiny = 1.3877787807814457e-17;                     // 2^-56
  bool coupled = false, bad = false, hard = false;
  if (t > 0) {
    // R[(t,a)][(t-1,b)]: offset K + a - b (1 .. 2K-1), row t*K + a
#pragma unroll
    for (int a = 0; a < K; ++a) {
#pragma unroll
      for (int b = 0; b < K; ++b) {
        const double v = Rband[(int64_t)(K + a - b) * mR + (int64_t)t * K + a];
        const double db = Rband[(int64_t)(t - 1) * K + b];
        if (!(fabs(v) <= tiny * sqrt(fabs(D[a][a] * db)))) coupled = true;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const double piv = D[c][c];
    if (!(piv > IPX_PIVOT_RTOL * Rband[(int64_t)t * K + c])) { bad = true; hard |= !(piv > 0.0); }
    const double ip = 1.0 / piv;
#pragma unroll
    for (int b = 0; b < K; ++b) { D[c][b] *= ip; Inv[c][b] *= ip; }
#pragma unroll
    for (int a = 0; a < K; ++a) {
      if (a != c) {
        const double f = D[a][c];
#pragma unroll
        for (int b = 0; b < K; ++b) { D[a][b] -= f * D[c][b]; Inv[a][b] -= f * Inv[c][b]; }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < K; ++a)
#pragma unroll
    for (int b = 0; b < K; ++b) rinv[((int64_t)t * K + a) * K + b] = Inv[a][b];
  if (coupled) atomicOr(flag, 2);
  if (bad) atomicOr(flag, hard ? 5 : 1);
  // contraction bound of block Jacobi on the separator system (what replacing R by its
  // diagonal blocks costs): max row sum of |D_t^-1 E_t,t-1| + |D_t^-1 E_t,t+1|; only
  // evaluated by separators with a coupled neighbour
  bool coupled_up = false;
  if (eta && t + 1 < nsep) {
#pragma unroll
    for (int b = 0; b < K; ++b) {
      const double db = Rband[(int64_t)(t + 1) * K + b];
#pragma unroll
      for (int c = 0; c < K; ++c) {
        const double v = Rband[(int64_t)(K + b - c) * mR + (int64_t)(t + 1) * K + b];
        if (!(fabs(v) <= tiny * sqrt(fabs(db * Rband[(int64_t)t * K + c])))) coupled_up = true;
      }
    }
  }
  if (eta && (coupled || coupled_up)) {
    double worst = 0.0;
#pragma unroll
    for (int a = 0; a < K; ++a) {
      double rs = 0.0;
#pragma unroll
      for (int b = 0; b < K; ++b) {
        double lo = 0.0, up = 0.0;
#pragma unroll
        for (int c = 0; c < K; ++c) {
          // E_t,t-1[c][b] = R[(t,c)][(t-1,b)];  E_t,t+1[c][b] = R[(t+1,b)][(t,c)]
          if (t > 0) lo += Inv[a][c] * Rband[(int64_t)(K + c - b) * mR + (int64_t)t * K + c];
          if (t + 1 < nsep)
            up += Inv[a][c] * Rband[(int64_t)(K + b - c) * mR + (int64_t)(t + 1) * K + b];
        }
        rs += fabs(lo) + fabs(up);
      }
      worst = fmax(worst, rs);
    }
    if (worst > 0.0) eta_max(eta, worst);
  }
}

template <int K>
__device__ __forceinline__ double corrected_at(int i, int m, int c, int P, const double *V,
                                               const double *W, const SepValues &xs,
                                               const double *y) {
  const int q = c + K;
  int t = i / q;
  if (t > P - 1) t = P - 1;
  const int j = i - t * q;
  const int ct = chunk_rows(m, q, c, P, t);
  if (j >= ct) return xs(t * K + (j - ct));
  double v = y[i];
  if (t > 0) {
#pragma unroll
    for (int a = 0; a < K; ++a) v -= V[(int64_t)i * K + a] * xs((t - 1) * K + a);
  }
  if (t < P - 1) {
#pragma unroll
    for (int a = 0; a < K; ++a) v -= W[(int64_t)i * K + a] * xs(t * K + a);
  }
  return v;
}

template <int K, bool RESID>
__global__ void __launch_bounds__(IPX_BLOCK)
k_correct_oop(int m, int c, int P, const double *__restrict__ V, const double *__restrict__ W,
              SepValues xs, const double *__restrict__ y, double *__restrict__ x,
              const double *__restrict__ band, const double *__restrict__ w,
              double *__restrict__ partial, const double *__restrict__ guard) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  if (guard && *guard != 0.0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (i < m) {
    const double vi = corrected_at<K>(i, m, c, P, V, W, xs, y);
    x[i] = vi;
    if (RESID) {
      double s = band[i] * vi;
#pragma unroll
      for (int d = 1; d <= K; ++d) {
        if (i - d >= 0)
          s += band[(int64_t)d * m + i] * corrected_at<K>(i - d, m, c, P, V, W, xs, y);
        if (i + d < m)
          s += band[(int64_t)d * m + i + d] * corrected_at<K>(i + d, m, c, P, V, W, xs, y);
      }
      const double res = w[i] - s;
      acc = res * res;
    }
  }
  if (RESID) {
    const double a = ipx_block_reduce<IPX_SUM>(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = a;
  }
}

// ---- S = A A' in band storage: the merge join of the two sorted CSR rows perm[i] and
// perm[i-d], products added in the order of the columns.
// k_aat_band: one lane per (row i, offset d) straight out of global memory (any row order).
// k_aat_band_rows (rows in their own order): a workgroup takes 256 consecutive rows, stages
// their entries -- one contiguous piece of the CSR arrays -- in LDS with coalesced loads and
// joins out of LDS, a lane per row, all offsets; a piece that does not fit takes the global
// path inside the same launch.  (The global join is two dependent loads per step: 32 us at
// m = 1e5, a third of a refactorization; staged: see DESIGN.md section 3.)
template <typename Col, typename Val>
__device__ __forceinline__ double aat_join(int p, int pe, int u, int ue, Col col, Val val,
                                           const double *__restrict__ wcol) {
  double s = 0.0;
  while (p < pe && u < ue) {
    const int cp = col(p), cu = col(u);
    if (cp == cu) { s += wcol ? val(p) * val(u) * wcol[cp] : val(p) * val(u); ++p; ++u; }
    else if (cp < cu) ++p;
    else ++u;
  }
  return s;
}

__global__ void __launch_bounds__(IPX_BLOCK)
k_aat_band(int m, int k, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
           const double *__restrict__ val, const int32_t *__restrict__ perm,
           const double *__restrict__ wcol, double *__restrict__ band) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)m * (k + 1)) return;
  const int d = (int)(idx / m), i = (int)(idx % m);
  double s = 0.0;
  if (i - d >= 0) {
    const int r1 = perm ? perm[i] : i, r2 = perm ? perm[i - d] : i - d;
    s = aat_join(rowptr[r1], rowptr[r1 + 1], rowptr[r2], rowptr[r2 + 1],
                 [&](int e) { return colidx[e]; }, [&](int e) { return val[e]; }, wcol);
  }
  band[(int64_t)d * m + i] = s;
}

constexpr int AAT_CAP = 5120;        // entries a workgroup stages (256 + k rows of <= ~19 entries)

__global__ void __launch_bounds__(IPX_BLOCK)
k_aat_band_rows(int m, int k, const int32_t *__restrict__ rowptr,
                const int32_t *__restrict__ colidx, const double *__restrict__ val,
                const double *__restrict__ wcol, double *__restrict__ band,
                int *__restrict__ zero = nullptr, int nzero = 0) {
  __shared__ double sval[AAT_CAP];
  __shared__ int32_t scol[AAT_CAP];
  // (the flags of the factorization this band is made for: cleared here instead of by a memset)
  if (blockIdx.x == 0 && (int)threadIdx.x < nzero) zero[threadIdx.x] = 0;
  const int i0 = blockIdx.x * IPX_BLOCK, i = i0 + (int)threadIdx.x;
  const int e0 = rowptr[max(i0 - k, 0)], e1 = rowptr[min(i0 + IPX_BLOCK, m)];
  const bool staged = e1 - e0 <= AAT_CAP;                  // (uniform over the workgroup)
  if (staged) {
    for (int e = e0 + (int)threadIdx.x; e < e1; e += IPX_BLOCK) {
      scol[e - e0] = colidx[e];
      sval[e - e0] = val[e];
    }
  }
  __syncthreads();
  if (i >= m) return;
  const int p = rowptr[i], pe = rowptr[i + 1];
  for (int d = 0; d <= k; ++d) {
    double s = 0.0;
    if (i - d >= 0) {
      const int u = rowptr[i - d], ue = rowptr[i - d + 1];
      if (staged)
        s = aat_join(p - e0, pe - e0, u - e0, ue - e0, [&](int e) { return scol[e]; },
                     [&](int e) { return sval[e]; }, wcol);
      else
        s = aat_join(p, pe, u, ue, [&](int e) { return colidx[e]; },
                     [&](int e) { return val[e]; }, wcol);
    }
    band[(int64_t)d * m + i] = s;
  }
}


// ===================================================================== fast
// Three-launch solve.  The multi-launch sweep above pays one dependent global
// load per recurrence step (tens of microseconds at m = 1e5); here
//   k_down0   a workgroup stages the rows of its T chunks in LDS with coalesced
//             loads, each lane runs its chunk's LDL' recurrence out of LDS with
//             the multipliers prefetched in register blocks, and the two halves
//             of every reduced right-hand side entry are written out;
//   k_middle  ONE workgroup holds every remaining level in LDS and walks down
//             and back up with __syncthreads() instead of kernel boundaries;
//   k_correct the level-0 up sweep (elementwise, all CUs).
// Same arithmetic as the multi-launch path (explicit FMA in the recurrences: equal
// to ~1e-16).  When the separator system is numerically diagonal (k = 1, checked at
// every factorization) the whole solve is ONE launch, k_solve_decoupled, further
// down: no middle kernel, correction and residual out of LDS, optionally g = r - A'v
// as its tail.
constexpr int DOWN_T = 256;          // threads per workgroup in k_down0 (T <= 64 of them own a chunk)
constexpr int MID_T = 512;
constexpr int JB = 8;                // recurrence steps per register block
constexpr size_t LDS_LIMIT = 160 * 1024 - 512;

// Cooperative global -> LDS fill with U loads in flight per lane (a plain copy
// loop exposes one full memory latency per iteration).  `val(i)` produces
// element i (normally one global load).  Rounds of U predicated loads: an
// array of up to U * blockDim elements costs ONE memory latency.
template <int U, typename Val>
__device__ __forceinline__ void stage(double *dst, int n, Val val) {
  const int step = blockDim.x;
  for (int base = threadIdx.x; base < n; base += U * step) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * step;
      v[u] = i < n ? val(i) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * step;
      if (i < n) dst[i] = v[u];
    }
  }
}

// The same in two halves, so that several arrays can be in flight together:
// load() all of them, then store() all of them (one latency for the lot).
template <int U>
struct StageRegs {
  double v[U];
  template <typename Val>
  __device__ __forceinline__ void load(int n, Val val) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = threadIdx.x + u * blockDim.x;
      v[u] = i < n ? val(i) : 0.0;
    }
  }
  // elements past U * blockDim (unusual chunk sizes) take the slow road
  template <typename Val>
  __device__ __forceinline__ void store(double *dst, int n, Val val) const {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = threadIdx.x + u * blockDim.x;
      if (i < n) dst[i] = v[u];
    }
    for (int i = threadIdx.x + U * blockDim.x; i < n; i += blockDim.x) dst[i] = val(i);
  }
};

struct LevDev {
  int m, k, c, P, q, mR;
  const double *band, *Dinv, *L, *V, *W;
  int oB, oD, oL, oV, oW;     // offsets of the same tables inside the level slab (levels >= 1)
};
struct LevArgs {
  int nlev;
  LevDev lev[MAX_LEVELS];
};

// Dinv / L are indexed [step][column]; the column stride is CS when known at
// compile time (LDS-staged tables of k_down0) or P otherwise.  The tables have
// q + K zero-initialised rows and multipliers that would reach outside the
// chunk are stored as 0, so the recurrences need no bounds tests.  One lane
// issues every instruction of its chain, so the loops are written for
// instruction count: blocks of JB steps are loaded into registers (constant
// address offsets), the chain is one FMA per band entry, then the block is
// stored.  (FMA contraction is explicit here: the solve has no bitwise
// counterpart in the reference, unlike the SpMV / vector kernels.)
template <int K, int CS>
__device__ __forceinline__ void chunk_solve(double *__restrict__ b, int ct,
                                            const double *__restrict__ Dinv,
                                            const double *__restrict__ L, int P, int t) {
  const int cs = CS ? CS : P;
  const double *lp = L + t;
  const double *dp = Dinv + t;
  double zw[K];
#pragma unroll
  for (int r = 0; r < K; ++r) zw[r] = 0.0;
  // ---- forward: z_j = w_j - sum_d l_{j,j-d} z_{j-d}
  int j = 0;
  for (; j + JB <= ct; j += JB) {
    double wv[JB], lc[JB][K];
    const double *lj = lp + (int64_t)j * K * cs;
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      wv[jj] = b[j + jj];
#pragma unroll
      for (int d = 0; d < K; ++d) lc[jj][d] = lj[(jj * K + d) * cs];
    }
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      double z = wv[jj];
#pragma unroll
      for (int d = 0; d < K; ++d) z = __builtin_fma(-lc[jj][d], zw[d], z);
#pragma unroll
      for (int r = K - 1; r > 0; --r) zw[r] = zw[r - 1];
      zw[0] = z;
      wv[jj] = z;
    }
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) b[j + jj] = wv[jj];
  }
  for (; j < ct; ++j) {
    double z = b[j];
#pragma unroll
    for (int d = 0; d < K; ++d) z = __builtin_fma(-lp[((int64_t)j * K + d) * cs], zw[d], z);
#pragma unroll
    for (int r = K - 1; r > 0; --r) zw[r] = zw[r - 1];
    zw[0] = z;
    b[j] = z;
  }
  // ---- backward: y_j = z_j / d_j - sum_d l_{j+d,j} y_{j+d}
  double yw[K];
#pragma unroll
  for (int r = 0; r < K; ++r) yw[r] = 0.0;
  j = ct;
  for (; j - JB >= 0; j -= JB) {          // rows j-1 .. j-JB
    double wv[JB], lc[JB][K];
    const int jb = j - JB;
    const double *lj = lp + (int64_t)jb * K * cs;
    const double *dj = dp + (int64_t)jb * cs;
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {       // jj-th row of the block is jb + jj
      wv[jj] = b[jb + jj] * dj[jj * cs];
#pragma unroll
      for (int d = 1; d <= K; ++d) lc[jj][d - 1] = lj[((jj + d) * K + (d - 1)) * cs];
    }
#pragma unroll
    for (int jj = JB - 1; jj >= 0; --jj) {
      double v = wv[jj];
#pragma unroll
      for (int d = 0; d < K; ++d) v = __builtin_fma(-lc[jj][d], yw[d], v);
#pragma unroll
      for (int r = K - 1; r > 0; --r) yw[r] = yw[r - 1];
      yw[0] = v;
      wv[jj] = v;
    }
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) b[jb + jj] = wv[jj];
  }
  for (--j; j >= 0; --j) {
    double v = b[j] * dp[(int64_t)j * cs];
#pragma unroll
    for (int d = 1; d <= K; ++d)
      v = __builtin_fma(-lp[((int64_t)(j + d) * K + (d - 1)) * cs], yw[d - 1], v);
#pragma unroll
    for (int r = K - 1; r > 0; --r) yw[r] = yw[r - 1];
    yw[0] = v;
    b[j] = v;
  }
}

// reduced rhs halves for chunk t held at LDS address b (its separator rows,
// still holding w, follow the interior rows).  cE / cF are this chunk's K x K
// coupling blocks: cE[j*K+a] = E[j][a] (first K rows), cF[jj*K+a] = F[ct-K+jj][a].
template <int K>
__device__ __forceinline__ void chunk_rhs_halves(const double *b, int P, int t, int ct,
                                                 const double *cE, const double *cF,
                                                 double *gL, double *gR) {
  if (t < P - 1) {
#pragma unroll
    for (int a = 0; a < K; ++a) {
      double r = b[ct + a];
#pragma unroll
      for (int jj = 0; jj < K; ++jj)
        if (ct - K + jj >= 0) r -= cF[jj * K + a] * b[ct - K + jj];
      gL[t * K + a] = r;
    }
  }
  if (t > 0) {
#pragma unroll
    for (int a = 0; a < K; ++a) {
      double r = 0.0;
#pragma unroll
      for (int j = 0; j < K; ++j)
        if (j < ct) r -= cE[j * K + a] * b[j];
      gR[(t - 1) * K + a] = r;     // added to gL by the consumer: g = gL + gR
    }
  }
}

template <int K, int T>
__global__ void __launch_bounds__(DOWN_T)
k_down0(LevDev lv, const double *w, double *y, double *__restrict__ gL,
        double *__restrict__ gR, const double *__restrict__ guard) {
  extern __shared__ double sm[];
  const double stop = guard ? *guard : 0.0;   // requested with the staging loads, tested after
  const int q = lv.q, P = lv.P, m = lv.m, c = lv.c;
  const int qk = q + K;                  // table rows (K zero rows of padding)
  const int t0 = blockIdx.x * T;
  const int tcount = min(T, P - t0);
  const int row0 = t0 * q;
  const int rend = min(m, (t0 + tcount) * q);
  const int nrows = rend - row0;
  double *sw = sm;                       // T*q      rows of this workgroup's chunks
  double *sD = sw + (size_t)T * q;       // qk*T     1/d   [step][lane]
  double *sL = sD + (size_t)qk * T;      // qk*K*T   l     [step*K+d][lane]
  double *sE = sL + (size_t)qk * K * T;  // T*K*K    coupling to the left separator
  double *sF = sE + (size_t)T * K * K;   // T*K*K    coupling to the right separator
  // One staging phase: every global load of the kernel is issued before the
  // first LDS store (one memory latency for the lot).
  const double *wp = w + row0;
  // columns t0 .. t0+tcount-1 of the [step][chunk] tables (lanes past the last
  // chunk re-read column P-1; never used)
  const double *Dg = lv.Dinv, *Lg = lv.L, *Bg = lv.band;
  auto f_w = [=](int i) { return wp[i]; };
  auto f_D = [=](int i) {
    const int j = i / T, tl = i - j * T;
    return Dg[(int64_t)j * P + min(t0 + tl, P - 1)];
  };
  auto f_L = [=](int i) {
    const int j = i / T, tl = i - j * T;
    return Lg[(int64_t)j * P + min(t0 + tl, P - 1)];
  };
  auto f_E = [=](int i) {
    const int tl = i / (K * K), r = i - tl * K * K, j = r / K, a = r - j * K;
    const int t = min(t0 + tl, P - 1);
    return t > 0 ? coupE<K>(Bg, m, t * q, j, a) : 0.0;
  };
  auto f_F = [=](int i) {
    const int tl = i / (K * K), r = i - tl * K * K, jj = r / K, a = r - jj * K;
    const int t = min(t0 + tl, P - 1);
    const int ct = chunk_rows(m, q, c, P, t);
    return (t < P - 1 && ct - K + jj >= 0) ? coupF<K>(Bg, m, t * q, ct, ct - K + jj, a) : 0.0;
  };
  // register budgets sized for 64-row chunks with T = 16; anything larger
  // takes StageRegs::store's slow loop
  StageRegs<5> g_w, g_D;
  StageRegs<5 * K> g_L;
  StageRegs<(T * K * K + DOWN_T - 1) / DOWN_T> g_E, g_F;
  g_w.load(nrows, f_w);          g_D.load(qk * T, f_D);      g_L.load(qk * K * T, f_L);
  g_E.load(T * K * K, f_E);      g_F.load(T * K * K, f_F);
  if (stop != 0.0) return;
  g_w.store(sw, nrows, f_w);     g_D.store(sD, qk * T, f_D); g_L.store(sL, qk * K * T, f_L);
  g_E.store(sE, T * K * K, f_E); g_F.store(sF, T * K * K, f_F);
  __syncthreads();
  const int tl = threadIdx.x, t = t0 + tl;
  if (tl < tcount) {
    const int ct = chunk_rows(m, q, c, P, t);
    double *b = sw + tl * q;
    chunk_solve<K, T>(b, ct, sD, sL, T, tl);
    chunk_rhs_halves<K>(b, P, t, ct, sE + tl * K * K, sF + tl * K * K, gL, gR);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nrows; i += blockDim.x) y[row0 + i] = sw[i];
}

// LDS-staged factorization of T chunks per workgroup (the one-lane-per-chunk
// kernel k_factor_chunks pays a global-memory round trip per recurrence step:
// ~120 us at m = 1e5).  Band rows are staged with coalesced loads; one lane per
// chunk runs the LDL' recurrence out of LDS into [step][chunk] tables kept in
// LDS; then 2K lanes per chunk compute the spike columns B^-1 E / B^-1 F with
// the same chunk_solve the solve path uses; a cooperative sweep writes the
// tables and the spikes out.
template <int K>
constexpr int factor_chunks_per_wg() { return K <= 2 ? 16 : (K <= 4 ? 8 : 4); }

template <int K>
size_t factor_lds_doubles(int q) {
  constexpr int T = factor_chunks_per_wg<K>();
  const size_t qk = q + K;
  return (size_t)(K + 1) * T * q + qk * T + qk * K * T + (size_t)2 * K * T * q;
}

template <int K, int T>
__global__ void __launch_bounds__(DOWN_T)
k_factor_lds(int m, int c, int P, const double *__restrict__ band, double *__restrict__ Dinv,
             double *__restrict__ L, double *__restrict__ V, double *__restrict__ W, int *flag) {
  static_assert(2 * K * T <= DOWN_T, "one lane per spike column");
  extern __shared__ double sm[];
  const int q = c + K, qk = q + K, TQ = T * q;
  const int t0 = blockIdx.x * T;
  const int tcount = min(T, P - t0);
  const int row0 = t0 * q;
  const int nrows = min(m, (t0 + tcount) * q) - row0;
  double *sB = sm;                             // (K+1) x TQ   band rows  [d][local row]
  double *sD = sB + (size_t)(K + 1) * TQ;      // qk*T         1/d        [step][lane]
  double *sL = sD + (size_t)qk * T;            // qk*K*T       l          [step*K+d][lane]
  double *sX = sL + (size_t)qk * K * T;        // 2K*T x q     spike columns [task][row]
  stage<16>(sB, (K + 1) * TQ, [=](int i) {
    const int d = i / TQ, r = i - d * TQ;
    return r < nrows ? band[(int64_t)d * m + row0 + r] : 0.0;
  });
  for (int i = threadIdx.x; i < qk * T * (K + 1) + 2 * K * TQ; i += blockDim.x) sD[i] = 0.0;
  __syncthreads();

  // ---- LDL' of the interior blocks, one lane per chunk (k_factor_chunks' recurrence)
  if ((int)threadIdx.x < tcount) {
    const int tl = threadIdx.x, t = t0 + tl;
    const int ct = chunk_rows(m, q, c, P, t);
    const double *bp = sB + tl * q;
    double lw[K][K], dw[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
      dw[r] = 1.0;
#pragma unroll
      for (int d = 0; d < K; ++d) lw[r][d] = 0.0;
    }
    bool bad = false, hard = false;
    for (int j = 0; j < ct; ++j) {
      double lrow[K];
#pragma unroll
      for (int d = K; d >= 1; --d) {
        double sv = 0.0;
        if (d <= j) {
          sv = bp[d * TQ + j];
#pragma unroll
          for (int e = K; e > d; --e)
            if (e <= j) sv -= lrow[e - 1] * dw[e - 1] * lw[d - 1][e - d - 1];
          sv = sv / dw[d - 1];
        }
        lrow[d - 1] = sv;
      }
      double dj = bp[j];
#pragma unroll
      for (int e = K; e >= 1; --e)
        if (e <= j) dj -= lrow[e - 1] * lrow[e - 1] * dw[e - 1];
      bad |= !(dj > IPX_PIVOT_RTOL * bp[j]);
      hard |= !(dj > 0.0);
      sD[j * T + tl] = 1.0 / dj;
#pragma unroll
      for (int d = 1; d <= K; ++d) sL[(j * K + (d - 1)) * T + tl] = lrow[d - 1];
#pragma unroll
      for (int r = K - 1; r > 0; --r) {
        dw[r] = dw[r - 1];
#pragma unroll
        for (int d = 0; d < K; ++d) lw[r][d] = lw[r - 1][d];
      }
      dw[0] = dj;
#pragma unroll
      for (int d = 0; d < K; ++d) lw[0][d] = lrow[d];
    }
    if (bad) atomicOr(flag, hard ? 5 : 1);
  }
  __syncthreads();

  // ---- spikes: lane (chunk, side, a) solves B x = E[:, a] (side 0) or F[:, a] (side 1)
  if ((int)threadIdx.x < 2 * K * tcount) {
    const int id = threadIdx.x;
    const int tl = id / (2 * K), rem = id - tl * 2 * K, side = rem / K, a = rem - side * K;
    const int t = t0 + tl;
    if (!(side == 0 && t == 0) && !(side == 1 && t == P - 1)) {
      const int ct = chunk_rows(m, q, c, P, t);
      const double *bp = sB + tl * q;
      double *x = sX + (size_t)id * q;
      if (side == 0) {           // E[j][a] = T[base+j][base-K+a], d = j+K-a, rows j <= a
        for (int j = 0; j <= a && j < ct; ++j) x[j] = bp[(j + K - a) * TQ + j];
      } else {                   // F[j][a] = T[base+ct+a][base+j], d = ct+a-j <= K
        for (int j = max(0, ct + a - K); j < ct; ++j) x[j] = bp[(ct + a - j) * TQ + ct + a];
      }
      chunk_solve<K, T>(x, ct, sD, sL, T, tl);
    }
  }
  __syncthreads();

  // ---- write-out (tables are [step][chunk] with q+K rows; rows past the chunk stay 0)
  for (int i = threadIdx.x; i < qk * T; i += blockDim.x) {
    const int j = i / T, tl = i - j * T;
    if (tl < tcount) Dinv[(int64_t)j * P + t0 + tl] = sD[i];
  }
  for (int i = threadIdx.x; i < qk * K * T; i += blockDim.x) {
    const int j = i / T, tl = i - j * T;
    if (tl < tcount) L[(int64_t)j * P + t0 + tl] = sL[i];
  }
  for (int i = threadIdx.x; i < nrows * K; i += blockDim.x) {
    const int r = i / K, a = i - r * K;
    const int tl = r / q, j = r - tl * q, t = t0 + tl;
    if (j < chunk_rows(m, q, c, P, t)) {
      const double *xv = sX + (size_t)(tl * 2 * K + a) * q + j;
      if (t > 0) V[(int64_t)(row0 + r) * K + a] = xv[0];
      if (t < P - 1) W[(int64_t)(row0 + r) * K + a] = xv[(size_t)K * q];
    }
  }
}

// Factor data of the levels >= 1 lives in one contiguous slab; k_middle copies
// the whole slab into LDS with a single coalesced sweep (one memory latency),
// after which every phase below touches LDS only.  `fb` is the slab base the
// level tables are addressed from: the LDS copy (STAGED) or global memory.
template <int K, int DEPTH>
struct Mid {
  // buf holds this level's right-hand side; on return it holds the solution.
  // (DEPTH is the level index: static indexing keeps the descriptors in SGPRs)
  static __device__ __forceinline__ void run(const LevArgs &a, double *buf, const double *fb) {
    const LevDev &lv = a.lev[DEPTH];
    double *nbuf = buf + lv.m;               // next level lives right behind
    const double *Dp = fb + lv.oD, *Lp = fb + lv.oL, *Vp = fb + lv.oV, *Wp = fb + lv.oW;
    const double *Bp = fb + lv.oB;
    for (int t = threadIdx.x; t < lv.P; t += blockDim.x) {
      const int ct = chunk_rows(lv.m, lv.q, lv.c, lv.P, t);
      chunk_solve<K, 0>(buf + t * lv.q, ct, Dp, Lp, lv.P, t);
    }
    __syncthreads();
    if (lv.mR > 0) {
      for (int qrow = threadIdx.x; qrow < lv.mR; qrow += blockDim.x) {
        const int t = qrow / K, a2 = qrow % K;
        const int base = t * lv.q, ct = lv.c;
        double r = buf[base + ct + a2];
        for (int j = ct - K; j < ct; ++j)
          if (j >= 0) r -= coupF<K>(Bp, lv.m, base, ct, j, a2) * buf[base + j];
        const int base2 = (t + 1) * lv.q;
        const int ct2 = chunk_rows(lv.m, lv.q, lv.c, lv.P, t + 1);
        for (int j = 0; j < K && j < ct2; ++j)
          r -= coupE<K>(Bp, lv.m, base2, j, a2) * buf[base2 + j];
        nbuf[qrow] = r;
      }
      __syncthreads();
      if constexpr (2 * K - 1 <= KMAX && DEPTH + 1 < MAX_LEVELS)
        Mid<2 * K - 1, DEPTH + 1>::run(a, nbuf, fb);
      __syncthreads();
      for (int i = threadIdx.x; i < lv.m; i += blockDim.x) {
        int t = i / lv.q;
        if (t > lv.P - 1) t = lv.P - 1;
        const int j = i - t * lv.q;
        const int ct = chunk_rows(lv.m, lv.q, lv.c, lv.P, t);
        if (j >= ct) { buf[i] = nbuf[t * K + (j - ct)]; continue; }
        double v = buf[i];
        if (t > 0) {
#pragma unroll
          for (int a2 = 0; a2 < K; ++a2) v -= Vp[(int64_t)i * K + a2] * nbuf[(t - 1) * K + a2];
        }
        if (t < lv.P - 1) {
#pragma unroll
          for (int a2 = 0; a2 < K; ++a2) v -= Wp[(int64_t)i * K + a2] * nbuf[t * K + a2];
        }
        buf[i] = v;
      }
      __syncthreads();
    }
  }
};

template <int K1, bool STAGED>
__global__ void __launch_bounds__(MID_T)
k_middle(LevArgs a, int nbuf_total, const double *__restrict__ slab, int nslab,
         const double *__restrict__ gL, const double *__restrict__ gR, double *__restrict__ xs,
         const double *__restrict__ guard) {
  extern __shared__ double sm[];
  if (guard && *guard != 0.0) return;
  const int m1 = a.lev[1].m;
  stage<8>(sm, m1, [=](int i) { return gL[i] + gR[i]; });
  if constexpr (STAGED) {
    double *coef = sm + nbuf_total;
    stage<16>(coef, nslab, [=](int i) { return slab[i]; });
    __syncthreads();
    Mid<K1, 1>::run(a, sm, coef);
  } else {
    __syncthreads();
    Mid<K1, 1>::run(a, sm, slab);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < m1; i += blockDim.x) xs[i] = sm[i];
}

constexpr int DOWN_CHUNKS = 16;      // chunks (recurrence lanes) per workgroup in k_down0
// own chunks per workgroup in k_solve_decoupled / k_solve_pcr.  Not a tunable: the fused
// kernels' window tables and the sharded layout's row blocks (ipsolver/sharded.py ROW_BLOCK)
// are built for 4 (a build with 2 faults in the headline loop)
constexpr int DEC_CHUNKS = 4;

IPX_STAMP_DECL(ipx_dbg_stamps);
#define IPX_STAMP(k) IPX_STAMP_TO(ipx_dbg_stamps, k)

// Decoupled path in ONE launch.  When the separator system is diagonal
// (SepValues above) a separator value needs only the two chunks next to it, so
// a workgroup that also solves one chunk to the left and two to the right of
// its T own chunks (the recurrence lanes are otherwise idle) has everything
// for the corrected solution of its rows AND their normal-equation residual:
// no y / gL / gR round trip through memory and no second kernel.
//   local chunk l = 0 .. T+2  <->  global chunk t = t0 - 1 + l;  own: l = 1 .. T
// Optional tail of k_solve_decoupled: g = r - A'v on the variables this workgroup's
// constraint rows are the first to touch (tridiagonal A A' <=> a variable sees at most
// two constraints, and they are adjacent, so every v it needs is in this workgroup's
// LDS).  A' in CSR; vown[b] .. vown[b+1] = variables of workgroup b; part3 = per-
// workgroup partials of ||g||^2 (second half zero, like a rectangular SpMV's).
struct AtvJob {
  const int32_t *rowptr, *colidx;
  const double *val;
  const double *r_in;
  double *g_out;
  const int32_t *vown;
  double *part3;
  // the same rows of A' in ELL(2) form for a tridiagonal A A' (a variable sees at most two
  // constraints, adjacent rows): ell_val[t * n + j], t = 0 the first constraint of variable j,
  // t = 1 the row after it (absent: 0); ell_row[j] = the first constraint's row as an offset
  // from the first row of the workgroup that owns j.  Indexed by the variable alone: no
  // row-pointer round trip; 18 bytes per variable.
  const uint16_t *ell_row;
  const double *ell_val;
  int64_t ell_n;
  // +1: g = r_in - A'v;  -1: g = A'v - r_in, the negated projection by the same roundings (the
  // priming's first direction p = -Z r: csrc/cg.hip prime_project)
  double sign = 1.0;
};

// The right-hand side formed inside k_solve_pcr (barrier problems: w = A_R u, csrc/boxschur.hip)
// instead of by an SpMV launch of its own.  Needs rows of ONE length 2^logL <= 16 (entry j of
// row i at [i << logL | j]: no row pointers, so the products' loads depend on nothing); a
// workgroup forms the rows of its whole window (own rows + 2^L either side: 1.5x the products,
// which is what one launch, its tile-table round trip and the w round trip through HBM cost
// less than).  Row sums left to right like every SpMV of the library: same bits as k_csr_spmv.
struct RowsJob {
  const int32_t *col;
  const double *val;
  const double *x;
  int logL;
};
constexpr int ROWS_PRODUCTS = 8192;     // products per workgroup: a window of <= 512 rows of <= 16 entries

// Optional tail of the ROWS form (round 5; the barrier problem's projection, csrc/boxschur.hip):
// the per-item back substitution of k_pairs_post -- y_c = (A_R' v_R)_c from the column's two
// entries, v_S of the item's group, g on its columns, ||g||^2 partials -- done by the workgroup
// whose rows are the first to touch the item's column, with v_R out of LDS: one launch and one
// round trip of v_R through memory less per CG iteration.  Items in the order of
// ipx_boxschur_args (the ng groups, then the ngen other columns), tables in their compact
// forms only (grp2 + computed columns + item-indexed ELL(2) columns of A_R): own_g / own_e =
// per workgroup the range of groups / of other columns it owns (non-decreasing first rows: the
// host checks).  part: `count` partials per half -- workgroup w writes entry w and zeroes the
// entries w + k gridDim beyond the grid, so a consumer may fold all `count` of them.
constexpr int POST_PG = 6;              // group items per lane (<= 6 * 512 per workgroup)
struct PostJob {
  const int32_t *own_g, *own_e;
  int ng, nitems;
  ipx_group_tab T;
  const int32_t *yrow;
  const double *yval;
  const double *r;
  double *g;
  double *part;
  int count;
};

template <int K, int T, int QV>
__global__ void __launch_bounds__(DOWN_T)
k_solve_decoupled(LevDev lv, const double *__restrict__ w, double *__restrict__ x,
                  const double *__restrict__ rinv, double *__restrict__ partial,
                  const double *__restrict__ guard, AtvJob atv) {
  constexpr int NCH = T + 3;
  constexpr int QA = QV > 0 ? QV : 1;
  extern __shared__ double sm[];
  __shared__ double red_lds[DOWN_T / IPX_WAVE];
  IPX_STAMP(0);
  const double stop = guard ? *guard : 0.0;   // requested with the staging loads, tested after
  const int q = lv.q, P = lv.P, m = lv.m, c = lv.c;
  const int qk = q + K;
  const int t0 = blockIdx.x * T, tfirst = t0 - 1;
  const int64_t rfirst = (int64_t)tfirst * q;          // global row of local row 0 (may be < 0)
  const int NB = T * q + K;                            // band rows kept for the residual
  double *sw = sm;                          // NCH*q     rows: w, then y
  double *sx = sw + (size_t)NCH * q;        // NCH*q     corrected solution
  double *sD = sx + (size_t)NCH * q;        // qk*NCH    1/d   [step][l]
  double *sL = sD + (size_t)qk * NCH;       // qk*K*NCH  l     [step*K+d][l]
  double *sE = sL + (size_t)qk * K * NCH;   // NCH*K*K
  double *sF = sE + (size_t)NCH * K * K;    // NCH*K*K
  double *sB = sF + (size_t)NCH * K * K;    // (K+1)*NB  band rows of the own rows (+K)
  double *sgL = sB + (size_t)(K + 1) * NB;  // NCH*K
  double *sgR = sgL + (size_t)NCH * K + K;  // NCH*K, K entries of slack in front
  double *sxs = sgR + (size_t)NCH * K;      // NCH*K     separator values
  double *srinv = sxs + (size_t)NCH * K;    // NCH*K*K   inverse of the separators' diagonal blocks
  double *sw0 = srinv + (size_t)NCH * K * K;   // T*q    w of the own rows (sw turns into y)
  const int NV = (T + 2) * q - (q - K);     // rows whose corrected value is needed
  double *sV = sw0 + (size_t)T * q;         // NV*K      spikes of those rows
  double *sW = sV + (size_t)NV * K;         // NV*K
  const double *Dg = lv.Dinv, *Lg = lv.L, *Bg = lv.band, *Vg = lv.V, *Wg = lv.W;
  // every global load of the kernel is issued in this one staging phase
  auto f_w = [=](int i) {
    const int64_t g = rfirst + i;
    return (g >= 0 && g < m) ? w[g] : 0.0;
  };
  auto f_D = [=](int i) {
    const int j = i / NCH, l = i - j * NCH;
    return Dg[(int64_t)j * P + min(max(tfirst + l, 0), P - 1)];
  };
  auto f_L = [=](int i) {
    const int j = i / NCH, l = i - j * NCH;
    return Lg[(int64_t)j * P + min(max(tfirst + l, 0), P - 1)];
  };
  auto f_E = [=](int i) {
    const int l = i / (K * K), r = i - l * K * K, j = r / K, a = r - j * K;
    const int t = tfirst + l;
    return (t > 0 && t < P) ? coupE<K>(Bg, m, t * q, j, a) : 0.0;
  };
  auto f_F = [=](int i) {
    const int l = i / (K * K), r = i - l * K * K, jj = r / K, a = r - jj * K;
    const int t = tfirst + l;
    if (t < 0 || t >= P - 1) return 0.0;
    const int ct = chunk_rows(m, q, c, P, t);
    return ct - K + jj >= 0 ? coupF<K>(Bg, m, t * q, ct, ct - K + jj, a) : 0.0;
  };
  auto f_B = [=](int i) {
    const int d = i / NB, r = i - d * NB;
    const int64_t g = (int64_t)t0 * q + r;
    return g < m ? Bg[(int64_t)d * m + g] : 0.0;
  };
  auto f_w0 = [=](int i) {
    const int64_t g = (int64_t)t0 * q + i;
    return g < m ? w[g] : 0.0;
  };
  auto f_V = [=](int i) {
    const int64_t gi = (rfirst + q - K) * K + i;         // element index in the m x K spikes
    return (gi >= 0 && gi < (int64_t)m * K) ? Vg[gi] : 0.0;
  };
  auto f_W = [=](int i) {
    const int64_t gi = (rfirst + q - K) * K + i;
    return (gi >= 0 && gi < (int64_t)m * K) ? Wg[gi] : 0.0;
  };
  auto f_r = [=](int i) {
    const int l = i / (K * K), t = tfirst + l;
    return (t >= 0 && t < P - 1) ? rinv[(int64_t)t * K * K + (i - l * K * K)] : 0.0;
  };
  // register budgets sized for the default 64-row chunks (q <= QD); larger
  // arrays spill into StageRegs::store's slow loop
  constexpr int QD = 66 + K;
  constexpr int UW = (NCH * QD + DOWN_T - 1) / DOWN_T;              // rows incl. halo chunks
  constexpr int UT = (NCH * (QD + K) + DOWN_T - 1) / DOWN_T;        // one table column set
  constexpr int UO = (T * QD + K + DOWN_T - 1) / DOWN_T;            // own rows
  constexpr int US = (NCH * K * K + DOWN_T - 1) / DOWN_T;
  StageRegs<UW> g_w;
  StageRegs<UT> g_D;
  StageRegs<UO> g_w0;
  StageRegs<UT * K> g_L;
  StageRegs<UW * K> g_V, g_W;
  StageRegs<UO *(K + 1)> g_B;
  StageRegs<US> g_E, g_F, g_r;                 // K*K entries per chunk / separator each
  g_w.load(NCH * q, f_w);       g_D.load(qk * NCH, f_D);     g_L.load(qk * K * NCH, f_L);
  g_E.load(NCH * K * K, f_E);   g_F.load(NCH * K * K, f_F);  g_B.load((K + 1) * NB, f_B);
  g_w0.load(T * q, f_w0);       g_V.load(NV * K, f_V);       g_W.load(NV * K, f_W);
  g_r.load((NCH - 1) * K * K, f_r);
  // A'v tail: row pointers of this workgroup's variables travel with the staging loads
  // (waves 1..3 only: wave 0 owns the chunk recurrences and must not queue behind
  // the tail's load issue; ATV_T lanes share the workgroup's variables)
  constexpr int ATV_T = DOWN_T - IPX_WAVE;
  const int ta = (int)threadIdx.x - IPX_WAVE;
  int av0 = 0, avn = 0, ra[QA], rb[QA];
  if (QV > 0 && ta >= 0) {
    av0 = atv.vown[blockIdx.x];
    avn = atv.vown[blockIdx.x + 1] - av0;
#pragma unroll
    for (int k = 0; k < QA; ++k) {
      const int j = av0 + min(ta + k * ATV_T, max(avn - 1, 0));
      ra[k] = atv.rowptr[j];
      rb[k] = atv.rowptr[j + 1];
    }
  }
  if (stop != 0.0) return;
  IPX_STAMP(1);
  g_w.store(sw, NCH * q, f_w);       g_D.store(sD, qk * NCH, f_D);
  g_L.store(sL, qk * K * NCH, f_L);  g_E.store(sE, NCH * K * K, f_E);
  g_F.store(sF, NCH * K * K, f_F);   g_B.store(sB, (K + 1) * NB, f_B);
  g_w0.store(sw0, T * q, f_w0);      g_V.store(sV, NV * K, f_V);
  g_W.store(sW, NV * K, f_W);        g_r.store(srinv, (NCH - 1) * K * K, f_r);
  __syncthreads();
  IPX_STAMP(2);
  // A'v tail: the (at most two) entries of every row and r, requested now so that they
  // arrive while the chunk recurrences run
  int ac0[QA], ac1[QA];
  double aw0[QA], aw1[QA], ar[QA];
  if (QV > 0 && ta >= 0) {
#pragma unroll
    for (int k = 0; k < QA; ++k) {
      const int j = av0 + min(ta + k * ATV_T, max(avn - 1, 0));
      if (K == 1) {
        // (clamped: an empty row re-reads a neighbouring entry, never out of bounds)
        const int e0 = max(min(ra[k], rb[k] - 1), 0), e1 = max(rb[k] - 1, e0);
        ac0[k] = atv.colidx[e0];
        ac1[k] = atv.colidx[e1];
        aw0[k] = atv.val[e0];
        aw1[k] = atv.val[e1];
      }
      ar[k] = atv.r_in[j];
    }
  }

  // ---- chunk recurrences + the two halves of every separator's reduced rhs
  if ((int)threadIdx.x < NCH) {
    const int l = threadIdx.x, t = tfirst + l;
    if (t >= 0 && t < P) {
      const int ct = chunk_rows(m, q, c, P, t);
      double *b = sw + l * q;
      chunk_solve<K, NCH>(b, ct, sD, sL, NCH, l);
      chunk_rhs_halves<K>(b, P, t, ct, sE + l * K * K, sF + l * K * K, sgL - (int64_t)tfirst * K,
                          sgR - (int64_t)tfirst * K);
    }
  }
  IPX_STAMP(3);
  __syncthreads();
  IPX_STAMP(4);
  // ---- separator values  xs_t = (gL_t + gR_t) / R_tt   for t = t0-1 .. t0+T
  for (int i = threadIdx.x; i < (NCH - 1) * K; i += blockDim.x) {
    const int l = i / K, t = tfirst + l;
    double v = 0.0;
    if (t >= 0 && t < P - 1) {
      const int a = i - l * K;
#pragma unroll
      for (int b = 0; b < K; ++b)
        v += srinv[(l * K + a) * K + b] * (sgL[l * K + b] + sgR[l * K + b]);
    }
    sxs[i] = v;
  }
  __syncthreads();
  IPX_STAMP(5);
  // ---- corrected solution on the own rows and the K rows either side of them
  for (int li = q - K + (int)threadIdx.x; li < (T + 2) * q; li += blockDim.x) {
    const int l = li / q, j = li - l * q, t = tfirst + l;
    const int64_t g = rfirst + li;
    double v = 0.0;
    if (t >= 0 && t < P && g < m) {
      const int ct = chunk_rows(m, q, c, P, t);
      if (j >= ct) {
        v = sxs[l * K + (j - ct)];
      } else {
        v = sw[li];
        if (t > 0) {
#pragma unroll
          for (int a = 0; a < K; ++a) v -= sV[(li - (q - K)) * K + a] * sxs[(l - 1) * K + a];
        }
        if (t < P - 1) {
#pragma unroll
          for (int a = 0; a < K; ++a) v -= sW[(li - (q - K)) * K + a] * sxs[l * K + a];
        }
      }
      if (l >= 1 && l <= T) x[g] = v;
    }
    sx[li] = v;
  }
  if (!partial && QV == 0) return;
  __syncthreads();
  IPX_STAMP(6);
  if (QV > 0) {
    // ---- g = r - A'v on this workgroup's variables, v out of LDS (sx)
    double gacc = 0.0;
#pragma unroll
    for (int k = 0; k < QA; ++k) {
      const int jl = ta + k * ATV_T;
      if (ta >= 0 && jl < avn) {
        const int len = rb[k] - ra[k];
        double sum = 0.0;
        if (K == 1) {
          if (len > 0) sum += aw0[k] * sx[ac0[k] - rfirst];
          if (len > 1) sum += aw1[k] * sx[ac1[k] - rfirst];
        } else {
          // half bandwidth K: up to K + 1 constraints per variable, all within K rows of the
          // first (which lies in this workgroup's rows): read here, not held in registers
          for (int e = ra[k]; e < rb[k]; ++e) sum += atv.val[e] * sx[atv.colidx[e] - rfirst];
        }
        double y = -atv.sign * sum;
        y += atv.sign * ar[k];
        atv.g_out[av0 + jl] = y;
        gacc += y * y;
      }
    }
    const double gtot = ipx_block_reduce<IPX_SUM>(gacc, red_lds);
    if (threadIdx.x == 0) {
      atv.part3[blockIdx.x] = gtot;
      atv.part3[gridDim.x + blockIdx.x] = 0.0;
    }
    if (!partial) return;
  }
  // ---- residual of the own rows out of LDS:  w_i - sum_d S[i][i+-d] x[i+-d]
  double acc = 0.0;
  for (int r = threadIdx.x; r < T * q; r += blockDim.x) {
    const int64_t g = (int64_t)t0 * q + r;
    if (g < m) {
      const int li = q + r;
      double sum = sB[r] * sx[li];
#pragma unroll
      for (int d = 1; d <= K; ++d) {
        if (g - d >= 0) sum += sB[d * NB + r] * sx[li - d];
        if (g + d < m) sum += sB[d * NB + r + d] * sx[li + d];
      }
      const double res = sw0[r] - sum;
      acc += res * res;
    }
  }
  const double tot = ipx_block_reduce<IPX_SUM>(acc, red_lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
  IPX_STAMP(7);
}


// ---------------------------------------------------------------- PCR (k = 1)
// Tridiagonal S = A A' of the banded benchmark: instead of one lane sweeping a 64-row chunk
// forwards and backwards (130 dependent steps), the whole workgroup reduces its window of
// rows by PARALLEL CYCLIC REDUCTION: at level s every row eliminates its neighbours at
// distance 2^s,
//     alpha = -a_i / b_{i-h},  gamma = -c_i / b_{i+h}
//     a_i <- alpha a_{i-h},  c_i <- gamma c_{i+h},
//     b_i <- b_i + alpha c_{i-h} + gamma a_{i+h},  d_i <- d_i + alpha d_{i-h} + gamma d_{i+h},
// all rows at once out of LDS (ping-pong buffers, one LDS barrier per level).  Because the
// inverse of S decays geometrically, after L levels (L <= 7, found at every factorization by
// running the reduction on the matrix alone, k_pcr_check) the remaining couplings are
// below 2^-56 of the diagonal: x_i = d_i / b_i, and a window of the own rows plus 2^L rows on
// either side gives the own rows exactly -- the same decay the chunk decoupling relies on.
// No factor tables are read at all (the band is 2 doubles per row); same rows per workgroup
// as k_solve_decoupled (DEC_CHUNKS * q), so the residual / A'v tail bookkeeping is unchanged.
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int PCR_RMAX = 4 * 66 + 2 * (1 << PCR_LMAX) + 8;        // window rows (q <= 66)
constexpr int PCR_NR = (PCR_RMAX + IPX_BLOCK - 1) / IPX_BLOCK;    // rows per lane (k_pcr_check)
constexpr int PCR_TB = 512;                                       // lanes per workgroup of k_solve_pcr

// The matrix is symmetric: only the sub-diagonal a_i = S[i][i-h] is carried (c_i = a_{i+h}
// is read from the neighbour), with the reciprocal of the diagonal next to it so that a level
// costs one v_rcp_f64 + one Newton step per row instead of two IEEE divisions:
//     al = -a_i r_{i-h},  ga = -a_{i+h} r_{i+h}           (r = 1/b)
//     a_i <- al a_{i-h},  b_i <- b_i + al a_i + ga a_{i+h},  d_i <- d_i + al d_{i-h} + ga d_{i+h}
// Rows outside the window (or the matrix) are identity rows with a = 0: no bounds branches.
__device__ __forceinline__ double pcr_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  return __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);      // one Newton step: full precision
}

// Factor-time check in ONE launch: the reduction of the matrix alone (no right-hand side) on
// the same windows as the solve, all PCR_LMAX levels; a workgroup tests its own rows after
// every level: flags[s + 1] = a coupling at distance 2^s is still above 2^-56 of the diagonal,
// flags[PCR_LMAX + 1 + s + 1] = a reduced diagonal entry is not positive.  Plain stores of the
// constant 1, one word per (level, finding): every workgroup raises the first kind below the
// decoupling level, and atomics on ONE word serialise in L2 at ~45 ns each -- 71 us of this
// kernel's 78 at m = 1e5 (1540 waves x 7 levels), half of a refactorization.
__global__ void __launch_bounds__(IPX_BLOCK)
k_pcr_check(int m, int rows_wg, const double *__restrict__ band, int *flags) {
  constexpr int PAD = 1 << PCR_LMAX;
  constexpr int RS = PCR_RMAX + 2 * PAD;
  __shared__ double pa[2][RS], pr[2][RS];
  const int H = 1 << PCR_LMAX;
  const int R = rows_wg + 2 * H;
  const int64_t g0 = (int64_t)blockIdx.x * rows_wg - H;
  const int tid = threadIdx.x;
  double a[PCR_NR], b[PCR_NR], b0[PCR_NR];
  bool own[PCR_NR];
#pragma unroll
  for (int k = 0; k < PCR_NR; ++k) {
    const int r = tid + k * IPX_BLOCK;
    const int64_t g = g0 + r;
    const bool in = r < R && g >= 0 && g < m;
    const int64_t gc = min(max(g, (int64_t)0), (int64_t)m - 1);
    const double bv = band[gc], av = band[(int64_t)m + gc];
    a[k] = (in && g >= 1 && r >= 1) ? av : 0.0;
    b[k] = in ? bv : 1.0;
    b0[k] = b[k];
    own[k] = in && r >= H && r < H + rows_wg;
  }
  for (int i = tid; i < 2 * PAD; i += IPX_BLOCK) {
    const int r = i < PAD ? i : R + i;
#pragma unroll
    for (int u = 0; u < 2; ++u) { pa[u][r] = 0.0; pr[u][r] = 1.0; }
  }
  const double tiny = 1.3877787807814457e-17;                     // 2^-56
  for (int s = 0; s < PCR_LMAX; ++s) {
    const int h = 1 << s, cur = s & 1;
#pragma unroll
    for (int k = 0; k < PCR_NR; ++k) {
      const int r = tid + k * IPX_BLOCK;
      if (r < R) { pa[cur][PAD + r] = a[k]; pr[cur][PAD + r] = pcr_rcp(b[k]); }
    }
    ipx_lds_barrier();
    int bits = 0;
#pragma unroll
    for (int k = 0; k < PCR_NR; ++k) {
      const int r = PAD + min(tid + k * IPX_BLOCK, R - 1);
      const double alo = pa[cur][r - h], rlo = pr[cur][r - h];
      const double ahi = pa[cur][r + h], rhi = pr[cur][r + h];
      const double al = -a[k] * rlo, ga = -ahi * rhi;
      double bn = __builtin_fma(al, a[k], b[k]);
      bn = __builtin_fma(ga, ahi, bn);
      a[k] = al * alo; b[k] = bn;
      if (own[k]) {
        if (!(fabs(a[k]) <= tiny * bn)) bits |= 1;
        if (!(bn > 0.0)) bits |= 2;
        // a reduced diagonal entry -- 1 / (S^-1)_ii in the limit, never above the pivot an
        // elimination in any order leaves for row i -- lost 43 bits against the entry itself:
        // whenever the chunked LDL' would raise its soft finding, this is raised too
        if (!(bn > IPX_PIVOT_RTOL * b0[k])) bits |= 4;
      }
    }
    const bool any0 = __ballot(bits & 1) != 0, any1 = __ballot(bits & 2) != 0;
    const bool any2 = __ballot(bits & 4) != 0;
    if ((tid & (IPX_WAVE - 1)) == 0) {
      if (any0) flags[s + 1] = 1;
      if (any1) flags[PCR_LMAX + 1 + s + 1] = 1;
      if (any2) flags[PCR_LMAX + 1] = 1;
    }
  }
}

// TB lanes per workgroup: 512 (PCR_TB) -- measured against 256 and 1024 at n = 1e6 / 4e6: the
// solve with its tail 10.4 / 28.8 us against 10.8 / 29.9 (256) and 12.6 / 38.6 (1024: one
// workgroup per CU); the form that builds its own right-hand side (config 5) 7.9 us against
// 9.1 (256) and 8.0 (1024).  One window row per lane, half the tail's loads per lane.
template <int QV, int NR, bool ROWS, int TB, bool POST = false>
__global__ void __launch_bounds__(TB)
k_solve_pcr(int m, int rows_wg, int L, const double *__restrict__ band,
            const double *__restrict__ w, double *__restrict__ x, double *__restrict__ partial,
            const double *__restrict__ guard, AtvJob atv, RowsJob rows = RowsJob{},
            PostJob post = PostJob{}) {
  constexpr int QA = QV > 0 ? QV : 1;
  constexpr int PAD = 1 << PCR_LMAX;                 // identity rows either side of the window
  constexpr int RS = PCR_RMAX + 2 * PAD;
  __shared__ double pa[2][RS], pr[2][RS], pd[2][RS];
  __shared__ double sx[PCR_RMAX];
  __shared__ double red_lds[TB / IPX_WAVE];
  constexpr int RWU = ROWS_PRODUCTS / TB;          // products per lane (ROWS form)
  IPX_STAMP(0);
  // A'v tail: this workgroup's variables -- the FIRST loads of the kernel, so that the tail's
  // own loads (which need them) can be issued while the solve's inputs are still in flight
  int av0 = 0, avn = 0;
  if (QV > 0) {
    av0 = atv.vown[blockIdx.x];
    avn = atv.vown[blockIdx.x + 1] - av0;
  }
  const double stop = guard ? *guard : 0.0;
  const int H = 1 << L;
  const int R = rows_wg + 2 * H;
  const int64_t own0 = (int64_t)blockIdx.x * rows_wg;
  const int64_t g0 = own0 - H;                      // global row of window row 0
  const int tid = threadIdx.x;
  // ROWS: the entries of the window's rows, the head of the kernel's only dependent chain
  // (entries -> gather of x -> LDS -> row sums), requested before everything else
  extern __shared__ double rows_prod[];              // R rows of (2^logL + 1) doubles (padded)
  int rcol[ROWS ? RWU : 1];
  double rval[ROWS ? RWU : 1];
  if constexpr (ROWS) {
    const int64_t nnz = (int64_t)m << rows.logL;
    const int64_t e0 = g0 * (1 << rows.logL);        // (negative for the first workgroup)
    const int P = R << rows.logL;
#pragma unroll
    for (int u = 0; u < RWU; ++u) {
      const int e = min(tid + u * TB, P - 1);
      const int64_t idx = min(max(e0 + e, (int64_t)0), nnz - 1);
      rcol[u] = rows.col[idx];
      rval[u] = rows.val[idx];
    }
  }
  // the matrix rows (static data: requested first), then w (the predecessor's output)
  double a[NR], b[NR], d[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int64_t g = g0 + tid + k * TB;
    const bool in = tid + k * TB < R && g >= 0 && g < m;
    const int64_t gc = min(max(g, (int64_t)0), (int64_t)m - 1);
    const double bv = band[gc], av = band[(int64_t)m + gc];
    a[k] = (in && g >= 1 && tid + k * TB >= 1) ? av : 0.0;     // (row 0 of the window: cut)
    b[k] = in ? bv : 1.0;
    if constexpr (!ROWS) {
      const double wv = w[gc];
      d[k] = in ? wv : 0.0;
    }
  }
  if constexpr (ROWS) {
    const int Lr = 1 << rows.logL, P = R << rows.logL;
    double xg[RWU];
#pragma unroll
    for (int u = 0; u < RWU; ++u) xg[u] = rows.x[rcol[u]];
#pragma unroll
    for (int u = 0; u < RWU; ++u) {
      const int e = tid + u * TB;
      if (e < P) rows_prod[(e >> rows.logL) * (Lr + 1) + (e & (Lr - 1))] = rval[u] * xg[u];
    }
    ipx_lds_barrier();
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = tid + k * TB;
      const int64_t g = g0 + r;
      const bool in = r < R && g >= 0 && g < m;
      double sum = 0.0;
      if (in) {
        const double *pr_ = rows_prod + r * (Lr + 1);
        for (int j = 0; j < Lr; ++j) sum += pr_[j];
      }
      d[k] = in ? 1.0 * sum : 0.0;
    }
  }
  // A'v tail: the two ELL entries of every variable and r, requested now so that they arrive
  // while the reduction runs.  A lane takes PAIRS of consecutive variables (16-byte loads of
  // the values and of r, 8-byte loads of the columns: half the load instructions -- the tail
  // is bound by the CU's load issue, 2600 variables x 5 arrays per workgroup); pairs start at
  // an even variable so every load is naturally aligned (the planes are padded to even length)
  constexpr int QP = (QA + 1) / 2;
  const int64_t vb = av0 & ~1;                       // first pair (may start one before av0)
  unsigned ac[QP];                                   // (two 16-bit row offsets)
  v2d aw0[QP], aw1[QP], ar[QP];
  if (QV > 0) {
    const int64_t last = max((int64_t)av0 + avn - 1, vb) & ~(int64_t)1;
#pragma unroll
    for (int k = 0; k < QP; ++k) {
      const int64_t j = min(vb + 2 * (int64_t)(tid + k * TB), last);
      ac[k] = *reinterpret_cast<const unsigned *>(atv.ell_row + j);
      aw0[k] = *reinterpret_cast<const v2d *>(atv.ell_val + j);
      aw1[k] = *reinterpret_cast<const v2d *>(atv.ell_val + atv.ell_n + j);
      ar[k] = *reinterpret_cast<const v2d *>(atv.r_in + j);
    }
  }
  // POST tail: the operands of this workgroup's items, requested now (the products of the ROWS
  // form are in LDS, their registers free) so that they arrive while the reduction runs
  int pg0 = 0, pgn = 0, pe0 = 0, pen = 0;
  int py0[POST ? POST_PG + 1 : 1], py1[POST ? POST_PG + 1 : 1];
  double pw0[POST ? POST_PG + 1 : 1], pw1[POST ? POST_PG + 1 : 1];
  double pep[POST ? POST_PG : 1], peq[POST ? POST_PG : 1];
  double prc[POST ? POST_PG + 1 : 1], prp[POST ? POST_PG : 1], prq[POST ? POST_PG : 1];
  if constexpr (POST) {
    pg0 = post.own_g[blockIdx.x]; pgn = post.own_g[blockIdx.x + 1] - pg0;
    pe0 = post.own_e[blockIdx.x]; pen = post.own_e[blockIdx.x + 1] - pe0;
    const ipx_group_tab &T = post.T;
#pragma unroll
    for (int k = 0; k < POST_PG; ++k) {
      const int i = pg0 + min(tid + k * TB, max(pgn - 1, 0));       // (clamped: no branches)
      const int gi = min(i, max(post.ng - 1, 0));
      const int c = T.c0 + gi;
      py0[k] = post.yrow[gi]; py1[k] = post.yrow[post.nitems + gi];
      pw0[k] = post.yval[gi]; pw1[k] = post.yval[post.nitems + gi];
      pep[k] = T.grp2[2 * gi]; peq[k] = T.grp2[2 * gi + 1];
      prc[k] = post.r[c]; prp[k] = post.r[c + T.dp]; prq[k] = post.r[c + T.dq];
    }
    {
      const int e = min(pe0 + min(tid, max(pen - 1, 0)), max(post.nitems - post.ng - 1, 0));
      const int i = post.ng + e;
      py0[POST_PG] = post.yrow[min(i, post.nitems - 1)];
      py1[POST_PG] = post.yrow[post.nitems + min(i, post.nitems - 1)];
      pw0[POST_PG] = post.yval[min(i, post.nitems - 1)];
      pw1[POST_PG] = post.yval[post.nitems + min(i, post.nitems - 1)];
      prc[POST_PG] = post.r[T.gen0 + e];
    }
  }
  if (stop != 0.0) return;
  IPX_STAMP(1);
  // level-0 rows kept for the residual
  double a0[NR], b0[NR], w0[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) { a0[k] = a[k]; b0[k] = b[k]; w0[k] = d[k]; }
  IPX_STAMP(2);
  // identity padding: rows [-PAD, 0) and [R, R + PAD) of both buffers (a = 0, r = 1, d = 0)
  for (int i = tid; i < 2 * PAD; i += TB) {
    const int r = i < PAD ? i : R + i;              // storage index = window row + PAD
#pragma unroll
    for (int u = 0; u < 2; ++u) { pa[u][r] = 0.0; pr[u][r] = 1.0; pd[u][r] = 0.0; }
  }
  // (a of the row just past the window's end is its coupling into the window: cut too)
  for (int s = 0; s < L; ++s) {
    const int h = 1 << s, cur = s & 1;
    double rc[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = tid + k * TB;
      rc[k] = pcr_rcp(b[k]);
      if (r < R) { pa[cur][PAD + r] = a[k]; pr[cur][PAD + r] = rc[k]; pd[cur][PAD + r] = d[k]; }
    }
    ipx_lds_barrier();
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = PAD + min(tid + k * TB, R - 1);
      const double alo = pa[cur][r - h], rlo = pr[cur][r - h], dlo = pd[cur][r - h];
      const double ahi = pa[cur][r + h], rhi = pr[cur][r + h], dhi = pd[cur][r + h];
      const double al = -a[k] * rlo, ga = -ahi * rhi;
      double bn = __builtin_fma(al, a[k], b[k]);
      bn = __builtin_fma(ga, ahi, bn);
      double dn = __builtin_fma(al, dlo, d[k]);
      dn = __builtin_fma(ga, dhi, dn);
      a[k] = al * alo; b[k] = bn; d[k] = dn;
    }
  }
  IPX_STAMP(3);
  // ---- x = d / b; own rows to memory, the window to LDS for the tail / residual
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = tid + k * TB;
    if (r < R) {
      const double xv = d[k] / b[k];
      sx[r] = xv;
      const int64_t g = g0 + r;
      if (r >= H && r < H + rows_wg && g < m) x[g] = xv;
    }
  }
  if (!partial && QV == 0 && !POST) return;
  ipx_lds_barrier();
  IPX_STAMP(6);
  if (QV > 0) {
    // ---- g = r - A'v on this workgroup's variables, v out of LDS (sx)
    double gacc = 0.0;
#pragma unroll
    for (int k = 0; k < QP; ++k) {
      const int64_t j = vb + 2 * (int64_t)(tid + k * TB);
      // (an absent entry carries value 0: same sum as the CSR row; a pair that reaches into a
      // neighbour's variables reads that workgroup's offsets -- in range, result unused)
      const int c0 = H + (int)(ac[k] & 0xffffu), c1 = H + (int)(ac[k] >> 16);
      double y0 = -atv.sign * (aw0[k].x * sx[c0] + aw1[k].x * sx[c0 + 1]);
      y0 += atv.sign * ar[k].x;
      double y1 = -atv.sign * (aw0[k].y * sx[c1] + aw1[k].y * sx[c1 + 1]);
      y1 += atv.sign * ar[k].y;
      const bool in0 = j >= av0 && j < (int64_t)av0 + avn;
      const bool in1 = j + 1 >= av0 && j + 1 < (int64_t)av0 + avn;
      if (in0 && in1) {
        *reinterpret_cast<v2d *>(atv.g_out + j) = (v2d){y0, y1};
        gacc += y0 * y0;
        gacc += y1 * y1;
      } else if (in0) {
        atv.g_out[j] = y0;
        gacc += y0 * y0;
      } else if (in1) {
        atv.g_out[j + 1] = y1;
        gacc += y1 * y1;
      }
    }
    const double gtot = ipx_block_reduce<IPX_SUM>(gacc, red_lds);
    if (tid == 0) {
      atv.part3[blockIdx.x] = gtot;
      atv.part3[gridDim.x + blockIdx.x] = 0.0;
    }
    if (!partial) return;
  }
  if constexpr (POST) {
    // ---- k_pairs_post's arithmetic, item by item, v_R out of LDS (sx: the window's solution,
    // index = row - g0; an item's two rows are its first general row, in this workgroup's own
    // rows, and at most 2^L rows after it: the host checks)
    const ipx_group_tab &T = post.T;
    const int wlim = R - 1;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < POST_PG; ++k) {
      if (tid + k * TB < pgn) {
        const int gi = pg0 + tid + k * TB;
        const int c = T.c0 + gi, cp = c + T.dp, cq = c + T.dq;
        const double u0 = sx[min(max((int)(py0[k] - g0), 0), wlim)];
        const double u1 = sx[min(max((int)(py1[k] - g0), 0), wlim)];
        double sum = 0.0;
        sum += pw0[k] * u0;
        sum += pw1[k] * u1;
        const double yj = 1.0 * sum;
        const double ap = copysign(1.0, pep[k]), sp = fabs(pep[k]);
        const double aq = copysign(1.0, peq[k]), sq = fabs(peq[k]);
        const double rc = prc[k], rp = prp[k], rq = prq[k];
        double i11, i12, i22, wgt;
        ipx_group_inverse(true, ap, sp, aq, sq, i11, i12, i22, wgt);
        const double wp = cp > c ? ap * rc + sp * rp : sp * rp + ap * rc;
        const double wq = cq > c ? aq * rc + sq * rq : sq * rq + aq * rc;
        const double ep = ap * yj, eq = aq * yj;
        const double tp = i11 * wp + i12 * wq;
        const double tq = i12 * wp + i22 * wq;
        const double vp = tp - (i11 * ep + i12 * eq);
        const double vq = tq - (i12 * ep + i22 * eq);
        double s3 = yj + ap * vp;
        s3 += aq * vq;
        double gc = -1.0 * s3;
        gc += 1.0 * rc;
        double gp = -1.0 * (sp * vp);
        gp += 1.0 * rp;
        double gq = -1.0 * (sq * vq);
        gq += 1.0 * rq;
        post.g[c] = gc; post.g[cp] = gp; post.g[cq] = gq;
        acc += gc * gc;
        acc += gp * gp;
        acc += gq * gq;
      }
    }
    if (tid < pen) {
      const int e = pe0 + tid;
      const double u0 = sx[min(max((int)(py0[POST_PG] - g0), 0), wlim)];
      const double u1 = sx[min(max((int)(py1[POST_PG] - g0), 0), wlim)];
      double sum = 0.0;
      sum += pw0[POST_PG] * u0;
      sum += pw1[POST_PG] * u1;
      double gc = -1.0 * (1.0 * sum);
      gc += 1.0 * prc[POST_PG];
      post.g[T.gen0 + e] = gc;
      acc += gc * gc;
    }
    const double gtot = ipx_block_reduce<IPX_SUM>(acc, red_lds);
    if (tid == 0) {
      post.part[blockIdx.x] = gtot;
      post.part[post.count + blockIdx.x] = 0.0;
      for (int k = blockIdx.x + gridDim.x; k < post.count; k += gridDim.x) {
        post.part[k] = 0.0;
        post.part[post.count + k] = 0.0;
      }
    }
    if (!partial) return;
  }
  // ---- residual of the own rows:  w_i - (a_i x_{i-1} + b_i x_i + a_{i+1} x_{i+1})
  // (a_{i+1} of the row below: the level-0 sub-diagonal of the neighbouring lane, via LDS)
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = tid + k * TB;
    if (r < R) pa[0][PAD + r] = a0[k];
  }
  ipx_lds_barrier();
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int r = tid + k * TB;
    const int64_t g = g0 + r;
    if (r >= H && r < H + rows_wg && g < m) {
      double sum = b0[k] * sx[r];
      sum += a0[k] * sx[r - 1];
      sum += pa[0][PAD + r + 1] * sx[r + 1];
      const double res = w0[k] - sum;
      acc += res * res;
    }
  }
  const double tot = ipx_block_reduce<IPX_SUM>(acc, red_lds);
  if (tid == 0) partial[blockIdx.x] = tot;
  IPX_STAMP(7);
}

template <int QV>
int launch_solve_pcr_q(const LevDev &lv, int L, const double *w, double *x, double *partial,
                       int *npartial, const double *guard, const AtvJob &atv, hipStream_t st) {
  const int rows_wg = DEC_CHUNKS * lv.q;
  const int grid = (lv.P + DEC_CHUNKS - 1) / DEC_CHUNKS;
  if (npartial) *npartial = grid;
  // rows per lane: the window is rows_wg + 2^(L+1) rows
  constexpr int NRL = (PCR_RMAX + PCR_TB - 1) / PCR_TB;
  if (rows_wg + 2 * (1 << L) <= PCR_TB)
    hipLaunchKernelGGL((k_solve_pcr<QV, 1, false, PCR_TB>), dim3(grid), dim3(PCR_TB), 0, st, lv.m,
                       rows_wg, L, lv.band, w, x, partial, guard, atv, RowsJob{});
  else
    hipLaunchKernelGGL((k_solve_pcr<QV, NRL, false, PCR_TB>), dim3(grid), dim3(PCR_TB), 0, st,
                       lv.m, rows_wg, L, lv.band, w, x, partial, guard, atv, RowsJob{});
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// the solve with its right-hand side formed from rows of one length (RowsJob); IPX_EUNSUPPORTED
// when the window does not fit the kernel's fixed product count
int launch_solve_pcr_rows(const LevDev &lv, int L, const RowsJob &rows, double *x,
                          double *partial, int *npartial, const double *guard, hipStream_t st,
                          const PostJob *post = nullptr) {
  const int rows_wg = DEC_CHUNKS * lv.q;
  const int R = rows_wg + 2 * (1 << L);
  if (R > PCR_TB || ((int64_t)R << rows.logL) > (int64_t)ROWS_PRODUCTS)
    return IPX_EUNSUPPORTED;
  const int grid = (lv.P + DEC_CHUNKS - 1) / DEC_CHUNKS;
  if (npartial) *npartial = grid;
  const size_t lds = (size_t)R * ((1 << rows.logL) + 1) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    // (the kernel's static arrays take 41 KB of the CU's 160: the attribute is the dynamic part)
    (void)hipFuncSetAttribute((const void *)k_solve_pcr<0, 1, true, PCR_TB>,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(LDS_LIMIT - 44 * 1024));
    attr_set = true;
  }
  if (post) {
    static bool attr_post = false;
    if (!attr_post) {
      (void)hipFuncSetAttribute((const void *)k_solve_pcr<0, 1, true, PCR_TB, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(LDS_LIMIT - 44 * 1024));
      attr_post = true;
    }
    hipLaunchKernelGGL((k_solve_pcr<0, 1, true, PCR_TB, true>), dim3(grid), dim3(PCR_TB), lds, st,
                       lv.m, rows_wg, L, lv.band, nullptr, x, partial, guard, AtvJob{}, rows, *post);
  } else {
    hipLaunchKernelGGL((k_solve_pcr<0, 1, true, PCR_TB>), dim3(grid), dim3(PCR_TB), lds, st, lv.m,
                       rows_wg, L, lv.band, nullptr, x, partial, guard, AtvJob{}, rows);
  }
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

int launch_solve_pcr(const LevDev &lv, int L, const double *w, double *x, double *partial,
                     int *npartial, const double *guard, hipStream_t st,
                     const AtvJob *atv = nullptr, int qv = 0) {
  const AtvJob none{};
  if (!atv || qv <= 0)
    return launch_solve_pcr_q<0>(lv, L, w, x, partial, npartial, guard, none, st);
  // the tail's lanes are the whole 512-thread workgroup (not the 192 the tables count with):
  // fewer variables each (+1: pairs start at an even variable, possibly one before the
  // workgroup's first).  qv <= 16, the most the tables allow, is a workgroup that owns ~3000
  // variables -- e.g. a problem of a single 260-row block with 11 columns per row
  // (tests/fuzz_fused_loop.py): 6 per lane; the benchmark's 2600: 6 as well
  const int per = (qv * (DOWN_T - IPX_WAVE) + 1 + PCR_TB - 1) / PCR_TB;
  if (per <= 2) return launch_solve_pcr_q<2>(lv, L, w, x, partial, npartial, guard, *atv, st);
  if (per <= 4) return launch_solve_pcr_q<4>(lv, L, w, x, partial, npartial, guard, *atv, st);
  if (per <= 6) return launch_solve_pcr_q<6>(lv, L, w, x, partial, npartial, guard, *atv, st);
  if (per <= 8) return launch_solve_pcr_q<8>(lv, L, w, x, partial, npartial, guard, *atv, st);
  return IPX_EINVAL;
}


template <int K>
size_t decoupled_lds_doubles(int q) {
  constexpr int T = DEC_CHUNKS, NCH = T + 3;
  const size_t qk = q + K;
  return (size_t)2 * NCH * q + qk * NCH * (K + 1) + (size_t)2 * NCH * K * K +
         (size_t)(K + 1) * (T * q + K) + (size_t)3 * NCH * K + (size_t)NCH * K * K + K +
         (size_t)T * q +
         (size_t)2 * K * ((T + 2) * q - (q - K));
}

template <int K, int QV>
int launch_solve_decoupled_q(const LevDev &lv, const double *w, double *x, const double *rinv,
                             double *partial, int *npartial, const double *guard,
                             const AtvJob &atv, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_solve_decoupled<K, DEC_CHUNKS, QV>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
    attr_set = true;
  }
  const int grid = (lv.P + DEC_CHUNKS - 1) / DEC_CHUNKS;
  if (npartial) *npartial = grid;
  hipLaunchKernelGGL((k_solve_decoupled<K, DEC_CHUNKS, QV>), dim3(grid), dim3(DOWN_T),
                     decoupled_lds_doubles<K>(lv.q) * sizeof(double), st, lv, w, x, rinv, partial,
                     guard, atv);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// qv = variables per lane of the A'v tail (0: plain solve)
template <int K>
int launch_solve_decoupled(const LevDev &lv, const double *w, double *x, const double *rinv,
                           double *partial, int *npartial, const double *guard, hipStream_t st,
                           const AtvJob *atv = nullptr, int qv = 0) {
  const AtvJob none{};
  if (!atv || qv <= 0)
    return launch_solve_decoupled_q<K, 0>(lv, w, x, rinv, partial, npartial, guard, none, st);
  if constexpr (K > 4) return IPX_EINVAL;       // the A'v tail is compiled for half bandwidth <= 4
  else {
  if (qv <= 4) return launch_solve_decoupled_q<K, 4>(lv, w, x, rinv, partial, npartial, guard, *atv, st);
  if (qv <= 8) return launch_solve_decoupled_q<K, 8>(lv, w, x, rinv, partial, npartial, guard, *atv, st);
  if (qv <= 12) return launch_solve_decoupled_q<K, 12>(lv, w, x, rinv, partial, npartial, guard, *atv, st);
  if (qv <= 16) return launch_solve_decoupled_q<K, 16>(lv, w, x, rinv, partial, npartial, guard, *atv, st);
  return IPX_EINVAL;
  }
}

template <int K>
int launch_down0(const LevDev &lv, size_t lds, const double *w, double *y, double *gL,
                 double *gR, const double *guard, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_down0<K, DOWN_CHUNKS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
    attr_set = true;
  }
  const int grid = (lv.P + DOWN_CHUNKS - 1) / DOWN_CHUNKS;
  hipLaunchKernelGGL((k_down0<K, DOWN_CHUNKS>), dim3(grid), dim3(DOWN_T), lds, st, lv, w, y, gL,
                     gR, guard);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

template <int K1, bool STAGED>
int launch_middle_impl(const LevArgs &a, int nbuf_total, const double *slab, int nslab,
                       const double *gL, const double *gR, double *xs, const double *guard,
                       hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_middle<K1, STAGED>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
    attr_set = true;
  }
  const size_t lds = ((size_t)nbuf_total + (STAGED ? nslab : 0)) * sizeof(double);
  hipLaunchKernelGGL((k_middle<K1, STAGED>), dim3(1), dim3(MID_T), lds, st, a, nbuf_total, slab,
                     nslab, gL, gR, xs, guard);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

template <int K1>
int launch_middle(const LevArgs &a, int nbuf_total, const double *slab, int nslab, bool staged,
                  const double *gL, const double *gR, double *xs, const double *guard,
                  hipStream_t st) {
  return staged ? launch_middle_impl<K1, true>(a, nbuf_total, slab, nslab, gL, gR, xs, guard, st)
                : launch_middle_impl<K1, false>(a, nbuf_total, slab, nslab, gL, gR, xs, guard, st);
}

template <int K>
struct Launch {
  static int factor(Banded *h, int li, hipStream_t st) {
    Level &lv = h->lev[li];
    constexpr int T = factor_chunks_per_wg<K>();
    const size_t lds = factor_lds_doubles<K>(lv.q) * sizeof(double);
    if (lds <= LDS_LIMIT) {
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_factor_lds<K, T>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
        attr_set = true;
      }
      hipLaunchKernelGGL((k_factor_lds<K, T>), dim3((lv.P + T - 1) / T), dim3(DOWN_T), lds, st,
                         lv.m, lv.c, lv.P, lv.band, lv.Dinv, lv.L, lv.V, lv.W, h->flag);
    } else {
      int grid = (lv.P + IPX_WAVE - 1) / IPX_WAVE;
      hipLaunchKernelGGL(k_factor_chunks<K>, dim3(grid), dim3(IPX_WAVE), 0, st, lv.m, lv.c, lv.P,
                         lv.band, lv.Dinv, lv.L, lv.V, lv.W, h->flag);
    }
    IPX_CHECK_LAUNCH();
    if (lv.mR > 0) {
      Level &nx = h->lev[li + 1];
      int64_t tot = (int64_t)lv.mR * (2 * K);
      hipLaunchKernelGGL(k_reduced_matrix<K>, dim3((unsigned)((tot + IPX_BLOCK - 1) / IPX_BLOCK)),
                         dim3(IPX_BLOCK), 0, st, lv.m, lv.c, lv.P, lv.band, lv.V, lv.W, nx.band);
      IPX_CHECK_LAUNCH();
    }
    return IPX_OK;
  }
  static int down(Banded *h, int li, const double *w, double *y, const double *guard,
                  hipStream_t st) {
    Level &lv = h->lev[li];
    int grid = (lv.P + IPX_WAVE - 1) / IPX_WAVE;
    hipLaunchKernelGGL(k_solve_chunks<K>, dim3(grid), dim3(IPX_WAVE), 0, st, lv.m, lv.c, lv.P,
                       lv.Dinv, lv.L, w, y, guard);
    IPX_CHECK_LAUNCH();
    if (lv.mR > 0) {
      Level &nx = h->lev[li + 1];
      hipLaunchKernelGGL(k_reduced_rhs<K>, dim3((lv.mR + IPX_BLOCK - 1) / IPX_BLOCK),
                         dim3(IPX_BLOCK), 0, st, lv.m, lv.c, lv.P, lv.band, w, y, nx.rhs, guard);
      IPX_CHECK_LAUNCH();
    }
    return IPX_OK;
  }
  static int up(Banded *h, int li, double *x, const double *guard, hipStream_t st) {
    Level &lv = h->lev[li];
    Level &nx = h->lev[li + 1];
    hipLaunchKernelGGL(k_correct<K>, dim3((lv.m + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0,
                       st, lv.m, lv.c, lv.P, lv.V, lv.W, nx.sol, x, guard);
    IPX_CHECK_LAUNCH();
    return IPX_OK;
  }
};

#define DISPATCH_K(kk, CALL)                         \
  switch (kk) {                                      \
    case 1: return Launch<1>::CALL;                  \
    case 2: return Launch<2>::CALL;                  \
    case 3: return Launch<3>::CALL;                  \
    case 4: return Launch<4>::CALL;                  \
    case 5: return Launch<5>::CALL;                  \
    case 6: return Launch<6>::CALL;                  \
    case 7: return Launch<7>::CALL;                  \
    case 8: return Launch<8>::CALL;                  \
    default: return IPX_EINVAL;                      \
  }

int level_factor(Banded *h, int li, hipStream_t st) { DISPATCH_K(h->lev[li].k, factor(h, li, st)) }
int level_down(Banded *h, int li, const double *w, double *y, const double *guard, hipStream_t st) {
  DISPATCH_K(h->lev[li].k, down(h, li, w, y, guard, st))
}
int level_up(Banded *h, int li, double *x, const double *guard, hipStream_t st) {
  DISPATCH_K(h->lev[li].k, up(h, li, x, guard, st))
}

// LDS need of the single-launch decoupled solve for half bandwidth k, row pitch q
size_t decoupled_lds_for(int k, int q) {
  switch (k) {
#define DL(kk) case kk: return decoupled_lds_doubles<kk>(q) * sizeof(double)
    DL(1); DL(2); DL(3); DL(4); DL(5); DL(6); DL(7); DL(8);
#undef DL
  }
  return (size_t)-1;
}
size_t decoupled_lds_bytes(const Banded *h) { return decoupled_lds_for(h->lev[0].k, h->lev[0].q); }

bool decoupling_candidate(const Banded *h) {
  return h->fast && h->nlev >= 2 && h->rinv && decoupled_lds_bytes(h) <= LDS_LIMIT;
}

// the separator values as (gL + gR) / R_tt inside the three-launch path: scalar blocks only
bool decoupled_scalar(const Banded *h) { return h->decoupled && h->lev[0].k == 1; }

int factor_upper(Banded *h, hipStream_t st) {
  for (int li = 1; li < h->nlev; ++li) {
    int rc = level_factor(h, li, st);
    if (rc != IPX_OK) return rc;
  }
  h->upper_done = true;
  return IPX_OK;
}

// flag[0] = pivot / coupling bits; flag[1 ..] = the PCR level flags: ONE blocking read
// (through the library's pinned read-back, csrc/misc.hip: a copy into pageable memory costs two
// runtime synchronisations -- this read ends every factorization)
int read_flag(Banded *h, int *f, hipStream_t st, int count = 1) {
  return ipx_read_ints(h->flag, count, f, st) == IPX_OK ? IPX_OK : IPX_ELAUNCH;
}

// the verdict of ipx_banded_status from the flags, on the device (ipx_banded_status_deferred)
__global__ void k_status_verdict(const int *__restrict__ flag, const int *__restrict__ pcr_flags,
                                 int L_assumed, double *__restrict__ verdict) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int *pf = pcr_flags, *pn = pcr_flags + (PCR_LMAX + 1);
  // (pn[0]: the reduction's own soft finding -- with the chunk factorization deferred,
  // ipx_banded_refactor, it stands in for flag[0]'s bits, which nobody has raised yet)
  const int f = flag[0] | pn[0];
  int L = 0;
  for (int s = 1; s <= PCR_LMAX; ++s) {
    if (pn[s]) break;
    if (!pf[s]) { L = s; break; }
  }
  verdict[0] = (f == 0 && L == L_assumed) ? 0.0 : 1.0;
}

template <typename T>
T *dalloc(Banded *h, size_t n) {
  void *p = nullptr;
  if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
  h->allocs.push_back(p);
  return (T *)p;
}

}  // namespace

extern "C" {

int ipx_banded_kmax(void) { return KMAX; }

IPX_STAMP_EXPORT(ipx_debug_stamps, ipx_dbg_stamps)

// Plan the level hierarchy for an m x m SPD matrix of half bandwidth k.
// `chunk` = interior rows per chunk (0 = default).  Returns a handle or NULL.
void *ipx_banded_create(int64_t m64, int32_t k, int32_t chunk) {
  if (m64 < 1 || m64 > (1LL << 30) || k < 0 || k > KMAX) return nullptr;
  if (k == 0) k = 1;                 // diagonal matrices ride the k = 1 path
  Banded *h = new Banded();
  h->nlev = 0;
  h->gL = h->gR = h->slab = h->ybuf = h->rinv = nullptr;
  h->decoupled = false;
  h->fast = h->fast_plan = false;
  h->wide = false;
  h->mid_fits = false;
  h->iter_cand = false;
  h->iter_N = 0;
  h->iter_buf = h->eta = nullptr;
  h->eta_host = 0.0;
  h->pcr_flags = nullptr;
  h->pcr_L = 0;
  h->last_L = 0;
  h->chunk_pending = false;
  int m = (int)m64, kk = k;
  if (chunk <= 0) chunk = 64;
  bool ok = true;
  // ---- geometry of every level
  while (true) {
    if (h->nlev >= MAX_LEVELS || (kk > KMAX && !h->wide)) { ok = false; break; }
    Level &lv = h->lev[h->nlev++];
    lv.m = m; lv.k = kk;
    lv.c = chunk < kk ? kk : chunk;
    // odd row pitch q = c + k: lanes of a wave then hit distinct LDS banks
    if (((lv.c + kk) & 1) == 0) lv.c += 1;
    // The separator system of a level has half bandwidth 2k-1.  When that
    // would exceed the compiled kernels there is no recursion below this level:
    //  * level 0 of a long band (k = 5..8, beyond WIDE_MIN_ROWS rows) is still cut into
    //    chunks; its separator matrix is formed to test whether its K x K blocks decouple
    //    numerically (the usual case: the chunk is ~8 bandwidths long), and then the
    //    single-launch decoupled solve needs nothing above level 0.  If they do not,
    //    ipx_banded_status says IPX_EUNSUPPORTED and the caller takes another solver;
    //  * otherwise the level becomes one chunk swept by a single lane (short systems and
    //    upper levels, where the level is already small).
    if (h->wide) {                       // the formed-only separator level
      lv.c = m;
    } else if (2 * kk - 1 > KMAX && m > lv.c + kk) {
      if (h->nlev == 1 && m > WIDE_MIN_ROWS && decoupled_lds_for(kk, lv.c + kk) <= LDS_LIMIT)
        h->wide = true;
      else lv.c = m;
    }
    lv.q = lv.c + kk;
    // single chunk when everything fits in one (<= q rows)
    lv.P = (m <= lv.q) ? 1 : (m + lv.q - 1) / lv.q;
    lv.mR = (lv.P - 1) * kk;
    lv.band = lv.Dinv = lv.L = lv.V = lv.W = lv.rhs = lv.sol = nullptr;
    if (lv.P == 1) break;
    m = lv.mR;
    kk = 2 * kk - 1;
  }
  // ---- storage: level 0 on its own, levels >= 1 carved from one slab
  auto need = [](const Level &lv, size_t &nD, size_t &nL, size_t &nV, size_t &nB) {
    nD = (size_t)(lv.q + lv.k) * lv.P;   // k rows of zero padding behind the q steps
    nL = nD * lv.k; nV = (size_t)lv.m * lv.k; nB = (size_t)(lv.k + 1) * lv.m;
  };
  if (ok) {
    size_t nD, nL, nV, nB;
    Level &l0 = h->lev[0];
    need(l0, nD, nL, nV, nB);
    l0.Dinv = dalloc<double>(h, nD); l0.L = dalloc<double>(h, nL);
    l0.V = dalloc<double>(h, nV);    l0.W = dalloc<double>(h, nV);
    ok = l0.Dinv && l0.L && l0.V && l0.W;
    if (ok) ok = hipMemset(l0.Dinv, 0, nD * sizeof(double)) == hipSuccess &&
                 hipMemset(l0.L, 0, nL * sizeof(double)) == hipSuccess;
    size_t tot = 0;
    for (int li = 1; li < h->nlev; ++li) { need(h->lev[li], nD, nL, nV, nB); tot += nD + nL + 2 * nV + nB; }
    h->nslab = tot;
    if (ok && tot > 0) {
      h->slab = dalloc<double>(h, tot);
      ok = h->slab != nullptr && hipMemset(h->slab, 0, tot * sizeof(double)) == hipSuccess;
      double *p = h->slab;
      for (int li = 1; ok && li < h->nlev; ++li) {
        Level &lv = h->lev[li];
        need(lv, nD, nL, nV, nB);
        lv.Dinv = p; p += nD; lv.L = p; p += nL; lv.V = p; p += nV; lv.W = p; p += nV;
        lv.band = p; p += nB;
        lv.rhs = dalloc<double>(h, lv.m);
        lv.sol = dalloc<double>(h, lv.m);
        ok = lv.rhs && lv.sol;
      }
    }
  }
  h->flag = dalloc<int>(h, 1 + 2 * (PCR_LMAX + 1) + 1);   // (+1: ipx_read_ints reads whole doubles)
  if (ok && h->flag && h->lev[0].k == 1 && h->nlev >= 2 &&
      DEC_CHUNKS * h->lev[0].q + 2 * (1 << PCR_LMAX) <= PCR_RMAX) {
    h->pcr_flags = h->flag + 1;
  }
  if (ok) {
    const Level &l0 = h->lev[0];
    // k_down0 stages (k+2) tables of T*q doubles.  Few chunks per workgroup =
    // many workgroups, so the staging loads spread over the CUs.
    h->down_T = DOWN_CHUNKS;
    h->lds_down = ((size_t)DOWN_CHUNKS * l0.q + (size_t)(l0.k + 1) * DOWN_CHUNKS * (l0.q + l0.k) +
                   2 * (size_t)DOWN_CHUNKS * l0.k * l0.k) * sizeof(double);
    size_t vec = 0;
    for (int li = 1; li < h->nlev; ++li) vec += (size_t)h->lev[li].m;
    h->mid_buf = (int)vec;
    h->nslab_lds = ((vec + h->nslab) * sizeof(double) <= LDS_LIMIT) ? (int)h->nslab : 0;
    h->mid_fits = vec * sizeof(double) <= LDS_LIMIT;
    const bool long_top = h->wide || (h->nlev >= 2 && h->lev[h->nlev - 1].m > ITER_TOP_MIN_ROWS);
    // (a tridiagonal band is a candidate for the cyclic-reduction solve whatever the size of the
    // upper levels -- it never touches them: m = 1.6e6 used to fall back to the level-by-level
    // form because its separator vectors do not fit k_middle's LDS; when the reduction turns
    // out not to decouple, ipx_banded_status takes `fast` back)
    const bool pcr_cand = l0.k == 1 && h->nlev >= 2 && h->pcr_flags != nullptr;
    h->fast = h->lds_down <= LDS_LIMIT && (long_top || h->mid_fits || pcr_cand);
    h->fast_plan = h->fast;
    h->iter_cand = h->fast && long_top && decoupled_lds_for(l0.k, l0.q) <= LDS_LIMIT;
    if (h->iter_cand) {
      h->iter_buf = dalloc<double>(h, (size_t)4 * l0.m + 1);
      if (!h->iter_buf) ok = false;
      else h->eta = h->iter_buf + (size_t)4 * l0.m;
    }
    if (h->fast && l0.mR > 0) {
      h->gL = dalloc<double>(h, l0.mR);
      h->gR = dalloc<double>(h, l0.mR);
      h->ybuf = dalloc<double>(h, l0.m);
      h->rinv = dalloc<double>(h, (size_t)l0.mR * l0.k);       // K x K per separator
      if (!h->gL || !h->gR || !h->ybuf || !h->rinv) ok = false;
      else if (hipMemset(h->gR, 0, (size_t)l0.mR * sizeof(double)) != hipSuccess) ok = false;
    }
  }
  if (!ok || !h->flag) {
    for (void *p : h->allocs) (void)hipFree(p);
    delete h;
    return nullptr;
  }
  return h;
}

void ipx_banded_destroy(void *handle) {
  if (!handle) return;
  Banded *h = (Banded *)handle;
  for (void *p : h->allocs) (void)hipFree(p);
  delete h;
}

int ipx_banded_levels(void *handle) { return handle ? ((Banded *)handle)->nlev : IPX_EINVAL; }

// band: (k+1) x m lower band storage of S, band[d*m+i] = S[i][i-d]; must stay
// alive and unchanged until the last solve with this factorization.
}  // extern "C"

namespace {
// how many ints the flags are: the pivot word and, right behind it, the cyclic reduction's
// level flags (cleared together)
int flag_ints(const Banded *h) { return h->pcr_flags ? 1 + 2 * (PCR_LMAX + 1) : 1; }

int launch_pcr_check(Banded *h, hipStream_t st) {
  // the cyclic reduction of the matrix alone: at which level has it decoupled?
  const Level &l0 = h->lev[0];
  hipLaunchKernelGGL(k_pcr_check, dim3((l0.P + DEC_CHUNKS - 1) / DEC_CHUNKS), dim3(IPX_BLOCK), 0,
                     st, l0.m, DEC_CHUNKS * l0.q, l0.band, h->pcr_flags);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

int chunk_tail(Banded *h, hipStream_t st);

// Level 0 and the separator (Schur complement) matrix, then the separators' decoupling check
// (or, no candidate for the decoupled solve, the upper levels): everything of a factorization
// that the cyclic-reduction solve does not read.
int factor_chunks(Banded *h, hipStream_t st) {
  int rc = level_factor(h, 0, st);
  if (rc != IPX_OK) return rc;
  h->upper_done = false;
  return IPX_OK;
}

// ipx_banded_refactor left the chunk factorization out: a solve that needs it (in place, no
// ELL(2) tables for the tail, a general path) runs it now, with the blocking verdict of
// ipx_banded_status -- the assumption the refactorization made covers the cyclic reduction only
int ensure_chunks(Banded *h, hipStream_t st) {
  if (!h->chunk_pending) return IPX_OK;
  const int rc = ipx_banded_status(h, st);
  return rc == IPX_EILLCOND ? IPX_OK : rc;
}
}  // namespace

extern "C" {

int ipx_banded_factor(void *handle, const double *band, void *stream) {
  if (!handle || !band) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  hipStream_t st = (hipStream_t)stream;
  // (one memset; a length that is not a multiple of 8 bytes is two fill launches: the buffer is
  // allocated with a spare int)
  if (hipMemsetAsync(h->flag, 0, (size_t)((flag_ints(h) + 1) & ~1) * sizeof(int), st) != hipSuccess)
    return IPX_ELAUNCH;
  h->lev[0].band = const_cast<double *>(band);
  h->decoupled = false;
  h->chunk_pending = false;
  h->fast = h->fast_plan;
  // Level 0 and the separator (Schur complement) matrix first.  When the
  // decoupled path is a candidate the upper levels wait for its verdict
  // (ipx_banded_status): a decoupled solve never touches them.
  int rc = factor_chunks(h, st);
  if (rc != IPX_OK) return rc;
  h->pcr_L = 0;
  if (decoupling_candidate(h) && h->pcr_flags) {
    rc = launch_pcr_check(h, st);
    if (rc != IPX_OK) return rc;
  }
  h->iter_N = 0;
  return chunk_tail(h, st);
}

}  // extern "C"

namespace {
int chunk_tail(Banded *h, hipStream_t st) {
  if (decoupling_candidate(h)) {
    if (h->eta && hipMemsetAsync(h->eta, 0, sizeof(double), st) != hipSuccess) return IPX_ELAUNCH;
    const int mR = h->lev[0].mR, K0 = h->lev[0].k, nsep = mR / K0;
    const dim3 grid((nsep + IPX_BLOCK - 1) / IPX_BLOCK), block(IPX_BLOCK);
    switch (K0) {
      case 1:
        hipLaunchKernelGGL(k_decoupling_check, grid, block, 0, st, mR, h->lev[1].band, h->rinv,
                           h->flag, h->eta);
        break;
#define DC(kk)                                                                              \
  case kk:                                                                                  \
    hipLaunchKernelGGL(k_decoupling_check_block<kk>, grid, block, 0, st, nsep, h->lev[1].band, \
                       h->rinv, h->flag, h->eta);                                           \
    break
      DC(2); DC(3); DC(4); DC(5); DC(6); DC(7); DC(8);
#undef DC
      default: return IPX_EINVAL;
    }
    IPX_CHECK_LAUNCH();
    return IPX_OK;
  }
  return factor_upper(h, st);
}
}  // namespace

extern "C" {

// Blocking read of the pivot flag: IPX_OK or IPX_ENOTSPD.
int ipx_banded_status(void *handle, void *stream) {
  if (!handle) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  hipStream_t st = (hipStream_t)stream;
  if (h->chunk_pending) {          // (ipx_banded_refactor left it out: now, then the verdict)
    h->chunk_pending = false;
    int rc = factor_chunks(h, st);
    if (rc == IPX_OK) rc = chunk_tail(h, st);
    if (rc != IPX_OK) return rc;
  }
  int fl[1 + 2 * (PCR_LMAX + 1)] = {0};
  if (read_flag(h, fl, st, h->pcr_flags ? 1 + 2 * (PCR_LMAX + 1) : 1) != IPX_OK) return IPX_ELAUNCH;
  int f = fl[0];
  h->decoupled = decoupling_candidate(h) && !(f & 2);
  h->pcr_L = 0;
  if (h->decoupled && h->pcr_flags && !(f & 1)) {
    const int *pf = fl + 1, *pn = fl + 1 + (PCR_LMAX + 1);
    for (int s = 1; s <= PCR_LMAX; ++s) {
      if (pn[s]) break;                     // a non-positive reduced diagonal: not this path
      if (!pf[s]) { h->pcr_L = s; break; }
    }
  }
  h->last_L = (f == 0 && h->decoupled) ? h->pcr_L : 0;
  h->iter_N = 0;
  if (!h->decoupled && h->iter_cand && decoupling_candidate(h) && !(f & 1)) {
    // coupled separator blocks and a top level that is long (or not compiled): solves become
    // defect correction on the single-launch solve when block Jacobi contracts fast enough
    double eta = 0.0;
    if (hipMemcpyAsync(&eta, h->eta, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return IPX_ELAUNCH;
    h->eta_host = eta;
    if (eta > 0.0 && eta < ITER_ETA_MAX) {
      // x_0 = M^-1 w is off by eta, every step gains another factor: eta^(N+1) <= 2^-54
      int N = (int)ceil(54.0 * 0.6931471805599453 / -log(eta)) - 1;
      h->iter_N = N < 1 ? 1 : N;
    }
  }
  if (h->iter_N > 0) return IPX_OK;
  if (h->wide && !h->decoupled)
    return (f & 4) ? IPX_ENOTSPD : ((f & 1) ? IPX_EILLCOND : IPX_EUNSUPPORTED);
  if (!h->decoupled && !h->mid_fits) h->fast = false;     // level by level (k_middle needs LDS)
  if (!h->decoupled && !h->upper_done && !(f & 1)) {
    int rc = factor_upper(h, st);
    if (rc != IPX_OK) return rc;
    if (read_flag(h, &f, st) != IPX_OK) return IPX_ELAUNCH;
  }
  // bit 4: a pivot was not positive (A A' not SPD: rank-deficient A); bit 1 alone: a pivot
  // lost IPX_PIVOT_RTOL against its diagonal entry -- the factorization is complete and
  // usable, the caller decides (SVD exit when the matrix is small enough, refinement otherwise)
  return (f & 4) ? IPX_ENOTSPD : ((f & 1) ? IPX_EILLCOND : IPX_OK);
}

// ipx_banded_status WITHOUT its blocking read, for a handle whose previous factorization was
// clean and runs the cyclic-reduction solve (tridiagonal A A', decoupled at level L): the host
// ASSUMES the same verdict for the factorization just enqueued -- the usual case from one
// accepted step of the outer loop to the next -- and a one-thread kernel derives the real one
// from the flags on the device: verdict[0] = 0 when it is the assumed one (no pivot finding,
// separators decoupled, the reduction decoupled at the same level), else 1.  The caller enqueues
// its solves, reads `verdict` with whatever it reads next anyway (the outer iteration's block,
// csrc/sqp.hip) and, on a 1, calls ipx_banded_status and repeats them.  IPX_EUNSUPPORTED: no
// clean previous verdict on this handle -- call ipx_banded_status.
int ipx_banded_status_deferred(void *handle, double *verdict, void *stream) {
  if (!handle || !verdict) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  if (h->last_L <= 0 || !h->pcr_flags || !decoupling_candidate(h)) return IPX_EUNSUPPORTED;
  hipLaunchKernelGGL(k_status_verdict, dim3(1), dim3(IPX_WAVE), 0, (hipStream_t)stream, h->flag,
                     h->pcr_flags, h->last_L, verdict);
  IPX_CHECK_LAUNCH();
  h->decoupled = true;
  h->pcr_L = h->last_L;
  h->iter_N = 0;
  return IPX_OK;
}

// 1 when solves skip the middle kernel (separator system diagonal to working
// precision); valid after ipx_banded_status.
// A numeric refresh on a handle whose previous factorization was clean and runs the cyclic-
// reduction solve, in THREE launches: the band of A A' (rows in their own order; the kernel also
// clears the flags), the reduction's check of the matrix alone -- decoupling level, positive
// reduced diagonal, no entry of it below 2^-43 of the diagonal entry --, and the verdict kernel
// of ipx_banded_status_deferred.  The chunked LDL' of level 0, the separator matrix and their
// decoupling check (three more launches, half the refresh's GPU time) are left out: the
// cyclic-reduction solves read the band only.  A solve that does need them -- in place, a tail
// without ELL(2) tables, any path once the verdict was bad -- runs them first, with the blocking
// verdict (ensure_chunks).  Returns 1 when it took this form (the caller reads `verdict` with
// whatever it reads next and, on a 1 there, calls ipx_banded_status and repeats its solves);
// 0: the handle does not qualify, nothing was enqueued -- ipx_aat_band_w + ipx_banded_factor +
// ipx_banded_status as before.
int ipx_banded_refactor(void *handle, int64_t m, int32_t k, const int32_t *rowptr,
                        const int32_t *colidx, const double *val, const double *wcol,
                        double *band, double *verdict, void *stream) {
  if (!handle || !rowptr || !colidx || !val || !band || !verdict) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  hipStream_t st = (hipStream_t)stream;
  const Level &l0 = h->lev[0];
  if (h->last_L <= 0 || !h->pcr_flags || !h->fast_plan || l0.k != 1 || k != 1 || l0.m != m ||
      flag_ints(h) > IPX_BLOCK)
    return 0;
  h->fast = h->fast_plan;
  if (!decoupling_candidate(h)) return 0;
  hipLaunchKernelGGL(k_aat_band_rows, dim3((unsigned)((m + IPX_BLOCK - 1) / IPX_BLOCK)),
                     dim3(IPX_BLOCK), 0, st, (int)m, k, rowptr, colidx, val, wcol, band, h->flag,
                     flag_ints(h));
  IPX_CHECK_LAUNCH();
  h->lev[0].band = band;
  h->upper_done = false;
  h->iter_N = 0;
  int rc = launch_pcr_check(h, st);
  if (rc != IPX_OK) return rc;
  hipLaunchKernelGGL(k_status_verdict, dim3(1), dim3(IPX_WAVE), 0, st, h->flag, h->pcr_flags,
                     h->last_L, verdict);
  IPX_CHECK_LAUNCH();
  h->decoupled = true;
  h->pcr_L = h->last_L;
  h->chunk_pending = true;
  return 1;
}

int ipx_banded_decoupled(void *handle) { return handle && ((Banded *)handle)->decoupled; }

int ipx_banded_set_decoupling(void *handle, int allow) {
  if (!handle) return IPX_EINVAL;
  if (!allow) ((Banded *)handle)->decoupled = false;
  if (allow == 2) ((Banded *)handle)->pcr_L = 0;     // keep the chunk form of the decoupled solve
  // test hook: force the reduction to stop at level allow - 16 (an INEXACT solve when that is
  // below the measured level: gives the fused residual something real to measure)
  if (allow > 16 && allow <= 16 + PCR_LMAX && ((Banded *)handle)->pcr_L > 0)
    ((Banded *)handle)->pcr_L = allow - 16;
  return IPX_OK;
}

// Correction steps per solve when this factorization runs defect correction on the
// single-launch solve (coupled separator blocks, no usable separator level), else 0;
// *eta (optional) receives the measured block-Jacobi contraction bound.
int ipx_banded_refine_steps(void *handle, double *eta) {
  if (!handle) return 0;
  if (eta) *eta = ((Banded *)handle)->eta_host;
  return ((Banded *)handle)->iter_N;
}

// Level at which the cyclic reduction of this factorization's matrix has decoupled (the
// single-launch solve then runs as parallel cyclic reduction); 0 when that path is off.
int ipx_banded_pcr_level(void *handle) { return handle ? ((Banded *)handle)->pcr_L : 0; }

// x = S^-1 w.  w and x are length m; x may alias w.
int ipx_banded_solve(void *handle, const double *w, double *x, void *stream) {
  return ipx_banded_solve_guarded(handle, w, x, nullptr, (hipStream_t)stream);
}

// band[d*m+i] = (A A')[pi, p(i-d)] with rows taken in the order perm (NULL =
// identity); reference: the matrix CHOLMOD factors in projections.py:62.
int ipx_aat_band_w(int64_t m, int32_t k, const int32_t *rowptr, const int32_t *colidx,
                   const double *val, const int32_t *perm, const double *wcol, double *band,
                   void *stream) {
  if (m < 0 || k < 0 || !rowptr || !band) return IPX_EINVAL;
  if (m == 0) return IPX_OK;
  int64_t tot = m * (k + 1);
  if (!perm)
    hipLaunchKernelGGL(k_aat_band_rows, dim3((unsigned)((m + IPX_BLOCK - 1) / IPX_BLOCK)),
                       dim3(IPX_BLOCK), 0, (hipStream_t)stream, (int)m, k, rowptr, colidx, val,
                       wcol, band);
  else
    hipLaunchKernelGGL(k_aat_band, dim3((unsigned)((tot + IPX_BLOCK - 1) / IPX_BLOCK)),
                       dim3(IPX_BLOCK), 0, (hipStream_t)stream, (int)m, k, rowptr, colidx, val, perm,
                       wcol, band);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

int ipx_aat_band(int64_t m, int32_t k, const int32_t *rowptr, const int32_t *colidx,
                 const double *val, const int32_t *perm, double *band, void *stream) {
  return ipx_aat_band_w(m, k, rowptr, colidx, val, perm, nullptr, band, stream);
}

}  // extern "C"

namespace {

LevDev to_dev(const Level &lv, const double *slab) {
  LevDev d{lv.m, lv.k, lv.c, lv.P, lv.q, lv.mR, lv.band, lv.Dinv, lv.L, lv.V, lv.W, 0, 0, 0, 0, 0};
  if (slab && lv.Dinv >= slab) {
    d.oB = (int)(lv.band - slab); d.oD = (int)(lv.Dinv - slab); d.oL = (int)(lv.L - slab);
    d.oV = (int)(lv.V - slab);    d.oW = (int)(lv.W - slab);
  }
  return d;
}

int fast_down0(Banded *h, const double *w, double *y, const double *guard, hipStream_t st) {
  const LevDev lv = to_dev(h->lev[0], nullptr);
  switch (lv.k) {
#define D0(kk) case kk: return launch_down0<kk>(lv, h->lds_down, w, y, h->gL, h->gR, guard, st)
    D0(1); D0(2); D0(3); D0(4); D0(5); D0(6); D0(7); D0(8);
#undef D0
  }
  return IPX_EINVAL;
}

int fast_middle(Banded *h, const double *guard, hipStream_t st) {
  LevArgs a;
  a.nlev = h->nlev;
  for (int li = 0; li < h->nlev; ++li) a.lev[li] = to_dev(h->lev[li], li ? h->slab : nullptr);
  double *xs = h->lev[1].sol;
  switch (h->lev[1].k) {
#define MD(kk) case kk: return launch_middle<kk>(a, h->mid_buf, h->slab, (int)h->nslab, h->nslab_lds > 0, h->gL, h->gR, xs, guard, st)
    MD(1); MD(3); MD(5); MD(7);
#undef MD
  }
  return IPX_EINVAL;
}

}  // namespace

extern "C" int ipx_banded_solve_multilaunch(void *handle, const double *w, double *x,
                                            void *stream);

namespace {
template <int K>
int launch_correct_oop(Banded *h, double *x, const double *w, double *partial, int *npartial,
                       const double *guard, hipStream_t st) {
  const Level &lv = h->lev[0];
  const int grid = (lv.m + IPX_BLOCK - 1) / IPX_BLOCK;
  if (npartial) *npartial = grid;
  const SepValues xs = decoupled_scalar(h) ? SepValues{nullptr, h->gL, h->gR, h->rinv}
                                    : SepValues{h->lev[1].sol, nullptr, nullptr, nullptr};
  if (partial)
    hipLaunchKernelGGL((k_correct_oop<K, true>), dim3(grid), dim3(IPX_BLOCK), 0, st, lv.m, lv.c,
                       lv.P, lv.V, lv.W, xs, h->ybuf, x, lv.band, w, partial, guard);
  else
    hipLaunchKernelGGL((k_correct_oop<K, false>), dim3(grid), dim3(IPX_BLOCK), 0, st, lv.m, lv.c,
                       lv.P, lv.V, lv.W, xs, h->ybuf, x, lv.band, w, partial, guard);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// r = w - S (xa + d), xb = xa + d   (d == nullptr: r = w - S xa only)
__global__ void __launch_bounds__(IPX_BLOCK)
k_band_update_resid(int m, int k, const double *__restrict__ band, const double *__restrict__ w,
                    const double *__restrict__ xa, const double *__restrict__ d,
                    double *__restrict__ xb, double *__restrict__ r,
                    const double *__restrict__ guard) {
  if (guard && *guard != 0.0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  auto X = [&](int j) { return d ? xa[j] + d[j] : xa[j]; };
  const double xi = X(i);
  double s = band[i] * xi;
  for (int dd = 1; dd <= k; ++dd) {
    if (i - dd >= 0) s += band[(int64_t)dd * m + i] * X(i - dd);
    if (i + dd < m) s += band[(int64_t)dd * m + i + dd] * X(i + dd);
  }
  if (xb) xb[i] = xi;
  r[i] = w[i] - s;
}

__global__ void __launch_bounds__(IPX_BLOCK)
k_band_add(int m, const double *__restrict__ xa, const double *__restrict__ d,
           double *__restrict__ x, const double *__restrict__ guard) {
  if (guard && *guard != 0.0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) x[i] = xa[i] + d[i];
}

int solve_decoupled_k(const LevDev &lv, const double *w, double *x, const double *rinv,
                      const double *guard, hipStream_t st) {
  switch (lv.k) {
#define SD(kk) case kk: return launch_solve_decoupled<kk>(lv, w, x, rinv, nullptr, nullptr, guard, st)
    SD(1); SD(2); SD(3); SD(4); SD(5); SD(6); SD(7); SD(8);
#undef SD
  }
  return IPX_EINVAL;
}

// Defect correction on the single-launch solve.  With coupled separator blocks that kernel
// applies M^-1, where M is S with the separator system R replaced by its diagonal blocks
// (exactly: every workgroup computes its separators from its own halo chunks), and
// I - M^-1 S contracts like block Jacobi on R: rho <= eta, measured at the factorization
// (k_decoupling_check_block).  N steps of  x <- x + M^-1 (w - S x)  with N fixed from eta
// (ipx_banded_status) give S^-1 w to working precision without a separator level, without
// host synchronisation and bitwise reproducibly -- instead of a serial sweep over the
// (P-1) k separator rows by one lane (1.7 ms at m = 1e5, k = 3; no compiled kernel at all
// for k >= 5).
int iter_solve(Banded *h, const double *w, double *x, double *partial, int *npartial,
               const double *guard, hipStream_t st) {
  const LevDev lv = to_dev(h->lev[0], nullptr);
  const int m = lv.m;
  double *xa = h->iter_buf, *xb = xa + m, *r = xb + m, *d = r + m;
  const dim3 grid((m + IPX_BLOCK - 1) / IPX_BLOCK), block(IPX_BLOCK);
  int rc = solve_decoupled_k(lv, w, xa, h->rinv, guard, st);
  if (rc != IPX_OK) return rc;
  hipLaunchKernelGGL(k_band_update_resid, grid, block, 0, st, m, lv.k, lv.band, w, xa,
                     (const double *)nullptr, (double *)nullptr, r, guard);
  IPX_CHECK_LAUNCH();
  for (int it = 1; it <= h->iter_N; ++it) {
    rc = solve_decoupled_k(lv, r, d, h->rinv, guard, st);
    if (rc != IPX_OK) return rc;
    if (it < h->iter_N) {
      hipLaunchKernelGGL(k_band_update_resid, grid, block, 0, st, m, lv.k, lv.band, w, xa, d, xb, r,
                         guard);
      double *t = xa; xa = xb; xb = t;
    } else {
      hipLaunchKernelGGL(k_band_add, grid, block, 0, st, m, xa, d, x, guard);
    }
    IPX_CHECK_LAUNCH();
  }
  if (partial) return ipx_banded_residual_launch(h, w, x, partial, npartial, guard, st);
  return IPX_OK;
}

// Fast path.  `partial` != NULL additionally yields per-workgroup sums of
// ||w - S x||^2 (needs x != w).
int fast_solve(Banded *h, const double *w, double *x, double *partial, int *npartial,
               const double *guard, hipStream_t st) {
  if (h->nlev == 1) {              // one chunk holds everything: no separators
    int rc = fast_down0(h, w, x, guard, st);
    if (rc != IPX_OK || !partial) return rc;
    return ipx_banded_residual_launch(h, w, x, partial, npartial, guard, st);
  }
  if (h->decoupled && w != x && h->pcr_L > 0 && h->lev[0].k == 1)
    // tridiagonal: parallel cyclic reduction (reads the band only)
    return launch_solve_pcr(to_dev(h->lev[0], nullptr), h->pcr_L, w, x, partial, npartial, guard,
                            st);
  if (h->chunk_pending) {
    const int rc = ensure_chunks(h, st);
    if (rc != IPX_OK) return rc;
    return fast_solve(h, w, x, partial, npartial, guard, st);     // (by the verdict just read)
  }
  if (h->decoupled && w != x) {
    const LevDev lv = to_dev(h->lev[0], nullptr);
    switch (lv.k) {
#define SD(kk) case kk: return launch_solve_decoupled<kk>(lv, w, x, h->rinv, partial, npartial, guard, st)
      SD(1); SD(2); SD(3); SD(4); SD(5); SD(6); SD(7); SD(8);
#undef SD
    }
    return IPX_EINVAL;
  }
  if (h->iter_N > 0 && !h->decoupled) {
    // (x may alias w as the contract of ipx_banded_solve says: the correction steps read w
    // and write scratch, x is only written by the last launch; the residual partials read w
    // after that, so they need the two apart)
    if (w == x && partial) return IPX_EINVAL;
    return iter_solve(h, w, x, partial, npartial, guard, st);
  }
  if (h->wide) return IPX_EUNSUPPORTED;      // no compiled separator level (see ipx_banded_create)
  int rc = fast_down0(h, w, h->ybuf, guard, st);
  if (rc != IPX_OK) return rc;
  if (!decoupled_scalar(h)) {
    if (!h->upper_done) {           // deferred by ipx_banded_factor (decoupled candidate)
      rc = factor_upper(h, st);
      if (rc != IPX_OK) return rc;
    }
    rc = fast_middle(h, guard, st);
    if (rc != IPX_OK) return rc;
  }
  switch (h->lev[0].k) {
#define CO(kk) case kk: return launch_correct_oop<kk>(h, x, w, partial, npartial, guard, st)
    CO(1); CO(2); CO(3); CO(4); CO(5); CO(6); CO(7); CO(8);
#undef CO
  }
  return IPX_EINVAL;
}
}  // namespace

// How many residual partials ipx_banded_solve_resid_launch writes for this
// factorization (depends on the path the solve takes).
int ipx_banded_resid_count(void *handle) {
  if (!handle) return 0;
  Banded *h = (Banded *)handle;
  const Level &l0 = h->lev[0];
  if (h->fast && h->nlev > 1 && h->decoupled)
    return (l0.P + DEC_CHUNKS - 1) / DEC_CHUNKS;
  if (h->fast && h->nlev > 1 && h->iter_N == 0) return (l0.m + IPX_BLOCK - 1) / IPX_BLOCK;
  const int grid = (l0.m + IPX_BLOCK - 1) / IPX_BLOCK;           // k_band_residual
  return grid > 256 ? 256 : grid;
}

// Geometry of the single-launch decoupled solve: out[0] = constraint rows per workgroup,
// out[1] = workgroups; returns 1 when that path is the one solves take, else 0.
extern "C" int ipx_banded_decoupled_geometry(void *handle, int32_t *out) {
  if (!handle || !out) return 0;
  Banded *h = (Banded *)handle;
  const Level &l0 = h->lev[0];
  if (!(h->fast && h->nlev > 1 && h->decoupled))
    return 0;
  out[0] = DEC_CHUNKS * l0.q;
  out[1] = (l0.P + DEC_CHUNKS - 1) / DEC_CHUNKS;
  return 1;
}

// Decoupled solve + residual partials + g = r_in - A'v (see AtvJob) in one launch.
// IPX_EINVAL when the factorization is not on the decoupled path.
int ipx_banded_solve_resid_atv_launch(void *handle, const double *w, double *x, double *partial,
                                      int *npartial, const int32_t *At_rowptr,
                                      const int32_t *At_colidx, const double *At_val,
                                      const double *r_in, double *g_out, const int32_t *vown,
                                      int qv, double *part3, const double *guard,
                                      hipStream_t st, const uint16_t *ell_row,
                                      const double *ell_val, int64_t ell_n, double sign) {
  if (!handle || !w || !x || !partial || w == x || (sign != 1.0 && sign != -1.0)) return IPX_EINVAL;
  int32_t geo[2];
  if (!ipx_banded_decoupled_geometry(handle, geo)) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  const AtvJob job{At_rowptr, At_colidx, At_val, r_in, g_out, vown, part3, ell_row, ell_val, ell_n,
                   sign};
  if (h->pcr_L > 0 && ell_row && ell_val)
    return launch_solve_pcr(to_dev(h->lev[0], nullptr), h->pcr_L, w, x, partial, npartial, guard,
                            st, &job, qv);
  if (h->chunk_pending) {
    const int rc = ensure_chunks(h, st);
    if (rc != IPX_OK) return rc;
    if (!ipx_banded_decoupled_geometry(handle, geo)) return IPX_EINVAL;
  }
  const LevDev lv = to_dev(h->lev[0], nullptr);
  switch (lv.k) {
    case 1: return launch_solve_decoupled<1>(lv, w, x, h->rinv, partial, npartial, guard, st, &job, qv);
    case 2: return launch_solve_decoupled<2>(lv, w, x, h->rinv, partial, npartial, guard, st, &job, qv);
    case 3: return launch_solve_decoupled<3>(lv, w, x, h->rinv, partial, npartial, guard, st, &job, qv);
    case 4: return launch_solve_decoupled<4>(lv, w, x, h->rinv, partial, npartial, guard, st, &job, qv);
  }
  return IPX_EINVAL;
}

// The cyclic-reduction form of the single-launch solve as the resident CG kernel
// (csrc/resident.hip) needs it: rows per workgroup, workgroups, levels, the band.  0 when the
// factorization's solves take another path.
int ipx_banded_pcr_view(void *handle, ipx_pcr_view *out) {
  if (!handle || !out) return 0;
  int32_t g2[2];
  if (!ipx_banded_decoupled_geometry(handle, g2)) return 0;
  Banded *h = (Banded *)handle;
  if (h->pcr_L <= 0) return 0;
  out->m = h->lev[0].m;
  out->rows_wg = g2[0];
  out->nwg = g2[1];
  out->L = h->pcr_L;
  out->band = h->lev[0].band;
  return 1;
}

// Solve + residual partials in one go (the CG loop's projection step).
int ipx_banded_solve_resid_launch(void *handle, const double *w, double *x, double *partial,
                                  int *npartial, const double *guard, hipStream_t st) {
  if (!handle || !w || !x || !partial || w == x) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  if (h->fast) return fast_solve(h, w, x, partial, npartial, guard, st);
  int rc = ipx_banded_solve_guarded(handle, w, x, guard, st);
  if (rc != IPX_OK) return rc;
  return ipx_banded_residual_launch(handle, w, x, partial, npartial, guard, st);
}

// ipx_banded_solve_resid_launch with w = (rows) x formed inside the solve kernel: only on the
// cyclic-reduction path (tridiagonal, decoupled); IPX_EUNSUPPORTED otherwise -- the caller then
// forms w by an SpMV and calls the plain entry.
int ipx_banded_solve_rows_launch(void *handle, const int32_t *col, const double *val,
                                 const double *xin, int logL, double *x, double *partial,
                                 int *npartial, const double *guard, hipStream_t st,
                                 const ipx_post_job *pj) {
  if (!handle || !col || !val || !xin || !x || !partial || logL < 0 || logL > 4) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  if (!(h->fast && h->nlev > 1 && h->decoupled && h->pcr_L > 0)) return IPX_EUNSUPPORTED;
  const LevDev lv = to_dev(h->lev[0], nullptr);
  if (lv.k != 1) return IPX_EUNSUPPORTED;
  if (!pj)
    return launch_solve_pcr_rows(lv, h->pcr_L, RowsJob{col, val, xin, logL}, x, partial, npartial,
                                 guard, st);
  // the tail's tables were laid out for ONE geometry of the solve (rows per workgroup, level).
  // reach: v_R of an item's second general row is read out of the window's LDS copy, of which
  // only the FIRST halo row past the own rows is exact (the window is cut with identity padding:
  // halo row j carries an error that grows with j) -- the QV tail's own bound.
  // count: workgroup b writes part[b] and part[count + b]; with more workgroups than the
  // consumer folds entries per half those stores would land in the other half / past the end
  // (boxed problems with few variables per general row): the separate launch then.
  if (pj->rows_wg != DEC_CHUNKS * lv.q || pj->reach > 1 || (1 << h->pcr_L) < pj->reach)
    return IPX_EUNSUPPORTED;
  if ((lv.P + DEC_CHUNKS - 1) / DEC_CHUNKS > pj->count) return IPX_EUNSUPPORTED;
  PostJob post{pj->own_g, pj->own_e, pj->ng, pj->nitems, pj->T, pj->yrow, pj->yval, pj->r, pj->g,
               pj->part, pj->count};
  return launch_solve_pcr_rows(lv, h->pcr_L, RowsJob{col, val, xin, logL}, x, partial, npartial,
                               guard, st, &post);
}

int ipx_banded_solve_guarded(void *handle, const double *w, double *x, const double *guard,
                             hipStream_t st) {
  if (!handle || !w || !x) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  if (h->fast) return fast_solve(h, w, x, nullptr, nullptr, guard, st);
  if (h->wide) return IPX_EUNSUPPORTED;
  if (!h->upper_done) {             // deferred by ipx_banded_factor (decoupled candidate)
    int rc = factor_upper(h, st);
    if (rc != IPX_OK) return rc;
  }
  for (int li = 0; li < h->nlev; ++li) {
    const double *in = li == 0 ? w : h->lev[li].rhs;
    double *out = li == 0 ? x : h->lev[li].sol;
    int rc = level_down(h, li, in, out, guard, st);
    if (rc != IPX_OK) return rc;
  }
  for (int li = h->nlev - 2; li >= 0; --li) {
    double *out = li == 0 ? x : h->lev[li].sol;
    int rc = level_up(h, li, out, guard, st);
    if (rc != IPX_OK) return rc;
  }
  return IPX_OK;
}

// Residual of the normal equations in constraint space:  partial sums of
// ||w - S v||^2, one per workgroup.  Since g = r - A'v, this IS ||A g||^2 of the
// reference's orthogonality test (projections.py:52) -- A g = A r - (A A') v =
// w - S v -- evaluated on m-vectors and the band of S (3 MB at m = 1e5) instead
// of a second 27 MB pass over A.  Differs from the SpMV form by rounding only
// (~1e-16 ||A|| ||g||, against a 1e-12 threshold).
__global__ void __launch_bounds__(IPX_BLOCK)
k_band_residual(int m, int k, const double *__restrict__ band, const double *__restrict__ w,
                const double *__restrict__ v, double *__restrict__ partial,
                const double *__restrict__ guard) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  if (guard && *guard != 0.0) return;
  double acc = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    double s = band[i] * v[i];
    for (int d = 1; d <= k; ++d) {
      if (i - d >= 0) s += band[(int64_t)d * m + i] * v[i - d];
      if (i + d < m) s += band[(int64_t)d * m + i + d] * v[i + d];
    }
    const double res = w[i] - s;
    acc += res * res;
  }
  const double a = ipx_block_reduce<IPX_SUM>(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}

int ipx_banded_residual_launch(void *handle, const double *w, const double *v, double *partial,
                               int *npartial, const double *guard, hipStream_t st) {
  if (!handle || !w || !v || !partial) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  const Level &l0 = h->lev[0];
  int grid = (l0.m + IPX_BLOCK - 1) / IPX_BLOCK;
  if (grid > 256) grid = 256;
  if (npartial) *npartial = grid;
  hipLaunchKernelGGL(k_band_residual, dim3(grid), dim3(IPX_BLOCK), 0, st, l0.m, l0.k, l0.band, w, v,
                     partial, guard);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// ipx_banded_solve that turns into a no-op when *guard != 0 (device stop flag
// of the CG loops).
extern "C" int ipx_banded_solve_guarded_c(void *handle, const double *w, double *x,
                                          const double *guard, void *stream) {
  return ipx_banded_solve_guarded(handle, w, x, guard, (hipStream_t)stream);
}

// Solve plus the per-workgroup partial sums of ||w - (A A') x||^2 (the
// constraint-space form of the orthogonality test);
// partial needs ceil(m / 256) doubles, *npartial receives how many were written.
extern "C" int ipx_banded_solve_resid(void *handle, const double *w, double *x, double *partial,
                                      int32_t *npartial, const double *guard, void *stream) {
  if (!partial || !npartial) return IPX_EINVAL;
  int np = 0;
  int rc = ipx_banded_solve_resid_launch(handle, w, x, partial, &np, guard, (hipStream_t)stream);
  *npartial = np;
  return rc;
}

// The original one-kernel-per-level sweep (kept for cross-checking the fast
// path and as the fallback when a level does not fit in LDS).
extern "C" int ipx_banded_solve_multilaunch(void *handle, const double *w, double *x,
                                            void *stream) {
  if (!handle || !w || !x) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  if (h->chunk_pending) {
    const int rc = ensure_chunks(h, (hipStream_t)stream);
    if (rc != IPX_OK) return rc;
  }
  const bool keep = h->fast;
  h->fast = false;
  int rc = ipx_banded_solve_guarded(handle, w, x, nullptr, (hipStream_t)stream);
  h->fast = keep;
  return rc;
}
