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
  std::vector<void *> allocs;
};

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
    if (!(dj > 0.0)) atomicOr(flag, 1);
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

// ---- S = A A' in band storage: one lane per (row i, offset d), merge join
// of the two sorted CSR rows perm[i] and perm[i-d].
__global__ void __launch_bounds__(IPX_BLOCK)
k_aat_band(int m, int k, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
           const double *__restrict__ val, const int32_t *__restrict__ perm,
           double *__restrict__ band) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)m * (k + 1)) return;
  const int d = (int)(idx / m), i = (int)(idx % m);
  double s = 0.0;
  if (i - d >= 0) {
    const int r1 = perm ? perm[i] : i, r2 = perm ? perm[i - d] : i - d;
    int p = rowptr[r1], pe = rowptr[r1 + 1];
    int u = rowptr[r2], ue = rowptr[r2 + 1];
    while (p < pe && u < ue) {
      const int cp = colidx[p], cu = colidx[u];
      if (cp == cu) { s += val[p] * val[u]; ++p; ++u; }
      else if (cp < cu) ++p;
      else ++u;
    }
  }
  band[(int64_t)d * m + i] = s;
}

template <int K>
struct Launch {
  static int factor(Banded *h, int li, hipStream_t st) {
    Level &lv = h->lev[li];
    int grid = (lv.P + IPX_WAVE - 1) / IPX_WAVE;
    hipLaunchKernelGGL(k_factor_chunks<K>, dim3(grid), dim3(IPX_WAVE), 0, st, lv.m, lv.c, lv.P,
                       lv.band, lv.Dinv, lv.L, lv.V, lv.W, h->flag);
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

// Plan the level hierarchy for an m x m SPD matrix of half bandwidth k.
// `chunk` = interior rows per chunk (0 = default).  Returns a handle or NULL.
void *ipx_banded_create(int64_t m64, int32_t k, int32_t chunk) {
  if (m64 < 1 || m64 > (1LL << 30) || k < 0 || k > KMAX) return nullptr;
  if (k == 0) k = 1;                 // diagonal matrices ride the k = 1 path
  Banded *h = new Banded();
  h->nlev = 0;
  int m = (int)m64, kk = k;
  if (chunk <= 0) chunk = 32;
  bool ok = true;
  while (true) {
    if (h->nlev >= MAX_LEVELS || kk > KMAX) { ok = false; break; }
    Level &lv = h->lev[h->nlev++];
    lv.m = m; lv.k = kk;
    lv.c = chunk < kk ? kk : chunk;
    // The separator system of a level has half bandwidth 2k-1.  When that
    // would exceed the compiled kernels, stop recursing: this level becomes
    // one chunk swept by a single lane (only reached for wide bands, where
    // the level is already small).
    if (2 * kk - 1 > KMAX && m > lv.c + kk) lv.c = m;
    lv.q = lv.c + kk;
    // single chunk when everything fits in one (<= q rows)
    lv.P = (m <= lv.q) ? 1 : (m + lv.q - 1) / lv.q;
    // the last chunk must keep >= 1 interior row: (P-1)*q < m holds by ceil
    lv.mR = (lv.P - 1) * kk;
    lv.band = nullptr;
    const int steps = lv.c + kk;
    lv.Dinv = dalloc<double>(h, (size_t)steps * lv.P);
    lv.L = dalloc<double>(h, (size_t)steps * kk * lv.P);
    lv.V = dalloc<double>(h, (size_t)m * kk);
    lv.W = dalloc<double>(h, (size_t)m * kk);
    lv.rhs = lv.sol = nullptr;
    if (h->nlev > 1) {
      lv.band = dalloc<double>(h, (size_t)(kk + 1) * m);
      lv.rhs = dalloc<double>(h, m);
      lv.sol = dalloc<double>(h, m);
      if (!lv.band || !lv.rhs || !lv.sol) { ok = false; break; }
    }
    if (!lv.Dinv || !lv.L || !lv.V || !lv.W) { ok = false; break; }
    if (lv.P == 1) break;
    m = lv.mR;
    kk = 2 * kk - 1;
  }
  h->flag = dalloc<int>(h, 1);
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
int ipx_banded_factor(void *handle, const double *band, void *stream) {
  if (!handle || !band) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(h->flag, 0, sizeof(int), st) != hipSuccess) return IPX_ELAUNCH;
  h->lev[0].band = const_cast<double *>(band);
  for (int li = 0; li < h->nlev; ++li) {
    int rc = level_factor(h, li, st);
    if (rc != IPX_OK) return rc;
  }
  return IPX_OK;
}

// Blocking read of the pivot flag: IPX_OK or IPX_ENOTSPD.
int ipx_banded_status(void *handle, void *stream) {
  if (!handle) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
  int f = 0;
  if (hipMemcpyAsync(&f, h->flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) !=
      hipSuccess)
    return IPX_ELAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return IPX_ELAUNCH;
  return f ? IPX_ENOTSPD : IPX_OK;
}

// x = S^-1 w.  w and x are length m; x may alias w.
int ipx_banded_solve(void *handle, const double *w, double *x, void *stream) {
  return ipx_banded_solve_guarded(handle, w, x, nullptr, (hipStream_t)stream);
}

// band[d*m+i] = (A A')[pi, p(i-d)] with rows taken in the order perm (NULL =
// identity); reference: the matrix CHOLMOD factors in projections.py:62.
int ipx_aat_band(int64_t m, int32_t k, const int32_t *rowptr, const int32_t *colidx,
                 const double *val, const int32_t *perm, double *band, void *stream) {
  if (m < 0 || k < 0 || !rowptr || !band) return IPX_EINVAL;
  if (m == 0) return IPX_OK;
  int64_t tot = m * (k + 1);
  hipLaunchKernelGGL(k_aat_band, dim3((unsigned)((tot + IPX_BLOCK - 1) / IPX_BLOCK)),
                     dim3(IPX_BLOCK), 0, (hipStream_t)stream, (int)m, k, rowptr, colidx, val, perm,
                     band);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

}  // extern "C"

int ipx_banded_solve_guarded(void *handle, const double *w, double *x, const double *guard,
                             hipStream_t st) {
  if (!handle || !w || !x) return IPX_EINVAL;
  Banded *h = (Banded *)handle;
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
