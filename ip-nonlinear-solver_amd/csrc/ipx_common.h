// Shared device helpers for the ipx kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ipx.h"

#define IPX_BLOCK 256            // 4 waves: one per SIMD of a CU
#define IPX_WAVE 64
#define IPX_VEC_GRID_CAP 1024    // grid-stride cap for streaming kernels (4 WG/CU)

// Records the HIP error text for ipx_last_error() (defined in misc.hip).
void ipx_note_error(hipError_t e, const char *file, int line);

#define IPX_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) {                                  \
      ipx_note_error(e_, __FILE__, __LINE__);                \
      return IPX_ELAUNCH;                                    \
    }                                                        \
  } while (0)

static inline int ipx_grid_for(int64_t n, int per_block, int cap = IPX_VEC_GRID_CAP) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ---- fixed-order reductions -------------------------------------------
// Wave: butterfly over 64 lanes (xor 32,16,...,1); every lane ends with the
// same bits.  Block: wave results through LDS, summed in wave order.
__device__ __forceinline__ double ipx_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, IPX_WAVE);
  return v;
}
__device__ __forceinline__ double ipx_wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, IPX_WAVE));
  return v;
}
__device__ __forceinline__ double ipx_wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, IPX_WAVE));
  return v;
}

enum { IPX_SUM = 0, IPX_MAX = 1, IPX_MIN = 2 };

template <int OP>
__device__ __forceinline__ double ipx_combine(double a, double b) {
  if (OP == IPX_SUM) return a + b;
  if (OP == IPX_MAX) return fmax(a, b);
  return fmin(a, b);
}
template <int OP>
__device__ __forceinline__ double ipx_identity() {
  if (OP == IPX_SUM) return 0.0;
  if (OP == IPX_MAX) return -__builtin_inf();
  return __builtin_inf();
}
template <int OP>
__device__ __forceinline__ double ipx_wave_reduce(double v) {
  if (OP == IPX_SUM) return ipx_wave_sum(v);
  if (OP == IPX_MAX) return ipx_wave_max(v);
  return ipx_wave_min(v);
}

// All threads of the block get the result.  `lds` needs blockDim/64 doubles
// per concurrently reduced quantity; call sites pass distinct slices.
template <int OP>
__device__ __forceinline__ double ipx_block_reduce(double v, double *lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = ipx_wave_reduce<OP>(v);
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double r = lds[0];
  for (int w = 1; w < nw; ++w) r = ipx_combine<OP>(r, lds[w]);
  __syncthreads();
  return r;
}

// Sum `count` partials (written by a previous kernel) in a fixed order that
// is identical in every block, so all blocks derive bit-identical scalars.
template <int OP>
__device__ __forceinline__ double ipx_sum_partials(const double *part, int count,
                                                   double *lds) {
  double v = ipx_identity<OP>();
  for (int i = threadIdx.x; i < count; i += blockDim.x)
    v = ipx_combine<OP>(v, part[i]);
  return ipx_block_reduce<OP>(v, lds);
}

// ---- internal (non-ABI) launchers shared between translation units --------
struct ipx_csr_view {
  int nrows, ncols;
  const int32_t *rowptr, *colidx;
  const double *val;
  const int32_t *tiles;
  int ntiles;
};
int ipx_spmv_launch(const ipx_csr_view &A, const double *x, double alpha, const double *diag,
                    double beta, const double *yin, double *yout, double *partial,
                    const double *guard, hipStream_t st, const double *xrow_override = nullptr);
int ipx_banded_solve_guarded(void *handle, const double *w, double *x, const double *guard,
                             hipStream_t st);
