// Streaming vector kernels: elementwise algebra and fixed-order reductions.
// HBM-bound; 16-byte (double2) accesses when every pointer is 16-B aligned,
// 8-byte otherwise (views into z = [x; s] start at arbitrary offsets).
#include "ipx_common.h"

namespace {

__host__ inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------- elementwise ------------------------------------------
struct OpAxpby {
  double a, b; const double *x, *y;
  __device__ double operator()(int64_t i) const {
    double r = a * x[i];
    return y ? r + b * y[i] : r;
  }
};
struct OpMul {
  const double *x, *y;
  __device__ double operator()(int64_t i) const { return x[i] * y[i]; }
};
struct OpFill {
  double v;
  __device__ double operator()(int64_t) const { return v; }
};
struct OpClip {
  const double *x, *lb, *ub;
  __device__ double operator()(int64_t i) const {
    // np.minimum(np.maximum(x, lb), ub)
    return fmin(fmax(x[i], lb[i]), ub[i]);
  }
};
struct OpAffine {
  double a, b; const double *x;
  __device__ double operator()(int64_t i) const { return a * x[i] + b; }
};

struct OpGather {   // out[i] = sign[i] * (x[idx[i]] - shift[i])
  const double *x; const int32_t *idx; const double *sign, *shift;
  __device__ double operator()(int64_t i) const {
    double v = x[idx[i]];
    if (shift) v -= shift[i];
    return sign ? sign[i] * v : v;
  }
};

struct OpMaxScalar {   // np.maximum(x, c)
  const double *x; double c;
  __device__ double operator()(int64_t i) const { return fmax(x[i], c); }
};
struct OpWherePos {    // np.where(v > 0, a, c)
  const double *v, *a; double c;
  __device__ double operator()(int64_t i) const { return v[i] > 0.0 ? a[i] : c; }
};
struct OpNegMasked {   // s[mask] = -c[mask], other entries unchanged
  const double *s, *mask, *c;
  __device__ double operator()(int64_t i) const { return mask[i] != 0.0 ? -c[i] : s[i]; }
};

__global__ void __launch_bounds__(IPX_BLOCK)
k_scatter_add(int64_t n, const double *__restrict__ x, const int32_t *__restrict__ idx,
              double *out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[idx[i]] += x[i];
}

__global__ void __launch_bounds__(IPX_BLOCK)
k_scatter(int64_t n, const double *__restrict__ x, const int32_t *__restrict__ idx, double *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[idx[i]] = x[i];
}

template <typename F>
__global__ void __launch_bounds__(IPX_BLOCK) k_map(int64_t n, F f, double *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = f(i);
}

// axpby specialisation with 16-byte accesses (the hot elementwise op).
__global__ void __launch_bounds__(IPX_BLOCK)
k_axpby2(int64_t n2, double a, const double2 *x, double b, const double2 *y, double2 *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 xv = x[i], r;
    r.x = a * xv.x; r.y = a * xv.y;
    if (y) { double2 yv = y[i]; r.x += b * yv.x; r.y += b * yv.y; }
    out[i] = r;
  }
}

template <typename F>
int launch_map(int64_t n, F f, double *out, void *stream) {
  if (n <= 0) return IPX_OK;
  int grid = ipx_grid_for(n, IPX_BLOCK * 4);
  hipLaunchKernelGGL(k_map<F>, dim3(grid), dim3(IPX_BLOCK), 0, (hipStream_t)stream, n, f, out);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// ---------------- reductions -------------------------------------------
// Stage 1: each block reduces NQ quantities over a grid-stride range and
// writes partial[q * gridDim.x + blockIdx.x].  Stage 2 (one block) folds the
// partials in index order into out[q].
template <int NQ> struct Acc { double v[NQ]; };

struct RedDot {
  static constexpr int NQ = 1;
  const double *x, *y;
  __device__ static int op(int) { return IPX_SUM; }
  __device__ void init(Acc<1> &a) const { a.v[0] = 0.0; }
  __device__ void step(Acc<1> &a, int64_t i) const { a.v[0] += x[i] * y[i]; }
};
struct RedNorms {   // sum of squares, max |x|
  static constexpr int NQ = 2;
  const double *x;
  __device__ void init(Acc<2> &a) const { a.v[0] = 0.0; a.v[1] = 0.0; }
  __device__ void step(Acc<2> &a, int64_t i) const {
    double t = x[i]; a.v[0] += t * t; a.v[1] = fmax(a.v[1], fabs(t));
  }
};
struct RedBoxInside {   // number of coordinates outside [lb, ub]
  static constexpr int NQ = 1;
  const double *x, *lb, *ub;
  __device__ void init(Acc<1> &a) const { a.v[0] = 0.0; }
  __device__ void step(Acc<1> &a, int64_t i) const {
    double t = x[i];
    // not (lb <= x) or not (x <= ub); NaN counts as outside like numpy's .all()
    bool in = (lb[i] <= t) && (t <= ub[i]);
    a.v[0] += in ? 0.0 : 1.0;
  }
};
struct RedBoxSphere {   // see ipx.h
  static constexpr int NQ = 7;
  const double *z, *d, *lb, *ub; double dscale;
  __device__ void init(Acc<7> &a) const {
    a.v[0] = a.v[1] = a.v[2] = 0.0;
    a.v[3] = -__builtin_inf(); a.v[4] = __builtin_inf();
    a.v[5] = 0.0; a.v[6] = 0.0;
  }
  __device__ void step(Acc<7> &a, int64_t i) const {
    const double zi = z[i], di = dscale * d[i];
    a.v[0] += di * di; a.v[1] += zi * di; a.v[2] += zi * zi;
    const double lo = lb ? lb[i] : -__builtin_inf();
    const double hi = ub ? ub[i] : __builtin_inf();
    if (di == 0.0) {
      a.v[5] += (zi < lo || zi > hi) ? 1.0 : 0.0;
    } else {
      const double tl = (lo - zi) / di, tu = (hi - zi) / di;
      a.v[3] = fmax(a.v[3], fmin(tl, tu));
      a.v[4] = fmin(a.v[4], fmax(tl, tu));
      a.v[6] += 1.0;
    }
  }
};

struct RedSumLog {   // sum(log s) and the number of s <= 0 (tr_interior_point.py:93)
  static constexpr int NQ = 2;
  const double *s;
  __device__ void init(Acc<2> &a) const { a.v[0] = 0.0; a.v[1] = 0.0; }
  __device__ void step(Acc<2> &a, int64_t i) const {
    const double t = s[i];
    if (t > 0.0) a.v[0] += log(t); else a.v[1] += 1.0;
  }
};

template <typename R> struct RedOps;
template <> struct RedOps<RedSumLog> { __device__ static constexpr int op(int) { return IPX_SUM; } };
template <> struct RedOps<RedDot> { __device__ static constexpr int op(int) { return IPX_SUM; } };
template <> struct RedOps<RedNorms> { __device__ static constexpr int op(int q) { return q == 1 ? IPX_MAX : IPX_SUM; } };
template <> struct RedOps<RedBoxInside> { __device__ static constexpr int op(int) { return IPX_SUM; } };
template <> struct RedOps<RedBoxSphere> {
  __device__ static constexpr int op(int q) { return q == 3 ? IPX_MAX : (q == 4 ? IPX_MIN : IPX_SUM); }
};

template <int OPC>
__device__ __forceinline__ double reduce_dyn(double v, double *lds) { return ipx_block_reduce<OPC>(v, lds); }

template <typename R, int Q>
struct FoldBlock {
  __device__ static void run(const Acc<R::NQ> &a, double *lds, double *dst, int stride, int idx) {
    constexpr int op = RedOps<R>::op(Q);
    double r = ipx_block_reduce<op>(a.v[Q], lds);
    if (threadIdx.x == 0) dst[Q * stride + idx] = r;
    if constexpr (Q + 1 < R::NQ) FoldBlock<R, Q + 1>::run(a, lds, dst, stride, idx);
  }
};

template <typename R>
__global__ void __launch_bounds__(IPX_BLOCK) k_reduce1(int64_t n, R r, double *partial) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  Acc<R::NQ> a; r.init(a);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    r.step(a, i);
  FoldBlock<R, 0>::run(a, lds, partial, gridDim.x, blockIdx.x);
}

template <typename R, int Q>
struct FoldFinal {
  __device__ static void run(const double *partial, int count, double *lds, double *out) {
    constexpr int op = RedOps<R>::op(Q);
    double r = ipx_sum_partials<op>(partial + (int64_t)Q * count, count, lds);
    if (threadIdx.x == 0) out[Q] = r;
    if constexpr (Q + 1 < R::NQ) FoldFinal<R, Q + 1>::run(partial, count, lds, out);
  }
};

template <typename R>
__global__ void __launch_bounds__(IPX_BLOCK) k_reduce2(const double *partial, int count, double *out) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  FoldFinal<R, 0>::run(partial, count, lds, out);
}

template <typename R>
int launch_reduce(int64_t n, R r, double *out, double *ws, void *stream) {
  if (!out || !ws) return IPX_EINVAL;
  int grid = ipx_grid_for(n, IPX_BLOCK * 4);
  static_assert(R::NQ * IPX_VEC_GRID_CAP <= IPX_WS_DOUBLES, "workspace too small");
  hipLaunchKernelGGL(k_reduce1<R>, dim3(grid), dim3(IPX_BLOCK), 0, (hipStream_t)stream, n, r, ws);
  IPX_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_reduce2<R>, dim3(1), dim3(IPX_BLOCK), 0, (hipStream_t)stream, ws, grid, out);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// the first stage alone: `grid` partials per quantity at part[q * grid + block] (the fold is
// left to the read-back: ipx_read_folded)
template <typename R>
int launch_reduce_partials(int64_t n, R r, double *part, void *stream) {
  if (!part) return IPX_EINVAL;
  const int grid = ipx_grid_for(n, IPX_BLOCK * 4);
  hipLaunchKernelGGL(k_reduce1<R>, dim3(grid), dim3(IPX_BLOCK), 0, (hipStream_t)stream, n, r, part);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

}  // namespace

extern "C" {

// ipx_dot / ipx_norms without their second launch: the partial sums of the first stage, one per
// workgroup and quantity (ipx_reduce_grid(n) workgroups; norms: sum of squares, then max |x|),
// for a caller that folds them where it reads them (ipx_read_folded: the same fixed order).
int ipx_reduce_grid(int64_t n) { return ipx_grid_for(n, IPX_BLOCK * 4); }
int ipx_dot_partials(int64_t n, const double *x, const double *y, double *part, void *stream) {
  if (n <= 0 || !x || !y) return IPX_EINVAL;
  return launch_reduce_partials(n, RedDot{x, y}, part, stream);
}
int ipx_norms_partials(int64_t n, const double *x, double *part, void *stream) {
  if (n <= 0 || !x) return IPX_EINVAL;
  return launch_reduce_partials(n, RedNorms{x}, part, stream);
}

int ipx_axpby(int64_t n, double a, const double *x, double b, const double *y,
              double *out, void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !x || !out) return IPX_EINVAL;
  if (b == 0.0) y = nullptr;
  if ((n & 1) == 0 && aligned16(x) && aligned16(out) && (!y || aligned16(y))) {
    int grid = ipx_grid_for(n / 2, IPX_BLOCK * 2);
    hipLaunchKernelGGL(k_axpby2, dim3(grid), dim3(IPX_BLOCK), 0, (hipStream_t)stream, n / 2, a,
                       (const double2 *)x, b, (const double2 *)y, (double2 *)out);
    IPX_CHECK_LAUNCH();
    return IPX_OK;
  }
  return launch_map(n, OpAxpby{a, b, x, y}, out, stream);
}

int ipx_mul(int64_t n, const double *x, const double *y, double *out, void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !x || !y || !out) return IPX_EINVAL;
  return launch_map(n, OpMul{x, y}, out, stream);
}

int ipx_fill(int64_t n, double value, double *out, void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !out) return IPX_EINVAL;
  return launch_map(n, OpFill{value}, out, stream);
}

int ipx_clip(int64_t n, const double *x, const double *lb, const double *ub, double *out,
             void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !x || !lb || !ub || !out) return IPX_EINVAL;
  return launch_map(n, OpClip{x, lb, ub}, out, stream);
}

int ipx_affine(int64_t n, double a, const double *x, double b, double *out, void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !x || !out) return IPX_EINVAL;
  return launch_map(n, OpAffine{a, b, x}, out, stream);
}

int ipx_gather(int64_t n, const double *x, const int32_t *idx, const double *sign,
               const double *shift, double *out, void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !x || !idx || !out) return IPX_EINVAL;
  return launch_map(n, OpGather{x, idx, sign, shift}, out, stream);
}

int ipx_scatter(int64_t n, const double *x, const int32_t *idx, double *out, void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !x || !idx || !out) return IPX_EINVAL;
  if (n == 0) return IPX_OK;
  hipLaunchKernelGGL(k_scatter, dim3(ipx_grid_for(n, IPX_BLOCK * 4)), dim3(IPX_BLOCK), 0,
                     (hipStream_t)stream, n, x, idx, out);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// out[idx[i]] += x[i]; idx must not repeat (no atomics).  Merges the values of
// several sparse terms into their union pattern (backend_hip.hessian_operator).
int ipx_scatter_add(int64_t n, const double *x, const int32_t *idx, double *out, void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !x || !idx || !out) return IPX_EINVAL;
  hipLaunchKernelGGL(k_scatter_add, dim3(ipx_grid_for(n, IPX_BLOCK * 4)), dim3(IPX_BLOCK), 0,
                     (hipStream_t)stream, n, x, idx, out);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

int ipx_max_scalar(int64_t n, const double *x, double c, double *out, void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !x || !out) return IPX_EINVAL;
  return launch_map(n, OpMaxScalar{x, c}, out, stream);
}

int ipx_where_positive(int64_t n, const double *v, const double *a, double c, double *out,
                       void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !v || !a || !out) return IPX_EINVAL;
  return launch_map(n, OpWherePos{v, a, c}, out, stream);
}

int ipx_assign_negated_where(int64_t n, double *s, const double *mask, const double *c,
                             void *stream) {
  if (n == 0) return IPX_OK;
  if (n < 0 || !s || !mask || !c) return IPX_EINVAL;
  return launch_map(n, OpNegMasked{s, mask, c}, s, stream);
}

int ipx_sum_log(int64_t n, const double *s, double *out, double *ws, void *stream) {
  if (n < 0 || (n > 0 && !s)) return IPX_EINVAL;
  return launch_reduce(n, RedSumLog{s}, out, ws, stream);
}

int ipx_dot(int64_t n, const double *x, const double *y, double *out, double *ws, void *stream) {
  if (n < 0 || (n > 0 && (!x || !y))) return IPX_EINVAL;
  return launch_reduce(n, RedDot{x, y}, out, ws, stream);
}

int ipx_norms(int64_t n, const double *x, double *out, double *ws, void *stream) {
  if (n < 0 || (n > 0 && !x)) return IPX_EINVAL;
  return launch_reduce(n, RedNorms{x}, out, ws, stream);
}

int ipx_box_inside(int64_t n, const double *x, const double *lb, const double *ub, double *out,
                   double *ws, void *stream) {
  if (n < 0 || (n > 0 && (!x || !lb || !ub))) return IPX_EINVAL;
  return launch_reduce(n, RedBoxInside{x, lb, ub}, out, ws, stream);
}

int ipx_box_sphere_reduce(int64_t n, const double *z, const double *d, double dscale,
                          const double *lb, const double *ub, double *out, double *ws,
                          void *stream) {
  if (n < 0 || (n > 0 && (!z || !d))) return IPX_EINVAL;
  return launch_reduce(n, RedBoxSphere{z, d, lb, ub, dscale}, out, ws, stream);
}

}  // extern "C"
