// One outer iteration of the trust-region SQP method as a chain of launches with its decisions
// taken on the device (reference equality_constrained_sqp.py:102-250; the normal step of
// qp_subproblem.py:320-413, the tangential step of :416-643 through csrc/cg.hip).
//
// The reference's outer loop is scalar logic between operator calls: the norm of the normal
// step sizes the tangential trust region (:125-129), five reductions feed the quadratic model,
// the penalty update and the predicted reduction (:135-153), one more the accept / reject test
// and the trust-radius ladder (:156-212).  Read one by one they are ~18 blocking device-to-host
// copies per iteration; rounds 3-5 packed them into four.  Here an iteration is THREE entry
// points, each ending in a one-workgroup "decide" kernel that folds the partial sums of the
// launches before it and writes a block of scalars (SQ_*) -- the host reads that block once per
// entry and only to learn what it must know to call the user's callbacks:
//
//   ipx_sqp_front    normal step (Newton point of the dogleg, accepted on the device when it is
//                    inside the box and the 0.8-radius ball; otherwise the whole dogleg, on the
//                    device as well), c_t = H dn + c, the shifted bounds, the priming of the
//                    projected CG with the tangential radius taken from the block, its first
//                    batch of iterations, the loop's trust-region / negative-curvature exits
//                    (:565-576, :585-596: intersections, step, clip), then d = dn + dt,
//                    x_next = x + S d, (H d).d, c.d, ||A d + b||, ||d||, ||dt|| and the model
//                    / penalty / predicted reduction
//   ipx_sqp_judge    ||b_next||, actual / predicted, the second-order-correction test, the
//                    trust-radius ladder, accept / reject
//   ipx_sqp_refresh  after an accepted step (and at every entry of the outer loop):
//                    v = -(A A')^-1 A c, ||c + A'v||_inf, ||b||_inf, ||b||, ||A||_F^2 and the
//                    verdict on the factorization just made (csrc/banded.hip deferred status)
//
// The scalar arithmetic of the decisions is ONE set of functions compiled for the device and
// for the host (ipx_sqp_*_host: the host-driven outer loop of backends without these chains --
// dense Jacobians, the row-sharded solver, the CPU tests -- calls the same code, so both forms
// take the same branches on the same numbers).
#include "ipx_common.h"
#include <math.h>
#include <utility>
#include <vector>

// ---- the block ----------------------------------------------------------------------------
// (mirrored by ipsolver/sqp.py; doubles)
enum {
  SQ_RADIUS = 0,        // trust radius: in; out of `radius` (the ladder's result)
  SQ_PENALTY = 1,       // penalty: in; raised by `model`, restored by `radius` on a rejection
  SQ_F = 2,             // f(x)
  SQ_NORM_B = 3,        // ||b(x)||
  SQ_NORM_DN = 4,       // ||dn||
  SQ_RADIUS_T = 5,      // sqrt(radius^2 - ||dn||^2)
  SQ_NORMAL_KIND = 6,   // 1: the Newton point was accepted; 2: the dogleg ran; 0: the host must
  SQ_NVIOL = 7,         // box violations of the Newton point
  SQ_HDD = 8, SQ_CD = 9, SQ_LIN = 10, SQ_NORM_D = 11, SQ_NORM_DT = 12,
  SQ_QMODEL = 13, SQ_VPRED = 14, SQ_PREV_PENALTY = 15, SQ_PRED = 16, SQ_MERIT = 17,
  SQ_F_NEXT = 18, SQ_NORM_B_NEXT = 19, SQ_ACTUAL = 20, SQ_RATIO = 21,
  SQ_SOC = 22,          // 1: the second-order correction is due (the host's: rare)
  SQ_ACCEPT = 23,
  SQ_OPT = 24, SQ_VIOL = 25, SQ_NORM_A2 = 26,
  SQ_FACTOR_BAD = 27,   // != 0: the deferred verdict on the factorization differs from the one
                        // the solves were enqueued under (the host repeats it, blocking)
  SQ_EXIT_TAU = 28,     // the step along p taken by the CG loop's boundary exit (diagnostic)
  SQ_EXIT_DONE = 29,    // 1: the exit of stop code 2 / 3 was finished on the device
  SQ_X_OUTSIDE = 30,    // box violations of the final CG iterate (:636-638)
  SQ_PRIME_STEPS = 31,  // correction steps the priming's projections took on the device (0..2)
  SQ_CG = 32,           // the CG loop's state block (ST_*, 16 doubles) as of the model kernel
  SQ_DOGLEG = 48,       // scalars of the device dogleg (diagnostics): coef, alphas, norms
  SQ_SIZE = 64
};

// constants of equality_constrained_sqp.py:50-60
#define SQC_PENALTY_FACTOR 0.3
#define SQC_LARGE 0.9
#define SQC_INTERMEDIARY 0.3
#define SQC_SUFFICIENT 1e-8
#define SQC_ENLARGE_L 7.0
#define SQC_ENLARGE_S 2.0
#define SQC_MAX_REDUCTION 0.5
#define SQC_MIN_REDUCTION 0.1
#define SQC_SOC_THRESHOLD 0.1

// ---- the decisions (host + device, -ffp-contract=off on both) --------------------------------
// :135-153 -- quadratic model, linearised constraint decrease, penalty, predicted reduction, merit
__host__ __device__ inline void sqp_model(double *q) {
  const double qm = 1.0 / 2.0 * q[SQ_HDD] + q[SQ_CD];
  const double vpred = fmax(1e-16, q[SQ_NORM_B] - q[SQ_LIN]);
  double penalty = q[SQ_PENALTY];
  q[SQ_PREV_PENALTY] = penalty;
  if (qm > 0.0) penalty = fmax(penalty, qm / ((1.0 - SQC_PENALTY_FACTOR) * vpred));
  q[SQ_QMODEL] = qm;
  q[SQ_VPRED] = vpred;
  q[SQ_PENALTY] = penalty;
  q[SQ_PRED] = -qm + penalty * vpred;
  q[SQ_MERIT] = q[SQ_F] + penalty * q[SQ_NORM_B];
}
// :156-173 -- actual against predicted; is the second-order correction due?
__host__ __device__ inline void sqp_ratio(double *q) {
  const double actual = q[SQ_MERIT] - (q[SQ_F_NEXT] + q[SQ_PENALTY] * q[SQ_NORM_B_NEXT]);
  const double ratio = actual / q[SQ_PRED];
  q[SQ_ACTUAL] = actual;
  q[SQ_RATIO] = ratio;
  q[SQ_SOC] = (ratio < SQC_SUFFICIENT && q[SQ_NORM_DN] <= SQC_SOC_THRESHOLD * q[SQ_NORM_DT])
                  ? 1.0 : 0.0;
}
// :196-242 -- the trust-radius ladder, accept / reject (a rejected step takes the penalty back)
__host__ __device__ inline void sqp_radius(double *q) {
  const double ratio = q[SQ_RATIO], norm_d = q[SQ_NORM_D];
  double radius = q[SQ_RADIUS];
  if (ratio >= SQC_LARGE) {
    radius = fmax(SQC_ENLARGE_L * norm_d, radius);
  } else if (ratio >= SQC_INTERMEDIARY) {
    radius = fmax(SQC_ENLARGE_S * norm_d, radius);
  } else if (ratio < SQC_SUFFICIENT) {
    const double reduction = (1.0 - SQC_SUFFICIENT) / (1.0 - ratio);
    const double shrunk = reduction * norm_d;
    if (shrunk >= SQC_MAX_REDUCTION * radius) radius *= SQC_MAX_REDUCTION;
    else if (shrunk >= SQC_MIN_REDUCTION * radius) radius = shrunk;
    else radius *= SQC_MIN_REDUCTION;
  }
  q[SQ_RADIUS] = radius;
  const bool accept = ratio >= SQC_SUFFICIENT;
  q[SQ_ACCEPT] = accept ? 1.0 : 0.0;
  if (!accept) q[SQ_PENALTY] = q[SQ_PREV_PENALTY];
}

// ---- scalar tails of the intersection routines (qp_subproblem.py:99-149, 194-234, 286-296) ----
// from the seven sums of RedBoxSphere: d.d, z.d, z.z, max-of-min, min-of-max, #(d == 0 outside)
struct sq_interval { double ta, tb; bool hit; };
__host__ __device__ inline sq_interval sq_sphere(double dd, double zd, double zz, double radius,
                                                 bool entire_line) {
  if (dd == 0.0) return sq_interval{0.0, 0.0, false};
  const double r2 = radius * radius;
  if (isinf(r2)) {
    if (entire_line) return sq_interval{-HUGE_VAL, HUGE_VAL, true};
    return sq_interval{0.0, 1.0, true};
  }
  const double a = dd, b = 2.0 * zd, c = zz - r2;
  const double disc = b * b - 4.0 * a * c;
  if (disc < 0.0) return sq_interval{0.0, 0.0, false};
  const double aux = b + copysign(sqrt(disc), b);
  const double t1 = -aux / (2.0 * a), t2 = -2.0 * c / aux;
  // sorted([t1, t2]) (a NaN -- aux == 0 -- keeps its place like Python's sort does)
  double ta = t1, tb = t2;
  if (t2 < t1) { ta = t2; tb = t1; }
  if (entire_line) return sq_interval{ta, tb, true};
  if (tb < 0.0 || ta > 1.0) return sq_interval{0.0, 0.0, false};
  return sq_interval{fmax(0.0, ta), fmin(1.0, tb), true};
}
__host__ __device__ inline sq_interval sq_box(double dd, double ta, double tb, double zero_d_outside,
                                              bool entire_line) {
  if (dd == 0.0) return sq_interval{0.0, 0.0, false};
  if (zero_d_outside > 0.0) return sq_interval{0.0, 0.0, false};
  const bool hit = ta <= tb;
  if (!entire_line) {
    if (tb < 0.0 || ta > 1.0) return sq_interval{0.0, 0.0, false};
    ta = fmax(0.0, ta);
    tb = fmin(1.0, tb);
  }
  return sq_interval{ta, tb, hit};
}
// box_sphere_intersections :286-296 from the seven sums r[0..7)
__host__ __device__ inline sq_interval sq_box_sphere(const double *r, double radius,
                                                     bool entire_line) {
  const sq_interval b = sq_box(r[0], r[3], r[4], r[5], entire_line);
  const sq_interval s = sq_sphere(r[0], r[1], r[2], radius, entire_line);
  // np.maximum / np.minimum propagate NaN; fmax / fmin do not: keep numpy's semantics
  double ta = (b.ta != b.ta || s.ta != s.ta) ? NAN : fmax(b.ta, s.ta);
  double tb = (b.tb != b.tb || s.tb != s.tb) ? NAN : fmin(b.tb, s.tb);
  return sq_interval{ta, tb, b.hit && s.hit && ta <= tb};
}

namespace {

constexpr int RB = IPX_BLOCK;
constexpr int RU = 4;                     // elements per lane and trip of the vector kernels

static int sq_grid(int64_t n) { return ipx_grid_for(n, RB * RU); }

// ---- the decide step inside the kernel that produces its sums -------------------------------
// A reduction's workgroups leave their partial sums; ONE of them (sq_folds) folds all of them --
// in the fixed order of ipx_sum_partials: the bits of the one-workgroup kernel that used to
// follow -- and runs the decision.  No fence anywhere (a device-scope release / acquire pair per
// workgroup writes back and invalidates L2: measured, +50 us per launch at n = 1e6): a partial
// travels as ONE 16-byte write-through store that validates itself -- (low word, tag, high
// word, tag), the launch's sequence number as the tag, the format of csrc/resident.hip's
// hand-offs -- and the folding workgroup reads past its caches (sc1) until the tag is there.
typedef unsigned int sq_u4 __attribute__((ext_vector_type(4)));
struct SqSync { unsigned int tag; };
__device__ __forceinline__ void sq_put(sq_u4 *slot, double v, unsigned int tag) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  sq_u4 w;
  w.x = (unsigned)(bits & 0xffffffffull); w.y = tag;
  w.z = (unsigned)(bits >> 32);           w.w = tag;
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot), "v"(w) : "memory");
}
__device__ __forceinline__ double sq_value(const sq_u4 &w) {
  return __longlong_as_double((long long)((unsigned long long)w.x | ((unsigned long long)w.z << 32)));
}
// ipx_sum_partials over tagged partials: the same lanes take the same entries in the same order
// (IPX_FOLD_U granules per lane and trip, requested together, polled until all carry the tag;
// bounded: ~1 s of polling, then NaN -- a decision on NaN fails loudly on the host, a kernel
// that never ends takes the machine with it)
template <int OP>
__device__ __forceinline__ double sq_fold(const sq_u4 *gr, int count, unsigned int tag,
                                          double *lds) {
  static_assert(IPX_FOLD_U == 4, "the load burst below is written for four granules per lane");
  double v = ipx_identity<OP>();
  bool missing = false;
  for (int base = threadIdx.x; base < count; base += IPX_FOLD_U * blockDim.x) {
    const sq_u4 *p0 = gr + base, *p1 = gr + min(base + (int)blockDim.x, count - 1),
                *p2 = gr + min(base + 2 * (int)blockDim.x, count - 1),
                *p3 = gr + min(base + 3 * (int)blockDim.x, count - 1);
    sq_u4 w0, w1, w2, w3;
    bool ok = false;
    for (int spin = 0; spin < (1 << 20) && !ok; ++spin) {
      asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                   "global_load_dwordx4 %1, %5, off sc1\n\t"
                   "global_load_dwordx4 %2, %6, off sc1\n\t"
                   "global_load_dwordx4 %3, %7, off sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                   : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                   : "memory");
      ok = w0.y == tag && w0.w == tag && w1.y == tag && w1.w == tag && w2.y == tag &&
           w2.w == tag && w3.y == tag && w3.w == tag;
      if (!ok) __builtin_amdgcn_s_sleep(1);
    }
    missing = missing || !ok;
    const double t[IPX_FOLD_U] = {sq_value(w0), sq_value(w1), sq_value(w2), sq_value(w3)};
#pragma unroll
    for (int u = 0; u < IPX_FOLD_U; ++u) {
      const int i = base + u * blockDim.x;
      v = ipx_combine<OP>(v, i < count ? t[u] : ipx_identity<OP>());
    }
  }
  const double r = ipx_block_reduce<OP>(v, lds);
  // (a maximum would swallow a NaN operand: the whole result is NaN when any lane gave up)
  return __syncthreads_or(missing ? 1 : 0) ? __builtin_nan("") : r;
}
// Who folds: the workgroup with the highest index -- dispatched behind all others, or at least
// never in their way (it holds one of the device's > 2000 workgroup slots while it polls).  No
// counter: ~1100 workgroups counting themselves off on one word serialise in L2 (measured:
// +8..10 us per launch, more than the launch this saves).
__device__ __forceinline__ bool sq_folds() { return blockIdx.x == gridDim.x - 1; }

// ... and hands the finished block to the host itself: the 64 doubles as self-validating
// granules into the pinned buffer of the read that waits for them (csrc/misc.hip k_publish's
// format), by the workgroup that completed the block -- no publish launch behind the chain.
struct SqPub { unsigned int *dst; unsigned int tag; };
__device__ __forceinline__ void sq_publish(const double *q, SqPub pub) {
  __syncthreads();                               // (thread 0's entries of q: visible)
  if (!pub.dst || threadIdx.x >= SQ_SIZE) return;
  const unsigned long long bits =
      (unsigned long long)__double_as_longlong(
          __hip_atomic_load(q + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  sq_u4 w;
  w.x = (unsigned)(bits & 0xffffffffull); w.y = pub.tag;
  w.z = (unsigned)(bits >> 32);           w.w = pub.tag;
  sq_u4 *d = (sq_u4 *)pub.dst + threadIdx.x;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d), "v"(w) : "memory");
}

// ---- vector kernels ---------------------------------------------------------------------------
// the Newton point's sum of squares (csrc/vec.hip RedNorms' accumulation: the bits of the host
// form's norm) and its violations of [f lb, f ub] (either bound may be NULL):
// part[block] / viol[block]
__device__ __forceinline__ void sq_after_newton(double nn2, double nviol, double radius,
                                                double tr_factor, double *q);
__global__ void __launch_bounds__(RB)
k_sq_newton_check(int64_t n, const double *__restrict__ v, const double *__restrict__ lb,
                  const double *__restrict__ ub, double f, sq_u4 *gran, SqSync sync, int boxed,
                  double radius, double tr_factor, double *__restrict__ q,
                  double *__restrict__ zero_out) {
  __shared__ double lds[RB / IPX_WAVE];
  double cnt = 0.0, s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
    const double t = v[i];
    zero_out[i] = 0.0;          // (the CG's start x0 = 0: one pass over n less than a memset)
    s += t * t;
    if (lb || ub) {
      const double lo = lb ? f * lb[i] : -HUGE_VAL, hi = ub ? f * ub[i] : HUGE_VAL;
      cnt += ((lo <= t) && (t <= hi)) ? 0.0 : 1.0;
    }
  }
  const double a = ipx_block_reduce<IPX_SUM>(s, lds);
  const double r = ipx_block_reduce<IPX_SUM>(cnt, lds);
  const int G = gridDim.x;
  if (threadIdx.x == 0) {
    sq_put(gran + blockIdx.x, a, sync.tag);
    sq_put(gran + G + blockIdx.x, r, sync.tag);
  }
  if (!sq_folds()) return;
  const double nn2 = sq_fold<IPX_SUM>(gran, G, sync.tag, lds);
  const double nviol = boxed ? sq_fold<IPX_SUM>(gran + G, G, sync.tag, lds) : 0.0;
  sq_after_newton(nn2, nviol, radius, tr_factor, q);
}

// two reductions of different lengths in one launch, each with the grid and the accumulation
// its stand-alone kernel has (csrc/vec.hip RedDot / RedNorms: same bits): blocks [0, g1) the
// first, [g1, g1 + g2) the second.  DOTS: sum x y; else sum x^2 (y unused).
template <bool DOT1, bool DOT2>
__global__ void __launch_bounds__(RB)
k_sq_two_sums(int64_t n1, const double *__restrict__ x1, const double *__restrict__ y1, int g1,
              double *__restrict__ part1, int64_t n2, const double *__restrict__ x2,
              const double *__restrict__ y2, int g2, double *__restrict__ part2,
              const double *__restrict__ guard) {
  __shared__ double lds[RB / IPX_WAVE];
  if (guard && *guard != 0.0) return;
  const bool second = (int)blockIdx.x >= g1;
  const int b = second ? blockIdx.x - g1 : blockIdx.x, G = second ? g2 : g1;
  const int64_t n = second ? n2 : n1;
  const double *x = second ? x2 : x1, *y = second ? y2 : y1;
  const bool dot = second ? DOT2 : DOT1;
  double s = 0.0;
  const int64_t stride = (int64_t)G * RB;
  for (int64_t i = (int64_t)b * RB + threadIdx.x; i < n; i += stride) {
    const double t = x[i];
    s += dot ? t * y[i] : t * t;
  }
  const double r = ipx_block_reduce<IPX_SUM>(s, lds);
  if (threadIdx.x == 0) (second ? part2 : part1)[b] = r;
}

// sum of squares + max |.| partials (part[block], part[grid + block])
__global__ void __launch_bounds__(RB)
k_sq_norms(int64_t n, const double *__restrict__ v, double *__restrict__ part) {
  __shared__ double lds[RB / IPX_WAVE];
  double s = 0.0, mx = 0.0;
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
    const double t = v[i];
    s += t * t;
    mx = fmax(mx, fabs(t));
  }
  const double a = ipx_block_reduce<IPX_SUM>(s, lds);
  const double b = ipx_block_reduce<IPX_MAX>(mx, lds);
  if (threadIdx.x == 0) { part[blockIdx.x] = a; part[gridDim.x + blockIdx.x] = b; }
}

// lbt = lb - dn, ubt = ub - dn (either side may be absent)
__global__ void __launch_bounds__(RB)
k_sq_shift_bounds(int64_t n, const double *__restrict__ lb, const double *__restrict__ ub,
                  const double *__restrict__ dn, double *__restrict__ lbt,
                  double *__restrict__ ubt) {
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
    const double t = dn[i];
    if (lb) lbt[i] = lb[i] - t;
    if (ub) ubt[i] = ub[i] - t;
  }
}

// the seven sums of box_sphere_intersections over (z, dscale * d) -- csrc/vec.hip RedBoxSphere --
// as partials part[q * grid + block], run only when the CG loop stopped on code 2 or 3 (`st`);
// dscale = alpha for code 2 (:585), 1 for code 3 (:562, entire line)
__global__ void __launch_bounds__(RB)
k_sq_exit_reduce(int64_t n, const double *__restrict__ st, const double *__restrict__ z,
                 const double *__restrict__ d, const double *__restrict__ lb,
                 const double *__restrict__ ub, sq_u4 *gran, SqSync sync,
                 double *__restrict__ q) {
  __shared__ double lds[RB / IPX_WAVE];
  const double stop = st[ST_STOP];
  if (stop != 2.0 && stop != 3.0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { q[SQ_EXIT_TAU] = 0.0; q[SQ_EXIT_DONE] = 0.0; }
    return;
  }
  const double dscale = stop == 2.0 ? st[ST_ALPHA] : 1.0;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = -HUGE_VAL, a4 = HUGE_VAL, a5 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
    const double zi = z[i], di = dscale * d[i];
    a0 += di * di; a1 += zi * di; a2 += zi * zi;
    const double lo = lb ? lb[i] : -HUGE_VAL, hi = ub ? ub[i] : HUGE_VAL;
    if (di == 0.0) {
      a5 += (zi < lo || zi > hi) ? 1.0 : 0.0;
    } else {
      const double tl = (lo - zi) / di, tu = (hi - zi) / di;
      a3 = fmax(a3, fmin(tl, tu));
      a4 = fmin(a4, fmax(tl, tu));
    }
  }
  const int g = gridDim.x, b = blockIdx.x;
  const unsigned int tag = sync.tag;
  double r;
  r = ipx_block_reduce<IPX_SUM>(a0, lds); if (threadIdx.x == 0) sq_put(gran + b, r, tag);
  r = ipx_block_reduce<IPX_SUM>(a1, lds); if (threadIdx.x == 0) sq_put(gran + g + b, r, tag);
  r = ipx_block_reduce<IPX_SUM>(a2, lds); if (threadIdx.x == 0) sq_put(gran + 2 * g + b, r, tag);
  r = ipx_block_reduce<IPX_MAX>(a3, lds); if (threadIdx.x == 0) sq_put(gran + 3 * g + b, r, tag);
  r = ipx_block_reduce<IPX_MIN>(a4, lds); if (threadIdx.x == 0) sq_put(gran + 4 * g + b, r, tag);
  r = ipx_block_reduce<IPX_SUM>(a5, lds); if (threadIdx.x == 0) sq_put(gran + 5 * g + b, r, tag);
  if (!sq_folds()) return;
  // the scalar tail of that exit, by the workgroup that arrives last: tau = theta * alpha
  // (code 2) / the line's far intersection (code 3), 0 when the segment misses
  double rr[7];
  rr[0] = sq_fold<IPX_SUM>(gran, g, tag, lds);
  rr[1] = sq_fold<IPX_SUM>(gran + g, g, tag, lds);
  rr[2] = sq_fold<IPX_SUM>(gran + 2 * g, g, tag, lds);
  rr[3] = sq_fold<IPX_MAX>(gran + 3 * g, g, tag, lds);
  rr[4] = sq_fold<IPX_MIN>(gran + 4 * g, g, tag, lds);
  rr[5] = sq_fold<IPX_SUM>(gran + 5 * g, g, tag, lds);
  rr[6] = 0.0;
  if (threadIdx.x != 0) return;
  const double radius = st[ST_RADIUS];
  const sq_interval iv = sq_box_sphere(rr, radius, stop == 3.0);
  // :565-568 x + alpha p with alpha = tb of the entire line; :588-590 x + theta alpha p
  const double tau = !iv.hit ? 0.0 : (stop == 2.0 ? iv.tb * st[ST_ALPHA] : iv.tb);
  q[SQ_EXIT_TAU] = tau;
  q[SQ_EXIT_DONE] = 1.0;
}

// dt <- clip(dt + tau p) when the exit above ran (tau == 0: dt itself, clipped -- :569,:591 clip
// in both cases), then the step's vectors and sums:
//   d = dn + dt,  x_next = x + (scale ? scale * d : d),
//   partials of ||d||^2, ||dt||^2, c.d and of the box violations of dt (:636)
__global__ void __launch_bounds__(RB)
k_sq_step_vectors(int64_t n, int use_exit, const double *__restrict__ q,
                  const double *__restrict__ dn,
                  double *__restrict__ dt, const double *__restrict__ p,
                  const double *__restrict__ lbt, const double *__restrict__ ubt,
                  const double *__restrict__ x, const double *__restrict__ c,
                  const double *__restrict__ scale, double *__restrict__ d,
                  double *__restrict__ x_next, double *__restrict__ part) {
  __shared__ double lds[RB / IPX_WAVE];
  // (use_exit == 0: the host finished the CG loop itself, boundary exits included)
  const bool exit_done = use_exit && q[SQ_EXIT_DONE] != 0.0;
  const double tau = use_exit ? q[SQ_EXIT_TAU] : 0.0;
  double s_d = 0.0, s_dt = 0.0, s_cd = 0.0, s_out = 0.0;
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
    double t = dt[i];
    const double lo = lbt ? lbt[i] : -HUGE_VAL, hi = ubt ? ubt[i] : HUGE_VAL;
    if (exit_done) {
      if (tau != 0.0) t = t + tau * p[i];
      if (lbt || ubt) t = fmin(fmax(t, lo), hi);
      dt[i] = t;
    }
    if (lbt || ubt) s_out += ((lo <= t) && (t <= hi)) ? 0.0 : 1.0;
    const double di = dn[i] + t;
    d[i] = di;
    x_next[i] = x[i] + (scale ? scale[i] * di : di);
    s_d += di * di;
    s_dt += t * t;
    s_cd += c[i] * di;
  }
  const int g = gridDim.x, b = blockIdx.x;
  double r;
  r = ipx_block_reduce<IPX_SUM>(s_d, lds);   if (threadIdx.x == 0) part[b] = r;
  r = ipx_block_reduce<IPX_SUM>(s_dt, lds);  if (threadIdx.x == 0) part[g + b] = r;
  r = ipx_block_reduce<IPX_SUM>(s_cd, lds);  if (threadIdx.x == 0) part[2 * g + b] = r;
  r = ipx_block_reduce<IPX_SUM>(s_out, lds); if (threadIdx.x == 0) part[3 * g + b] = r;
}

// ---- the dogleg proper on the device (qp_subproblem.py:375-413) -------------------------------
// Runs behind k_sq_after_newton when the Newton point was turned down (every kernel returns at
// once when q[SQ_NORMAL_KIND] != 0).  g = A'b and A g come from two guarded SpMVs whose epilogue
// partials give g.g and (A g).(A g).
//
// k_sq_dogleg_reduce: cauchy = coef g with coef = -(g.g) / (Ag.Ag) (:380), and the seven sums of
// box_sphere_intersections for the three segments the routine may search (:386-407) in ONE pass
// over g and the Newton point: (cauchy, newton - cauchy), (0, cauchy), (0, newton) -- 18 partial
// arrays (6 per segment, csrc/vec.hip RedBoxSphere's accumulation and order).
struct sq_seg { double a0, a1, a2, a3, a4, a5; };
__device__ __forceinline__ void sq_seg_init(sq_seg &s) {
  s.a0 = s.a1 = s.a2 = 0.0; s.a3 = -HUGE_VAL; s.a4 = HUGE_VAL; s.a5 = 0.0;
}
__device__ __forceinline__ void sq_seg_step(sq_seg &s, double zi, double di, double lo, double hi) {
  s.a0 += di * di; s.a1 += zi * di; s.a2 += zi * zi;
  if (di == 0.0) {
    s.a5 += (zi < lo || zi > hi) ? 1.0 : 0.0;
  } else {
    const double tl = (lo - zi) / di, tu = (hi - zi) / di;
    s.a3 = fmax(s.a3, fmin(tl, tu));
    s.a4 = fmin(s.a4, fmax(tl, tu));
  }
}
__device__ __forceinline__ void sq_seg_store(const sq_seg &s, double *lds, double *part, int g) {
  const int b = blockIdx.x;
  double r;
  r = ipx_block_reduce<IPX_SUM>(s.a0, lds); if (threadIdx.x == 0) part[b] = r;
  r = ipx_block_reduce<IPX_SUM>(s.a1, lds); if (threadIdx.x == 0) part[g + b] = r;
  r = ipx_block_reduce<IPX_SUM>(s.a2, lds); if (threadIdx.x == 0) part[2 * g + b] = r;
  r = ipx_block_reduce<IPX_MAX>(s.a3, lds); if (threadIdx.x == 0) part[3 * g + b] = r;
  r = ipx_block_reduce<IPX_MIN>(s.a4, lds); if (threadIdx.x == 0) part[4 * g + b] = r;
  r = ipx_block_reduce<IPX_SUM>(s.a5, lds); if (threadIdx.x == 0) part[5 * g + b] = r;
}
__device__ __forceinline__ void sq_seg_fold(const double *part, int g, double *lds, double *r7) {
  r7[0] = ipx_sum_partials<IPX_SUM>(part, g, lds);
  r7[1] = ipx_sum_partials<IPX_SUM>(part + g, g, lds);
  r7[2] = ipx_sum_partials<IPX_SUM>(part + 2 * g, g, lds);
  r7[3] = ipx_sum_partials<IPX_MAX>(part + 3 * g, g, lds);
  r7[4] = ipx_sum_partials<IPX_MIN>(part + 4 * g, g, lds);
  r7[5] = ipx_sum_partials<IPX_SUM>(part + 5 * g, g, lds);
  r7[6] = 0.0;
}

__global__ void __launch_bounds__(RB)
k_sq_dogleg_reduce(int64_t n, double *__restrict__ q, const double *__restrict__ p_gg, int n_gg,
                   const double *__restrict__ p_ag, int n_ag, const double *__restrict__ g,
                   const double *__restrict__ newton, const double *__restrict__ lb,
                   const double *__restrict__ ub, double bf, double *__restrict__ part) {
  __shared__ double lds[2 * (RB / IPX_WAVE)];
  if (q[SQ_NORMAL_KIND] != 0.0) return;
  const double *parts[2] = {p_gg, p_ag};
  const int counts[2] = {n_gg, n_ag};
  double s2[2];
  ipx_sum_partials_multi<2>(parts, counts, lds, s2);
  const double coef = -s2[0] / s2[1];                                   // :380
  if (blockIdx.x == 0 && threadIdx.x == 0) q[SQ_DOGLEG] = coef;
  sq_seg s1, sb, sc;
  sq_seg_init(s1); sq_seg_init(sb); sq_seg_init(sc);
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
    const double ni = newton[i], ci = coef * g[i];
    const double lo = lb ? bf * lb[i] : -HUGE_VAL, hi = ub ? bf * ub[i] : HUGE_VAL;
    sq_seg_step(s1, ci, ni - ci, lo, hi);
    sq_seg_step(sb, 0.0, ci, lo, hi);
    sq_seg_step(sc, 0.0, ni, lo, hi);
  }
  const int G = gridDim.x;
  sq_seg_store(s1, lds, part, G);
  sq_seg_store(sb, lds, part + 6 * (int64_t)G, G);
  sq_seg_store(sc, lds, part + 12 * (int64_t)G, G);
}

// the three intervals and what they make of the two candidate points (:386-407):
//   x1 = cauchy + alpha (newton - cauchy)   when that segment meets the region, else alpha cauchy
//   x2 = alpha newton
// as a mode and two step lengths for k_sq_dogleg_points
__global__ void __launch_bounds__(RB)
k_sq_dogleg_decide(double *__restrict__ q, const double *__restrict__ part, int G, double radius) {
  __shared__ double lds[RB / IPX_WAVE];
  if (q[SQ_NORMAL_KIND] != 0.0) return;
  double r1[7], r2[7], r3[7];
  sq_seg_fold(part, G, lds, r1);
  sq_seg_fold(part + 6 * (int64_t)G, G, lds, r2);
  sq_seg_fold(part + 12 * (int64_t)G, G, lds, r3);
  if (threadIdx.x != 0) return;
  const sq_interval i1 = sq_box_sphere(r1, radius, false);
  const sq_interval i2 = sq_box_sphere(r2, radius, false);
  const sq_interval i3 = sq_box_sphere(r3, radius, false);
  q[SQ_DOGLEG + 1] = i1.hit ? 1.0 : 0.0;
  q[SQ_DOGLEG + 2] = i1.hit ? i1.tb : i2.tb;
  q[SQ_DOGLEG + 3] = i3.tb;
}

__global__ void __launch_bounds__(RB)
k_sq_dogleg_points(int64_t n, const double *__restrict__ q, const double *__restrict__ g,
                   const double *__restrict__ newton, double *__restrict__ x1,
                   double *__restrict__ x2) {
  if (q[SQ_NORMAL_KIND] != 0.0) return;
  const double coef = q[SQ_DOGLEG], a1 = q[SQ_DOGLEG + 2], a3 = q[SQ_DOGLEG + 3];
  const bool along = q[SQ_DOGLEG + 1] != 0.0;
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
    const double ni = newton[i], ci = coef * g[i];
    x1[i] = along ? ci + a1 * (ni - ci) : 0.0 + a1 * ci;     // :388 / :393 (origin + alpha cauchy)
    x2[i] = 0.0 + a3 * ni;                                   // :405
  }
}

// the smaller ||A x + b|| wins (strictly; a tie is x2's, :410-413); dn <- the winner, ||dn||^2
__global__ void __launch_bounds__(RB)
k_sq_dogleg_take(int64_t n, double *__restrict__ q, const double *__restrict__ p1, int n1,
                 const double *__restrict__ p2, int n2, const double *__restrict__ x1,
                 const double *__restrict__ x2, double *__restrict__ dn,
                 double *__restrict__ part) {
  __shared__ double lds[2 * (RB / IPX_WAVE)];
  if (q[SQ_NORMAL_KIND] != 0.0) return;
  const double *parts[2] = {p1, p2};
  const int counts[2] = {n1, n2};
  double s2[2];
  ipx_sum_partials_multi<2>(parts, counts, lds, s2);
  const bool first = sqrt(s2[0]) < sqrt(s2[1]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    q[SQ_DOGLEG + 4] = first ? 1.0 : 2.0;
    q[SQ_DOGLEG + 5] = sqrt(s2[0]);
    q[SQ_DOGLEG + 6] = sqrt(s2[1]);
  }
  double s = 0.0, mx = 0.0;
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
    const double t = first ? x1[i] : x2[i];
    dn[i] = t;
    s += t * t;
    mx = fmax(mx, fabs(t));
  }
  const double a = ipx_block_reduce<IPX_SUM>(s, lds);
  const double b = ipx_block_reduce<IPX_MAX>(mx, lds);
  if (threadIdx.x == 0) { part[blockIdx.x] = a; part[gridDim.x + blockIdx.x] = b; }
}

__global__ void __launch_bounds__(RB)
k_sq_after_dogleg(const double *__restrict__ p_nn, int n_nn, double radius,
                  double *__restrict__ q) {
  __shared__ double lds[RB / IPX_WAVE];
  if (q[SQ_NORMAL_KIND] != 0.0) return;
  const double nn2 = ipx_sum_partials<IPX_SUM>(p_nn, n_nn, lds);
  if (threadIdx.x != 0) return;
  const double norm_dn = sqrt(nn2);
  q[SQ_NORM_DN] = norm_dn;
  q[SQ_NORMAL_KIND] = 2.0;
  q[SQ_RADIUS_T] = sqrt(radius * radius - norm_dn * norm_dn);
}

// ---- decide kernels (one workgroup each) ----------------------------------------------------
// after the Newton point: ||dn||, inside the box and the ball? the tangential radius
__device__ __forceinline__ void sq_after_newton(double nn2, double nviol, double radius,
                                                double tr_factor, double *q) {
  if (threadIdx.x != 0) return;
  const double norm_dn = sqrt(nn2);
  const bool ok = nviol == 0.0 && norm_dn <= tr_factor * radius;      // qp_subproblem.py:370-373
  q[SQ_NVIOL] = nviol;
  q[SQ_NORM_DN] = norm_dn;
  q[SQ_NORMAL_KIND] = ok ? 1.0 : 0.0;
  q[SQ_RADIUS] = radius;
  q[SQ_RADIUS_T] = sqrt(radius * radius - norm_dn * norm_dn);         // :127
}

// a normal step the host computed (the dogleg proper): its norm and the tangential radius
__global__ void __launch_bounds__(RB)
k_sq_after_given(const double *__restrict__ p_nn, int n_nn, double radius,
                 double *__restrict__ q) {
  __shared__ double lds[RB / IPX_WAVE];
  const double nn2 = ipx_sum_partials<IPX_SUM>(p_nn, n_nn, lds);
  if (threadIdx.x != 0) return;
  const double norm_dn = sqrt(nn2);
  q[SQ_NVIOL] = 0.0;
  q[SQ_NORM_DN] = norm_dn;
  q[SQ_NORMAL_KIND] = 2.0;
  q[SQ_RADIUS] = radius;
  q[SQ_RADIUS_T] = sqrt(radius * radius - norm_dn * norm_dn);
}

// the model: folds the step's sums, :135-153, and a copy of the CG loop's state block -- run by
// the last workgroup of the launch that forms (H d).d and ||A d + b||^2 (k_sq_two_sums<true,
// false>'s sums, grids and bits), which then hands the block to the host
struct SqModelIn {
  const double *p_vec; int g_vec;
  const double *cg_state, *red;
  double penalty, f, norm_b;
};
__global__ void __launch_bounds__(RB)
k_sq_model_sums(int64_t n1, const double *__restrict__ x1, const double *__restrict__ y1, int g1,
                int64_t n2, const double *__restrict__ x2, int g2, sq_u4 *gran, SqSync sync,
                SqModelIn in, double *q, SqPub pub) {
  __shared__ double lds[RB / IPX_WAVE];
  const bool second = (int)blockIdx.x >= g1;
  const int b = second ? blockIdx.x - g1 : blockIdx.x, G = second ? g2 : g1;
  const int64_t n = second ? n2 : n1;
  const double *x = second ? x2 : x1;
  double s = 0.0;
  const int64_t stride = (int64_t)G * RB;
  for (int64_t i = (int64_t)b * RB + threadIdx.x; i < n; i += stride) {
    const double t = x[i];
    s += second ? t * t : t * y1[i];
  }
  const double r = ipx_block_reduce<IPX_SUM>(s, lds);
  if (threadIdx.x == 0) sq_put(gran + blockIdx.x, r, sync.tag);      // ([0, g1): H d . d; then A d + b)
  if (!sq_folds()) return;
  // (each quantity folded on its own: the order ipx_sum_partials_multi gives it)
  const double *pv = in.p_vec;
  const int gv = in.g_vec;
  const double d2 = ipx_sum_partials<IPX_SUM>(pv, gv, lds);
  const double dt2 = ipx_sum_partials<IPX_SUM>(pv + gv, gv, lds);
  const double cd = ipx_sum_partials<IPX_SUM>(pv + 2 * gv, gv, lds);
  const double outside = ipx_sum_partials<IPX_SUM>(pv + 3 * gv, gv, lds);
  const double hdd = sq_fold<IPX_SUM>(gran, g1, sync.tag, lds);
  const double lin2 = sq_fold<IPX_SUM>(gran + g1, g2, sync.tag, lds);
  if (threadIdx.x < ST_SIZE) q[SQ_CG + threadIdx.x] = in.cg_state[threadIdx.x];
  if (threadIdx.x == 0) {
    q[SQ_NORM_D] = sqrt(d2);
    q[SQ_NORM_DT] = sqrt(dt2);
    q[SQ_CD] = cd;
    q[SQ_HDD] = hdd;
    q[SQ_LIN] = sqrt(lin2);
    q[SQ_X_OUTSIDE] = outside;
    q[SQ_PRIME_STEPS] = in.red ? in.red[16] + in.red[17] : 0.0;       // (csrc/cg.hip PR_TAKEN)
    q[SQ_PENALTY] = in.penalty;
    q[SQ_F] = in.f;
    q[SQ_NORM_B] = in.norm_b;
    sqp_model(q);
  }
  sq_publish(q, pub);
}

// the verdict: ||b_next||, :156-173, and -- unless the second-order correction is due, which
// the host runs -- the ladder and the accept test
__device__ __forceinline__ void sq_judge_fold(double bn2, double f_next, const double *f_next_dev,
                                              double *q) {
  if (threadIdx.x != 0) return;
  q[SQ_F_NEXT] = f_next_dev ? *f_next_dev : f_next;
  q[SQ_NORM_B_NEXT] = sqrt(bn2);
  sqp_ratio(q);
  if (q[SQ_SOC] == 0.0) sqp_radius(q);
  else q[SQ_ACCEPT] = 0.0;
}
// (no constraint rows: nothing to sum)
__global__ void __launch_bounds__(RB)
k_sq_judge(double f_next, const double *__restrict__ f_next_dev, double *q, SqPub pub) {
  sq_judge_fold(0.0, f_next, f_next_dev, q);
  sq_publish(q, pub);
}
// k_sq_norms over b_next (its grid, its sums) whose last workgroup is the judge
__global__ void __launch_bounds__(RB)
k_sq_judge_norms(int64_t n, const double *__restrict__ v, sq_u4 *gran, SqSync sync,
                 double f_next, const double *__restrict__ f_next_dev, double *q, SqPub pub) {
  __shared__ double lds[RB / IPX_WAVE];
  double s = 0.0, mx = 0.0;
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
    const double t = v[i];
    s += t * t;
    mx = fmax(mx, fabs(t));
  }
  const double a = ipx_block_reduce<IPX_SUM>(s, lds);
  const double b = ipx_block_reduce<IPX_MAX>(mx, lds);
  (void)b;                                 // (the verdict takes the 2-norm only)
  if (threadIdx.x == 0) sq_put(gran + blockIdx.x, a, sync.tag);
  if (!sq_folds()) return;
  const double bn2 = sq_fold<IPX_SUM>(gran, gridDim.x, sync.tag, lds);
  sq_judge_fold(bn2, f_next, f_next_dev, q);
  sq_publish(q, pub);
}

// the measures of a new iterate (:86-87, 238-239): ||c + A'v||_inf, ||b||_inf, ||b||, and
// ||A||_F^2 from the partials of ipx_norms_partials over A's values
// ... run by the last workgroup of ONE launch that takes both vectors' norms (k_sq_norms' grids
// and sums: blocks [0, g_t) over c + A'v, [g_t, g_t + g_b) over b)
__global__ void __launch_bounds__(RB)
k_sq_measure_norms(int64_t n1, const double *__restrict__ v1, int g1, int64_t n2,
                   const double *__restrict__ v2, int g2, sq_u4 *gran, SqSync sync,
                   const double *__restrict__ p_A, int g_A, const double *__restrict__ verdict,
                   double *q, SqPub pub, const double *__restrict__ neg_in,
                   double *__restrict__ neg_out) {
  __shared__ double lds[RB / IPX_WAVE];
  const bool second = (int)blockIdx.x >= g1;
  const int b = second ? blockIdx.x - g1 : blockIdx.x, G = second ? g2 : g1;
  const int64_t n = second ? n2 : n1;
  const double *v = second ? v2 : v1;
  double s = 0.0, mx = 0.0;
  const int64_t stride = (int64_t)G * RB;
  for (int64_t i = (int64_t)b * RB + threadIdx.x; i < n; i += stride) {
    const double t = v[i];
    s += t * t;
    mx = fmax(mx, fabs(t));
    if (second) neg_out[i] = -neg_in[i];       // (the multipliers v = -(A A')^-1 A c: same length)
  }
  const double a = ipx_block_reduce<IPX_SUM>(s, lds);
  const double c = ipx_block_reduce<IPX_MAX>(mx, lds);
  // granules: [0, g1) / [g1, 2 g1) sums / maxima of the first vector, then the second's
  sq_u4 *mine = gran + (second ? 2 * g1 : 0);
  if (threadIdx.x == 0) {
    sq_put(mine + b, a, sync.tag);
    sq_put(mine + G + b, c, sync.tag);
  }
  if (!sq_folds()) return;
  const sq_u4 *gt = gran, *gb = gran + 2 * g1;
  const double opt = g1 > 0 ? sq_fold<IPX_MAX>(gt + g1, g1, sync.tag, lds) : 0.0;
  const double viol = g2 > 0 ? sq_fold<IPX_MAX>(gb + g2, g2, sync.tag, lds) : 0.0;
  const double nb2 = g2 > 0 ? sq_fold<IPX_SUM>(gb, g2, sync.tag, lds) : 0.0;
  const double na2 = g_A > 0 ? ipx_sum_partials<IPX_SUM>(p_A, g_A, lds) : 0.0;
  if (threadIdx.x == 0) {
    q[SQ_OPT] = opt;
    q[SQ_VIOL] = viol;
    q[SQ_NORM_B] = sqrt(nb2);
    if (g_A > 0) q[SQ_NORM_A2] = na2;
    q[SQ_FACTOR_BAD] = verdict ? *verdict : 0.0;
  }
  sq_publish(q, pub);
}

}  // namespace

// ---- the argument block (all members 8 bytes; mirrored by ipsolver/sqp_chain.py) ------------
namespace {
// Measurement aid (bench.py: the in-solve projected-CG rate of SURVEY.md 8(d)(i)): with timing
// on, ipx_sqp_front brackets the CG's priming + first batch with HIP events on its stream.
struct CgTiming {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
} g_cg_timing;
}  // namespace

extern "C" {


int ipx_sqp_block_size(void) { return SQ_SIZE; }

// doubles of `part`: the vector kernels' partials (4 quantities, then 6 for the exit's sums), the
// SpMV partials of H d, A d + b, the Newton point; room for the refresh's three norm pairs
int64_t ipx_sqp_part_doubles(const ipx_sqp_args *s) {
  if (!s || !s->cg) return -1;
  const int64_t g = sq_grid(s->n), gm = sq_grid(s->m);
  return 4 * g + 6 * g + g + gm + 2 * g + 2 * gm + g + 18 * g + 64 + 2 + 2 * (6 * g + 2 * gm) +
         2 * (int64_t)s->cg->H_ntiles;
}

/* on != 0: start collecting; on == 0: stop, synchronise, *ms_total = GPU milliseconds between the
 * first launch of the projected CG's priming and the last of its first batch, summed over the
 * ipx_sqp_front calls since timing was switched on, *calls = how many. */
int ipx_sqp_cg_timing(int on, double *ms_total, int *calls) {
  if (on) {
    g_cg_timing.on = true;
    return IPX_OK;
  }
  g_cg_timing.on = false;
  double tot = 0.0;
  int n = 0;
  for (auto &pr : g_cg_timing.ev) {
    float ms = 0.f;
    if (hipEventSynchronize(pr.second) == hipSuccess &&
        hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { tot += ms; ++n; }
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  g_cg_timing.ev.clear();
  if (ms_total) *ms_total = tot;
  if (calls) *calls = n;
  return IPX_OK;
}

void ipx_sqp_model_host(double *q) { sqp_model(q); }
void ipx_sqp_ratio_host(double *q) { sqp_ratio(q); }
void ipx_sqp_radius_host(double *q) { sqp_radius(q); }
/* box_sphere_intersections' scalar tail from the seven sums of ipx_box_sphere_reduce:
 * out3 = (ta, tb, intersect) */
void ipx_sqp_box_sphere_host(const double *sums7, double radius, int entire_line, double *out3) {
  const sq_interval iv = sq_box_sphere(sums7, radius, entire_line != 0);
  out3[0] = iv.ta; out3[1] = iv.tb; out3[2] = iv.hit ? 1.0 : 0.0;
}

}  // extern "C"

namespace {

struct PartLayout {
  double *vec, *exit, *hd, *ad, *viol, *bn, *tn, *dog, *tail;
  sq_u4 *gran;                // tagged partials of the kernels that end in their own decision:
  int g, gm;                  // 6 g + 2 gm granules, one launch's at a time (allocated zeroed)
  double *ctn;                // epilogue partials of c_t = H dn + c (2 x H's tiles): ||c_t||^2
};
// the tag of one launch: its sequence number (never 0: the arena's initial state)
static unsigned int next_tag() {
  static unsigned int seq = 0;
  if (++seq == 0) ++seq;
  return seq;
}

// a chain's block on its way to the host: begin the read before the last launch, wait behind it
struct BlockRead {
  SqPub pub{nullptr, 0};
  double *host = nullptr;
  int begin(const ipx_sqp_args *s) {
    host = s->host_block;
    if (!host) return IPX_OK;
    return ipx_read_begin(&pub.dst, &pub.tag);
  }
  int wait(hipStream_t st) {
    return host ? ipx_read_wait(pub.dst, pub.tag, SQ_SIZE, host, st) : IPX_OK;
  }
};
static PartLayout layout(const ipx_sqp_args *s) {
  PartLayout L;
  L.g = sq_grid(s->n); L.gm = sq_grid(s->m);
  double *p = s->part;
  L.vec = p;  p += 4 * (int64_t)L.g;
  L.exit = p; p += 6 * (int64_t)L.g;
  L.hd = p;   p += L.g;
  L.ad = p;   p += L.gm;
  L.tn = p;   p += 2 * (int64_t)L.g;        // norms of an n-vector (refresh: c + A'v; given dn)
  L.bn = p;   p += 2 * (int64_t)L.gm;       // norms of an m-vector (b, b_next)
  L.viol = p; p += L.g;
  L.dog = p;  p += 18 * (int64_t)L.g;       // the dogleg's three segments
  L.tail = p;
  p += 64;
  p += ((p - s->part) & 1);                 // (granules are 16 bytes)
  L.gran = (sq_u4 *)p;
  p += 2 * (6 * (int64_t)L.g + 2 * (int64_t)L.gm);
  L.ctn = p;
  return L;
}

static int solve(const ipx_cg_args *a, const double *w, double *v, hipStream_t st) {
  if (a->solver_kind == 1)
    return ipx_boxschur_solve((const ipx_boxschur_args *)a->banded, w, v, nullptr, nullptr, nullptr,
                              st);
  return ipx_banded_solve(a->banded, w, v, st);
}

}  // namespace

extern "C" {

// The step's second half on its own: the CG loop's boundary exits, d, x_next, the five sums, the
// model.  `dt` = cg->x.  (ipx_sqp_front ends with it; the host calls it again after it had to
// finish the CG itself.)
int ipx_sqp_model(const ipx_sqp_args *s, double penalty, double f, double norm_b, int host_cg,
                  void *stream) {
  if (!s || !s->cg || !s->q || !s->part || !s->x_next) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const ipx_cg_args *a = s->cg;
  const PartLayout L = layout(s);
  const ipx_csr_view A{(int)s->m, (int)s->n, a->A_rowptr, a->A_colidx, a->A_val, s->A_tiles,
                       (int)s->A_ntiles};
  const ipx_csr_view Hm{(int)s->n, (int)s->n, a->H_rowptr, a->H_colidx, a->H_val, a->H_tiles,
                        (int)a->H_ntiles};
  if (!host_cg) {
    hipLaunchKernelGGL(k_sq_exit_reduce, dim3(L.g), dim3(RB), 0, st, s->n, a->state, a->x, a->p,
                       a->lb, a->ub, L.gran, SqSync{next_tag()}, s->q);
    IPX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_sq_step_vectors, dim3(L.g), dim3(RB), 0, st, s->n, host_cg ? 0 : 1, s->q,
                     s->dn, a->x, a->p,
                     a->lb, a->ub, s->x, s->c, s->scale, s->d, s->x_next, L.vec);
  IPX_CHECK_LAUNCH();
  int rc = ipx_spmv_launch(Hm, s->d, 1.0, a->H_diag, 0.0, nullptr, s->Hd, nullptr, nullptr, st);
  if (rc) return rc;
  rc = ipx_spmv_launch(A, s->d, 1.0, nullptr, 1.0, s->b, s->Ad, nullptr, nullptr, st);
  if (rc) return rc;
  // (H d).d and ||A d + b||^2 summed as the host form's dot / norm kernels sum them (the
  // products' own epilogue sums come in tile order: other bits), one launch for both
  // ... whose last workgroup folds the step's sums, decides (:135-153) and publishes the block
  BlockRead br;
  rc = br.begin(s);
  if (rc) return rc;
  const SqModelIn in{L.vec, L.g, a->state, host_cg ? nullptr : s->red, penalty, f, norm_b};
  hipLaunchKernelGGL(k_sq_model_sums, dim3(L.g + L.gm), dim3(RB), 0, st, s->n, s->Hd, s->d, L.g,
                     s->m, s->Ad, L.gm, L.gran, SqSync{next_tag()}, in, s->q,
                     br.pub);
  IPX_CHECK_LAUNCH();
  return br.wait(st);
}

// have_dn == 0: the Newton point of the dogleg into s->dn, its acceptance decided on the
// device (SQ_NORMAL_KIND 1; 0: the host runs the dogleg and calls again with have_dn = 1).
// Then c_t, the shifted bounds, the priming of the projected CG with the radius of the block,
// iterations [0, first_end) and ipx_sqp_model.  box_factor / tr_factor: :43-44 (0.5, 0.8).
int ipx_sqp_front(const ipx_sqp_args *s, int have_dn, int with_dogleg, int with_steps,
                  double radius, double penalty, double f, double norm_b, double tr_factor,
                  double box_factor, double tol_in, double norm_A, int32_t first_end,
                  void *stream) {
  if (!s || !s->cg || !s->q || !s->part || !s->dn || !s->ct || s->m <= 0) return IPX_EINVAL;
  if (with_dogleg && (!s->d || !s->Hd || !s->Ad)) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const ipx_cg_args *a = s->cg;
  const PartLayout L = layout(s);
  const ipx_csr_view At{(int)s->n, (int)s->m, a->At_rowptr, a->At_colidx, a->At_val, a->At_tiles,
                        (int)a->At_ntiles};
  const ipx_csr_view Hm{(int)s->n, (int)s->n, a->H_rowptr, a->H_colidx, a->H_val, a->H_tiles,
                        (int)a->H_ntiles};
  int rc;
  if (!have_dn) {
    // newton = -Y b = -A'(A A')^-1 b   (qp_subproblem.py:368)
    rc = solve(a, s->b, a->v, st);
    if (rc) return rc;
    rc = ipx_spmv_launch(At, a->v, -1.0, nullptr, 0.0, nullptr, s->dn, nullptr, nullptr, st);
    if (rc) return rc;
    const bool boxed = s->lb || s->ub;
    hipLaunchKernelGGL(k_sq_newton_check, dim3(L.g), dim3(RB), 0, st, s->n, s->dn, s->lb, s->ub,
                       box_factor, L.gran, SqSync{next_tag()}, boxed ? 1 : 0,
                       radius, tr_factor, s->q, a->x);
    IPX_CHECK_LAUNCH();
    if (with_dogleg) {
      // the dogleg proper behind it, every launch a no-op when the Newton point stands (the
      // SpMVs by their guard: *guard != 0 skips; s->ct, s->d, s->Hd, a->t, a->w, s->Ad are free
      // at this point of the chain)
      const ipx_csr_view A{(int)s->m, (int)s->n, a->A_rowptr, a->A_colidx, a->A_val, s->A_tiles,
                           (int)s->A_ntiles};
      const double *guard = s->q + SQ_NORMAL_KIND;
      double *g = s->ct, *x1 = s->d, *x2 = s->Hd;
      rc = ipx_spmv_launch(At, s->b, 1.0, nullptr, 0.0, nullptr, g, nullptr, guard, st);   // :376
      if (rc) return rc;
      rc = ipx_spmv_launch(A, g, 1.0, nullptr, 0.0, nullptr, a->w, nullptr, guard, st);  // :379
      if (rc) return rc;
      // g.g and (A g).(A g) as the host form's dot kernel sums them
      hipLaunchKernelGGL((k_sq_two_sums<true, true>), dim3(L.g + L.gm), dim3(RB), 0, st, s->n, g, g,
                         L.g, L.tn, s->m, a->w, a->w, L.gm, L.bn, guard);
      IPX_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_sq_dogleg_reduce, dim3(L.g), dim3(RB), 0, st, s->n, s->q, L.tn, L.g,
                         L.bn, L.gm, g, s->dn, s->lb, s->ub, box_factor, L.dog);
      IPX_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_sq_dogleg_decide, dim3(1), dim3(RB), 0, st, s->q, L.dog, L.g,
                         tr_factor * radius);
      IPX_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_sq_dogleg_points, dim3(L.g), dim3(RB), 0, st, s->n, s->q, g, s->dn, x1,
                         x2);
      IPX_CHECK_LAUNCH();
      rc = ipx_spmv_launch(A, x1, 1.0, nullptr, 1.0, s->b, s->Ad, nullptr, guard, st);
      if (rc) return rc;
      rc = ipx_spmv_launch(A, x2, 1.0, nullptr, 1.0, s->b, a->t, nullptr, guard, st);
      if (rc) return rc;
      hipLaunchKernelGGL((k_sq_two_sums<false, false>), dim3(2 * L.gm), dim3(RB), 0, st, s->m,
                         s->Ad, s->Ad, L.gm, L.bn, s->m, a->t, a->t, L.gm, L.bn + L.gm, guard);
      IPX_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_sq_dogleg_take, dim3(L.g), dim3(RB), 0, st, s->n, s->q, L.bn, L.gm,
                         L.bn + L.gm, L.gm, x1, x2, s->dn, L.tn);
      IPX_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_sq_after_dogleg, dim3(1), dim3(RB), 0, st, L.tn, L.g, radius, s->q);
      IPX_CHECK_LAUNCH();
    }
  } else {
    hipLaunchKernelGGL(k_sq_norms, dim3(L.g), dim3(RB), 0, st, s->n, s->dn, L.tn);
    IPX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sq_after_given, dim3(1), dim3(RB), 0, st, L.tn, L.g, radius, s->q);
    IPX_CHECK_LAUNCH();
  }
  // c_t = H dn + c   (:125)
  rc = ipx_spmv_launch(Hm, s->dn, 1.0, a->H_diag, 1.0, s->c, s->ct, L.ctn, nullptr, st);
  if (rc) return rc;
  if (s->lb || s->ub) {
    hipLaunchKernelGGL(k_sq_shift_bounds, dim3(L.g), dim3(RB), 0, st, s->n, s->lb, s->ub, s->dn,
                       s->lbt, s->ubt);
    IPX_CHECK_LAUNCH();
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_cg_timing.on && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
    (void)hipEventRecord(e0, st);
  rc = ipx_cg_prime_dev(a, s->A_tiles, (int32_t)s->A_ntiles, s->ct, nullptr, s->red, s->ws, tol_in,
                        0.0, s->q + SQ_RADIUS_T, s->orth_tol, norm_A, nullptr, s->cancellation,
                        first_end, with_steps, L.ctn, (int32_t)a->H_ntiles, have_dn ? 0 : 1, st);
  if (e0 && e1) {
    (void)hipEventRecord(e1, st);
    g_cg_timing.ev.emplace_back(e0, e1);
  }
  if (rc) return rc;
  return ipx_sqp_model(s, penalty, f, norm_b, 0, stream);
}

// ||b_next|| and the verdict (f_next by value, or -- f_next_dev non-NULL -- a device scalar the
// user's objective left)
int ipx_sqp_judge(const ipx_sqp_args *s, const double *b_next, double f_next,
                  const double *f_next_dev, void *stream) {
  if (!s || !s->q || !s->part || (s->m > 0 && !b_next)) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const PartLayout L = layout(s);
  BlockRead br;
  int rc = br.begin(s);
  if (rc) return rc;
  if (s->m > 0) {
    hipLaunchKernelGGL(k_sq_judge_norms, dim3(L.gm), dim3(RB), 0, st, s->m, b_next, L.gran,
                       SqSync{next_tag()}, f_next, f_next_dev, s->q, br.pub);
  } else {
    hipLaunchKernelGGL(k_sq_judge, dim3(1), dim3(RB), 0, st, f_next, f_next_dev, s->q, br.pub);
  }
  IPX_CHECK_LAUNCH();
  return br.wait(st);
}

// v = -LS c (:83,226), optimality, constraint violation, ||b|| (:86-87,238-239)
int ipx_sqp_refresh(const ipx_sqp_args *s, void *stream) {
  if (!s || !s->cg || !s->q || !s->part || !s->v_out || s->m <= 0) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const ipx_cg_args *a = s->cg;
  const PartLayout L = layout(s);
  const ipx_csr_view A{(int)s->m, (int)s->n, a->A_rowptr, a->A_colidx, a->A_val, s->A_tiles,
                       (int)s->A_ntiles};
  const ipx_csr_view At{(int)s->n, (int)s->m, a->At_rowptr, a->At_colidx, a->At_val, a->At_tiles,
                        (int)a->At_ntiles};
  int rc = ipx_spmv_launch(A, s->c, 1.0, nullptr, 0.0, nullptr, a->w, nullptr, nullptr, st);
  if (rc) return rc;
  rc = solve(a, a->w, a->v, st);
  if (rc) return rc;
  // c + A'(-v): the products and the row sums of a negated vector are the negated ones, bit for
  // bit, so the product runs on (A A')^-1 A c itself with alpha = -1
  rc = ipx_spmv_launch(At, a->v, -1.0, nullptr, 1.0, s->c, s->ct, nullptr, nullptr, st);
  if (rc) return rc;
  BlockRead br;
  rc = br.begin(s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sq_measure_norms, dim3(L.g + L.gm), dim3(RB), 0, st, s->n, s->ct, L.g,
                     s->m, s->b, L.gm, L.gran, SqSync{next_tag()},
                     s->A_norm_part,
                     s->A_norm_part ? (int)s->A_norm_grid : 0, s->verdict, s->q, br.pub, a->v,
                     s->v_out);
  IPX_CHECK_LAUNCH();
  return br.wait(st);
}

}  // extern "C"
