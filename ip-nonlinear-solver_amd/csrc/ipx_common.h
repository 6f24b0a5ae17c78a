// Shared device helpers for the ipx kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ipx.h"

#define IPX_BLOCK 256            // 4 waves: one per SIMD of a CU
#define IPX_WAVE 64
// A pivot of an LDL' / Cholesky factorization of S = A A' that has lost this many digits
// against its diagonal entry marks A as numerically rank deficient (IPX_ENOTSPD): the
// callers then fall back to the reference's SVD projections (projections.py:101-108).
#define IPX_PIVOT_RTOL 1.1368683772161603e-13   // 2^-43
#define IPX_VEC_GRID_CAP 1024    // grid-stride cap for streaming kernels (4 WG/CU)

// Records the HIP error text for ipx_last_error() (defined in misc.hip).
void ipx_note_error(hipError_t e, const char *file, int line);
// blocking read-back of k ints (csrc/misc.hip: ipx_read_doubles' mechanism); the device array
// must be readable up to the next multiple of 8 bytes
// The two halves of a blocking read for a kernel that publishes its results itself (csrc/sqp.hip:
// the last workgroup of a chain writes the block as tagged granules, csrc/misc.hip k_publish's
// format, into `pinned`): begin -> (the calling thread's pinned buffer, the tag to write),
// launch, wait -> k doubles.
int ipx_read_begin(unsigned int **pinned_out, unsigned int *tag_out);
int ipx_read_wait(unsigned int *pinned, unsigned int tag, int k, double *host_out, hipStream_t st);
int ipx_read_ints(const int *dev, int k, int *host_out, hipStream_t st);

// kernel launches of the library since it was loaded (ipx_launch_count; misc.hip): counted where
// every launch is checked
extern long long g_ipx_launches;

#define IPX_CHECK_LAUNCH()                                   \
  do {                                                       \
    ++g_ipx_launches;                                        \
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

// ---- XCD-aware work mapping ---------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), and
// each XCD has its own 4 MiB L2.  Every kernel of the CG loop therefore maps
// work item t (row tile, or chunk of vector elements) so that XCD k always
// owns the k-th eighth of the index space: the slices of x, p, r, Hp an XCD
// read or wrote in one kernel are the slices it needs in the next, and can be
// served from its own L2 instead of the fabric.  Launch ipx_xcd_grid(n) work-
// groups; ipx_xcd_item returns the item or -1 for the few padding groups.
static inline int ipx_xcd_grid(int nitems) { return 8 * ((nitems + 7) / 8); }
// Rounds of IPX_BLOCK items per workgroup (rmin..rmax) for an item-parallel kernel whose items
// are heavy on the CU's vector-memory path (k_cg_step1_box: ~20 loads per item).  A CU that gets
// one workgroup more than the others finishes that much later, and a CU with a single workgroup
// (one wave per SIMD) cannot overlap its rounds at all -- measured at config 5 (557 k items),
// rounds -> workgroups -> us: 3 -> 726 -> 17.6, 4 -> 544 -> 19.6, 5 -> 436 -> 18.9,
// 6 -> 363 -> 20.6, 8 -> 272 -> 22.0, 9 -> 242 -> 21.2.  Cost model: rounds x workgroups per
// CU (rounded up, at least 2); ties go to the smaller R (more workgroups in flight).
constexpr int IPX_NUM_CUS = 256;
static inline int ipx_balanced_rounds(int64_t items, int rmin, int rmax) {
  int best = rmin;
  int64_t cost = INT64_MAX;
  for (int R = rmin; R <= rmax; ++R) {
    const int64_t nb = (items + (int64_t)IPX_BLOCK * R - 1) / ((int64_t)IPX_BLOCK * R);
    const int64_t per_cu = (nb + IPX_NUM_CUS - 1) / IPX_NUM_CUS;
    const int64_t c = R * (per_cu < 2 ? 2 : per_cu);
    if (c < cost) { cost = c; best = R; }
  }
  return best;
}
__device__ __forceinline__ int ipx_xcd_item(int block, int nitems) {
  const int per = (nitems + 7) >> 3;
  const int t = (block & 7) * per + (block >> 3);
  return ((block >> 3) < per && t < nitems) ? t : -1;
}

// ---- fixed-order reductions -------------------------------------------
// Wave: DPP butterfly inside each row of 16 lanes (xor 1, xor 2, half-row
// mirror, row mirror: four v_mov_dpp pairs, no LDS crossbar), then the four
// row totals are read as scalars and combined in row order; every lane ends
// with the same bits.  (__shfl_xor compiles to ds_bpermute: ~6x the latency.)
// All 64 lanes must be active.  Block: wave results through LDS, in wave order.
template <int CTRL>
__device__ __forceinline__ double ipx_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ipx_readlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
#define IPX_WAVE_REDUCE(v, COMB)                                                  \
  do {                                                                            \
    v = COMB(v, ipx_dpp<0xB1>(v));  /* quad_perm [1,0,3,2] */                     \
    v = COMB(v, ipx_dpp<0x4E>(v));  /* quad_perm [2,3,0,1] */                     \
    v = COMB(v, ipx_dpp<0x141>(v)); /* row_half_mirror     */                     \
    v = COMB(v, ipx_dpp<0x140>(v)); /* row_mirror          */                     \
    const double r0 = ipx_readlane(v, 0), r1 = ipx_readlane(v, 16);               \
    const double r2 = ipx_readlane(v, 32), r3 = ipx_readlane(v, 48);              \
    v = COMB(COMB(r0, r1), COMB(r2, r3));                                         \
  } while (0)
#define IPX_ADD(a, b) ((a) + (b))
__device__ __forceinline__ double ipx_wave_sum(double v) {
  IPX_WAVE_REDUCE(v, IPX_ADD);
  return v;
}
__device__ __forceinline__ double ipx_wave_max(double v) {
  IPX_WAVE_REDUCE(v, fmax);
  return v;
}
__device__ __forceinline__ double ipx_wave_min(double v) {
  IPX_WAVE_REDUCE(v, fmin);
  return v;
}

// Workgroup barrier for exchanges that go through LDS only: waits for this
// wave's LDS traffic, NOT for its outstanding global loads/stores
// (__syncthreads() also drains vmcnt -- ~2 us when stores are in flight).
__device__ __forceinline__ void ipx_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

enum { IPX_SUM = 0, IPX_MAX = 1, IPX_MIN = 2 };
#define IPX_FOLD_U 4             // partial-fold loads in flight per lane and quantity

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
  ipx_lds_barrier();
  double r;
  if (nw == 4) {
    // (the usual workgroup: the four partner values requested together -- a loop to a
    // run-time wave count pays one dependent LDS round trip per wave; same order)
    const double t0 = lds[0], t1 = lds[1], t2 = lds[2], t3 = lds[3];
    r = ipx_combine<OP>(ipx_combine<OP>(ipx_combine<OP>(t0, t1), t2), t3);
  } else {
    r = lds[0];
    for (int w = 1; w < nw; ++w) r = ipx_combine<OP>(r, lds[w]);
  }
  ipx_lds_barrier();
  return r;
}

// Sum `count` partials (written by a previous kernel) in a fixed order that
// is identical in every block, so all blocks derive bit-identical scalars.
template <int OP>
__device__ __forceinline__ double ipx_sum_partials(const double *part, int count,
                                                   double *lds) {
  // rounds of IPX_FOLD_U predicated loads issued together (a load-add loop
  // would pay one memory latency per trip); same accumulation order
  double v = ipx_identity<OP>();
  for (int base = threadIdx.x; base < count; base += IPX_FOLD_U * blockDim.x) {
    double t[IPX_FOLD_U];
#pragma unroll
    for (int u = 0; u < IPX_FOLD_U; ++u) {
      // unconditional clamped load + select: a predicated load compiles to a
      // branch with a full s_waitcnt per load
      const int i = base + u * blockDim.x;
      const double ld = part[min(i, count - 1)];
      t[u] = i < count ? ld : ipx_identity<OP>();
    }
#pragma unroll
    for (int u = 0; u < IPX_FOLD_U; ++u) v = ipx_combine<OP>(v, t[u]);
  }
  return ipx_block_reduce<OP>(v, lds);
}

// Fold NQ partial arrays at once: all loads are issued together and the block
// reduction uses ONE barrier pair for all quantities (a prologue that folds
// them one after the other pays the memory latency and two barriers per
// quantity).  Same fixed summation order as ipx_sum_partials.  `lds` needs
// NQ * blockDim/64 doubles.
template <int NQ>
__device__ __forceinline__ void ipx_sum_partials_multi(const double *const (&part)[NQ],
                                                       const int (&count)[NQ], double *lds,
                                                       double (&out)[NQ]) {
  double v[NQ];
  int longest = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    v[q] = 0.0;
    longest = max(longest, count[q]);
  }
  for (int base = threadIdx.x; base < longest; base += IPX_FOLD_U * blockDim.x) {
    double t[NQ][IPX_FOLD_U];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
      for (int u = 0; u < IPX_FOLD_U; ++u) {
        const int i = base + u * blockDim.x;
        const double ld = part[q][max(min(i, count[q] - 1), 0)];   // part[q] readable even if count 0
        t[q][u] = i < count[q] ? ld : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
      for (int u = 0; u < IPX_FOLD_U; ++u) v[q] += t[q][u];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    v[q] = ipx_wave_sum(v[q]);
    if (lane == 0) lds[q * nw + wave] = v[q];
  }
  ipx_lds_barrier();
  if (nw == 4) {                 // (all partner values requested together: ipx_block_reduce)
    double t[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int w = 0; w < 4; ++w) t[q][w] = lds[q * 4 + w];
#pragma unroll
    for (int q = 0; q < NQ; ++q) out[q] = ((t[q][0] + t[q][1]) + t[q][2]) + t[q][3];
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double r = lds[q * nw];
      for (int w = 1; w < nw; ++w) r += lds[q * nw + w];
      out[q] = r;
    }
  }
  ipx_lds_barrier();
}

// The same fold in two halves: load() requests the first IPX_FOLD_U * blockDim
// entries of every array (no use yet), finish() adds them -- plus any entries
// beyond, the slow way -- and reduces.  Lets a kernel put the fold's loads in
// flight together with its first operand loads (one memory latency for both).
// Block-wide sums of NQ per-thread values in one barrier pair, every thread gets
// all results (fixed order: lanes by DPP butterfly, waves in wave order).
template <int NQ>
__device__ __forceinline__ void ipx_block_sum_multi(double (&v)[NQ], double *lds,
                                                    double (&out)[NQ]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    v[q] = ipx_wave_sum(v[q]);
    if (lane == 0) lds[q * nw + wave] = v[q];
  }
  ipx_lds_barrier();
  if (nw == 4) {                 // (all partner values requested together: ipx_block_reduce)
    double t[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int w = 0; w < 4; ++w) t[q][w] = lds[q * 4 + w];
#pragma unroll
    for (int q = 0; q < NQ; ++q) out[q] = ((t[q][0] + t[q][1]) + t[q][2]) + t[q][3];
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double r = lds[q * nw];
      for (int w = 1; w < nw; ++w) r += lds[q * nw + w];
      out[q] = r;
    }
  }
  ipx_lds_barrier();
}

template <int NQ, int FU = IPX_FOLD_U>
struct ipx_fold_regs {
  double t[NQ][FU];
  __device__ __forceinline__ void load(const double *const (&part)[NQ], const int (&count)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        // unconditional clamped load + select (see ipx_sum_partials); part[q]
        // must be a readable address even when count[q] is 0
        const int i = threadIdx.x + u * blockDim.x;
        const double ld = part[q][max(min(i, count[q] - 1), 0)];
        t[q][u] = i < count[q] ? ld : 0.0;
      }
    }
  }
  // this thread's share of every sum (entries beyond FU * blockDim the slow way)
  __device__ __forceinline__ void local(const double *const (&part)[NQ], const int (&count)[NQ],
                                        double *v) const {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      v[q] = 0.0;
#pragma unroll
      for (int u = 0; u < FU; ++u) v[q] += t[q][u];
      for (int i = threadIdx.x + FU * blockDim.x; i < count[q]; i += blockDim.x)
        v[q] += part[q][i];
    }
  }
  __device__ __forceinline__ void finish(const double *const (&part)[NQ], const int (&count)[NQ],
                                         double *lds, double (&out)[NQ]) const {
    double v[NQ];
    local(part, count, v);
    ipx_block_sum_multi<NQ>(v, lds, out);
  }
};

// The elements that count in a reduction: everything on one GPU; in the row-sharded loop a
// rank's OWN entries -- one range per segment of the local vector (x-space: one; the barrier
// problem's z = [x; s_nl; s_lb; s_ub]: four), halo copies in between.
struct ipx_own_ranges {
  int64_t lo[4], hi[4];
#ifdef __HIPCC__
  __device__ __forceinline__ bool has(int64_t i) const {
    return (i >= lo[0] && i < hi[0]) || (i >= lo[1] && i < hi[1]) || (i >= lo[2] && i < hi[2]) ||
           (i >= lo[3] && i < hi[3]);
  }
#endif
};

// ---- peer mailboxes of the row-sharded loop (csrc/peer.hip) ------------------------
// Every rank owns one mailbox in its own HBM (uncached allocation) that its peers map through
// hipIpc and write into directly over xGMI.  Values travel as "LL" words: a double is two
// 64-bit words, each carrying 32 data bits and a 32-bit sequence number; an aligned 8-byte
// store is indivisible, so a reader that sees the expected sequence number in both words has
// the value -- no flag, no fence, one store per word on the writer's side.
//   words [0, IPX_PEER_SCAL_WORDS): scalars  [slot 0..3][rank][8 quantities][2 words]
//   then the halo regions              [side: from the left / right neighbour][parity][cap][2 words]
#define IPX_MAX_PEERS 16
#define IPX_PEER_SLOTS 4
#define IPX_PEER_NQ 8
#define IPX_PEER_SCAL_WORDS (IPX_PEER_SLOTS * IPX_MAX_PEERS * IPX_PEER_NQ * 2)
#define IPX_PEER_TIMEOUT_TICKS 1000000000LL    // default deadline of a wait: 10 s of the 100 MHz
                                               // wall clock (ipx_peer_set_timeout changes it)
struct ipx_peer_view {
  int rank, world;
  int64_t cap;                                 // halo capacity (doubles) per side and parity
  long long timeout_ticks;                     // a wait longer than this raises stop code 7
  unsigned long long *mbox[IPX_MAX_PEERS];     // every rank's mailbox as mapped here (own: local)
};
// host-side object behind ipx_shard2_ext.peer
struct ipx_peer {
  ipx_peer_view view;
  uint32_t seq, hseq;                          // scalar / halo sequence numbers (never 0)
  void *opened[IPX_MAX_PEERS];                 // hipIpcOpenMemHandle results (to close)
  int64_t bytes;
  int64_t fused;                               // launches that did their collective in a prologue
  // hand-off buffers of the resident loop kernel's PEER form (csrc/resident.hip;
  // ipx_peer_attach_resident): every rank's buffer as mapped here, the same table in device
  // memory, the tag counter (identical on every rank: same launches, same batches)
  unsigned long long *res[IPX_MAX_PEERS];
  void *res_opened[IPX_MAX_PEERS];
  unsigned long long **res_tab;
  int64_t res_words;
  uint32_t rseq;
  int64_t res_launches;
};
#ifdef __HIPCC__
__device__ __forceinline__ int64_t ipx_peer_scal_word(int slot, int rank, int q) {
  return (((int64_t)slot * IPX_MAX_PEERS + rank) * IPX_PEER_NQ + q) * 2;
}
__device__ __forceinline__ int64_t ipx_peer_halo_word(int64_t cap, int side, int parity, int64_t i) {
  return IPX_PEER_SCAL_WORDS + (((int64_t)side * 2 + parity) * cap + i) * 2;
}
__device__ __forceinline__ void ipx_ll_store(unsigned long long *dst, double v, uint32_t seq) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  const unsigned long long tag = (unsigned long long)seq << 32;
  __hip_atomic_store(dst, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(dst + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// spins until both words carry `seq`; false past the deadline (a peer died or fell out of step)
__device__ __forceinline__ bool ipx_ll_load(const unsigned long long *src, uint32_t seq, double &v,
                                            long long deadline) {
  while (true) {
    const unsigned long long a = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long b = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((uint32_t)(a >> 32) == seq && (uint32_t)(b >> 32) == seq) {
      v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
      return true;
    }
    if ((long long)wall_clock64() > deadline) return false;
    __builtin_amdgcn_s_sleep(2);
  }
}
#endif
// What a loop kernel needs to do its rank's part of an all-reduce -- and, for the kernel that
// consumes g, of the halo exchange -- ITSELF, in its prologue (csrc/cg.hip, PEER variants): the
// mailboxes, the sequence numbers of this launch, the extents of the (single) segment.
// Segment k of the local vector is [seg_lo | own_lo .. own_hi | seg_hi) (x-space problems: one
// segment; the barrier problem's z = [x; s_nl; s_lb; s_ub]: four); in a mailbox's halo areas
// the segments follow each other: off_l / off_r = entries of the earlier segments' left /
// right halos (what this rank pulls), push_l / push_r = of their send counts (what it pushes).
struct ipx_peer_job {
  ipx_peer_view pv;
  uint32_t seq, hseq;
  int nseg;
  int seg_lo[4], own_lo[4], own_hi[4], seg_hi[4], send_left[4], send_right[4];
  int off_l[4], off_r[4], push_l[4], push_r[4];
  double *pack_out;                              // 4 doubles: the reduced sums, or NULL
#ifdef __HIPCC__
  // the halo word of this rank's mailbox that holds local entry `col`, or -1 (own / no peer)
  __device__ __forceinline__ int64_t pull_word(int col, int par) const {
    for (int k = 0; k < nseg; ++k) {
      if (pv.rank > 0 && col >= seg_lo[k] && col < own_lo[k])
        return ipx_peer_halo_word(pv.cap, 0, par, off_l[k] + (col - seg_lo[k]));
      if (pv.rank < pv.world - 1 && col >= own_hi[k] && col < seg_hi[k])
        return ipx_peer_halo_word(pv.cap, 1, par, off_r[k] + (col - own_hi[k]));
    }
    return -1;
  }
  __device__ __forceinline__ bool owns_rows(int r0, int r1) const {
    for (int k = 0; k < nseg; ++k)
      if (r0 >= own_lo[k] && r1 <= own_hi[k]) return true;
    return false;
  }
#endif
};
#ifdef __HIPCC__
// Sum over the ranks of NQ doubles that every workgroup of the launch holds (`mine`: the
// rank's own sums, the same bits in every workgroup): workgroup `sender` stores them as LL
// words into every rank's mailbox, every workgroup waits for the W contributions in its own
// rank's mailbox and adds them in rank order.  lds: NQ * IPX_MAX_PEERS + 1 doubles.  false when
// a wait timed out (a peer died): the caller raises stop code 7 and returns.
template <int NQ>
__device__ __forceinline__ bool ipx_peer_sum(const ipx_peer_view &pv, uint32_t seq, int q0,
                                             double (&mine)[NQ], bool sender, double *lds,
                                             double (&out)[NQ]) {
  const int tid = threadIdx.x, slot = seq & (IPX_PEER_SLOTS - 1);
  const int r = tid / NQ, q = tid - r * NQ;
  const bool lane = tid < NQ * pv.world;
  if (tid == 0) lds[NQ * IPX_MAX_PEERS] = 0.0;
  if (sender && lane) {
    double v = mine[0];
#pragma unroll
    for (int k = 1; k < NQ; ++k) v = q == k ? mine[k] : v;
    ipx_ll_store(pv.mbox[r] + ipx_peer_scal_word(slot, pv.rank, q0 + q), v, seq);
  }
  ipx_lds_barrier();
  if (lane) {
    const long long deadline = (long long)wall_clock64() + pv.timeout_ticks;
    double v = 0.0;
    if (!ipx_ll_load(pv.mbox[pv.rank] + ipx_peer_scal_word(slot, r, q0 + q), seq, v, deadline))
      lds[NQ * IPX_MAX_PEERS] = 1.0;
    lds[q * IPX_MAX_PEERS + r] = v;
  }
  ipx_lds_barrier();
  const bool ok = lds[NQ * IPX_MAX_PEERS] == 0.0;
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    double sum = 0.0;
    for (int w = 0; w < pv.world; ++w) sum += lds[k * IPX_MAX_PEERS + w];      // rank order
    out[k] = sum;
  }
  ipx_lds_barrier();
  return ok;
}
#endif

// ---- box-Schur group tables (csrc/boxschur.hip; also read by the CG loop's step1)
#ifdef __HIPCC__
struct ipx_group_tab {
  const int32_t *gcol;      // 3 per group: shared column, private column of p, of q
                            // (-1: the row has none; q: -2 = single-row group)
  const double *grp;        // 4 per group: ap, sp, aq, sq
  const double *grp2;       // 2 per group, or NULL: (sp, sq) carrying the signs of ap, aq --
                            // the form of box rows, whose shared-column entries are +-1 and
                            // whose slack entries are not negative (k_pairs_factor checks
                            // both): half the table bytes in the CG loop's two group kernels
  // affine != 0 (every variable bounded on both sides, the usual BoxConstraint): group g has
  // the columns (c0 + g, c0 + g + dp, c0 + g + dq) and the columns of no group are
  // gen0, gen0 + 1, ...: the two kernels compute them instead of reading gcol / gen_cols --
  // 12 bytes per group less and, more to the point, no table -> gather dependency: every load
  // of an item is issued at once and the vector loads coalesce
  int affine, c0, dp, dq, gen0;
};

inline ipx_group_tab ipx_boxschur_tab(const ipx_boxschur_args *a) {
  return ipx_group_tab{a->gcol, a->grp, a->grp2, (int)a->gaffine, (int)a->gc0, (int)a->gdp,
                       (int)a->gdq, (int)a->gen0};
}

// table forms of the CG loop's group kernels (template parameter MODE)
constexpr int IPX_GROUPS_FULL = 0;      // gcol + grp
constexpr int IPX_GROUPS_UNIT = 1;      // gcol + grp2
constexpr int IPX_GROUPS_AFFINE = 2;    // computed columns + grp2
__host__ __device__ inline int ipx_group_mode(const ipx_group_tab &T) {
  return !T.grp2 ? IPX_GROUPS_FULL : (T.affine ? IPX_GROUPS_AFFINE : IPX_GROUPS_UNIT);
}
template <int MODE>
__device__ __forceinline__ void ipx_group_cols(const ipx_group_tab &T, int g, int &c, int &cp,
                                               int &cq) {
  if constexpr (MODE == IPX_GROUPS_AFFINE) {
    c = T.c0 + g; cp = c + T.dp; cq = c + T.dq;
  } else {
    c = T.gcol[3 * g]; cp = T.gcol[3 * g + 1]; cq = T.gcol[3 * g + 2];
  }
}

// (ap, sp, aq, sq) of group g.  UNIT: from the compact table -- the same four doubles, bit
// for bit (|a| = 1 exactly, s >= 0 with its own sign bit clear)
template <bool UNIT>
__device__ __forceinline__ void ipx_group_coeffs(const ipx_group_tab &T, int g, double &ap,
                                                 double &sp, double &aq, double &sq) {
  if constexpr (UNIT) {
    const double ep = T.grp2[2 * g], eq = T.grp2[2 * g + 1];
    ap = copysign(1.0, ep); sp = fabs(ep);
    aq = copysign(1.0, eq); sq = fabs(eq);
  } else {
    ap = T.grp[4 * g]; sp = T.grp[4 * g + 1]; aq = T.grp[4 * g + 2]; sq = T.grp[4 * g + 3];
  }
}

// inverse (i11, i12, i22) of B = [[ap^2 + sp^2, ap aq], [ap aq, aq^2 + sq^2]] (single row: 1/b11)
// and the Schur column weight 1 - alpha' B^-1 alpha, both in cancellation-free form (slacks of
// active bounds are ~1e-8 next to ap = aq = 1).  One definition for the factorization and for
// the kernels that re-derive the inverse from (ap, sp, aq, sq) instead of reading it: same bits.
__device__ __forceinline__ bool ipx_group_inverse(bool has_q, double ap, double sp, double aq,
                                                  double sq, double &i11, double &i12,
                                                  double &i22, double &wgt) {
  const double b11 = ap * ap + sp * sp;
  if (!has_q) {
    i11 = 1.0 / b11; i12 = 0.0; i22 = 0.0;
    wgt = sp * sp / b11;
    return b11 > 0.0;
  }
  const double b22 = aq * aq + sq * sq, b12 = ap * aq;
  // (ap^2+sp^2)(aq^2+sq^2) - (ap aq)^2 without the cancellation
  const double det = ap * ap * (sq * sq) + sp * sp * (aq * aq) + sp * sp * (sq * sq);
  i11 = b22 / det; i12 = -b12 / det; i22 = b11 / det;
  wgt = (sp * sp) * (sq * sq) / det;
  return det > 0.0 && b11 > 0.0;
}

__device__ __forceinline__ void ipx_group_w(const ipx_group_tab &T, int g, const double *__restrict__ r,
                                        int &c, int &cp, int &cq, double &ap, double &sp,
                                        double &aq, double &sq, double &rc, double &rp, double &rq,
                                        double &wp, double &wq) {
  c = T.gcol[3 * g]; cp = T.gcol[3 * g + 1]; cq = T.gcol[3 * g + 2];
  ap = T.grp[4 * g]; sp = T.grp[4 * g + 1]; aq = T.grp[4 * g + 2]; sq = T.grp[4 * g + 3];
  rc = r[c];
  rp = cp >= 0 ? r[cp] : 0.0;
  rq = cq >= 0 ? r[cq] : 0.0;
  // row sums in column order (0 + first + second)
  wp = cp < 0 ? ap * rc : (cp > c ? ap * rc + sp * rp : sp * rp + ap * rc);
  wq = cq < 0 ? aq * rc : (cq > c ? aq * rc + sq * rq : sq * rq + aq * rc);
}

#endif

// ---- diagnostic build only (-DIPX_PHASE_TIMING, scripts/phase_timing.py): in-kernel
// wall-clock stamps of workgroup 0, one array per translation unit.
#ifdef IPX_PHASE_TIMING
#define IPX_STAMP_DECL(name) __device__ unsigned long long name[16]
#define IPX_STAMP_TO(name, k) \
  do { if (blockIdx.x == 0 && threadIdx.x == 0) name[k] = wall_clock64(); } while (0)
#define IPX_STAMP_EXPORT(fn, name)                                                         \
  extern "C" int fn(unsigned long long *out) {                                             \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(name), 16 * sizeof(unsigned long long)) ==  \
                   hipSuccess ? 0 : -1;                                                    \
  }
#else
#define IPX_STAMP_DECL(name)
#define IPX_STAMP_TO(name, k) do { } while (0)
#define IPX_STAMP_EXPORT(fn, name)
#endif

// ---- state block of the device-resident CG loop (doubles; csrc/cg.hip) -------------------
enum {
  ST_RTG0 = 0, ST_RTG1 = 1,   // rt_g, double buffered by iteration parity
  ST_TOL = 2, ST_RADIUS = 3, ST_ALPHA = 4, ST_STOP = 5, ST_NITER = 6, ST_BETA = 7,
  ST_PTHP = 8, ST_ORTH_RHS = 9,   // orth_tol * ||A||_F  (0 disables the check)
  ST_XNORM2 = 10, ST_VIOL = 11, ST_ORTH = 12, ST_IT_DONE = 13,
  ST_MARGIN = 14,   // written by the priming: min over its two projections of ||Z x||^2 / ||x||^2
                    // (how far they are from needing the cancellation step: 2^-20)
  ST_PRIME_STEPS = 15,   // ... and how many correction steps they took on the device (0..2)
  ST_SIZE = 16
};

struct ipx_prime_idx { int i[7]; };      // (csrc/cg.hip k_cg_prime_state)

// ---- resident projected-CG kernel (csrc/resident.hip): the cyclic-reduction geometry of a
// banded handle, and the launcher csrc/cg.hip's loop calls
struct ipx_pcr_view {
  int m, rows_wg, nwg, L;
  const double *band;
};
int ipx_banded_pcr_view(void *handle, ipx_pcr_view *out);
int ipx_cg_resident_launch(const ipx_cg_args *a, int32_t it_begin, int32_t it_end, int np1, int np2,
                           int np3, int np4, hipStream_t st);
int ipx_cg_shard2_resident_launch(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t it_begin,
                                  int32_t it_end, int np1, hipStream_t st);

// ipx_cg_prime with the trust radius and ||A||_F^2 optionally taken from device memory at
// execution time (radius_dev / norm_A2_dev non-NULL override the by-value arguments); steps != 0
// (b == NULL only): each projection may take one correction step on the device (k_prime_decide)
// -- eight more launches, no-ops when no step is due; steps == 0: a projection that needs one
// ends the priming with stop code 9 (the host's).  c_part (b == NULL, steps == 0): the 2 x
// c_npart epilogue partials of the product that formed c (||c||^2 without a norm launch, as
// the b != NULL path takes it from its own H x0 + c product); x_is_zero: the caller cleared
// a->x (no memset launch).
int ipx_cg_prime_dev(const ipx_cg_args *a, const int32_t *A_tiles, int32_t A_ntiles,
                     const double *c, const double *b, double *red, double *ws, double tol_in,
                     double radius, const double *radius_dev, double orth_tol, double norm_A,
                     const double *norm_A2_dev, double cancellation, int32_t first_end,
                     int steps, const double *c_part, int32_t c_npart, int x_is_zero,
                     hipStream_t stream);

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
#ifdef __HIPCC__
// the per-item tail of the Schur solve's kernel (csrc/banded.hip PostJob), as csrc/boxschur.hip
// hands it over: the tables' geometry (rows per workgroup of the solve they were laid out for,
// how many rows after its first one an item's second row may lie) travels with them
struct ipx_post_job {
  const int32_t *own_g, *own_e;
  int ng, nitems;
  ipx_group_tab T;
  const int32_t *yrow;
  const double *yval;
  const double *r;
  double *g;
  double *part;
  int count, rows_wg, reach;
};
int ipx_banded_solve_rows_launch(void *handle, const int32_t *col, const double *val,
                                 const double *xin, int logL, double *x, double *partial,
                                 int *npartial, const double *guard, hipStream_t st,
                                 const ipx_post_job *post = nullptr);
#endif
// ipx_boxschur_project with a->up (= r - alpha't) optionally prepared by the caller
int ipx_boxschur_project_from(const ipx_boxschur_args *a, const double *r, double *g,
                              double *part_g, int32_t *npart_g, double *part_res,
                              int32_t *npart_res, const double *guard, int have_up,
                              hipStream_t stream, const ipx_own_ranges *own = nullptr);
// yout = alpha A x [+ diag x] [+ beta yin] for a row-major dense A; partial (optional) gets
// per-workgroup sums of y^2 then of x.y (square A), *npartial entries per half
int ipx_dense_gemv_launch(int m, int n, const double *A, int64_t lda, const double *x,
                          double alpha, const double *diag, double beta, const double *yin,
                          double *yout, double *partial, int *npartial, const double *guard,
                          hipStream_t st);
// x = (A A')^-1 w and partial[0..*npartial) <- per-workgroup sums of ||w - (A A') x||^2
// in one go; the partial buffer needs ceil(m / 256) doubles.
int ipx_banded_resid_count(void *handle);
int ipx_banded_solve_resid_atv_launch(void *handle, const double *w, double *x, double *partial,
                                      int *npartial, const int32_t *At_rowptr,
                                      const int32_t *At_colidx, const double *At_val,
                                      const double *r_in, double *g_out, const int32_t *vown,
                                      int qv, double *part3, const double *guard,
                                      hipStream_t st, const uint16_t *ell_row = nullptr,
                                      const double *ell_val = nullptr, int64_t ell_n = 0,
                                      double sign = 1.0);
int ipx_banded_solve_resid_launch(void *handle, const double *w, double *x, double *partial,
                                  int *npartial, const double *guard, hipStream_t st);
// partial[0..*npartial) <- per-workgroup sums of ||w - (A A') v||^2 (<= 256 of them)
int ipx_banded_residual_launch(void *handle, const double *w, const double *v, double *partial,
                               int *npartial, const double *guard, hipStream_t st);
