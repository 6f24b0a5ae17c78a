// The whole projected-CG iteration (reference qp_subproblem.py:549-634) as ONE resident launch
// per BATCH of iterations, for problems small enough that every workgroup of the banded solve
// gets a CU of its own (<= 224 workgroups of 260 constraint rows: n <= ~5.8e5 on the
// benchmark's banded problem -- the per-rank sizes of a multi-GPU run).
//
// Why: at n = 1.25e5 the three dependent launches of csrc/cg.hip cost 21.8 us per iteration
// (profiles/r04_per_rank_sweep.json) for ~3 us of memory traffic: every launch pays its
// boundary, a state-word round trip, a partial fold before its first store and a block
// reduction + partial store after its last.  Here a workgroup owns 260 rows of A A' and the
// variables whose first constraint lies in them (the decomposition of k_solve_pcr's tail) for
// ALL of the iteration, keeps its matrix entries in registers (row-wise: a lane owns a row of
// the window of A, a few rows of H, a few columns of A') and its vectors in LDS for the whole
// batch, and talks to the other workgroups only through 8-byte tagged words ("LL"
// granules: 32 data bits + a 32-bit sequence number per word, one indivisible store each;
// ipx_common.h, the form the peer mailboxes use between GPUs) -- no fence, no flag, no grid
// barrier.  Two hops per iteration, the data dependencies of CG itself:
//
//   top      alpha = rt_g / p'Hp                                  (:551,:558,:579 on every WG)
//   phase P  r_next = r + alpha Hp on the window's columns; w = A r_next on the window's rows
//            (own rows + 2^L either side: a lane per row, sums left to right out of registers);
//            cyclic reduction; g = r_next - A'v on the own variables (A' in ELL(2) form, as in
//            k_solve_pcr's tail); partials of ||x + alpha p||^2, ||g||^2, ||w - (A A')v||^2
//   hop 2    every WG publishes its three partials and the first / last own entries of g its
//            neighbours' windows reach; waits for all partials (fixed-order fold: same bits in
//            every WG) and for its own halo of g
//   branch   :583 radius, orthogonality (projections.py:72), beta = ||g||^2 / rt_g
//   phase H  x += alpha p;  p = beta p - g on own + hmax;  Hp = H p on the own rows (row sums
//            left to right out of registers);  partial of p'Hp
//   hop 1    partial of p'Hp + the boundary entries of Hp -> everybody / the neighbours
//
// The batch's first iteration takes p'Hp from the partial array the previous launch (of any
// form) left; the last hop doubles as the commit: only a workgroup that has seen every other
// workgroup's final word writes x, p, r, Hp and the state block back, so a launch in which
// any wait timed out (stop code 8: a workgroup that never became resident) leaves memory
// exactly as it found it and the host repeats the batch with the separate launches.
//
// Everything a lane does in a phase is written as "all loads, then all arithmetic, then all
// stores": a workgroup has one or two waves per SIMD and nothing else hides an LDS round trip
// (the first version, one dependent LDS access after the other, took 27 us per iteration).
//
// Hand-off buffer.  Scalar records live in TWO arrays used in turn (hop h >= 1: array h & 1;
// hop 0 of a PEER launch has a third, see res_records): a
// workgroup can only publish hop h + 2 after every workgroup has published hop h + 1, i.e. has
// passed hop h -- so no word is overwritten while somebody still polls it, whatever the kind of
// the hops (round 4 alternated by KIND; the commit hop of a launch that stops at the top of an
// iteration then reused the words of the hop before it: ADVICE r4).  Halos: one slot of six
// areas per workgroup (Hp / g / p to the left / right neighbour), plus two slots for what the
// neighbour RANKS write (below).
//
// PEER form (the row-sharded loop, ipsolver/sharded.py; one process per GPU): the chain of
// workgroups simply continues across the ranks.  A rank launches one workgroup per OWN block of
// its extended local problem (tables indexed by the local block: J.wg0 + blockIdx.x); the scalar
// records are indexed by the GLOBAL workgroup and every workgroup stores its record into every
// rank's buffer (hipIpc-mapped, uncached; system-scope stores over xGMI) and polls its own
// rank's: one flat all-to-all per hop, folded in the same lane order on every rank -- the same
// bits as one GPU running all the workgroups.  The first / last workgroup of a rank exchange
// their halos with the neighbour rank's last / first through the two extra slots of the
// buffers.  A launch starts with a hop 0: the all-reduce of the p'Hp partials the previous
// launch (of any form) left on each rank, and the edge workgroups' halos of r, Hp and p (the
// local arrays' halo entries are not maintained by this kernel: the host synchronises them
// when it leaves the loop).  Budget: all workgroups of all ranks <= 512 (one record per lane).
//
// Arithmetic: element by element the expressions of k_cg_step1_ar / k_solve_pcr /
// k_cg_step2_hp, row sums left to right; the sums differ in their order: ||g||^2 and the
// residual are summed per workgroup by 256 lanes (the order k_solve_pcr had while it ran 256
// lanes; it runs 512 since), p'Hp and ||x + alpha p||^2 per workgroup of THIS decomposition
// instead of per row tile of H / A -- so alpha and beta can differ in the last bit (tests:
// 1e-13 relative over whole solves, identical branch decisions).
#include "ipx_common.h"
#include <algorithm>

namespace {

IPX_STAMP_DECL(ipx_dbg_res);
#ifdef IPX_PHASE_TIMING     // (an interior workgroup: the edge ones have no left / right halo)
#define RS_STAMP(k) \
  do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) ipx_dbg_res[k] = wall_clock64(); } while (0)
#define RS_STAMP_SYNC(k) do { ipx_lds_barrier(); RS_STAMP(k); } while (0)
#else
#define RS_STAMP(k) do { } while (0)
#define RS_STAMP_SYNC(k) do { } while (0)
#endif

typedef unsigned long long ull;

constexpr int RB = 512;       // threads per workgroup: two waves per SIMD, 256 registers per lane
                              // (256 threads hold twice the static data per lane: the arch VGPRs
                              // overflow into AGPR copies and 195 scratch words, reloaded in every
                              // iteration -- 25 us per iteration, measured)
constexpr int RNR = 1;        // window rows per lane    (<= 512)
constexpr int RLA = 16;       // entries per row of A held in registers
constexpr int RQS = 8;        // span doubles per lane   (<= 4096 columns)
constexpr int RQX = 6;        // own variables per lane  (<= 3072)
constexpr int RLH = 4;        // entries per row of H held in registers
constexpr int RHK = 4;        // halo entries per lane and hop (2 * hw <= RHK * RB)
constexpr int RQP = 3;        // pairs of own variables per lane
constexpr int R_MAXG = 512;                        // scalar records per array (one per lane)
constexpr int R_REC = 8;                           // words per record: 4 granules, 3 used
constexpr int R_HALO = 3 * R_REC * R_MAXG;         // word offset of the halo slots (three record arrays)
constexpr int R_AREAS = 6;                         // per slot: Hp, g, p to the left (even) / right (odd)
constexpr int R_MAXLOCAL = 224;                    // workgroups of one launch (all co-resident)

struct ResJob {
  double *st;
  int it_begin, it_end, n, m, rows_wg, L, nwg;
  const double *band;
  double *x, *p, *r, *Hp;
  const double *A_val;
  const uint16_t *A_off16;
  const int32_t *A_rowfirst;
  int rl;
  const int32_t *win, *vown;
  int nspan, navn;
  const uint16_t *ell_row;      // A' in ELL(2) form (ipx_cg_args.At_ell_row / At_ell_val)
  const double *ell_val;
  const int32_t *H_rowptr, *H_colidx;
  const double *H_val, *H_diag;
  int hmax;
  double *part1, *part2, *part3, *part4;
  int p1_off, p1_cnt;           // the p'Hp partials this launch folds / leaves: part1[p1_off ..+p1_cnt)
  int np2, np3, np4;
  ull *ll;
  int hw;
  uint32_t seq;
  int no_xn2;
  long long timeout;
  int stop_code;                // what a timed-out wait records: 8 (one GPU), 7 (PEER)
  // PEER form: nwg workgroups are launched for the local blocks wg0 .. wg0 + nwg - 1; they are
  // the global workgroups gwg0 .. of gnwg; pll[r] = rank r's buffer as mapped here (own: ll)
  int wg0, gwg0, gnwg, rank, world;
  ull *const *pll;
  double *pack_out;             // 4 doubles: the sums of the last projection for the host's resume
};

// (hop 0 -- the PEER form's opening all-reduce -- has an array of its own: a launch that ran all
// its iterations ends on an even hop, array 0, and a rank that read stop 0 relaunches without any
// agreement on the host: its hop-0 records must not land where a slow workgroup of a peer still
// polls the previous launch's last hop.  Hop 1 of the new launch can only be published after
// every workgroup of every rank has published ITS hop 0, i.e. has left the previous launch.)
__device__ __forceinline__ ull *res_records(ull *base, uint32_t hop) {
  return base + (hop == 0 ? 2 : (int)(hop & 1)) * (R_REC * R_MAXG);
}
__device__ __forceinline__ ull *res_area(ull *base, int slot, int area, int hw) {
  return base + R_HALO + ((int64_t)(slot * R_AREAS + area) * hw) * 2;
}

// The packed index registers are unpacked INSIDE the loop: without this the compiler hoists
// every unpacked LDS address out of it (loop invariant!) and holds ~100 more registers across
// the whole batch -- which then spill to scratch and are reloaded in every iteration.
__device__ __forceinline__ int res_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ double res_rcp(double b) {        // (csrc/banded.hip pcr_rcp)
  double r = __builtin_amdgcn_rcp(b);
  return __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
}
// One granule = the two tagged words of a double, written by ONE 16-byte write-through store
// (8-byte sc1 stores are one fabric write each and cost 2.7x per byte: MI355X_MICROARCH.md,
// stores table; every 8-byte half validates itself, so tearing between the halves is harmless).
// SYS: system scope (sc0 sc1) -- the destination may be another GPU's memory.
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <bool SYS>
__device__ __forceinline__ void ll_put(ull *dst, double v, uint32_t tag) {
  const ull bits = (ull)__double_as_longlong(v);
  u4 w;
  w.x = (unsigned)(bits & 0xffffffffull); w.y = tag;
  w.z = (unsigned)(bits >> 32);           w.w = tag;
  if constexpr (SYS)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(w) : "memory");
  else
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(w) : "memory");
}
__device__ __forceinline__ bool ll_hit(const u4 &w, uint32_t tag, double &v) {
  if (w.y == tag && w.w == tag) {
    v = __longlong_as_double((long long)((ull)w.x | ((ull)w.z << 32)));
    return true;
  }
  return false;
}
// where workgroup `wg` of this launch publishes area `area` (even: for its left neighbour, odd:
// for its right one): its own slot -- or, at a rank's edge, the neighbour rank's slot 1 / 0
template <bool PEER>
__device__ __forceinline__ ull *halo_dst(const ResJob &J, int wg, int area) {
  if constexpr (PEER) {
    if (!(area & 1) && wg == 0 && J.rank > 0) return res_area(J.pll[J.rank - 1], 1, area, J.hw);
    if ((area & 1) && wg == J.nwg - 1 && J.rank < J.world - 1)
      return res_area(J.pll[J.rank + 1], 0, area, J.hw);
  }
  return res_area(J.ll, wg + 2, area, J.hw);
}
// where it finds what its left / right neighbour published for it (area: the neighbour's)
__device__ __forceinline__ const ull *halo_from_left(const ResJob &J, int wg, int area) {
  return res_area(J.ll, wg == 0 ? 0 : wg + 1, area, J.hw);
}
__device__ __forceinline__ const ull *halo_from_right(const ResJob &J, int wg, int area) {
  return res_area(J.ll, wg == J.nwg - 1 ? 1 : wg + 3, area, J.hw);
}
#define RED_NEXT (red + 32 * (int)(red_use++ & 1u))
// ipx_block_sum_multi with the wave count a compile-time constant: the partner sums of ALL
// quantities are requested from LDS together and added in wave order.  (The library routine
// loops to blockDim / 64 with one dependent LDS round trip per trip: 7 trips x NQ quantities
// cost 1.1 us for NQ = 2 and 2.4 us for NQ = 4 here -- measured, the largest single item of
// the first profile.)  Same order of additions: same bits.
template <int NQ>
__device__ __forceinline__ void res_block_sum(double (&v)[NQ], double *lds, double (&out)[NQ]) {
  constexpr int NW = RB / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    v[q] = ipx_wave_sum(v[q]);
    if (lane == 0) lds[q * NW + wave] = v[q];
  }
  ipx_lds_barrier();
  double t[NQ][NW];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int w = 0; w < NW; ++w) t[q][w] = lds[q * NW + w];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double r = t[q][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r += t[q][w];
    out[q] = r;
  }
  // (no barrier behind the reads: consecutive reductions take the two halves of the buffer in
  // turn -- RED_NEXT --, and between two uses of one half lies the other use's barrier, which
  // a wave only reaches with its LDS reads complete)
}

// (The hop helpers take the lane index as an argument -- the caller's per-iteration opaque copy:
// computed from threadIdx.x their 20-odd word addresses are loop invariants the compiler
// hoists and then spills; passing it in took the kernel from 176 to 32 bytes of scratch per
// lane and the iteration at n = 5e5 from 18.9 to 15.4 us.)
// One hop: NS scalars per workgroup (lane t < nrec waits for record t of `scal`) and this
// workgroup's left / right halo (nl + nr entries, from the areas `left` / `right`) into LDS at
// dst_l / dst_r.  Every word of the lane is requested in ONE burst per pass (unconditional
// loads, no branch between them), then the tags are checked.  false: a wait timed out.
#define RES_BURST7(SC)                                                                           \
  asm volatile("global_load_dwordx4 %0, %7, off " SC "\n\t"                                      \
               "global_load_dwordx4 %1, %8, off " SC "\n\t"                                      \
               "global_load_dwordx4 %2, %9, off " SC "\n\t"                                      \
               "global_load_dwordx4 %3, %10, off " SC "\n\t"                                     \
               "global_load_dwordx4 %4, %11, off " SC "\n\t"                                     \
               "global_load_dwordx4 %5, %12, off " SC "\n\t"                                     \
               "global_load_dwordx4 %6, %13, off " SC "\n\t"                                     \
               "s_waitcnt vmcnt(0)"                                                              \
               : "=&v"(sw[0]), "=&v"(sw[1]), "=&v"(sw[2]), "=&v"(hw0), "=&v"(hw1), "=&v"(hw2),   \
                 "=&v"(hw3)                                                                      \
               : "v"(ssrc[0]), "v"(ssrc[1]), "v"(ssrc[2]), "v"(hsrc[0]), "v"(hsrc[1]),           \
                 "v"(hsrc[2]), "v"(hsrc[3])                                                      \
               : "memory")
#define RES_BURST5(SC)                                                                           \
  asm volatile("global_load_dwordx4 %0, %5, off " SC "\n\t"                                      \
               "global_load_dwordx4 %1, %6, off " SC "\n\t"                                      \
               "global_load_dwordx4 %2, %7, off " SC "\n\t"                                      \
               "global_load_dwordx4 %3, %8, off " SC "\n\t"                                      \
               "global_load_dwordx4 %4, %9, off " SC "\n\t"                                      \
               "s_waitcnt vmcnt(0)"                                                              \
               : "=&v"(sw[0]), "=&v"(hw0), "=&v"(hw1), "=&v"(hw2), "=&v"(hw3)                    \
               : "v"(ssrc[0]), "v"(hsrc[0]), "v"(hsrc[1]), "v"(hsrc[2]), "v"(hsrc[3])            \
               : "memory")
template <int NS, bool SYS>
__device__ __forceinline__ bool hop_wait(const ResJob &J, uint32_t tag, const ull *scal, int nrec,
                                         double (&sv)[NS], const ull *left, const ull *right,
                                         int nl, int nr, double *dst_l, double *dst_r,
                                         long long timeout, int tid) {
  const long long deadline = (long long)wall_clock64() + timeout;
  bool sdone[NS];
  const bool slane = tid < nrec;
#pragma unroll
  for (int q = 0; q < NS; ++q) { sdone[q] = !slane; sv[q] = 0.0; }
  bool hdone[RHK];
  double hv[RHK];
  const ull *hsrc[RHK], *ssrc[NS];
  const int nh = nl + nr;
#pragma unroll
  for (int q = 0; q < NS; ++q) ssrc[q] = scal + (int64_t)(slane ? tid : 0) * R_REC + 2 * q;
#pragma unroll
  for (int k = 0; k < RHK; ++k) {
    const int e = tid + k * RB;
    hdone[k] = e >= nh;
    hv[k] = 0.0;
    hsrc[k] = hdone[k] ? ssrc[0] : (e < nl ? left + 2 * e : right + 2 * (e - nl));
  }
  static_assert(RHK == 4, "the load burst below is written for four halo granules per lane");
  while (true) {
    // one burst of 16-byte loads (sc1: never served from this CU's L1; sc0 sc1: nor from an L2
    // line a peer's store went past), one wait (ONE asm statement: a result must not be touched
    // before the wait at its end)
    u4 sw[3], hw0, hw1, hw2, hw3;
    static_assert(NS == 1 || NS == 3, "hop_wait: one or three scalars per workgroup");
    if constexpr (NS == 3) {
      if constexpr (SYS) RES_BURST7("sc0 sc1"); else RES_BURST7("sc1");
    } else {
      if constexpr (SYS) RES_BURST5("sc0 sc1"); else RES_BURST5("sc1");
    }
    bool all = true;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      double v;
      if (!sdone[q] && ll_hit(sw[q], tag, v)) { sv[q] = v; sdone[q] = true; }
      all = all && sdone[q];
    }
    {
      const u4 hw[RHK] = {hw0, hw1, hw2, hw3};
#pragma unroll
      for (int k = 0; k < RHK; ++k) {
        double v;
        if (!hdone[k] && ll_hit(hw[k], tag, v)) { hv[k] = v; hdone[k] = true; }
        all = all && hdone[k];
      }
    }
    if (all) break;
    if ((long long)wall_clock64() > deadline) return false;
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int k = 0; k < RHK; ++k) {
    const int e = tid + k * RB;
    if (e < nh) {
      if (e < nl) dst_l[e] = hv[k]; else dst_r[e - nl] = hv[k];
    }
  }
  return true;
}

// this workgroup's record of a hop: NS doubles into the record array of every rank (PEER) /
// of this GPU
template <int NS, bool PEER>
__device__ __forceinline__ void rec_put(const ResJob &J, uint32_t hop, int gw, const double (&v)[NS],
                                        uint32_t tag, int tid) {
  if constexpr (PEER) {
    if (tid < NS * J.world) {
      const int r = tid / NS, q = tid - r * NS;
      double val = v[0];
#pragma unroll
      for (int k = 1; k < NS; ++k) val = q == k ? v[k] : val;
      ll_put<true>(res_records(J.pll[r], hop) + (int64_t)R_REC * gw + 2 * q, val, tag);
    }
  } else {
    if (tid < NS) {
      double val = v[0];
#pragma unroll
      for (int k = 1; k < NS; ++k) val = tid == k ? v[k] : val;
      ll_put<false>(res_records(J.ll, hop) + (int64_t)R_REC * gw + 2 * tid, val, tag);
    }
  }
}

// publish `cnt` doubles of LDS (src[0..cnt), cnt <= RHK/2 * RB) into one of this workgroup's areas
template <bool PEER>
__device__ __forceinline__ void halo_put(const ResJob &J, int wg, int area, const double *src,
                                         int cnt, uint32_t tag, int tid) {
  ull *dst = halo_dst<PEER>(J, wg, area);
  double v[RHK / 2];
#pragma unroll
  for (int k = 0; k < RHK / 2; ++k) v[k] = src[min(tid + k * RB, max(cnt - 1, 0))];
#pragma unroll
  for (int k = 0; k < RHK / 2; ++k) {
    const int j = tid + k * RB;
    if (j < cnt) ll_put<PEER>(dst + 2 * j, v[k], tag);
  }
}

template <bool NOXN2, bool HAS_DIAG, bool PEER>
__global__ void __launch_bounds__(RB, 2)
k_cg_resident(ResJob J) {
  extern __shared__ __attribute__((aligned(16))) double rs_lds[];
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int wgl = wg + J.wg0;                       // block of the (local) solve: the tables' index
  const int gw = wg + J.gwg0;                       // global workgroup: the scalar records' index
  const bool has_left = wg > 0 || (PEER && J.rank > 0);
  const bool has_right = wg < J.nwg - 1 || (PEER && J.rank < J.world - 1);
  const int H = 1 << J.L;
  const int R = J.rows_wg + 2 * H;
  const int RS = R + 2 * H;                         // PCR rows incl. identity padding
  const int nspanP = (J.nspan + 1) & ~1;
  const int npsp = (J.navn + 2 * J.hmax + 1) & ~1;
  double *rspan = rs_lds;                           // r / r_next / g on the span
  double *hspan = rspan + nspanP;                   // Hp on the span
  double *U = hspan + nspanP;                       // PCR ping-pong (6 RS) | squares of g, of the residual
  const int usize = 6 * RS;
  double *sx = U + usize;                           // [R]: w, then v
  double *pspan = sx + ((R + 1) & ~1);              // p on own +- hmax
  double *aval = pspan + npsp;                      // A's window rows: [R * rl] values
  double *red = aval + ((R * J.rl + 1) & ~1);       // [2 x 32] reductions (RED_NEXT)
  uint32_t red_use = 0;
  double *pa0 = U, *pa1 = U + RS, *pr0 = U + 2 * RS, *pr1 = U + 3 * RS, *pd0 = U + 4 * RS,
         *pd1 = U + 5 * RS;

  // ---- geometry
  const int c_lo = J.win[2 * wgl], c_hi = J.win[2 * wgl + 1];
  const int nspan = c_hi - c_lo;
  const int av0 = J.vown[wgl], av1 = J.vown[wgl + 1], avn = av1 - av0;
  const int nl = av0 - c_lo, nr = c_hi - av1;       // halo columns left / right of the own ones
  const int pl = has_left ? J.win[2 * (wgl - 1) + 1] - av0 : 0;           // own entries the left /
  const int pr = has_right ? av1 - J.win[2 * (wgl + 1)] : 0;              // right neighbour reads
  const int own_off = av0 - c_lo;                   // span index of the first own variable
  const int64_t g0 = (int64_t)wgl * J.rows_wg - H;  // (local) row of window row 0
  const int rl = J.rl;
  const int p_lo = av0 - J.hmax;                    // column of pspan[0]

  // ---- state words, p'Hp partials (requested first)
  const double stop0 = J.st[ST_STOP];
  double rt[2] = {J.st[ST_RTG0], J.st[ST_RTG1]};
  const double tol = J.st[ST_TOL], radius = J.st[ST_RADIUS], orth_rhs = J.st[ST_ORTH_RHS];
  const double *const fparts[1] = {J.part1 + J.p1_off};
  const int fcounts[1] = {J.p1_cnt};
  ipx_fold_regs<1, 4> fold;
  fold.load(fparts, fcounts);

  // ---- static data.  Where it lives was decided by measurement: everything in registers
  // overflows the 256 a lane has at two waves per SIMD (the compiler also hoists every loop
  // invariant address: res_opaque) and the overflow is reloaded from scratch one dependent
  // round trip at a time (3.9 us in the H.p phase alone); re-reading A and H row-wise from L2
  // every iteration is uncoalesced (a wave instruction touches 64 cache lines: +4 us per hop).
  // So: H's own rows in registers, A's window rows in LDS (filled coalesced, once), A' (ELL,
  // whose global layout IS lane-major) streamed from L2 under the cyclic reduction.
  // (i) window row tid of A: values in LDS, span indices packed two per register; the band
  double a0[RNR], b0[RNR];
  bool row_in[RNR];
  int ac2[RLA / 2];
  {
    const int r = tid;
    const int64_t grow = g0 + r;
    row_in[0] = r < R && grow >= 0 && grow < J.m;
    const int64_t gc = min(max(grow, (int64_t)0), (int64_t)J.m - 1);
    const int first = J.A_rowfirst[gc] - c_lo;
#pragma unroll
    for (int k = 0; k < RLA / 2; ++k) {
      const int c0 = row_in[0] ? first + (int)J.A_off16[gc * rl + min(2 * k, rl - 1)] : 0;
      const int c1 = row_in[0] ? first + (int)J.A_off16[gc * rl + min(2 * k + 1, rl - 1)] : 0;
      ac2[k] = c0 | (c1 << 16);
    }
    const double bv = J.band[gc], avv = J.band[(int64_t)J.m + gc];
    a0[0] = (row_in[0] && grow >= 1 && r >= 1) ? avv : 0.0;     // (row 0 of the window: cut)
    b0[0] = row_in[0] ? bv : 1.0;
  }
  for (int i = tid; i < R * rl; i += RB) {
    const int64_t kk = g0 * rl + i;                  // (window row i / rl, its entry i % rl)
    aval[i] = (kk >= 0 && kk < (int64_t)J.m * rl) ? J.A_val[kk] : 0.0;
  }
  // (the sub-diagonal entry of the row below, for the residual of the own rows)
  double a0n;
  {
    const int64_t gn = g0 + tid + 1;
    const bool in = tid + 1 < R && gn >= 1 && gn < J.m;
    a0n = in ? J.band[(int64_t)J.m + min(max(gn, (int64_t)0), (int64_t)J.m - 1)] : 0.0;
  }
  __builtin_amdgcn_sched_barrier(0);      // (keep the set-up's load groups apart: their
                                          //  addresses and clamps would all be live at once)
  // (ii) A' on the own variables, ELL(2): lane t takes the pairs (vb + 2 (t + RB k), +1) -- pairs
  // start at an even variable like in k_solve_pcr's tail; the window rows of the entries in
  // registers, the values re-read every iteration (16-byte loads, consecutive lanes consecutive
  // pairs: the array's own layout)
  const int64_t vb = av0 & ~1;
  int ec2[RQP][2];               // window rows of (entry 0 | entry 1 << 16) of variables j, j + 1
  {
    const int64_t lastj = max((int64_t)av1 - 1, vb) & ~(int64_t)1;
#pragma unroll
    for (int k = 0; k < RQP; ++k) {
      const int64_t j = min(vb + 2 * (int64_t)(tid + k * RB), lastj);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t jj = min(j + u, (int64_t)J.n - 1);
        const int c0 = min(H + (int)J.ell_row[jj], R - 1);
        const int c1 = min(c0 + 1, R - 1);
        ec2[k][u] = c0 | (c1 << 16);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);      // (keep the set-up's load groups apart: their
                                          //  addresses and clamps would all be live at once)
  // (iii) own rows tid + k RB of H (absent entries: value 0 on a valid column, so the row sums
  // need no length), x and the diagonal term on them
  double hv[RQX][RLH], xo[RQX], dg[RQX];
  int hc2[RQX][RLH / 2];         // pspan indices of the entries, two per register
#pragma unroll
  for (int k = 0; k < RQX; ++k) {
    const int i = min(tid + k * RB, max(avn - 1, 0));
    const int row = av0 + i;
    const int a = J.H_rowptr[row], b = J.H_rowptr[row + 1];
    const bool on = tid + k * RB < avn;
    int cc[RLH];
#pragma unroll
    for (int t = 0; t < RLH; ++t) {
      const int kk = min(a + t, max(b - 1, a));
      const bool have = on && a + t < b;
      const double v = J.H_val[kk];
      const int c = J.H_colidx[kk];
      hv[k][t] = have ? v : 0.0;
      cc[t] = have ? c - p_lo : J.hmax;
    }
#pragma unroll
    for (int t = 0; t < RLH / 2; ++t) hc2[k][t] = cc[2 * t] | (cc[2 * t + 1] << 16);
    xo[k] = J.x[row];
    dg[k] = HAS_DIAG ? J.H_diag[row] : 0.0;
  }
  __builtin_amdgcn_sched_barrier(0);      // (keep the set-up's load groups apart: their
                                          //  addresses and clamps would all be live at once)
  // ---- the vectors into LDS
#pragma unroll
  for (int k = 0; k < RQS; ++k) {
    const int j = tid + k * RB;
    if (j < nspan) { rspan[j] = J.r[c_lo + j]; hspan[j] = J.Hp[c_lo + j]; }
  }
  for (int j = tid; j < avn + 2 * J.hmax; j += RB) {
    const int col = p_lo + j;
    pspan[j] = (col >= 0 && col < J.n) ? J.p[col] : 0.0;
  }
  __builtin_amdgcn_sched_barrier(0);      // (keep the set-up's load groups apart: their
                                          //  addresses and clamps would all be live at once)
  if (stop0 != 0.0) return;                         // (every workgroup alike: nothing was written)
  const bool lead = wg == 0 && tid == 0;
  double ptHp;
  {
    double fout[1];
    fold.finish(fparts, fcounts, RED_NEXT, fout);
    ptHp = fout[0];
  }
  // bookkeeping (written back by the lead lane after the commit hop)
  double st_alpha = J.st[ST_ALPHA], st_beta = J.st[ST_BETA], st_pthp = J.st[ST_PTHP];
  double st_xn2 = J.st[ST_XNORM2], st_orth = J.st[ST_ORTH];
  int niter_inc = 0, done_inc = 0, stop = 0;
  uint32_t hop = 0;                                 // hops so far: tag = J.seq + hop
  double part_xn2 = 0.0, part_gg = 0.0, part_tt = 0.0;   // totals of the last projection
  bool have_proj = false;
  if constexpr (PEER) {
    // ---- hop 0: p'Hp over the ranks (workgroup 0 of a rank speaks for its partials) and, at a
    // rank's edges, the neighbour rank's entries of r, Hp and p
    const uint32_t tag = J.seq;
    const bool le = wg == 0 && J.rank > 0, re = wg == J.nwg - 1 && J.rank < J.world - 1;
    ipx_lds_barrier();                              // (the vectors are in LDS)
    if (le) {
      halo_put<PEER>(J, wg, 2, rspan + own_off, pl, tag, tid);
      halo_put<PEER>(J, wg, 0, hspan + own_off, pl, tag, tid);
      halo_put<PEER>(J, wg, 4, pspan + J.hmax, J.hmax, tag, tid);
    }
    if (re) {
      halo_put<PEER>(J, wg, 3, rspan + own_off + avn - pr, pr, tag, tid);
      halo_put<PEER>(J, wg, 1, hspan + own_off + avn - pr, pr, tag, tid);
      halo_put<PEER>(J, wg, 5, pspan + avn, J.hmax, tag, tid);
    }
    const double mine0[1] = {wg == 0 ? ptHp : 0.0};
    rec_put<1, PEER>(J, hop, gw, mine0, tag, tid);
    double s1[1], dum[1];
    bool ok = hop_wait<1, PEER>(J, tag, res_records(J.ll, hop), J.gnwg, s1,
                                halo_from_left(J, wg, 3), halo_from_right(J, wg, 2),
                                le ? nl : 0, re ? nr : 0, rspan, rspan + own_off + avn, J.timeout, tid);
    if (le || re) {
      ok = hop_wait<1, PEER>(J, tag, J.ll, 0, dum, halo_from_left(J, wg, 1), halo_from_right(J, wg, 0),
                             le ? nl : 0, re ? nr : 0, hspan, hspan + own_off + avn, J.timeout, tid) && ok;
      ok = hop_wait<1, PEER>(J, tag, J.ll, 0, dum, halo_from_left(J, wg, 5), halo_from_right(J, wg, 4),
                             le ? J.hmax : 0, re ? J.hmax : 0, pspan, pspan + J.hmax + avn,
                             J.timeout, tid) && ok;
    }
    double sv[2] = {s1[0], ok ? 0.0 : 1.0}, tot[2];
    res_block_sum<2>(sv, RED_NEXT, tot);                 // (barriers inside: the halos are in LDS)
    if (tot[1] != 0.0) {
      if (tid == 0) { J.st[ST_VIOL] = 80.0; J.st[ST_STOP] = (double)J.stop_code; }
      return;
    }
    ptHp = tot[0];
  }

  const int tid0 = tid;
  for (int it = J.it_begin; it < J.it_end; ++it) {
    // (an opaque copy of the lane index per iteration: nothing computed from it -- LDS
    // addresses, clamps, predicates -- is a loop invariant the compiler could hoist and hold in
    // registers across the batch; measured: 130 of 262 loop-carried registers were such)
    const int tid = res_opaque(tid0);
    const int par = it & 1;
    const double rtg = rt[par];
    if (rtg < tol) { stop = 4; break; }                      // qp_subproblem.py:551
    if (ptHp <= 0.0) { niter_inc += 1; st_pthp = ptHp; stop = 3; break; }   // :558
    const double alpha = rtg / ptHp;                          // :579
    niter_inc += 1; st_pthp = ptHp; st_alpha = alpha;
    RS_STAMP(0);
    // ================= phase P: r_next, w = A r_next, cyclic reduction, g ====================
    ipx_lds_barrier();
#pragma unroll
    for (int k0 = 0; k0 < RQS; k0 += 4) {
      double rv[4], hh[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = min(tid + (k0 + k) * RB, nspan - 1);
        rv[k] = rspan[j]; hh[k] = hspan[j];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = tid + (k0 + k) * RB;
        if (j < nspan) rspan[j] = rv[k] + alpha * hh[k];                 // :622
      }
    }
    double sxx = 0.0;
    if (!NOXN2) {
#pragma unroll
      for (int k0 = 0; k0 < RQX; k0 += 6) {
        double pv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) pv[k] = pspan[J.hmax + min(tid + (k0 + k) * RB, avn - 1)];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          if (tid + (k0 + k) * RB < avn) {
            const double xn = xo[k0 + k] + alpha * pv[k];                // :580 (not stored)
            sxx += xn * xn;
          }
        }
      }
    }
    ipx_lds_barrier();
    RS_STAMP(1);
    // w on the window rows: row sums left to right out of registers (scipy's csr_matvec order)
    double a[RNR], b[RNR], d[RNR], w0[RNR];
    {
      const double *arow = aval + min(tid, R - 1) * rl;
      double sum = 0.0;
#pragma unroll
      for (int k0 = 0; k0 < RLA; k0 += 8) {
        double rr[8], aa[8];
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          const int pk = res_opaque(ac2[(k0 + k) >> 1]);
          rr[k] = rspan[pk & 0xffff];
          rr[k + 1] = rspan[(pk >> 16) & 0xffff];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) aa[k] = arow[min(k0 + k, rl - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k0 + k < rl) sum += aa[k] * rr[k];
      }
      d[0] = row_in[0] ? 1.0 * sum : 0.0;
      w0[0] = d[0]; a[0] = a0[0]; b[0] = b0[0];
    }
    RS_STAMP(2);
    // (the values of A' for the tail, requested now, used after the cyclic reduction)
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d e0[RQP], e1[RQP];
    {
      const int64_t lastj = max((int64_t)av1 - 1, vb) & ~(int64_t)1;
#pragma unroll
      for (int k = 0; k < RQP; ++k) {
        const int64_t j = min(vb + 2 * (int64_t)(tid + k * RB), lastj);
        e0[k] = *reinterpret_cast<const v2d *>(J.ell_val + j);
        e1[k] = *reinterpret_cast<const v2d *>(J.ell_val + (int64_t)J.n + j);
      }
    }
    // cyclic reduction on the window (k_solve_pcr's arithmetic)
    for (int i = tid; i < 2 * H; i += RB) {
      const int r = i < H ? i : R + i;                // storage index = window row + H
      pa0[r] = 0.0; pr0[r] = 1.0; pd0[r] = 0.0;
      pa1[r] = 0.0; pr1[r] = 1.0; pd1[r] = 0.0;
    }
    for (int sl = 0; sl < J.L; ++sl) {
      const int h = 1 << sl;
      double *pa = (sl & 1) ? pa1 : pa0, *pr_ = (sl & 1) ? pr1 : pr0, *pd = (sl & 1) ? pd1 : pd0;
#pragma unroll
      for (int q = 0; q < RNR; ++q) {
        const int r = tid + q * RB;
        const double rc = res_rcp(b[q]);
        if (r < R) { pa[H + r] = a[q]; pr_[H + r] = rc; pd[H + r] = d[q]; }
      }
      ipx_lds_barrier();
      double alo[RNR], rlo_[RNR], dlo[RNR], ahi[RNR], rhi_[RNR], dhi[RNR];
#pragma unroll
      for (int q = 0; q < RNR; ++q) {
        const int r = H + min(tid + q * RB, R - 1);
        alo[q] = pa[r - h]; rlo_[q] = pr_[r - h]; dlo[q] = pd[r - h];
        ahi[q] = pa[r + h]; rhi_[q] = pr_[r + h]; dhi[q] = pd[r + h];
      }
#pragma unroll
      for (int q = 0; q < RNR; ++q) {
        const double al = -a[q] * rlo_[q], ga = -ahi[q] * rhi_[q];
        double bn = __builtin_fma(al, a[q], b[q]);
        bn = __builtin_fma(ga, ahi[q], bn);
        double dn = __builtin_fma(al, dlo[q], d[q]);
        dn = __builtin_fma(ga, dhi[q], dn);
        a[q] = al * alo[q]; b[q] = bn; d[q] = dn;
      }
    }
#pragma unroll
    for (int q = 0; q < RNR; ++q) {
      const int r = tid + q * RB;
      if (r < R) sx[r] = d[q] / b[q];
    }
    ipx_lds_barrier();
    RS_STAMP(3);
    // g = r_next - A'v on the own variables: the expressions of k_solve_pcr's tail; g replaces r
    // on the span; a lane adds up the squares of its own entries (pairs tid, tid + RB, ...)
    double gacc = 0.0;
    {
      double v4[RQP][4], rn[RQP][2];
#pragma unroll
      for (int k = 0; k < RQP; ++k) {
        const int64_t j = vb + 2 * (int64_t)(tid + k * RB);
        const int s0 = (int)min(max(j - c_lo, (int64_t)0), (int64_t)nspan - 1);
        const int s1 = (int)min(max(j + 1 - c_lo, (int64_t)0), (int64_t)nspan - 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pk = res_opaque(ec2[k][u]);
          v4[k][2 * u] = sx[pk & 0xffff];
          v4[k][2 * u + 1] = sx[(pk >> 16) & 0xffff];
        }
        rn[k][0] = rspan[s0]; rn[k][1] = rspan[s1];
      }
#pragma unroll
      for (int k = 0; k < RQP; ++k) {
        const int64_t j = vb + 2 * (int64_t)(tid + k * RB);
        double y0 = -1.0 * (e0[k].x * v4[k][0] + e1[k].x * v4[k][1]);
        y0 += 1.0 * rn[k][0];
        double y1 = -1.0 * (e0[k].y * v4[k][2] + e1[k].y * v4[k][3]);
        y1 += 1.0 * rn[k][1];
        const bool in0 = j >= av0 && j < (int64_t)av1;
        const bool in1 = j + 1 >= av0 && j + 1 < (int64_t)av1;
        if (in0) rspan[j - c_lo] = y0;
        if (in1) rspan[j + 1 - c_lo] = y1;
        gacc += in0 ? y0 * y0 : 0.0;
        gacc += in1 ? y1 * y1 : 0.0;
      }
    }
    RS_STAMP(4);
    // the halo of g leaves as soon as g is complete (the scalars follow after the sums below:
    // the transfer overlaps them); hop 2's tag
    ipx_lds_barrier();
    halo_put<PEER>(J, wg, 2, rspan + own_off, pl, J.seq + hop + 1, tid);                 // to the left neighbour
    halo_put<PEER>(J, wg, 3, rspan + own_off + avn - pr, pr, J.seq + hop + 1, tid);      // to the right neighbour
    // residual of the own rows:  w_i - (a_i v_{i-1} + b_i v_i + a_{i+1} v_{i+1}), squared (a lane
    // per window row)
    double acc = 0.0;
    {
      const int r = min(max(tid, 1), R - 2);
      const double vm = sx[r - 1], vc = sx[r], vp = sx[r + 1];
      double res2 = 0.0;
      if (tid >= H && tid < H + J.rows_wg && g0 + tid < J.m) {
        double sum = b0[0] * vc;
        sum += a0[0] * vm;
        sum += a0n * vp;
        const double res = w0[0] - sum;
        res2 = res * res;
      }
      acc = res2;
    }
    // ||x + alpha p||^2, ||g||^2 and the residual: wave sums, waves in order
    double mine3[3];
    {
      double loc3[3] = {sxx, gacc, acc};
      res_block_sum<3>(loc3, RED_NEXT, mine3);
    }
    RS_STAMP(5);
    // ================= hop 2: partials + halo of g ===========================================
    ++hop;
    {
      const uint32_t tag = J.seq + hop;
      rec_put<3, PEER>(J, hop, gw, mine3, tag, tid);
      double sv[4];
      {
        double s3[3];
        const bool ok = hop_wait<3, PEER>(J, tag, res_records(J.ll, hop), J.gnwg, s3,
                                          halo_from_left(J, wg, 3), halo_from_right(J, wg, 2), nl, nr,
                                          rspan, rspan + own_off + avn, J.timeout, tid);
        sv[0] = s3[0]; sv[1] = s3[1]; sv[2] = s3[2]; sv[3] = ok ? 0.0 : 1.0;
      }
      RS_STAMP(6);
      RS_STAMP_SYNC(12);
      double tot[4];
      res_block_sum<4>(sv, RED_NEXT, tot);                // (barriers inside: the halo is in LDS)
      if (tot[3] != 0.0) {
        if (tid == 0) { J.st[ST_VIOL] = 82.0; J.st[ST_STOP] = (double)J.stop_code; }
        return;
      }
      part_xn2 = tot[0]; part_gg = tot[1]; part_tt = tot[2];
      have_proj = true;
    }
    if (!NOXN2) {
      if (sqrt(part_xn2) >= radius) { st_xn2 = part_xn2; stop = 2; break; }      // :583
    }
    const double gg = part_gg;
    if (orth_rhs > 0.0 && gg > 0.0 && sqrt(part_tt) > orth_rhs * sqrt(gg)) {     // projections.py:72
      st_orth = sqrt(part_tt) / sqrt(gg); stop = 6; break;
    }
    const double beta = gg / rtg;                             // :627
    rt[par ^ 1] = gg;                                         // :633
    st_beta = beta;
    done_inc += 1;
    RS_STAMP(7);
    // ================= phase H: x, p, Hp = H p ================================================
    {
      // x_next on the own variables, then p_next on own +- hmax (lane: entries tid + k RB of
      // pspan): every old value is read before the barrier, every new one written after it
      double po[RQX + 1], gv[RQX + 1];
      const int np_ = avn + 2 * J.hmax;
#pragma unroll
      for (int k0 = 0; k0 < RQX; k0 += 6) {
        double px[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) px[k] = pspan[J.hmax + min(tid + (k0 + k) * RB, avn - 1)];
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (tid + (k0 + k) * RB < avn) xo[k0 + k] = xo[k0 + k] + alpha * px[k];   // :580,630
      }
#pragma unroll
      for (int k = 0; k < RQX + 1; ++k) {
        const int j = min(tid + k * RB, np_ - 1);
        const int col = min(max(p_lo + j, 0), J.n - 1);
        po[k] = pspan[j];
        gv[k] = rspan[col - c_lo];
      }
      ipx_lds_barrier();
#pragma unroll
      for (int k = 0; k < RQX + 1; ++k) {
        const int j = tid + k * RB;
        const int col = p_lo + j;
        if (j < np_ && col >= 0 && col < J.n) pspan[j] = beta * po[k] - gv[k];     // :628
      }
    }
    ipx_lds_barrier();
    RS_STAMP(8);
    double acc_xy = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < RQX; k0 += 3) {
      double pp[3][RLH], xr[3], y[3];
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int k = k0 + kk;
#pragma unroll
        for (int t = 0; t < RLH; t += 2) {
          const int pk = res_opaque(hc2[k][t >> 1]);
          pp[kk][t] = pspan[pk & 0xffff];
          pp[kk][t + 1] = pspan[(pk >> 16) & 0xffff];
        }
        xr[kk] = pspan[J.hmax + min(tid + k * RB, avn - 1)];
      }
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int k = k0 + kk;
        double sum = 0.0;
#pragma unroll
        for (int t = 0; t < RLH; ++t) sum += hv[k][t] * pp[kk][t];       // (absent entries: + 0.0)
        y[kk] = 1.0 * sum;
        if (HAS_DIAG) y[kk] += dg[k] * xr[kk];
        if (tid + k * RB < avn) acc_xy += xr[kk] * y[kk];
      }
      // (hspan is written here and read nowhere in this phase)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk)
        if (tid + (k0 + kk) * RB < avn) hspan[own_off + tid + (k0 + kk) * RB] = y[kk];
    }
    if (it + 1 != J.it_end) {                         // (the halo of Hp: hop 1's tag; as above)
      ipx_lds_barrier();
      halo_put<PEER>(J, wg, 0, hspan + own_off, pl, J.seq + hop + 1, tid);
      halo_put<PEER>(J, wg, 1, hspan + own_off + avn - pr, pr, J.seq + hop + 1, tid);
    }
    double mine1[1], loc1[1] = {acc_xy};
    res_block_sum<1>(loc1, RED_NEXT, mine1);         // (barriers inside: Hp is complete on the own part)
    RS_STAMP(9);
    // ================= hop 1: p'Hp + halo of Hp (the last one of the launch: commit) =========
    ++hop;
    {
      const uint32_t tag = J.seq + hop;
      const bool last = it + 1 == J.it_end;
      rec_put<1, PEER>(J, hop, gw, mine1, tag, tid);
      double sv[2];
      {
        // (the launch's last hop is its commit: every workgroup has been seen running by then,
        // so a word that is late is late, not missing -- eight times the patience, which keeps
        // "some workgroups commit, one times out" out of reach: ADVICE r4)
        double s1[1];
        const bool ok = hop_wait<1, PEER>(J, tag, res_records(J.ll, hop), J.gnwg, s1,
                                          halo_from_left(J, wg, 1), halo_from_right(J, wg, 0),
                                          last ? 0 : nl, last ? 0 : nr, hspan, hspan + own_off + avn,
                                          last ? 8 * J.timeout : J.timeout, tid);
        sv[0] = s1[0]; sv[1] = ok ? 0.0 : 1.0;
      }
      RS_STAMP(10);
      RS_STAMP_SYNC(13);
      double tot[2];
      res_block_sum<2>(sv, RED_NEXT, tot);
      if (tot[1] != 0.0) {
        if (tid == 0) { J.st[ST_VIOL] = 81.0; J.st[ST_STOP] = (double)J.stop_code; }
        return;
      }
      ptHp = tot[0];
      RS_STAMP(11);
    }
  }
  if (stop != 0) {
    // commit hop of a stopped launch: every workgroup took the same branch at the same
    // iteration (the reduced scalars are the same bits everywhere)
    ++hop;
    const uint32_t tag = J.seq + hop;
    const double zero1[1] = {0.0};
    rec_put<1, PEER>(J, hop, gw, zero1, tag, tid);
    double sv[2], s1[1];
    const bool ok = hop_wait<1, PEER>(J, tag, res_records(J.ll, hop), J.gnwg, s1, J.ll, J.ll, 0, 0,
                                      hspan, hspan, hop > 1 ? 8 * J.timeout : J.timeout, tid);
    sv[0] = 0.0; sv[1] = ok ? 0.0 : 1.0;
    double tot[2];
    res_block_sum<2>(sv, RED_NEXT, tot);
    if (tot[1] != 0.0) {
      if (tid == 0) { J.st[ST_VIOL] = 83.0; J.st[ST_STOP] = (double)J.stop_code; }
      return;
    }
  }
  // ================= write-back (every workgroup has passed the launch's last hop) ===========
  ipx_lds_barrier();
#pragma unroll
  for (int k = 0; k < RQX; ++k) {
    const int i = tid + k * RB;
    if (i < avn) {
      J.x[av0 + i] = xo[k];
      J.p[av0 + i] = pspan[J.hmax + i];
      J.r[av0 + i] = rspan[own_off + i];
      J.Hp[av0 + i] = hspan[own_off + i];
    }
  }
  // partial arrays as the separate launches' consumers fold them: the total in entry 0, zeros
  // behind it (x + 0.0 = x: the next fold returns the same bits)
  // (PEER: the consumers sum a rank's own range and then over the ranks -- the total sits on
  // rank 0, zeros elsewhere; the sums of a stopped projection go to the host's pack, which the
  // resumed step2 reads)
  const int gtid = wg * RB + tid, gn = J.nwg * RB;
  for (int i = gtid + 1; i < J.p1_cnt; i += gn) J.part1[J.p1_off + i] = 0.0;
  if (!PEER && have_proj && (stop == 2 || stop == 6)) {
    for (int i = gtid + 1; i < 2 * J.np2; i += gn) J.part2[i] = 0.0;
    for (int i = gtid + 1; i < J.np3; i += gn) J.part3[i] = 0.0;
    for (int i = gtid + 1; i < J.np4; i += gn) J.part4[i] = 0.0;
  }
  if (lead) {
    J.part1[J.p1_off] = (!PEER || J.rank == 0) ? ptHp : 0.0;
    if (have_proj && (stop == 2 || stop == 6)) {
      if constexpr (PEER) {
        J.pack_out[0] = part_xn2; J.pack_out[1] = 0.0; J.pack_out[2] = part_gg; J.pack_out[3] = part_tt;
      } else {
        J.part2[0] = part_xn2; J.part3[0] = part_gg; J.part4[0] = part_tt;
      }
    }
    J.st[ST_RTG0] = rt[0]; J.st[ST_RTG1] = rt[1];
    J.st[ST_ALPHA] = st_alpha; J.st[ST_BETA] = st_beta; J.st[ST_PTHP] = st_pthp;
    J.st[ST_XNORM2] = st_xn2; J.st[ST_ORTH] = st_orth;
    J.st[ST_NITER] += (double)niter_inc;
    J.st[ST_IT_DONE] += (double)done_inc;
    if (stop != 0) J.st[ST_STOP] = (double)stop;
  }
}

size_t resident_lds_bytes(int nspan, int navn, int hmax, int rows_wg, int L, int rl) {
  const int H = 1 << L, R = rows_wg + 2 * H, RS = R + 2 * H;
  const int nspanP = (nspan + 1) & ~1;
  const int npsp = (navn + 2 * hmax + 1) & ~1;
  const int usize = 6 * RS;
  return sizeof(double) * (size_t)(2 * nspanP + usize + ((R + 1) & ~1) + npsp + ((R * rl + 1) & ~1) + 64);
}

}  // namespace

IPX_STAMP_EXPORT(ipx_debug_stamps_res, ipx_dbg_res)

// words of the hand-off buffer for `nwg` launched workgroups with halos of up to `hw` entries
// (two record arrays, a slot of six areas per workgroup + the two slots the neighbour ranks fill)
extern "C" int64_t ipx_cg_resident_ll_words(int32_t nwg, int32_t hw) {
  return (int64_t)R_HALO + ((int64_t)nwg + 2) * R_AREAS * hw * 2;
}

// the kernel's budgets, for the host code that builds its tables (ipsolver/cg_fused.py
// fuse_project): workgroups per launch, threads, span / own variables / window rows per
// workgroup, entries per row of A and of H, halo entries, workgroups of all ranks together
extern "C" void ipx_cg_resident_limits(int32_t *out8) {
  out8[0] = R_MAXLOCAL; out8[1] = RB; out8[2] = RQS * RB; out8[3] = RQX * RB; out8[4] = RNR * RB;
  out8[5] = RLA; out8[6] = RLH; out8[7] = RHK * RB / 2;
}
extern "C" int32_t ipx_cg_resident_max_global(void) { return R_MAXG; }

namespace {

// compute units of the current device (one workgroup per CU: every workgroup of a launch must be
// resident at once) and the dynamic-LDS attribute, once per device
int res_device_cus() {
  static int cus[64];
  static bool attr[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cus[dev] == 0) {
    hipDeviceProp_t p;
    cus[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
                   ? p.multiProcessorCount : -1;
  }
  if (!attr[dev]) {
    const int lim = 158 * 1024;
    hipError_t e = hipSuccess;
#define RES_ATTR(K) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, lim)
    RES_ATTR((k_cg_resident<false, false, false>)); RES_ATTR((k_cg_resident<false, true, false>));
    RES_ATTR((k_cg_resident<true, false, false>));  RES_ATTR((k_cg_resident<true, true, false>));
    RES_ATTR((k_cg_resident<false, false, true>));  RES_ATTR((k_cg_resident<false, true, true>));
    RES_ATTR((k_cg_resident<true, false, true>));   RES_ATTR((k_cg_resident<true, true, true>));
#undef RES_ATTR
    if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return 0; }
    attr[dev] = true;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}

// the tables of the argument block fit the kernel, for a launch of `nlaunch` of the solve's
// pv.nwg blocks
bool res_tables_ok(const ipx_cg_args *a, const ipx_pcr_view &pv, int nlaunch) {
  if (!a || a->solver_kind != 0 || a->lb || a->m <= 0 || !a->P_win || !a->A_off16 ||
      !a->A_rowfirst || a->A_rl < 1 || a->A_rl > RLA || !a->At_vown || !a->At_ell_row ||
      !a->At_ell_val || a->H_operator || !a->H_rowptr || a->H_hmax < 1 || a->H_hmax > 64 ||
      (a->n & 1) || a->n > (1 << 26) || a->m * a->A_rl > (1ll << 30))
    return false;
  const int H = 1 << pv.L, R = pv.rows_wg + 2 * H;
  const int navnE = ((int)a->P_navn + 2) & ~1;
  if (nlaunch < 1 || nlaunch > pv.nwg || nlaunch > R_MAXLOCAL || R > RNR * RB || H < 1 ||
      a->P_nspan > RQS * RB || a->P_nspan < 1 || a->P_navn < 1 || a->P_navn > RQX * RB ||
      navnE > 2 * RB * RQP || 2 * a->R_hw > RHK * RB || a->R_hw < 1 || a->H_hmax > a->R_hw ||
      a->P_navn + 2 * a->H_hmax > (RQX + 1) * RB)
    return false;
  if (resident_lds_bytes((int)a->P_nspan, (int)a->P_navn, (int)a->H_hmax, pv.rows_wg, pv.L, (int)a->A_rl) > 158 * 1024)
    return false;
  return nlaunch <= res_device_cus();
}

void res_job_common(ResJob &J, const ipx_cg_args *a, const ipx_pcr_view &pv, int32_t it_begin,
                    int32_t it_end) {
  J.st = a->state; J.it_begin = it_begin; J.it_end = it_end; J.n = (int)a->n; J.m = pv.m;
  J.rows_wg = pv.rows_wg; J.L = pv.L; J.band = pv.band;
  J.x = a->x; J.p = a->p; J.r = a->r; J.Hp = a->Hp;
  J.A_val = a->A_val; J.A_off16 = (const uint16_t *)a->A_off16; J.A_rowfirst = a->A_rowfirst;
  J.rl = (int)a->A_rl; J.win = a->P_win; J.vown = a->At_vown; J.nspan = (int)a->P_nspan;
  J.navn = (int)a->P_navn;
  J.ell_row = a->At_ell_row; J.ell_val = a->At_ell_val;
  J.H_rowptr = a->H_rowptr; J.H_colidx = a->H_colidx; J.H_val = a->H_val; J.H_diag = a->H_diag;
  J.hmax = (int)a->H_hmax;
  J.part1 = a->part1; J.part2 = a->part2; J.part3 = a->part3; J.part4 = a->part4;
  J.hw = (int)a->R_hw;
  J.no_xn2 = a->no_radius != 0;
}

template <bool PEER>
void res_launch(const ResJob &J, size_t lds, hipStream_t st) {
  const dim3 grid(J.nwg), block(RB);
  if (J.no_xn2) {
    if (J.H_diag) hipLaunchKernelGGL((k_cg_resident<true, true, PEER>), grid, block, lds, st, J);
    else hipLaunchKernelGGL((k_cg_resident<true, false, PEER>), grid, block, lds, st, J);
  } else {
    if (J.H_diag) hipLaunchKernelGGL((k_cg_resident<false, true, PEER>), grid, block, lds, st, J);
    else hipLaunchKernelGGL((k_cg_resident<false, false, PEER>), grid, block, lds, st, J);
  }
}

}  // namespace

// 1 when the argument block can run the resident form (sizes within the kernel's budgets, one
// compute unit per workgroup on this device)
extern "C" int ipx_cg_resident_ok(const ipx_cg_args *a) {
  if (!a || !a->resident || !a->R_ll || !a->R_seq) return 0;
  ipx_pcr_view pv;
  if (!ipx_banded_pcr_view(a->banded, &pv)) return 0;
  return res_tables_ok(a, pv, pv.nwg) ? 1 : 0;
}

// iterations [it_begin, it_end) in one resident launch (see the top of this file)
int ipx_cg_resident_launch(const ipx_cg_args *a, int32_t it_begin, int32_t it_end, int np1, int np2,
                           int np3, int np4, hipStream_t st) {
  if (!ipx_cg_resident_ok(a) || it_end <= it_begin) return IPX_EINVAL;
  ipx_pcr_view pv;
  ipx_banded_pcr_view(a->banded, &pv);
  ResJob J{};
  res_job_common(J, a, pv, it_begin, it_end);
  J.nwg = pv.nwg;
  J.p1_off = np1; J.p1_cnt = np1; J.np2 = np2; J.np3 = np3; J.np4 = np4;
  J.ll = (ull *)a->R_ll;
  J.timeout = 200000000LL;                           // 2 s of the 100 MHz wall clock
  J.stop_code = 8;
  J.wg0 = 0; J.gwg0 = 0; J.gnwg = pv.nwg; J.rank = 0; J.world = 1; J.pll = nullptr; J.pack_out = nullptr;
  // tags: strictly increasing over the life of the buffer (hop 0, 2 per iteration, the commit)
  const int64_t need = 2 * (int64_t)(it_end - it_begin) + 2;
  if (*a->R_seq + need >= 0xfffffff0LL) {
    if (hipMemsetAsync(a->R_ll, 0, (size_t)ipx_cg_resident_ll_words(pv.nwg, J.hw) * 8, st) != hipSuccess)
      return IPX_ELAUNCH;
    *a->R_seq = 0;
  }
  J.seq = (uint32_t)*a->R_seq;
  *a->R_seq += need;
  res_launch<false>(J, resident_lds_bytes(J.nspan, J.navn, J.hmax, J.rows_wg, J.L, J.rl), st);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// ---- PEER form: one rank's part of a batch of the row-sharded loop (ipsolver/sharded.py).
// e->res_*: this rank's own blocks [res_wg0, res_wg0 + res_nwg) of its local solve are the
// global workgroups res_gwg0 .. of res_gnwg; e->peer carries the hand-off buffers of all ranks
// (ipx_peer_attach_resident).  The GROUP decides to take this form (every rank's
// ipx_cg_shard2_resident_ok, minimum over the ranks): a rank on another form would leave the
// others waiting for its records.
extern "C" int ipx_cg_shard2_resident_ok(const ipx_cg_args *a, const ipx_shard2_ext *e) {
  if (!a || !e || !e->peer || e->nseg != 1 || e->res_nwg < 1) return 0;
  const ipx_peer *peer = (const ipx_peer *)e->peer;
  ipx_pcr_view pv;
  if (!ipx_banded_pcr_view(a->banded, &pv)) return 0;
  if (e->res_wg0 < 0 || e->res_wg0 + e->res_nwg > pv.nwg) return 0;
  // (a rank's edge workgroups read the tables of the block beyond: the halo blocks)
  if ((peer->view.rank > 0 && e->res_wg0 < 1) ||
      (peer->view.rank < peer->view.world - 1 && e->res_wg0 + e->res_nwg >= pv.nwg))
    return 0;
  return res_tables_ok(a, pv, (int)e->res_nwg) ? 1 : 0;
}

int ipx_cg_shard2_resident_launch(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t it_begin,
                                  int32_t it_end, int np1, hipStream_t stream) {
  if (!ipx_cg_shard2_resident_ok(a, e) || it_end <= it_begin || !e->pack) return IPX_EINVAL;
  ipx_peer *peer = (ipx_peer *)e->peer;
  const int world = peer->view.world;
  if (!peer->res_tab || !ipx_peer_ready(peer) || e->res_gnwg > R_MAXG || e->res_gnwg < e->res_nwg ||
      e->res_gwg0 < 0 || e->res_gwg0 + e->res_nwg > e->res_gnwg ||
      ipx_cg_resident_ll_words((int)e->res_nwg, (int)a->R_hw) > peer->res_words ||
      e->p1_lo[0] < 0 || e->p1_hi[0] > np1 || e->p1_hi[0] <= e->p1_lo[0])
    return IPX_EINVAL;
  const int64_t need = 2 * (int64_t)(it_end - it_begin) + 2;
  if ((int64_t)peer->rseq + need >= 0xfffffff0LL) return IPX_EINVAL;   // (2e9 iterations: the
                                                    // caller goes back to the separate launches)
  ipx_pcr_view pv;
  ipx_banded_pcr_view(a->banded, &pv);
  ResJob J{};
  res_job_common(J, a, pv, it_begin, it_end);
  J.nwg = (int)e->res_nwg;
  J.p1_off = np1 + (int)e->p1_lo[0]; J.p1_cnt = (int)(e->p1_hi[0] - e->p1_lo[0]);
  J.np2 = J.np3 = J.np4 = 0;
  J.ll = peer->res[peer->view.rank];
  J.timeout = peer->view.timeout_ticks;
  J.stop_code = 7;
  J.wg0 = (int)e->res_wg0; J.gwg0 = (int)e->res_gwg0; J.gnwg = (int)e->res_gnwg;
  J.rank = peer->view.rank; J.world = world;
  J.pll = peer->res_tab;
  J.pack_out = e->pack;
  J.seq = peer->rseq;
  peer->rseq += (uint32_t)need;
  ++peer->res_launches;
  res_launch<true>(J, resident_lds_bytes(J.nspan, J.navn, J.hmax, J.rows_wg, J.L, J.rl), stream);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}
