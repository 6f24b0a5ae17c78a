// Device-resident projected CG iteration (Steihaug-Toint; reference
// qp_subproblem.py:549-634) for gfx950.
//
// The reference loop has three scalar-gated branches per iteration
// (rt_g < tol :551, p'Hp <= 0 :558, ||x_next|| >= radius :583).  Reading those
// scalars on the host costs more than the HBM time of the whole iteration, so
// here they never leave the device: every kernel's prologue folds the
// per-workgroup partial sums left by its predecessor in a FIXED order (so all
// workgroups derive bit-identical scalars), takes the branch, and workgroup 0
// records it in a small state block.  A non-zero `stop` field turns every
// later kernel of the stream into a no-op; the host reads the state block
// only once per batch of iterations.
//
// One iteration, step by step (general sparsity: one launch each):
//
//   step1  alpha = rt_g / p'Hp;  r += alpha*Hp;  partials of ||x+alpha p||^2,
//          # of box violations of x + alpha p            (x itself untouched)
//   spmv   w = A r
//   banded v = (AA')^-1 w, with ||A g||^2 = ||w - (AA')v||^2 partials from the same
//          launch(es)                                    (orthogonality, projections.py:52)
//   spmv   r = r - A'v  (= g_next; the reference sets r = g, :632), ||g||^2 partials
//   step2  checks: ||x_next|| >= radius -> stop 2; box violated -> stop 5 (host
//          finishes the iteration); orthogonality > tol -> stop 6 (host refines);
//          else beta = ||g||^2/rt_g;  x += alpha p;  p = beta p - g
//   spmv   Hp = H p (+ diag*p), p'Hp partials
//
// For banded problems the steps pair up into THREE launches (same arithmetic):
//   k_cg_step1_ar      step1 inside the A.r SpMV        (banded A, no box)
//   k_solve_decoupled  the banded solve with g = r - A'v as its tail (banded.hip;
//                      tridiagonal A A', separator system numerically diagonal)
//   k_cg_step2_hp      step2 inside the H.p SpMV        (banded H)
// each falling back to the separate launches when its structural condition fails
// (checked symbolically by ipsolver/cg_fused.py).
//
// Because x is only advanced in step2, every early exit leaves (x, p, alpha)
// exactly as the reference's exit paths (:565-576, :585-596) need them.
#include "ipx_common.h"
#include <algorithm>
#include <vector>

// (state block layout: ipx_common.h ST_*)

namespace {

IPX_STAMP_DECL(ipx_dbg_cg);
#define CG_STAMP(k) IPX_STAMP_TO(ipx_dbg_cg, k)

constexpr int VB = IPX_BLOCK;

constexpr int VU = 8;        // elements per lane per trip, loads issued together (one trip at n = 1e6)

typedef ipx_own_ranges OwnRanges;      // (ipx_common.h)
// kernel argument of the PEER forms: the peer job; an empty struct for the others (the job is
// ~350 bytes of kernel arguments the single-GPU launches should not carry: measured +0.2 us on
// the dominant kernel)
struct ipx_no_peer {};
template <bool PEER> struct peer_arg { typedef ipx_no_peer type; };
template <> struct peer_arg<true> { typedef ipx_peer_job type; };
static OwnRanges own_all(int64_t n) {
  OwnRanges o;
  for (int k = 0; k < 4; ++k) { o.lo[k] = 0; o.hi[k] = 0; }
  o.hi[0] = n;
  return o;
}

__global__ void __launch_bounds__(VB)
k_cg_step1(int64_t n, double *st, int parity, const double *__restrict__ p1, int np1,
           const double *__restrict__ x, const double *__restrict__ p, double *r,
           const double *__restrict__ Hp, const double *__restrict__ lb,
           const double *__restrict__ ub, double *__restrict__ p2, int nchunks, OwnRanges own) {
  // own: the elements that count in ||x + alpha p||^2 and the box test (all of them on one
  // GPU; a rank's own entries in the row-sharded loop, whose halo copies are updated by the
  // same launch)
  __shared__ double lds[VB / IPX_WAVE];
  // chunk c of nchunks: a contiguous run of elements, owned by a fixed XCD
  const int c = ipx_xcd_item(blockIdx.x, nchunks);
  if (c < 0) return;
  const int64_t len = (n + nchunks - 1) / nchunks;
  const int64_t lo_i = (int64_t)c * len, hi_i = min(n, lo_i + len);
  int64_t i0 = lo_i + threadIdx.x;
  // Request order = arrival order (vmcnt counts in order): first the state
  // words and the partials of p'Hp (second half of the SpMV's partials: x.y
  // sums), then the first trip's operands.  The scalar fold below then runs
  // while the operand loads are still streaming in.
  const double *const fparts[1] = {p1 + np1};
  const int fcounts[1] = {np1};
  const double stop = st[ST_STOP];
  const double rtg = st[parity ? ST_RTG1 : ST_RTG0];
  const double tol = st[ST_TOL];
  ipx_fold_regs<1> fold;
  fold.load(fparts, fcounts);
  double xv[VU], pv[VU], rv[VU], hv[VU], lo[VU], hi[VU];
#pragma unroll
  for (int u = 0; u < VU; ++u) {
    const int64_t i = min(i0 + u * VB, n - 1);
    xv[u] = x[i]; pv[u] = p[i]; rv[u] = r[i]; hv[u] = Hp[i];
    if (lb) { lo[u] = lb[i]; hi[u] = ub ? ub[i] : HUGE_VAL; }
  }
  if (stop != 0.0) return;
  const bool lead = c == 0 && threadIdx.x == 0;
  double fout[1];
  fold.finish(fparts, fcounts, lds, fout);
  const double ptHp = fout[0];
  if (rtg < tol) {                                   // qp_subproblem.py:551
    if (lead) st[ST_STOP] = 4.0;
    return;
  }
  if (ptHp <= 0.0) {                                 // :558
    if (lead) { st[ST_NITER] += 1.0; st[ST_PTHP] = ptHp; st[ST_STOP] = 3.0; }
    return;
  }
  const double alpha = rtg / ptHp;                   // :579
  if (lead) { st[ST_NITER] += 1.0; st[ST_PTHP] = ptHp; st[ST_ALPHA] = alpha; }
  double sx = 0.0, viol = 0.0;
  while (true) {
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int64_t i = i0 + u * VB;
      if (i < hi_i) {
        const double xn = xv[u] + alpha * pv[u];     // :580 (not stored)
        if (own.has(i)) {
          sx += xn * xn;
          if (lb) viol += ((lo[u] <= xn) && (xn <= hi[u])) ? 0.0 : 1.0;   // :599
        }
        r[i] = rv[u] + alpha * hv[u];                // :622
      }
    }
    i0 += VU * VB;
    if (i0 >= hi_i) break;
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int64_t i = min(i0 + u * VB, n - 1);
      xv[u] = x[i]; pv[u] = p[i]; rv[u] = r[i]; hv[u] = Hp[i];
      if (lb) { lo[u] = lb[i]; hi[u] = ub ? ub[i] : HUGE_VAL; }
    }
  }
  const double a = ipx_block_reduce<IPX_SUM>(sx, lds);
  const double b = ipx_block_reduce<IPX_SUM>(viol, lds);
  if (threadIdx.x == 0) { p2[c] = a; p2[nchunks + c] = b; }
}

// step1 for the barrier problem's box-Schur projection (solver_kind 1 with group tables):
// the same updates and sums as k_cg_step1, but the elements are visited group by group
// (shared column + the rows' private columns; then the columns of no group), so that the
// first stage of the projection -- w_S from r_next, t = B^-1 w_S, up = r_next - alpha't
// (csrc/boxschur.hip k_pairs_pre) -- happens on the values just formed instead of in a
// launch of its own that reads r again.  `rounds` items per thread, SB_ITEMS or a few more
// (ipx_balanced_rounds: one partial per 2048+ items).
constexpr int SB_ITEMS = 8;
constexpr int SB_RMIN = 5;       // fewest rounds (part2 holds n / 1024 + 2 entries per half:
                                 // cg_fused.py; 3 rounds are as fast but triple the partials)
// items whose loads are in flight together in the computed-columns form (measured at config 5
// with 8 rounds: 1 -> 22.0, 2 -> 22.3, 4 -> 23.9, 8 -> 36 us: the kernel wants workgroups per CU,
// not loads per lane -- DESIGN.md section 6)
constexpr int SB_BATCH = 1;

// MODE: the form of the group tables (ipx_group_tab: full, compact coefficients, compact
// coefficients + computed columns; the same numbers bit for bit).
// PEER (row-sharded loop on the peer mailboxes): p1 holds zeros for the tiles of halo rows (the
// PEER form of k_cg_step2_hp and ipx_cg_shard2_fold_hp see to that), so its sum is the rank's
// own p'Hp; summed over the ranks here (ipx_peer_sum), unguarded like in k_cg_step1_ar.
template <bool PEER, int MODE>
__global__ void __launch_bounds__(IPX_BLOCK)
k_cg_step1_box(double *st, int parity, const double *__restrict__ p1, int np1,
               const double *__restrict__ x, const double *__restrict__ p, double *__restrict__ r,
               const double *__restrict__ Hp, const double *__restrict__ lb,
               const double *__restrict__ ub, double *__restrict__ p2, int nblk, int ng, int ngen,
               ipx_group_tab T, const int32_t *__restrict__ gen_cols, int ny,
               double *__restrict__ up, OwnRanges own, int rounds,
               typename peer_arg<PEER>::type pj) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  __shared__ double plds[PEER ? IPX_MAX_PEERS + 1 : 1];
  const int blk = ipx_xcd_item(blockIdx.x, nblk);
  if (blk < 0) return;
  const double *const fparts[1] = {p1 + np1};
  const int fcounts[1] = {np1};
  const double stop = st[ST_STOP];
  const double rtg = st[parity ? ST_RTG1 : ST_RTG0];
  const double tol = st[ST_TOL];
  ipx_fold_regs<1> fold;
  fold.load(fparts, fcounts);
  if (!PEER && stop != 0.0) return;
  if (PEER && stop == 7.0) return;                   // (a mailbox wait timed out: k_cg_step2_hp)
  const bool lead = blk == 0 && threadIdx.x == 0;
  double fout[1];
  fold.finish(fparts, fcounts, lds, fout);
  if constexpr (PEER) {
    double tot[1];
    if (!ipx_peer_sum<1>(pj.pv, pj.seq, 1, fout, blk == 0, plds, tot)) {
      if (threadIdx.x == 0) { st[ST_STOP] = 7.0; st[ST_VIOL] = 4.0 + 10.0 * blk; }
      return;
    }
    fout[0] = tot[0];
    if (stop != 0.0) return;
  }
  const double ptHp = fout[0];
  if (rtg < tol) {                                   // qp_subproblem.py:551
    if (lead) st[ST_STOP] = 4.0;
    return;
  }
  if (ptHp <= 0.0) {                                 // :558
    if (lead) { st[ST_NITER] += 1.0; st[ST_PTHP] = ptHp; st[ST_STOP] = 3.0; }
    return;
  }
  const double alpha = rtg / ptHp;                   // :579
  if (lead) { st[ST_NITER] += 1.0; st[ST_PTHP] = ptHp; st[ST_ALPHA] = alpha; }
  double sx = 0.0, viol = 0.0;
  const int nitems = ng + ngen;
  // one item: the updates of its (up to three) elements and the first stage of the projection
  // (e[t] < 0: no such element); the same expressions whichever way the operands were loaded
  auto item = [&](const int (&e)[3], bool grp, double ap, double sp, double aq, double sq,
                  const double (&xv)[3], const double (&pv)[3], const double (&rv)[3],
                  const double (&hv)[3], const double (&lo)[3], const double (&hi)[3]) {
    double rn[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const double xn = xv[t] + alpha * pv[t];                 // :580 (not stored)
      rn[t] = rv[t] + alpha * hv[t];                           // :622
      if (e[t] >= 0) {
        if (own.has(e[t])) {                                   // (sharded: own entries only)
          sx += xn * xn;
          if (lb) viol += ((lo[t] <= xn) && (xn <= hi[t])) ? 0.0 : 1.0;   // :599
        }
        r[e[t]] = rn[t];
      }
    }
    const int c = e[0];
    if (grp) {
      const int cp = e[1], cq = e[2];
      const double rc = rn[0], rp = cp >= 0 ? rn[1] : 0.0, rq = cq >= 0 ? rn[2] : 0.0;
      const double wp = cp < 0 ? ap * rc : (cp > c ? ap * rc + sp * rp : sp * rp + ap * rc);
      const double wq = cq < 0 ? aq * rc : (cq > c ? aq * rc + sq * rq : sq * rq + aq * rc);
      double i11, i12, i22, wgt, ut;
      ipx_group_inverse(cq != -2, ap, sp, aq, sq, i11, i12, i22, wgt);
      if (cq == -2) {
        ut = ap * (i11 * wp);
      } else {
        const double tp = i11 * wp + i12 * wq;
        const double tq = i12 * wp + i22 * wq;
        ut = ap * tp + aq * tq;
      }
      if (c < ny) up[c] = rc - ut;
    } else if (c < ny) {
      up[c] = rn[0];
    }
  };
  if constexpr (MODE == IPX_GROUPS_AFFINE) {
    // computed columns: nothing depends on a table, so all the loads of an item (SB_BATCH items)
    // are requested at once, indices clamped instead of branches -- with the table forms below
    // every item is a table -> gather round trip
    for (int k0 = 0; k0 < rounds; k0 += SB_BATCH) {
      int e[SB_BATCH][3];
      bool grp[SB_BATCH], valid[SB_BATCH];
      double ep[SB_BATCH], eq[SB_BATCH];
      double xv[SB_BATCH][3], pv[SB_BATCH][3], rv[SB_BATCH][3], hv[SB_BATCH][3];
      double lo[SB_BATCH][3], hi[SB_BATCH][3];
#pragma unroll
      for (int k = 0; k < SB_BATCH; ++k) {
        const int i = (blk * rounds + k0 + k) * IPX_BLOCK + threadIdx.x;
        const bool on = k0 + k < rounds;                      // (uniform)
        valid[k] = on && i < nitems;
        const int ic = min(i, nitems - 1);
        grp[k] = ic < ng;
        const int gi = min(ic, ng - 1);
        const int e0 = grp[k] ? T.c0 + ic : T.gen0 + (ic - ng);
        e[k][0] = e0; e[k][1] = grp[k] ? e0 + T.dp : e0; e[k][2] = grp[k] ? e0 + T.dq : e0;
        if (on) {
          ep[k] = T.grp2[2 * gi]; eq[k] = T.grp2[2 * gi + 1];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int j = e[k][t];
            xv[k][t] = x[j]; pv[k][t] = p[j]; rv[k][t] = r[j]; hv[k][t] = Hp[j];
          }
        } else {
          ep[k] = eq[k] = 0.0;
#pragma unroll
          for (int t = 0; t < 3; ++t) xv[k][t] = pv[k][t] = rv[k][t] = hv[k][t] = 0.0;
        }
      }
      if (lb) {
#pragma unroll
        for (int k = 0; k < SB_BATCH; ++k)
#pragma unroll
          for (int t = 0; t < 3; ++t) lo[k][t] = k0 + k < rounds ? lb[e[k][t]] : 0.0;
      } else {
#pragma unroll
        for (int k = 0; k < SB_BATCH; ++k)
#pragma unroll
          for (int t = 0; t < 3; ++t) lo[k][t] = -HUGE_VAL;
      }
      if (ub) {
#pragma unroll
        for (int k = 0; k < SB_BATCH; ++k)
#pragma unroll
          for (int t = 0; t < 3; ++t) hi[k][t] = k0 + k < rounds ? ub[e[k][t]] : 0.0;
      } else {
#pragma unroll
        for (int k = 0; k < SB_BATCH; ++k)
#pragma unroll
          for (int t = 0; t < 3; ++t) hi[k][t] = HUGE_VAL;
      }
#pragma unroll
      for (int k = 0; k < SB_BATCH; ++k) {
        if (!valid[k]) continue;
        const int el[3] = {e[k][0], grp[k] ? e[k][1] : -1, grp[k] ? e[k][2] : -1};
        item(el, grp[k], copysign(1.0, ep[k]), fabs(ep[k]), copysign(1.0, eq[k]), fabs(eq[k]),
             xv[k], pv[k], rv[k], hv[k], lo[k], hi[k]);
      }
    }
  } else {
#pragma unroll 2
    for (int k = 0; k < rounds; ++k) {
      const int i = (blk * rounds + k) * IPX_BLOCK + threadIdx.x;
      if (i >= nitems) continue;
      // every load of the item is requested before the first use: the tables, then the (up
      // to three) elements' x, p, r, Hp and bounds
      const bool grp = i < ng;
      int e[3];
      double ap = 0.0, sp = 0.0, aq = 0.0, sq = 0.0;
      if (grp) {
        ipx_group_cols<MODE>(T, i, e[0], e[1], e[2]);
        ipx_group_coeffs<MODE != IPX_GROUPS_FULL>(T, i, ap, sp, aq, sq);
      } else {
        e[0] = gen_cols[i - ng]; e[1] = -1; e[2] = -1;
      }
      double xv[3], pv[3], rv[3], hv[3], lo[3], hi[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int j = e[t] >= 0 ? e[t] : e[0];
        xv[t] = x[j]; pv[t] = p[j]; rv[t] = r[j]; hv[t] = Hp[j];
        lo[t] = lb ? lb[j] : -HUGE_VAL;
        hi[t] = ub ? ub[j] : HUGE_VAL;
      }
      item(e, grp, ap, sp, aq, sq, xv, pv, rv, hv, lo, hi);
    }
  }
  const double a = ipx_block_reduce<IPX_SUM>(sx, lds);
  const double b = ipx_block_reduce<IPX_SUM>(viol, lds);
  if (threadIdx.x == 0) { p2[blk] = a; p2[nblk + blk] = b; }
}

// mode bit0: skip the radius / box checks (host already handled them)
//      bit1: skip the orthogonality check (host already refined)
__global__ void __launch_bounds__(VB)
k_cg_step2(int64_t n, double *st, int parity, int mode, const double *__restrict__ p2, int np2,
           const double *__restrict__ p3, int np3, const double *__restrict__ p4, int np4,
           double *x, double *p, const double *__restrict__ g, int nchunks) {
  __shared__ double lds[4 * (VB / IPX_WAVE)];
  CG_STAMP(0);
  const int c = ipx_xcd_item(blockIdx.x, nchunks);   // same element -> XCD map as step1
  if (c < 0) return;
  const int64_t len = (n + nchunks - 1) / nchunks;
  const int64_t lo_i = (int64_t)c * len, hi_i = min(n, lo_i + len);
  int64_t i0 = lo_i + threadIdx.x;
  // Request order = arrival order: state words and the four partial arrays
  // (||x+ap||^2, #violations, ||g||^2, ||A g||^2) first, operands after, so the
  // fold runs while the operands stream in.
  const double *const parts[4] = {p2, p2 + np2, p3, p4};
  const int counts[4] = {(mode & 1) ? 0 : np2, (mode & 1) ? 0 : np2, np3, (mode & 2) ? 0 : np4};
  const double stop = st[ST_STOP];
  const double radius = st[ST_RADIUS], orth_rhs = st[ST_ORTH_RHS];
  const double rtg = st[parity ? ST_RTG1 : ST_RTG0];
  const double alpha = st[ST_ALPHA];
  ipx_fold_regs<4> fold;
  fold.load(parts, counts);
  double xv[VU], pv[VU], gv[VU];
#pragma unroll
  for (int u = 0; u < VU; ++u) {
    const int64_t i = min(i0 + u * VB, n - 1);
    xv[u] = x[i]; pv[u] = p[i]; gv[u] = g[i];
  }
  if (stop != 0.0) return;
  CG_STAMP(1);
  const bool lead = c == 0 && threadIdx.x == 0;
  double red[4];
  fold.finish(parts, counts, lds, red);
  CG_STAMP(2);
  if (!(mode & 1)) {
    const double xn2 = red[0], viol = red[1];
    if (sqrt(xn2) >= radius) {                       // :583
      if (lead) { st[ST_XNORM2] = xn2; st[ST_STOP] = 2.0; }
      return;
    }
    if (viol > 0.0) {                                // :599-616 continues on the host
      if (lead) { st[ST_VIOL] = viol; st[ST_STOP] = 5.0; }
      return;
    }
  }
  const double gg = red[2];                          // ||g_next||^2
  if (!(mode & 2)) {
    const double tt = red[3];                        // ||A g_next||^2
    const double rhs = orth_rhs;
    // orthogonality(A, g) > orth_tol  <=>  ||A g|| > orth_tol ||A||_F ||g||
    if (rhs > 0.0 && gg > 0.0 && sqrt(tt) > rhs * sqrt(gg)) {
      if (lead) { st[ST_ORTH] = sqrt(tt) / sqrt(gg); st[ST_STOP] = 6.0; }
      return;
    }
  }
  const double beta = gg / rtg;                      // :627
  if (lead) {
    st[parity ? ST_RTG0 : ST_RTG1] = gg;             // :633
    st[ST_BETA] = beta;
    st[ST_IT_DONE] += 1.0;
  }
  while (true) {
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int64_t i = i0 + u * VB;
      if (i < hi_i) {
        x[i] = xv[u] + alpha * pv[u];                // :580,630
        p[i] = beta * pv[u] - gv[u];                 // :628
      }
    }
    i0 += VU * VB;
    if (i0 >= hi_i) break;
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int64_t i = min(i0 + u * VB, n - 1);
      xv[u] = x[i]; pv[u] = p[i]; gv[u] = g[i];
    }
  }
  CG_STAMP(3);
}

// ---- step2 fused into the H.p SpMV (banded Hessians) ------------------------
// step2 rewrites p (and x) element by element and the SpMV right behind it reads
// p back; for a Hessian whose row tiles touch only a few columns outside
// their own row range (halo <= hmax on either side: every banded H) the two
// are one kernel: a workgroup forms p_next = beta p - g on its row range plus
// halo in LDS, stores its own part of p_next and x_next, and takes the SpMV's
// gathers out of LDS.  The halo entries of p are owned (and overwritten in
// place) by the neighbouring workgroups, so their OLD values come from `pb`:
// every tile's first/last hmax entries of p, saved by the kernel that produced
// p (double buffered by iteration parity).  Same expressions in the same
// order as k_cg_step2 + k_csr_spmv: bit-identical results.
// (halo <= 64 and <= the shortest tile: checked by the host binding, cg_fused.fuse_halo)
constexpr int FT_NNZ = IPX_SPMV_TILE_NNZ;

// C16: the column indices come as 16-bit offsets into the tile's span (col - c_lo, built once
// per pattern by the host binding) and the row pointers of a tile whose rows all have the
// same length are not read at all (rowlen[tile] >= 0): 2 + 4/rowlen bytes less per nonzero of
// the 12 a CSR entry costs -- 10 of the 96 MB this kernel moves at n = 1e6.
// PEER (row-sharded loop on the peer mailboxes, one segment): p2 / p3 / p4 are the rank's OWN
// ranges of the partial arrays; every workgroup folds them, workgroup 0 sends the four sums to
// the peers and every workgroup adds up the ranks' contributions in rank order (ipx_peer_sum)
// -- the all-reduce of qp_subproblem.py:583,599,626 without a launch of its own.  The halo of
// g travels in the same prologue: two early workgroups store the rank's first / last own
// entries into the neighbours' mailboxes, the tiles whose span reaches into the rank's halo
// take those entries from the mailbox (and write them into g, whose halo the next iteration's
// step1 reads).
template <bool HAS_DIAG, int Q, int QS, bool BOX, bool C16, bool PEER>
__global__ void __launch_bounds__(IPX_BLOCK)
k_cg_step2_hp(int n, double *st, int parity, int mode, const double *__restrict__ p2, int np2,
              const double *__restrict__ p3, int np3, const double *__restrict__ p4, int np4,
              double *x, double *p, const double *__restrict__ g,
              const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
              const double *__restrict__ val, const int32_t *__restrict__ tiles, int ntiles,
              const double *__restrict__ diag, double *__restrict__ Hp,
              double *__restrict__ partial, int hmax, const double *__restrict__ pb_in,
              double *__restrict__ pb_out,
              const uint16_t *__restrict__ col16, const int32_t *__restrict__ rowlen,
              typename peer_arg<PEER>::type pj, double *g_halo) {
  __shared__ double prod[FT_NNZ];
  __shared__ double span[QS * IPX_BLOCK];
  __shared__ int rp[Q * IPX_BLOCK + 1];
  __shared__ double lds[4 * (IPX_BLOCK / IPX_WAVE)];
  __shared__ double plds[PEER ? 4 * IPX_MAX_PEERS + 1 : 1];
  CG_STAMP(0);
  const int tile = ipx_xcd_item(blockIdx.x, ntiles);
  if (tile < 0) return;
  const int tid = threadIdx.x;
  // request order = arrival order: state words and partials first (the fold
  // runs while the operands stream in), then the tile's data
  // (register budget: the partial arrays get as many in-flight loads as their usual
  // lengths need -- step1's <= 512, the SpMV's ~n/1024, the solve's ~m/260 -- not 4 each)
  // BOX = false (no bounds): the violation counts are all zero and are not read; the
  // ||x + alpha p||^2 partials -- one per row tile of A when step1 is fused -- get the
  // freed register instead
  const double *const partsA[2] = {BOX ? p2 + np2 : p2, p4};
  const int countsA[2] = {(mode & 1) ? 0 : np2, (mode & 2) ? 0 : np4};
  const double *const partsB[1] = {p3};
  const int countsB[1] = {np3};
  const double *const partsC[1] = {BOX ? p2 : p2 + np2};     // BOX: xn2;  else unused (count 0)
  const int countsC[1] = {(BOX && !(mode & 1)) ? np2 : 0};
  const double stop = st[ST_STOP];
  const double radius = st[ST_RADIUS], orth_rhs = st[ST_ORTH_RHS];
  const double rtg = st[parity ? ST_RTG1 : ST_RTG0];
  const double alpha = st[ST_ALPHA];
  ipx_fold_regs<2, BOX ? 2 : 3> foldA;         // BOX: {viol, tt}      else {xn2, tt}
  ipx_fold_regs<1, 4> foldB;                   // gg
  ipx_fold_regs<1, BOX ? 2 : 1> foldC;         // BOX: xn2
  foldA.load(partsA, countsA);
  foldB.load(partsB, countsB);
  if (BOX) foldC.load(partsC, countsC);
  const int r0 = tiles[tile], r1 = tiles[tile + 1];
  const int s = tiles[ntiles + 1 + tile], e = tiles[ntiles + 2 + tile];
  const int nrows = r1 - r0;
  const int c_lo = max(r0 - hmax, 0), c_hi = min(r1 + hmax, n);
  const int nspan = c_hi - c_lo;
  // Q / QS: row / span elements per lane, sized by the launcher from the longest tile
  constexpr int U = FT_NNZ / IPX_BLOCK;
  int c[U];
  double v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int jj = min(s + tid + u * IPX_BLOCK, max(e - 1, 0));   // empty tile: any valid entry
    c[u] = C16 ? (int)col16[jj] : colidx[jj] - c_lo;
    v[u] = val[jj];
  }
  const int rl = C16 ? rowlen[tile] : -1;       // >= 0: every row of the tile has rl entries
  // span operands: element j of the span is column c_lo + j
  double sp[QS], sg[QS];
  const double *pbl = pb_in + (int64_t)(tile - 1) * 2 * hmax + hmax;   // right part of tile-1
  const double *pbr = pb_in + (int64_t)(tile + 1) * 2 * hmax;          // left part of tile+1
#pragma unroll
  for (int k = 0; k < QS; ++k) {
    const int j = min(tid + k * IPX_BLOCK, nspan - 1);
    const int col = c_lo + j;
    sg[k] = g[col];
    // one unconditional load through a selected address (a predicated load would
    // cost a branch and a full wait): left halo = last hmax entries of the previous
    // tile, right halo = first hmax entries of the next one, else p itself
    const double *src = col < r0 ? pbl + (col - (r0 - hmax)) : (col >= r1 ? pbr + (col - r1) : p + col);
    sp[k] = *src;
  }
  // (PEER: the collectives below run even when the loop has stopped -- like the pack kernels
  // they replace -- so that every rank's stores find their readers whatever a workgroup that
  // starts late reads in the stop word: the lead workgroup of THIS launch may have set it).
  // Except after a mailbox wait has timed out (code 7: a peer is gone or cannot run -- two ranks
  // sharing one GPU that is full of the other's spinning workgroups): every later launch of
  // the batch would wait out the 3 s again; the batch is void, the host gives up the transport
  if (!PEER && stop != 0.0) return;
  if (PEER && stop == 7.0) return;
  CG_STAMP(1);
  const bool lead = tile == 0 && tid == 0;
  double red[4], loc[4];
  {
    double la[2], lb1[1], lc[1] = {0.0};
    foldA.local(partsA, countsA, la);
    foldB.local(partsB, countsB, lb1);
    if (BOX) foldC.local(partsC, countsC, lc);
    loc[0] = BOX ? lc[0] : la[0];                  // xn2      (k_cg_step2's order:
    loc[1] = BOX ? la[0] : 0.0;                    // viol      xn2, viol, gg, tt)
    loc[2] = lb1[0];                               // gg
    loc[3] = la[1];                                // tt
  }
  ipx_block_sum_multi<4>(loc, lds, red);
  if constexpr (PEER) {
    const ipx_peer_view &pv = pj.pv;
    const int par = pj.hseq & 1;
    // push: the rank's first / last own entries of g go to the neighbours' halo areas, by
    // two of the FIRST workgroups dispatched (not by the tiles that own those entries: with
    // more workgroups than the chip holds at once the last tiles start only after the first
    // ones have finished, and the first ones wait for the left neighbour's last entries -- a
    // chain through all the ranks)
    if (tile == min(1, ntiles - 1) && pv.rank > 0)
      for (int k = 0; k < pj.nseg; ++k)
        for (int j = tid; j < pj.send_left[k]; j += IPX_BLOCK)
          ipx_ll_store(pv.mbox[pv.rank - 1] + ipx_peer_halo_word(pv.cap, 1, par, pj.push_l[k] + j),
                       g[pj.own_lo[k] + j], pj.hseq);
    if (tile == min(2, ntiles - 1) && pv.rank < pv.world - 1)
      for (int k = 0; k < pj.nseg; ++k)
        for (int j = tid; j < pj.send_right[k]; j += IPX_BLOCK)
          ipx_ll_store(pv.mbox[pv.rank + 1] + ipx_peer_halo_word(pv.cap, 0, par, pj.push_r[k] + j),
                       g[pj.own_hi[k] - pj.send_right[k] + j], pj.hseq);
    double tot[4];
    bool ok = ipx_peer_sum<4>(pv, pj.seq, 0, red, tile == 0, plds, tot);
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q] = tot[q];
    // (where the launch-per-collective form leaves them: the host's event handlers and the
    // resumed step2 read the reduced sums there)
    if (lead && pj.pack_out) {
#pragma unroll
      for (int q = 0; q < 4; ++q) pj.pack_out[q] = tot[q];
    }
    // pull: span entries inside the rank's halo come from the neighbours
    const long long deadline = (long long)wall_clock64() + pv.timeout_ticks;
    const unsigned long long *mine = pv.mbox[pv.rank];
    bool okh = true;
#pragma unroll
    for (int k = 0; k < QS; ++k) {
      const int col = c_lo + tid + k * IPX_BLOCK;
      if (tid + k * IPX_BLOCK < nspan) {
        const int64_t word = pj.pull_word(col, par);
        if (word >= 0) {
          double v = 0.0;
          okh = ipx_ll_load(mine + word, pj.hseq, v, deadline) && okh;
          sg[k] = v;
          if (col >= r0 && col < r1) g_halo[col] = v;
        }
      }
    }
    if (!okh) plds[4 * IPX_MAX_PEERS] = 1.0;         // (reset by ipx_peer_sum, read below)
    ipx_lds_barrier();
    const bool okh_all = plds[4 * IPX_MAX_PEERS] == 0.0;
    if (!ok || !okh_all) {                           // a peer died: stop code 7, every rank alike
      // (ST_VIOL: which wait -- 1 the sums, 2 the halo -- and on which tile)
      if (tid == 0) { st[ST_STOP] = 7.0; st[ST_VIOL] = (ok ? 2.0 : 1.0) + 10.0 * tile; }
      return;
    }
    if (stop != 0.0) return;
  }
  if (!(mode & 1)) {
    const double xn2 = red[0], viol = red[1];
    if (sqrt(xn2) >= radius) {                       // :583
      if (lead) { st[ST_XNORM2] = xn2; st[ST_STOP] = 2.0; }
      return;
    }
    if (viol > 0.0) {                                // :599-616 continues on the host
      if (lead) { st[ST_VIOL] = viol; st[ST_STOP] = 5.0; }
      return;
    }
  }
  const double gg = red[2];                          // ||g_next||^2
  if (!(mode & 2)) {
    const double tt = red[3];                        // ||A g_next||^2
    if (orth_rhs > 0.0 && gg > 0.0 && sqrt(tt) > orth_rhs * sqrt(gg)) {
      if (lead) { st[ST_ORTH] = sqrt(tt) / sqrt(gg); st[ST_STOP] = 6.0; }
      return;
    }
  }
  CG_STAMP(2);
  const double beta = gg / rtg;                      // :627
  if (lead) {
    st[parity ? ST_RTG0 : ST_RTG1] = gg;             // :633
    st[ST_BETA] = beta;
    st[ST_IT_DONE] += 1.0;
  }
  // p_next on the span (LDS), x_next / p_next / boundary copies on the own rows
  double *pbo = pb_out + (int64_t)tile * 2 * hmax;
#pragma unroll
  for (int k = 0; k < QS; ++k) {
    const int j = tid + k * IPX_BLOCK;
    if (j < nspan) {
      const int col = c_lo + j;
      const double pn = beta * sp[k] - sg[k];        // :628
      span[j] = pn;
      if (col >= r0 && col < r1) {
        p[col] = pn;
        if (col - r0 < hmax) pbo[col - r0] = pn;
        if (r1 - col <= hmax) pbo[hmax + (col - (r1 - hmax))] = pn;
      }
    }
  }
  // row pointers, x and the diagonal entries of this lane's rows: requested only now (their registers
  // are not held while the bulk loads are in flight), landed by the time they are needed
  double sx[QS];
#pragma unroll
  for (int k = 0; k < QS; ++k) sx[k] = x[min(max(c_lo + tid + k * IPX_BLOCK, r0), r1 - 1)];
  double dg[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q)
    dg[q] = HAS_DIAG ? diag[r0 + min(tid + q * IPX_BLOCK, nrows - 1)] : 0.0;
  int rpv[Q + 1];
  if (C16 && rl >= 0) {
#pragma unroll
    for (int q = 0; q <= Q; ++q) rpv[q] = min(tid + q * IPX_BLOCK, nrows) * rl;
  } else {
#pragma unroll
    for (int q = 0; q <= Q; ++q) rpv[q] = rowptr[r0 + min(tid + q * IPX_BLOCK, nrows)] - s;
  }
  CG_STAMP(3);
  ipx_lds_barrier();
  CG_STAMP(4);
  // SpMV phase 1: products with the gathers served from the span
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int jj = s + tid + u * IPX_BLOCK;
    if (jj < e) prod[jj - s] = v[u] * span[c[u]];
  }
#pragma unroll
  for (int q = 0; q <= Q; ++q) {
    const int i = tid + q * IPX_BLOCK;
    if (i <= nrows) rp[i] = rpv[q];
  }
  ipx_lds_barrier();
  CG_STAMP(5);
  // phase 2: row sums, diagonal term, partials of y'y and p'y
  double acc_yy = 0.0, acc_xy = 0.0, yq[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = tid + q * IPX_BLOCK;
    yq[q] = 0.0;
    if (i < nrows) {
      const int a = rp[i], b = rp[i + 1];
      double sum = 0.0;
      for (int k = a; k < b; ++k) sum += prod[k];
      const double xr = span[r0 - c_lo + i];
      double y = 1.0 * sum;
      if (HAS_DIAG) y += dg[q] * xr;
      yq[q] = y;
      acc_yy += y * y;
      acc_xy += xr * y;
    }
  }
  CG_STAMP(6);
  {
    double red2[2] = {acc_yy, acc_xy}, out2[2];
    ipx_block_sum_multi<2>(red2, lds, out2);
    // (PEER: the tiles of halo rows leave zeros, so that the consumer may fold the whole array
    // for the rank's own sum -- the barrier problem's own rows are up to four ranges)
    bool counts = true;
    if constexpr (PEER) counts = pj.owns_rows(r0, r1);
    if (tid == 0) {
      partial[tile] = counts ? out2[0] : 0.0;
      partial[ntiles + tile] = counts ? out2[1] : 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = tid + q * IPX_BLOCK;
    if (i < nrows) Hp[r0 + i] = yq[q];
  }
  // x_next = x + alpha p on the own rows (x was requested before the SpMV phases and has
  // landed)
#pragma unroll
  for (int k = 0; k < QS; ++k) {
    const int col = c_lo + tid + k * IPX_BLOCK;
    if (col >= r0 && col < r1) x[col] = sx[k] + alpha * sp[k];       // :580,630
  }
  CG_STAMP(7);
}

// ---- step1 fused into the A.r SpMV (banded Jacobians, no box) ----------------
// step1 rewrites r element by element and the SpMV right behind it reads r
// back.  When the row tiles of A sweep the columns monotonically (every banded
// A) the columns are dealt to the tiles -- tile t owns [own[t], own[t+1]) -- and
// a workgroup forms r_next = r + alpha Hp on its own columns plus the few
// columns beyond them that its rows still touch, in LDS, stores its own part
// and takes the SpMV's gathers out of LDS.  r_next goes to a second buffer
// (the columns beyond the own range belong to the next tile, which must still
// see the old r); the r - A'v SpMV that follows reads it and writes g back into
// r, so outside the iteration nothing changes.  Same expressions in the same
// order as k_cg_step1 + k_csr_spmv; the ||x + alpha p||^2 partials are per tile
// instead of per vector chunk (same values up to the order of summation).
// NOXN2: trust radius +inf -- ||x + alpha p||^2 can only feed a comparison that is always false
// (:583) and is not formed: x and p are not read at all (16 of the 56 MB this kernel moves at
// n = 1e6).
// C16: column indices as 16-bit offsets into the tile's span (col - own[tile]; the host
// binding builds them once per pattern), 2 bytes instead of 4 per nonzero.
// PEER: p1 is the rank's own range of the p'Hp partials; summed over the ranks in the prologue
// (ipx_peer_sum: workgroup 0 sends, every workgroup adds the contributions in rank order).
template <int QS, int TN, bool NOXN2, bool C16, bool PEER>
__global__ void __launch_bounds__(IPX_BLOCK)
k_cg_step1_ar(int n, double *st, int parity, const double *__restrict__ p1, int np1,
              const double *__restrict__ x, const double *__restrict__ p,
              const double *__restrict__ r, double *__restrict__ r_next,
              const double *__restrict__ Hp, const int32_t *__restrict__ rowptr,
              const int32_t *__restrict__ colidx, const double *__restrict__ val,
              const int32_t *__restrict__ tiles, int ntiles, const int32_t *__restrict__ own,
              double *__restrict__ w, double *__restrict__ part2,
              const uint16_t *__restrict__ col16, typename peer_arg<PEER>::type pj) {
  __shared__ double prod[TN];
  __shared__ double span[QS * IPX_BLOCK];
  __shared__ int rp[IPX_SPMV_TILE_ROWS + 1];
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  __shared__ double plds[PEER ? IPX_MAX_PEERS + 1 : 1];
  CG_STAMP(8);
  const int tile = ipx_xcd_item(blockIdx.x, ntiles);
  if (tile < 0) return;
  const int tid = threadIdx.x;
  const double *const fparts[1] = {p1 + np1};
  const int fcounts[1] = {np1};
  const double stop = st[ST_STOP];
  const double rtg = st[parity ? ST_RTG1 : ST_RTG0];
  const double tol = st[ST_TOL];
  ipx_fold_regs<1, 6> fold;                      // p'Hp partials: one per row tile of H
  fold.load(fparts, fcounts);
  const int r0 = tiles[tile], r1 = tiles[tile + 1];
  const int s = tiles[ntiles + 1 + tile], e = tiles[ntiles + 2 + tile];
  const int nrows = r1 - r0;
  const int o0 = own[tile], o1 = own[tile + 1];          // own columns
  const int c_hi = own[ntiles + 1 + tile];               // one past the last column touched (>= o1)
  const int nspan = c_hi - o0;
  constexpr int U = TN / IPX_BLOCK;
  int c[U];
  double v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int jj = min(s + tid + u * IPX_BLOCK, max(e - 1, 0));   // empty tile: any valid entry
    c[u] = C16 ? (int)col16[jj] : colidx[jj] - o0;
    v[u] = val[jj];
  }
  double sr[QS], sh[QS], sxv[QS], spv[QS];
#pragma unroll
  for (int k = 0; k < QS; ++k) {
    // (a tile without nonzeros owns nothing: nspan = 0, the clamps keep its loads in range)
    const int col = min(o0 + max(min(tid + k * IPX_BLOCK, nspan - 1), 0), n - 1);
    sr[k] = r[col];
    sh[k] = Hp[col];
    sxv[k] = NOXN2 ? 0.0 : x[col];
    spv[k] = NOXN2 ? 0.0 : p[col];
  }
  if (!PEER && stop != 0.0) return;                  // (PEER: see k_cg_step2_hp)
  if (PEER && stop == 7.0) return;
  CG_STAMP(9);
  const bool lead = tile == 0 && tid == 0;
  double fout[1];
  fold.finish(fparts, fcounts, lds, fout);
  if constexpr (PEER) {
    double tot[1];
    if (!ipx_peer_sum<1>(pj.pv, pj.seq, 1, fout, tile == 0, plds, tot)) {
      if (tid == 0) { st[ST_STOP] = 7.0; st[ST_VIOL] = 3.0 + 10.0 * tile; }
      return;
    }
    fout[0] = tot[0];
    if (stop != 0.0) return;
  }
  const double ptHp = fout[0];
  if (rtg < tol) {                                   // qp_subproblem.py:551
    if (lead) st[ST_STOP] = 4.0;
    return;
  }
  if (ptHp <= 0.0) {                                 // :558
    if (lead) { st[ST_NITER] += 1.0; st[ST_PTHP] = ptHp; st[ST_STOP] = 3.0; }
    return;
  }
  CG_STAMP(10);
  const double alpha = rtg / ptHp;                   // :579
  if (lead) { st[ST_NITER] += 1.0; st[ST_PTHP] = ptHp; st[ST_ALPHA] = alpha; }
  double sx = 0.0;
#pragma unroll
  for (int k = 0; k < QS; ++k) {
    const int j = tid + k * IPX_BLOCK;
    if (j < nspan) {
      const double rn = sr[k] + alpha * sh[k];       // :622
      span[j] = rn;
      if (o0 + j < o1) {                             // own column
        if (!NOXN2) {
          const double xn = sxv[k] + alpha * spv[k]; // :580 (not stored)
          sx += xn * xn;
        }
        r_next[o0 + j] = rn;
      }
    }
  }
  int rpv[(IPX_SPMV_TILE_ROWS / IPX_BLOCK) + 1];
#pragma unroll
  for (int q = 0; q <= IPX_SPMV_TILE_ROWS / IPX_BLOCK; ++q)
    rpv[q] = rowptr[r0 + min(tid + q * IPX_BLOCK, nrows)] - s;
  CG_STAMP(11);
  ipx_lds_barrier();
  CG_STAMP(12);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int jj = s + tid + u * IPX_BLOCK;
    if (jj < e) prod[jj - s] = v[u] * span[c[u]];
  }
#pragma unroll
  for (int q = 0; q <= IPX_SPMV_TILE_ROWS / IPX_BLOCK; ++q) {
    const int i = tid + q * IPX_BLOCK;
    if (i <= nrows) rp[i] = rpv[q];
  }
  ipx_lds_barrier();
  double yq[IPX_SPMV_TILE_ROWS / IPX_BLOCK];
#pragma unroll
  for (int q = 0; q < IPX_SPMV_TILE_ROWS / IPX_BLOCK; ++q) {
    const int i = tid + q * IPX_BLOCK;
    yq[q] = 0.0;
    if (i < nrows) {
      const int a = rp[i], b = rp[i + 1];
      double sum = 0.0;
      for (int k = a; k < b; ++k) sum += prod[k];
      yq[q] = 1.0 * sum;
    }
  }
  CG_STAMP(13);
  if (!NOXN2) {
    const double tot = ipx_block_reduce<IPX_SUM>(sx, lds);
    if (tid == 0) { part2[tile] = tot; part2[ntiles + tile] = 0.0; }
  }
#pragma unroll
  for (int q = 0; q < IPX_SPMV_TILE_ROWS / IPX_BLOCK; ++q) {
    const int i = tid + q * IPX_BLOCK;
    if (i < nrows) w[r0 + i] = yq[q];
  }
  CG_STAMP(14);
}

// Large problems: the per-tile partial arrays grow with n while every consumer
// workgroup folds all of them, so beyond ~1000 entries they are first compacted
// to one entry per 1024 (fixed order: deterministic).  Up to three arrays per
// launch; array k takes blocks [first[k], first[k+1]): its half h, chunk c is
// block first[k] + h*G[k] + c.
struct CompactJob {
  const double *in[3];
  double *out[3];
  int count[3], halves[3], G[3], first[4];
};
__global__ void __launch_bounds__(IPX_BLOCK)
k_compact_partials(CompactJob job, const double *__restrict__ guard) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  if (guard && *guard != 0.0) return;
  int k = 0;
  while (k < 2 && (int)blockIdx.x >= job.first[k + 1]) ++k;
  const int b = blockIdx.x - job.first[k];
  const int half = b / job.G[k], chunk = b - half * job.G[k];
  const int lo = chunk * 1024, len = min(1024, job.count[k] - lo);
  const double v = ipx_sum_partials<IPX_SUM>(job.in[k] + (int64_t)half * job.count[k] + lo, len, lds);
  if (threadIdx.x == 0) job.out[k][half * job.G[k] + chunk] = v;
}

// First / last hmax entries of p for every row tile of H (both parities): what
// k_cg_step2_hp reads as its halo.  Launched by the unfused producers of p.
__global__ void __launch_bounds__(IPX_BLOCK)
k_cg_save_pb(const double *__restrict__ p, const int32_t *__restrict__ tiles, int ntiles,
             int hmax, double *__restrict__ pb) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ntiles * 2 * hmax) return;
  const int t = idx / (2 * hmax), j = idx - t * 2 * hmax;
  const int r0 = tiles[t], r1 = tiles[t + 1];
  const double v = j < hmax ? p[r0 + j] : p[r1 - hmax + (j - hmax)];
  pb[idx] = v;
  pb[(int64_t)ntiles * 2 * hmax + idx] = v;
}

// Partitioned row-sharded loop (ipsolver/sharded.py): up to four sub-ranges of partial
// arrays -- the entries produced by a rank's OWN tiles / workgroups -- summed in a fixed
// order into out[0..4): the rank's contribution to an all-reduce.
struct RangeJob {
  const double *ptr[4][4];   // [piece][quantity]; readable even where the count is 0
  int count[4][4];
  int npieces;
  int active;                // bit q: out[q] is written
};

__global__ void __launch_bounds__(256)
k_cg_range_pack(RangeJob job, double *__restrict__ out, const double *__restrict__ guard) {
  __shared__ double lds[4 * 4];
  if (guard && *guard != 0.0) return;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int pc = 0; pc < job.npieces; ++pc) {          // one piece per segment, in order
    double red[4];
    ipx_sum_partials_multi<4>(job.ptr[pc], job.count[pc], lds, red);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += red[q];
  }
  if (threadIdx.x < 4 && ((job.active >> threadIdx.x) & 1)) out[threadIdx.x] = acc[threadIdx.x];
}

// The same own-range sums, all-reduced over the ranks INSIDE the kernel (peer mailboxes,
// csrc/peer.hip): workgroup 0 adds up the rank's partials, stores the sums as LL words into
// every rank's mailbox (its own included), waits for the W contributions to its own mailbox
// and adds them in rank order -- every rank derives the same bits -- into out[q].  The other
// workgroups (when a halo plan is given) move the halo of g the same way: the first / last own
// entries of every segment go to the neighbours' mailboxes, the neighbours' arrive in this
// rank's and are copied into the halo entries of g.  Everything a later kernel of this rank
// reads (out, g) is written by this kernel to local memory: no fences, no flags beyond the
// sequence numbers inside the LL words.  A wait that times out (a peer died) raises stop code
// 7 in the state block.
struct HaloPlan {
  int nseg;                                   // 0: no halo exchange in this launch
  int seg_lo[4], own_lo[4], own_hi[4], seg_hi[4], send_left[4], send_right[4];
};

__global__ void __launch_bounds__(256)
k_cg_pack_comm(RangeJob job, double *__restrict__ out, ipx_peer_view pv, uint32_t seq,
               HaloPlan hp, double *__restrict__ g, uint32_t hseq, double *__restrict__ st) {
  __shared__ double lds[4 * 4];
  __shared__ double vals[IPX_MAX_PEERS * 4];
  const int tid = threadIdx.x;
  if (st[ST_STOP] == 7.0) return;        // an earlier wait timed out: do not wait it out again
  const long long deadline = (long long)wall_clock64() + pv.timeout_ticks;
  if (blockIdx.x == 0) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int pc = 0; pc < job.npieces; ++pc) {          // one piece per segment, in order
      double red[4];
      ipx_sum_partials_multi<4>(job.ptr[pc], job.count[pc], lds, red);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += red[q];
    }
    const int slot = seq & (IPX_PEER_SLOTS - 1);
    if (tid < 4 * pv.world) {
      const int r = tid >> 2, q = tid & 3;
      if ((job.active >> q) & 1) {
        const double mine = q == 0 ? acc[0] : (q == 1 ? acc[1] : (q == 2 ? acc[2] : acc[3]));
        ipx_ll_store(pv.mbox[r] + ipx_peer_scal_word(slot, pv.rank, q), mine, seq);
      }
    }
    bool ok = true;
    if (tid < 4 * pv.world) {
      const int r = tid >> 2, q = tid & 3;
      double v = 0.0;
      if ((job.active >> q) & 1)
        ok = ipx_ll_load(pv.mbox[pv.rank] + ipx_peer_scal_word(slot, r, q), seq, v, deadline);
      vals[tid] = v;
    }
    if (!ok) st[ST_STOP] = 7.0;
    __syncthreads();
    if (tid < 4 && ((job.active >> tid) & 1)) {
      double s = 0.0;
      for (int r = 0; r < pv.world; ++r) s += vals[4 * r + tid];        // rank order
      out[tid] = s;
    }
    return;
  }
  // ---- halo of g
  const int par = hseq & 1;
  const int nth = (gridDim.x - 1) * blockDim.x, me = (blockIdx.x - 1) * blockDim.x + tid;
  int64_t offL = 0, offR = 0;                   // cumulative counts towards the left / right neighbour
  for (int k = 0; k < hp.nseg; ++k) {
    if (pv.rank > 0)
      for (int j = me; j < hp.send_left[k]; j += nth)       // my first own entries: the left
        ipx_ll_store(pv.mbox[pv.rank - 1] + ipx_peer_halo_word(pv.cap, 1, par, offL + j),   // neighbour's right halo
                     g[hp.own_lo[k] + j], hseq);
    if (pv.rank < pv.world - 1)
      for (int j = me; j < hp.send_right[k]; j += nth)      // my last own entries: the right
        ipx_ll_store(pv.mbox[pv.rank + 1] + ipx_peer_halo_word(pv.cap, 0, par, offR + j),   // neighbour's left halo
                     g[hp.own_hi[k] - hp.send_right[k] + j], hseq);
    offL += hp.send_left[k];
    offR += hp.send_right[k];
  }
  bool ok = true;
  offL = offR = 0;
  const unsigned long long *mine = pv.mbox[pv.rank];
  for (int k = 0; k < hp.nseg; ++k) {
    const int nl = hp.own_lo[k] - hp.seg_lo[k], nr = hp.seg_hi[k] - hp.own_hi[k];
    if (pv.rank > 0)
      for (int j = me; j < nl; j += nth) {
        double v = 0.0;
        ok = ipx_ll_load(mine + ipx_peer_halo_word(pv.cap, 0, par, offL + j), hseq, v, deadline) && ok;
        g[hp.seg_lo[k] + j] = v;
      }
    if (pv.rank < pv.world - 1)
      for (int j = me; j < nr; j += nth) {
        double v = 0.0;
        ok = ipx_ll_load(mine + ipx_peer_halo_word(pv.cap, 1, par, offR + j), hseq, v, deadline) && ok;
        g[hp.own_hi[k] + j] = v;
      }
    offL += nl;
    offR += nr;
  }
  if (!ok) st[ST_STOP] = 7.0;
}

static int step1_box_rounds(const ipx_boxschur_args *b) {
  return ipx_balanced_rounds(b->ng + b->ngen, SB_RMIN, 2 * SB_ITEMS);
}
template <bool PEER>
static void launch_step1_box(const ipx_cg_args *a, const ipx_boxschur_args *b, int it,
                             const double *p1, int np1, int nblk, const OwnRanges &own,
                             typename peer_arg<PEER>::type pj, hipStream_t st) {
  const ipx_group_tab T = ipx_boxschur_tab(b);
#define IPX_S1BOX(M)                                                                            \
  hipLaunchKernelGGL((k_cg_step1_box<PEER, M>), dim3(ipx_xcd_grid(nblk)), dim3(IPX_BLOCK), 0,  \
                     st, a->state, it & 1, p1, np1, a->x, a->p, a->r, a->Hp, a->lb, a->ub,     \
                     a->part2, nblk, (int)b->ng, (int)b->ngen, T, b->gen_cols, (int)b->ny,     \
                     b->up, own, step1_box_rounds(b), pj)
  switch (ipx_group_mode(T)) {
    case IPX_GROUPS_AFFINE: IPX_S1BOX(IPX_GROUPS_AFFINE); break;
    case IPX_GROUPS_UNIT: IPX_S1BOX(IPX_GROUPS_UNIT); break;
    default: IPX_S1BOX(IPX_GROUPS_FULL);
  }
#undef IPX_S1BOX
}
}  // namespace

IPX_STAMP_EXPORT(ipx_debug_stamps_cg, ipx_dbg_cg)

extern "C" {

int ipx_cg_state_size(void) { return ST_SIZE; }
// 2 workgroups per CU: measured best for step2 at n = 1e6 (9.3 us vs 10.8 us with 1024)
int ipx_cg_vec_grid(int64_t n) { return ipx_grid_for(n, VB * 2, 512); }

static bool fused_hp(const ipx_cg_args *a) {
  return a->pb != nullptr && a->H_hmax > 0 && !a->H_operator;
}

// Unfused H.p; when the fused step2+H.p kernel is in use it also saves the tile
// boundaries of the p it was given (that kernel's halo source).
static bool dense_loop(const ipx_cg_args *a) { return a->solver_kind == 2; }
// entries per half of part1: row tiles of a CSR Hessian / workgroups of the dense matvec
static int part1_count(const ipx_cg_args *a) {
  if (a->H_operator) return 1;                   // [unused, p'Hp] written by the caller
  if (dense_loop(a) && !a->H_rowptr) {
    const int g = (int)((a->n + 3) / 4);
    return g > 2048 ? 2048 : g;
  }
  return (int)a->H_ntiles;
}

static int launch_hp(const ipx_cg_args *a, const double *guard, hipStream_t st) {
  if (a->H_operator) return IPX_OK;              // the caller applies H between the iterations
  if (dense_loop(a) && !a->H_rowptr) {           // dense Hessian (row major n x n in H_val)
    int np = 0;
    return ipx_dense_gemv_launch((int)a->n, (int)a->n, a->H_val, a->n, a->p, 1.0, a->H_diag, 0.0,
                                 nullptr, a->Hp, a->part1, &np, guard, st);
  }
  ipx_csr_view H{(int)a->n, (int)a->n, a->H_rowptr, a->H_colidx, a->H_val, a->H_tiles, (int)a->H_ntiles};
  int rc = ipx_spmv_launch(H, a->p, 1.0, a->H_diag, 0.0, nullptr, a->Hp, a->part1, guard, st);
  if (rc || !fused_hp(a)) return rc;
  const int tot = (int)(a->H_ntiles * 2 * a->H_hmax);
  hipLaunchKernelGGL(k_cg_save_pb, dim3((tot + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0, st,
                     a->p, a->H_tiles, (int)a->H_ntiles, (int)a->H_hmax, a->pb);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// Host side of k_compact_partials: queue arrays, launch once, hand back the
// (pointer, entries per half) the consumer should fold.
struct Compactor {
  const ipx_cg_args *a;
  CompactJob job;
  int n = 0, used = 0;
  explicit Compactor(const ipx_cg_args *args) : a(args) { job.first[0] = 0; }
  // returns true (and rewrites ptr / count) when the array will be compacted
  bool add(const double *&ptr, int &count, int halves, int limit) {
    if (!a->fold_ws || count <= limit || n >= 3) return false;
    const int G = (count + 1023) / 1024;
    if (used + halves * G > IPX_FOLD_WS_DOUBLES) return false;
    job.in[n] = ptr; job.count[n] = count; job.halves[n] = halves; job.G[n] = G;
    job.out[n] = a->fold_ws + used;
    job.first[n + 1] = job.first[n] + halves * G;
    ptr = a->fold_ws + used;
    count = G;
    used += halves * G;
    ++n;
    return true;
  }
  int launch(const double *guard, hipStream_t st) {
    if (n == 0) return IPX_OK;
    for (int k = n; k < 3; ++k) job.first[k + 1] = job.first[n];
    hipLaunchKernelGGL(k_compact_partials, dim3(job.first[n]), dim3(IPX_BLOCK), 0, st, job, guard);
    IPX_CHECK_LAUNCH();
    n = 0;                       // `used` keeps growing: later arrays get fresh scratch
    job.first[0] = 0;
    return IPX_OK;
  }
};

static bool fused_ar(const ipx_cg_args *a);
static bool fused_hp(const ipx_cg_args *a);

// solver_kind 1 with the group tables present: the fused box-Schur projection
static bool box_project(const ipx_cg_args *a) {
  if (a->solver_kind != 1 || !a->banded) return false;
  const ipx_boxschur_args *b = (const ipx_boxschur_args *)a->banded;
  return b->gcol != nullptr && b->grp != nullptr && b->up != nullptr && !fused_ar(a);
}

static bool fused_ar(const ipx_cg_args *a) {
  return a->r_next != nullptr && a->A_own != nullptr && a->A_span > 0 && !a->lb && a->m > 0;
}
// entries per half of part2: one per row tile of A (fused step1) or per vector chunk
static int step1_box_blocks(const ipx_cg_args *a) {
  const ipx_boxschur_args *b = (const ipx_boxschur_args *)a->banded;
  const int64_t per = (int64_t)IPX_BLOCK * step1_box_rounds(b);
  return (int)((b->ng + b->ngen + per - 1) / per);
}
static int part4_count(const ipx_cg_args *a) {
  return a->m > 0 ? ipx_banded_resid_count(a->solver_kind == 1
                        ? ((const ipx_boxschur_args *)a->banded)->inner : a->banded) : 1;
}
static int part2_count(const ipx_cg_args *a) {
  if (box_project(a)) return step1_box_blocks(a);
  return fused_ar(a) ? (int)a->A_ntiles : (int)a->vec_grid;
}
// entries per half of part3: one per row tile of A', or per workgroup of the solve when
// g = r - A'v is its tail
static int part3_count(const ipx_cg_args *a) {
  if (box_project(a)) return ipx_boxschur_project_count((const ipx_boxschur_args *)a->banded);
  return (a->solver_kind == 0 && a->At_vown && a->At_qv > 0) ? part4_count(a) : (int)a->At_ntiles;
}

// step1 + A.r in one launch (see k_cg_step1_ar): r_next <- r + alpha Hp, w <- A r_next
// no_xn2: trust radius +inf and no box -- ||x + alpha p||^2 is not formed at all (it can only
// feed a comparison that is always false): x and p are not read, 16 of the kernel's 56 MB at
// n = 1e6
static int launch_step1_ar(const ipx_cg_args *a, int it, const double *p1, int np1,
                           hipStream_t st, bool no_xn2 = false,
                           const ipx_peer_job *peer = nullptr) {
  const dim3 grid(ipx_xcd_grid((int)a->A_ntiles)), block(IPX_BLOCK);
  const ipx_peer_job pj = peer ? *peer : ipx_peer_job{};
  const ipx_no_peer none{};
#define FUSED_ARGS                                                                         \
  (int)a->n, a->state, it & 1, p1, np1, a->x, a->p, a->r, a->r_next,                       \
      a->Hp, a->A_rowptr, a->A_colidx, a->A_val, a->A_tiles, (int)a->A_ntiles, a->A_own,   \
      a->w, a->part2, (const uint16_t *)a->A_col16
  const int qs = (int)((a->A_span + IPX_BLOCK - 1) / IPX_BLOCK);
  if (a->A_tile_nnz != 0) return IPX_EINVAL;    // (the 1024-nonzero tile form was removed)
  if (peer && !a->A_col16) return IPX_EINVAL;   // (see peer_fusable)
#define GO(Q)                                                                              \
  do {                                                                                     \
    if (peer && no_xn2) hipLaunchKernelGGL((k_cg_step1_ar<Q, FT_NNZ, true, true, true>), grid, block, 0, st, FUSED_ARGS, pj); \
    else if (peer) hipLaunchKernelGGL((k_cg_step1_ar<Q, FT_NNZ, false, true, true>), grid, block, 0, st, FUSED_ARGS, pj); \
    else if (no_xn2 && a->A_col16) hipLaunchKernelGGL((k_cg_step1_ar<Q, FT_NNZ, true, true, false>), grid, block, 0, st, FUSED_ARGS, none); \
    else if (no_xn2) hipLaunchKernelGGL((k_cg_step1_ar<Q, FT_NNZ, true, false, false>), grid, block, 0, st, FUSED_ARGS, none); \
    else if (a->A_col16) hipLaunchKernelGGL((k_cg_step1_ar<Q, FT_NNZ, false, true, false>), grid, block, 0, st, FUSED_ARGS, none); \
    else hipLaunchKernelGGL((k_cg_step1_ar<Q, FT_NNZ, false, false, false>), grid, block, 0, st, FUSED_ARGS, none);    \
  } while (0)
  switch (qs) {
    case 1: case 2: GO(2); break;
    case 3: case 4: GO(4); break;
    case 5: case 6: GO(6); break;
    case 7: case 8: GO(8); break;
    default: return IPX_EINVAL;
  }
#undef GO
#undef FUSED_ARGS
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// step2 + H.p in one launch (see k_cg_step2_hp); iteration `it` reads the halo
// copies of parity it & 1 and leaves those of p_next in the other one.
static int launch_step2_hp(const ipx_cg_args *a, int it, int mode, const double *p2, int np2,
                           const double *p3, int np3, const double *p4, int np4,
                           hipStream_t st, const double *g = nullptr,
                           const ipx_peer_job *peer = nullptr) {
  if (!g) g = a->r;
  const ipx_peer_job pj = peer ? *peer : ipx_peer_job{};
  const ipx_no_peer none{};
  double *g_halo = peer ? a->r : nullptr;
  const int64_t half = a->H_ntiles * 2 * a->H_hmax;
  const double *pb_in = a->pb + (it & 1) * half;
  double *pb_out = a->pb + ((it + 1) & 1) * half;
  const dim3 grid(ipx_xcd_grid((int)a->H_ntiles)), block(IPX_BLOCK);
#define FUSED_ARGS                                                                            \
  (int)a->n, a->state, it & 1, mode, p2, np2, p3, np3, p4, np4, a->x, a->p, g,                \
      a->H_rowptr, a->H_colidx, a->H_val, a->H_tiles,                                         \
      (int)a->H_ntiles, a->H_diag, a->Hp, a->part1, (int)a->H_hmax, pb_in, pb_out,           \
      (const uint16_t *)a->H_col16, a->H_rowlen
  // H_hmax carries the longest tile's row count in its upper half (set by the host
  // binding): short tiles (3 nonzeros per row -> 683 rows) take the 3-elements-per-lane
  // instantiation, which needs fewer registers
  const bool small = a->H_tile_rows > 0 && a->H_tile_rows + 2 * a->H_hmax <= 3 * IPX_BLOCK;
  const bool box = a->lb != nullptr;
  const bool c16 = a->H_col16 != nullptr && a->H_rowlen != nullptr;
  if (peer && !c16) return IPX_EINVAL;                        // (see peer_fusable)
#define GO(D, QQ, QSS)                                                                       \
  do {                                                                                       \
    if (peer && box) hipLaunchKernelGGL((k_cg_step2_hp<D, QQ, QSS, true, true, true>), grid, block, 0, st, FUSED_ARGS, pj, g_halo); \
    else if (peer) hipLaunchKernelGGL((k_cg_step2_hp<D, QQ, QSS, false, true, true>), grid, block, 0, st, FUSED_ARGS, pj, g_halo); \
    else if (box && c16) hipLaunchKernelGGL((k_cg_step2_hp<D, QQ, QSS, true, true, false>), grid, block, 0, st, FUSED_ARGS, none, g_halo); \
    else if (box) hipLaunchKernelGGL((k_cg_step2_hp<D, QQ, QSS, true, false, false>), grid, block, 0, st, FUSED_ARGS, none, g_halo); \
    else if (c16) hipLaunchKernelGGL((k_cg_step2_hp<D, QQ, QSS, false, true, false>), grid, block, 0, st, FUSED_ARGS, none, g_halo); \
    else hipLaunchKernelGGL((k_cg_step2_hp<D, QQ, QSS, false, false, false>), grid, block, 0, st, FUSED_ARGS, none, g_halo);    \
  } while (0)
  if (a->H_diag) {
    if (small) GO(true, 3, 3); else GO(true, 4, 5);
  } else {
    if (small) GO(false, 3, 3); else GO(false, 4, 5);
  }
#undef GO
#undef FUSED_ARGS
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// Single kernels of the loop with explicit partial buffers, for drivers that
// interleave their own steps.
int ipx_cg_step1(int64_t n, double *state, int32_t it, const double *p1, int32_t np1,
                 const double *x, const double *p, double *r, const double *Hp, const double *lb,
                 const double *ub, double *part2, int32_t grid, void *stream) {
  if (n < 0 || !state || !p1 || !part2 || grid < 1) return IPX_EINVAL;
  hipLaunchKernelGGL(k_cg_step1, dim3(ipx_xcd_grid(grid)), dim3(VB), 0, (hipStream_t)stream, n,
                     state, it & 1, p1, np1, x, p, r, Hp, lb, ub, part2, grid, own_all(n));
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

int ipx_cg_step2(int64_t n, double *state, int32_t it, int32_t mode, const double *p2,
                 int32_t np2, const double *p3, int32_t np3, const double *p4, int32_t np4,
                 double *x, double *p, const double *g, int32_t grid, void *stream) {
  if (n < 0 || !state || grid < 1) return IPX_EINVAL;
  hipLaunchKernelGGL(k_cg_step2, dim3(ipx_xcd_grid(grid)), dim3(VB), 0, (hipStream_t)stream, n,
                     state, it & 1, mode, p2, np2, p3, np3, p4, np4, x, p, g, grid);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// One local segment of an iteration of the PARTITIONED row-sharded loop (ipsolver/sharded.py
// FusedShardedCG).  The argument block describes the rank's extended local problem (own
// rows / variables plus halo copies): the kernels are those of the single-GPU loop; what is
// specific is (a) their reduction inputs are the all-reduced scalars (e->s1, e->pack, one
// entry each) instead of the predecessor's partial arrays, and (b) the partials they leave
// are summed over the rank's own tiles only.
//   phase 0 (after the all-reduce of s1): step1 (+ A.r), solve (+ g = r - A'v), own sums of
//           ||x+ap||^2, #violations, ||g||^2, ||A g||^2 -> e->pack[0..4)
//   phase 1 (after the all-reduce of pack and the halo exchange of g): step2 (+ H.p),
//           own sum of p'Hp -> e->s1[1]
static OwnRanges own_of(const ipx_shard2_ext *e) {
  OwnRanges o;
  for (int k = 0; k < 4; ++k) {
    o.lo[k] = k < e->nseg ? e->own_lo[k] : 0;
    o.hi[k] = k < e->nseg ? e->own_hi[k] : 0;
  }
  return o;
}

// The rank's own sums to `out`; with a peer mailbox (e->peer) all-reduced over the ranks in
// the same launch, and -- with_halo -- the halo of g (a->r) exchanged with the neighbours.
static int launch_pack(const ipx_cg_args *a, const ipx_shard2_ext *e, const RangeJob &job,
                       double *out, bool with_halo, hipStream_t st) {
  ipx_peer *peer = (ipx_peer *)e->peer;
  if (!peer) {
    hipLaunchKernelGGL(k_cg_range_pack, dim3(1), dim3(256), 0, st, job, out,
                       (const double *)nullptr);
    IPX_CHECK_LAUNCH();
    return IPX_OK;
  }
  if (peer->view.world > IPX_MAX_PEERS || !ipx_peer_ready(peer)) return IPX_EINVAL;
  HaloPlan hp;
  hp.nseg = 0;
  int64_t moved = 0;
  if (with_halo && peer->view.world > 1) {
    hp.nseg = (int)e->nseg;
    int64_t inl = 0, inr = 0, outl = 0, outr = 0;
    for (int k = 0; k < hp.nseg; ++k) {
      hp.seg_lo[k] = (int)e->seg_lo[k]; hp.seg_hi[k] = (int)e->seg_hi[k];
      hp.own_lo[k] = (int)e->own_lo[k]; hp.own_hi[k] = (int)e->own_hi[k];
      hp.send_left[k] = (int)e->send_left[k]; hp.send_right[k] = (int)e->send_right[k];
      if (hp.seg_lo[k] > hp.own_lo[k] || hp.own_hi[k] > hp.seg_hi[k] || hp.send_left[k] < 0 ||
          hp.send_right[k] < 0 || hp.send_left[k] + hp.send_right[k] > 2 * (hp.own_hi[k] - hp.own_lo[k]))
        return IPX_EINVAL;
      inl += hp.own_lo[k] - hp.seg_lo[k]; inr += hp.seg_hi[k] - hp.own_hi[k];
      outl += hp.send_left[k]; outr += hp.send_right[k];
    }
    // (the neighbours' capacities equal this rank's: one group, one ipx_peer_create argument)
    if (inl > peer->view.cap || inr > peer->view.cap || outl > peer->view.cap || outr > peer->view.cap)
      return IPX_EINVAL;
    moved = inl + inr + outl + outr;
    if (++peer->hseq == 0) ++peer->hseq;
  }
  if (++peer->seq == 0) ++peer->seq;
  const int halo_wg = hp.nseg ? (int)std::min<int64_t>(16, std::max<int64_t>(1, (moved + 511) / 512)) : 0;
  hipLaunchKernelGGL(k_cg_pack_comm, dim3(1 + halo_wg), dim3(256), 0, st, job, out, peer->view,
                     peer->seq, hp, a->r, peer->hseq, a->state);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// zeros over the entries of `part` outside the given ranges (the p'Hp partials of halo tiles
// after an unfused H.p: the PEER form of k_cg_step1_box folds the whole array)
__global__ void __launch_bounds__(IPX_BLOCK)
k_zero_outside(double *__restrict__ part, int n, OwnRanges keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !keep.has(i)) part[i] = 0.0;
}

// own sum of the p'Hp partials (second half of part1) over the segments' own row tiles
static int pack_hp(const ipx_cg_args *a, const ipx_shard2_ext *e, hipStream_t st) {
  const int np1 = part1_count(a);
  RangeJob job;
  job.npieces = (int)e->nseg;
  job.active = 2;
  for (int pc = 0; pc < 4; ++pc)
    for (int q = 0; q < 4; ++q) { job.ptr[pc][q] = a->part1; job.count[pc][q] = 0; }
  for (int pc = 0; pc < (int)e->nseg; ++pc) {
    if (e->p1_hi[pc] > np1) return IPX_EINVAL;
    job.ptr[pc][1] = a->part1 + np1 + e->p1_lo[pc];
    job.count[pc][1] = (int)(e->p1_hi[pc] - e->p1_lo[pc]);
  }
  // NOT guarded by the stop flag: once an iteration has raised it, the remaining iterations
  // of the batch are no-ops but their all-reduces still run; re-packing the (unchanged)
  // own sums keeps the reduced values those of the iteration that stopped
  int rc = launch_pack(a, e, job, e->s1, false, st);
  if (rc || !e->peer || !e->fuse_comm || !box_project(a)) return rc;
  // the prologue form of k_cg_step1_box folds ALL of part1's second half: zero the entries
  // an unfused H.p left for the tiles of halo rows
  OwnRanges keep;
  for (int k = 0; k < 4; ++k) {
    keep.lo[k] = k < (int)e->nseg ? e->p1_lo[k] : 0;
    keep.hi[k] = k < (int)e->nseg ? e->p1_hi[k] : 0;
  }
  hipLaunchKernelGGL(k_zero_outside, dim3((np1 + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0, st,
                     a->part1 + np1, np1, keep);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// The collectives INSIDE the loop's own kernels (k_cg_step1_ar / k_cg_step2_hp, PEER forms): 3
// launches per iteration instead of 5.  One segment (x-space problems), both fused SpMV
// kernels in their 16-bit index forms, g = r - A'v as the solve's tail, no box.
static bool peer_fusable(const ipx_cg_args *a, const ipx_shard2_ext *e) {
  ipx_peer *peer = (ipx_peer *)e->peer;
  if (!(peer && e->fuse_comm && peer->view.world > 1 && peer->view.world <= IPX_MAX_PEERS &&
        ipx_peer_ready(peer) && fused_hp(a) && a->H_col16 && a->H_rowlen && a->m > 0))
    return false;
  // x-space problems: both fused SpMV kernels, g = r - A'v as the solve's tail, no box
  if (e->nseg == 1 && fused_ar(a) && !a->lb && a->A_col16 &&
      a->solver_kind == 0 && a->At_vown && a->At_qv > 0)
    return true;
  // the barrier problem's z-space: the projection without the box rows as matrix rows
  return box_project(a) && a->lb != nullptr;
}

// the segments' extents and their places in the mailboxes' halo areas
static int peer_job_geometry(const ipx_shard2_ext *e, const ipx_peer *peer, ipx_peer_job *pj) {
  pj->nseg = (int)e->nseg;
  int inl = 0, inr = 0, outl = 0, outr = 0;
  for (int k = 0; k < pj->nseg; ++k) {
    pj->seg_lo[k] = (int)e->seg_lo[k]; pj->own_lo[k] = (int)e->own_lo[k];
    pj->own_hi[k] = (int)e->own_hi[k]; pj->seg_hi[k] = (int)e->seg_hi[k];
    pj->send_left[k] = (int)e->send_left[k]; pj->send_right[k] = (int)e->send_right[k];
    if (pj->seg_lo[k] > pj->own_lo[k] || pj->own_hi[k] > pj->seg_hi[k] || pj->send_left[k] < 0 ||
        pj->send_right[k] < 0)
      return IPX_EINVAL;
    pj->off_l[k] = inl; pj->off_r[k] = inr; pj->push_l[k] = outl; pj->push_r[k] = outr;
    inl += pj->own_lo[k] - pj->seg_lo[k]; inr += pj->seg_hi[k] - pj->own_hi[k];
    outl += pj->send_left[k]; outr += pj->send_right[k];
  }
  const int64_t cap = peer->view.cap;
  return (inl > cap || inr > cap || outl > cap || outr > cap) ? IPX_EINVAL : IPX_OK;
}

static int shard2_segment(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t phase,
                          int32_t it, int32_t mode, void *stream, bool fuse_comm);

int ipx_cg_shard2_segment(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t phase,
                          int32_t it, int32_t mode, void *stream) {
  return shard2_segment(a, e, phase, it, mode, stream, false);
}

static int shard2_segment(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t phase,
                          int32_t it, int32_t mode, void *stream, bool fuse_comm) {
  if (!a || !e || phase < 0 || phase > 1 || !e->s1 || !e->pack || a->solver_kind > 1 ||
      e->nseg < 1 || e->nseg > 4)
    return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (fuse_comm) {
    ipx_peer *peer = (ipx_peer *)e->peer;
    ipx_peer_job pj{};
    pj.pv = peer->view;
    const bool boxp = box_project(a);
    const double *stopw = a->state + ST_STOP;
    if (phase == 0) {
      // step1 with the all-reduce of p'Hp in its prologue, then the projection; no pack launch
      const int np1 = part1_count(a);
      // (every argument check comes BEFORE the sequence numbers move: a rank that returns
      // IPX_EINVAL must not leave its counters out of step with the group's)
      if (!boxp && (e->p1_hi[0] > np1 || e->p1_hi[0] < e->p1_lo[0])) return IPX_EINVAL;
      if (++peer->seq == 0) ++peer->seq;
      pj.seq = peer->seq;
      ++peer->fused;
      if (boxp) {
        const ipx_boxschur_args *b = (const ipx_boxschur_args *)a->banded;
        const OwnRanges own = own_of(e);
        const int nblk = step1_box_blocks(a);
        launch_step1_box<true>(a, b, it, a->part1, np1, nblk, own, pj, st);
        IPX_CHECK_LAUNCH();
        int32_t n3 = 0, n4 = 0;
        return ipx_boxschur_project_from(b, a->r, a->r, a->part3, &n3, a->part4, &n4, stopw, 1, st,
                                         &own);
      }
      const int cnt = (int)(e->p1_hi[0] - e->p1_lo[0]);
      // (the kernel folds [p1 + np1, p1 + 2 np1): hand it the own range as that window)
      const double *p1 = a->part1 + np1 + e->p1_lo[0] - cnt;
      int rc = launch_step1_ar(a, it, p1, cnt, st, false, &pj);
      if (rc) return rc;
      int np4 = 0;
      return ipx_banded_solve_resid_atv_launch(a->banded, a->w, a->v, a->part4, &np4, a->At_rowptr,
                                               a->At_colidx, a->At_val, a->r_next, a->r,
                                               a->At_vown, (int)a->At_qv, a->part3, stopw, st,
                                               a->At_ell_row, a->At_ell_val, a->n);
    }
    // step2 + H.p with the all-reduce of the four sums and the halo exchange of g in its
    // prologue; the own sums of p'Hp stay in part1 for the next step1
    const int np4 = part4_count(a);
    if (e->p4_hi > np4) return IPX_EINVAL;
    if (!boxp && (e->p2_hi > (int)a->A_ntiles || e->p3_hi[0] > np4)) return IPX_EINVAL;
    int rc = peer_job_geometry(e, peer, &pj);
    if (rc) return rc;
    if (++peer->seq == 0) ++peer->seq;
    if (++peer->hseq == 0) ++peer->hseq;
    pj.seq = peer->seq;
    pj.hseq = peer->hseq;
    pj.pack_out = e->pack;
    ++peer->fused;
    if (boxp)       // the kernels counted own entries only: the own sums are the whole arrays
      return launch_step2_hp(a, it, mode, a->part2, step1_box_blocks(a), a->part3,
                             ipx_boxschur_project_count((const ipx_boxschur_args *)a->banded),
                             a->part4 + e->p4_lo, (int)(e->p4_hi - e->p4_lo), st, nullptr, &pj);
    return launch_step2_hp(a, it, mode, a->part2 + e->p2_lo, (int)(e->p2_hi - e->p2_lo),
                           a->part3 + e->p3_lo[0], (int)(e->p3_hi[0] - e->p3_lo[0]),
                           a->part4 + e->p4_lo, (int)(e->p4_hi - e->p4_lo), st, nullptr, &pj);
  }
  const double *guard = a->state + ST_STOP;
  const int grid = (int)a->vec_grid;
  int rc = IPX_OK;
  if (phase == 0) {
    const bool fuse1 = fused_ar(a);
    const bool boxp = box_project(a);
    int np4 = 0, np3 = (int)a->At_ntiles, np2 = fuse1 ? (int)a->A_ntiles : grid;
    if (boxp) {
      // barrier problem: the projection without the box rows as matrix rows (csrc/boxschur.hip);
      // the kernels count a rank's own entries only (element mask), so the own sums are the
      // sums over ALL their partials
      const ipx_boxschur_args *b = (const ipx_boxschur_args *)a->banded;
      const OwnRanges own = own_of(e);
      np2 = step1_box_blocks(a);
      launch_step1_box<false>(a, b, it, e->s1, 1, np2, own, ipx_no_peer{}, st);
      IPX_CHECK_LAUNCH();
      int32_t n3 = 0, n4 = 0;
      rc = ipx_boxschur_project_from(b, a->r, a->r, a->part3, &n3, a->part4, &n4, guard, 1, st, &own);
      if (rc) return rc;
      np3 = n3;
      np4 = n4;
    } else {
      if (fuse1) {
        rc = launch_step1_ar(a, it, e->s1, 1, st);
        if (rc) return rc;
      } else {
        hipLaunchKernelGGL(k_cg_step1, dim3(ipx_xcd_grid(grid)), dim3(VB), 0, st, a->n, a->state,
                           it & 1, e->s1, 1, a->x, a->p, a->r, a->Hp, a->lb, a->ub, a->part2, grid,
                           own_of(e));
        IPX_CHECK_LAUNCH();
        ipx_csr_view A{(int)a->m, (int)a->n, a->A_rowptr, a->A_colidx, a->A_val, a->A_tiles,
                       (int)a->A_ntiles};
        rc = ipx_spmv_launch(A, a->r, 1.0, nullptr, 0.0, nullptr, a->w, nullptr, guard, st);
        if (rc) return rc;
      }
      const double *r_in = fuse1 ? a->r_next : a->r;
      if (a->solver_kind == 0 && a->At_vown && a->At_qv > 0) {
        rc = ipx_banded_solve_resid_atv_launch(a->banded, a->w, a->v, a->part4, &np4,
                                               a->At_rowptr, a->At_colidx, a->At_val, r_in, a->r,
                                               a->At_vown, (int)a->At_qv, a->part3, guard, st,
                                               a->At_ell_row, a->At_ell_val, a->n);
        if (rc) return rc;
        np3 = np4;
      } else {
        if (a->solver_kind == 1)    // box rows eliminated analytically, banded Schur complement
          rc = ipx_boxschur_solve((const ipx_boxschur_args *)a->banded, a->w, a->v, a->part4,
                                  &np4, guard, st);
        else
          rc = ipx_banded_solve_resid_launch(a->banded, a->w, a->v, a->part4, &np4, guard, st);
        if (rc) return rc;
        ipx_csr_view At{(int)a->n, (int)a->m, a->At_rowptr, a->At_colidx, a->At_val, a->At_tiles,
                        (int)a->At_ntiles};
        rc = ipx_spmv_launch(At, a->v, -1.0, nullptr, 1.0, r_in, a->r, a->part3, guard, st);
        if (rc) return rc;
      }
    }
    if (e->p4_hi > np4 || (!boxp && e->p2_hi > np2)) return IPX_EINVAL;
    RangeJob job;
    job.npieces = (int)e->nseg;
    job.active = 15;
    for (int pc = 0; pc < 4; ++pc)
      for (int q = 0; q < 4; ++q) { job.ptr[pc][q] = a->part2; job.count[pc][q] = 0; }
    // ||x+ap||^2, #violations and ||A g||^2 are single ranges (piece 0); ||g||^2 one per segment
    const int64_t p2lo = boxp ? 0 : e->p2_lo, p2hi = boxp ? np2 : e->p2_hi;
    job.ptr[0][0] = a->part2 + p2lo;        job.count[0][0] = (int)(p2hi - p2lo);
    job.ptr[0][1] = a->part2 + np2 + p2lo;  job.count[0][1] = (int)(p2hi - p2lo);
    job.ptr[0][3] = a->part4 + e->p4_lo;    job.count[0][3] = (int)(e->p4_hi - e->p4_lo);
    if (boxp) {
      job.ptr[0][2] = a->part3;
      job.count[0][2] = np3;
    } else {
      for (int pc = 0; pc < (int)e->nseg; ++pc) {
        if (e->p3_hi[pc] > np3) return IPX_EINVAL;
        job.ptr[pc][2] = a->part3 + e->p3_lo[pc];
        job.count[pc][2] = (int)(e->p3_hi[pc] - e->p3_lo[pc]);
      }
    }
    return launch_pack(a, e, job, e->pack, true, st);       // unguarded: see pack_hp
  }
  // phase 1
  if (fused_hp(a)) {
    rc = launch_step2_hp(a, it, mode, e->pack, 1, e->pack + 2, 1, e->pack + 3, 1, st);
  } else {
    hipLaunchKernelGGL(k_cg_step2, dim3(ipx_xcd_grid(grid)), dim3(VB), 0, st, a->n, a->state,
                       it & 1, mode, e->pack, 1, e->pack + 2, 1, e->pack + 3, 1, a->x, a->p, a->r,
                       grid);
    IPX_CHECK_LAUNCH();
    rc = launch_hp(a, guard, st);
  }
  if (rc) return rc;
  return pack_hp(a, e, st);
}

// Iterations [it_begin, it_end) of the sharded loop in ONE call: needs the peer mailbox
// (e->peer), through which the pack kernels all-reduce the scalars and exchange the halo of g
// themselves -- no collective call, no host between the iterations.
int ipx_cg_shard2_iterate(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t it_begin,
                          int32_t it_end, void *stream) {
  if (!a || !e || !e->peer || it_end < it_begin) return IPX_EINVAL;
  // e->fuse_comm is the GROUP's decision (ipx_cg_shard2_fusable on every rank, the minimum
  // taken over the ranks by the caller): the two forms order the collectives of an iteration
  // differently, so a rank may not pick one from its own slice of the matrices
  const bool fuse = e->fuse_comm != 0;
  if (fuse && !peer_fusable(a, e)) return IPX_EINVAL;
  for (int it = it_begin; it < it_end; ++it) {
    int rc = shard2_segment(a, e, 0, it, 0, stream, fuse);
    if (rc) return rc;
    rc = shard2_segment(a, e, 1, it, 0, stream, fuse);
    if (rc) return rc;
  }
  return IPX_OK;
}

// The batch as one resident launch per rank (csrc/resident.hip, PEER form).
int ipx_cg_shard2_resident(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t it_begin,
                           int32_t it_end, void *stream) {
  if (!a || !e) return IPX_EINVAL;
  return ipx_cg_shard2_resident_launch(a, e, it_begin, it_end, part1_count(a), (hipStream_t)stream);
}

// p at the row-tile boundaries of H (the fused step2 + H.p kernel reads its neighbours' OLD p
// there) from the current a->p
int ipx_cg_save_pb(const ipx_cg_args *a, void *stream) {
  if (!a) return IPX_EINVAL;
  if (!fused_hp(a)) return IPX_OK;
  const int tot = (int)(a->H_ntiles * 2 * a->H_hmax);
  hipLaunchKernelGGL(k_cg_save_pb, dim3((tot + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0,
                     (hipStream_t)stream, a->p, a->H_tiles, (int)a->H_ntiles, (int)a->H_hmax, a->pb);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// 1 when THIS rank's argument block allows the collectives in the prologues of the loop's own
// kernels (3 launches per iteration; peer_fusable: the tables are functions of the rank's
// slice of A and H).  The ranks must agree before anyone sets e->fuse_comm: the caller
// reduces this value with MIN over the group (ipsolver/sharded.py FusedShardedCG).
int ipx_cg_shard2_fusable(const ipx_cg_args *a, const ipx_shard2_ext *e) {
  if (!a || !e || !e->peer) return 0;
  ipx_shard2_ext asked = *e;
  asked.fuse_comm = 1;
  return peer_fusable(a, &asked) ? 1 : 0;
}

// The own-range sum of the p'Hp partials on its own (priming the sharded loop).
int ipx_cg_shard2_fold_hp(const ipx_cg_args *a, const ipx_shard2_ext *e, void *stream) {
  if (!a || !e || !e->s1 || e->nseg < 1 || e->nseg > 4) return IPX_EINVAL;
  return pack_hp(a, e, (hipStream_t)stream);
}

// The state block of a new call written ON THE DEVICE from reductions its priming left in
// device memory, so that a projected_cg call needs no host read before its first batch
// (reference qp_subproblem.py:502-542 evaluates these scalars on the host: five blocking reads
// here, ~0.3 ms of a call's fixed cost).  red[idx[.]]:
//   0: ||x0||^2;   1, 2, 3: ||t||^2, ||r0||^2, ||A r0||^2 of r0 = Z t;
//   4, 5, 6: ||r0||^2, ||g0||^2, ||A g0||^2 of g0 = Z r0   (idx < 0: quantity is 0)
// Writes rt_g = ||g0||^2, tol (tol_in, or the default rule :529-530 when tol_in is NaN), the
// radius, the orthogonality threshold -- and stop code 9 when the host must take over: a
// projection that needs refinement (projections.py:72-78) or the cancellation step
// (projector.null_space), or no room to the trust-region boundary (:515-526).
// taken (may be NULL): taken[pj] != 0 when projection pj already had its one correction step on
// the device (ipx_cg_prime: k_prime_decide) -- like the host's loop after its first step
// (projector.null_space, k >= 1) only the orthogonality test then stands.
__device__ void prime_state_body(double *st, const double *red, const ipx_prime_idx &ix,
                                 double tol_in, double radius, double orth_tol, double norm_A,
                                 double canc2, const double *taken = nullptr) {
  double q[7];
  for (int k = 0; k < 7; ++k) q[k] = ix.i[k] >= 0 ? red[ix.i[k]] : 0.0;
  bool bad = false;
  double margin = HUGE_VAL;
  for (int pj = 0; pj < 2; ++pj) {
    const double nx2 = q[1 + 3 * pj], nz2 = q[2 + 3 * pj], naz2 = q[3 + 3 * pj];
    const double nz = sqrt(nz2), naz = sqrt(naz2);
    const double orth = (nz == 0.0 || norm_A == 0.0) ? 0.0 : naz / (norm_A * nz);
    const bool stepped = taken && taken[pj] != 0.0;
    if (orth > orth_tol || (!stepped && nz2 < canc2 * nx2)) bad = true;
    if (!stepped && nx2 > 0.0) margin = fmin(margin, nz2 / nx2);
  }
  const double rt_g = q[5];
  const double tr_distance = radius - sqrt(q[0]);
  if (!(tr_distance >= 1e-25)) bad = true;
  double tol = tol_in;
  if (tol_in != tol_in) tol = fmax(fmin(0.01 * sqrt(rt_g), 0.1 * rt_g), 1e-25);
  for (int k = 0; k < ST_SIZE; ++k) st[k] = 0.0;
  st[ST_RTG0] = rt_g;
  st[ST_TOL] = tol;
  st[ST_RADIUS] = radius;
  st[ST_ORTH_RHS] = orth_tol * norm_A;
  st[ST_MARGIN] = margin;
  st[ST_PRIME_STEPS] = taken ? taken[0] + taken[1] : 0.0;
  st[ST_STOP] = bad ? 9.0 : 0.0;
}

__global__ void k_cg_prime_state(double *st, const double *__restrict__ red, ipx_prime_idx ix,
                                 double tol_in, double radius, double orth_tol, double norm_A,
                                 double canc2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  prime_state_body(st, red, ix, tol_in, radius, orth_tol, norm_A, canc2);
}

// The same behind the folds of the priming's SpMV partials (ipx_cg_prime: up to six products
// leave per-tile partials in separate regions of the workspace; folded here by the fold
// kernel's own routine, in its order, instead of by six launches of it)
struct PrimeFolds {
  const double *part[6];
  int count[6], slot[6], n;
  int single[6];            // 1: ONE sum per entry (the fused solve's tail), not the SpMV's pair
};
// radius_dev / norm_A2_dev (optional): the trust radius and ||A||_F^2 as DEVICE scalars that
// kernels earlier in the stream have written -- the outer iteration's chain (csrc/sqp.hip), whose
// tangential radius sqrt(Delta^2 - ||dn||^2) never visits the host.
// The correction step of a projection on the device (ipx_cg_prime).  After z = x - A'(A A')^-1 A x
// and t = A z, k_prime_decide folds their partials into red[base], red[base + 2] and decides
// what the host's loop decides from the same numbers (projector.null_space: orthogonality above
// the tolerance, or z cancelled below 2^-10 |x|): red[PR_SKIP + pj] = 0 lets the three guarded
// launches behind it run -- v = (A A')^-1 t, z <- z - A'v, t = A z -- whose partials (cz, ct) the
// NEXT decide kernel (or the state kernel) folds over red[base], red[base + 2]; taken:
// red[PR_TAKEN + pj] = 1.  One step per projection; what it does not settle is the host's (9).
#define PR_SKIP 14
#define PR_TAKEN 16
struct PrimeStep {          // one projection's correction: where its partials wait
  const double *cz, *ct;    // partials of the corrected ||z||^2 / of ||A z||^2
  int ncz, nct, base, pj;
  int pairs;                // 1: SpMV epilogue arrays [sum y^2 | sum x y]; 0: one sum each (the
                            // fused solve's tail: ||z||^2 per workgroup, residual partials)
};
__device__ void prime_fold4(const double *pz, int npz, const double *pt, int npt, int pairs,
                            double *lds, double (&out)[4]) {
  const double *parts[4] = {pz, pairs ? pz + npz : pz, pt, pairs ? pt + npt : pt};
  const int counts[4] = {npz, pairs ? npz : 0, npt, pairs ? npt : 0};
  ipx_sum_partials_multi<4>(parts, counts, lds, out);
}
__device__ void prime_fold_step(double *red, const PrimeStep &c, double *lds) {
  // (uniform across the block: red[PR_SKIP + pj] was written by an earlier kernel)
  if (red[PR_SKIP + c.pj] != 0.0) return;
  double out[4];
  prime_fold4(c.cz, c.ncz, c.ct, c.nct, c.pairs, lds, out);
  if (threadIdx.x == 0) {
    red[c.base] = out[0]; red[c.base + 1] = out[1];
    red[c.base + 2] = out[2]; red[c.base + 3] = out[3];
    red[PR_TAKEN + c.pj] = 1.0;
  }
  __syncthreads();
}
__global__ void __launch_bounds__(IPX_BLOCK)
k_prime_decide(double *red, const double *pz, int npz, const double *pt, int npt, int pairs,
               int base, int xslot, int pj, double orth_tol, double norm_A, double canc2,
               const double *__restrict__ norm_A2_dev, PrimeStep prev, int have_prev) {
  __shared__ double lds[4 * (IPX_BLOCK / IPX_WAVE)];
  if (have_prev) prime_fold_step(red, prev, lds);       // (||x||^2 of this projection: red[xslot])
  if (norm_A2_dev) norm_A = sqrt(*norm_A2_dev);
  double out[4];
  prime_fold4(pz, npz, pt, npt, pairs, lds, out);
  if (threadIdx.x != 0) return;
  red[base] = out[0]; red[base + 1] = out[1];
  red[base + 2] = out[2]; red[base + 3] = out[3];
  const double nx2 = red[xslot], nz2 = out[0], naz2 = out[2];
  const double nz = sqrt(nz2), naz = sqrt(naz2);
  const double orth = (nz == 0.0 || norm_A == 0.0) ? 0.0 : naz / (norm_A * nz);
  const bool need = orth > orth_tol || nz2 < canc2 * nx2;
  red[PR_SKIP + pj] = need ? 0.0 : 1.0;
  red[PR_TAKEN + pj] = 0.0;
}

__global__ void __launch_bounds__(IPX_BLOCK)
k_cg_prime_state_folds(double *st, double *red, PrimeFolds f, ipx_prime_idx ix, double tol_in,
                       double radius, double orth_tol, double norm_A, double canc2,
                       const double *__restrict__ radius_dev,
                       const double *__restrict__ norm_A2_dev, PrimeStep last, int have_last) {
  if (radius_dev) radius = *radius_dev;
  if (norm_A2_dev) norm_A = sqrt(*norm_A2_dev);
  // all twelve sums (six jobs x [sum y^2 | sum x y]) in one pass: the loads of every array in
  // flight together, one barrier pair (one after the other: 12 us at n = 1e6; ~3 us this way);
  // the fold kernel's routine and order per array
  __shared__ double lds[12 * (IPX_BLOCK / IPX_WAVE)];
  const double *parts[12];
  int counts[12];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const bool on = j < f.n, two = on && !f.single[j];
    parts[2 * j] = on ? f.part[j] : f.part[0];
    parts[2 * j + 1] = two ? f.part[j] + f.count[j] : f.part[0];
    counts[2 * j] = on ? f.count[j] : 0;
    counts[2 * j + 1] = two ? f.count[j] : 0;
  }
  double out[12];
  if (f.n > 0) {
    ipx_sum_partials_multi<12>(parts, counts, lds, out);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j < f.n) { red[f.slot[j]] = out[2 * j]; red[f.slot[j] + 1] = out[2 * j + 1]; }
    }
    __syncthreads();
  }
  if (have_last) prime_fold_step(red, last, lds);
  // (thread 0 reads back what it wrote itself)
  if (threadIdx.x == 0)
    prime_state_body(st, red, ix, tol_in, radius, orth_tol, norm_A, canc2,
                     have_last ? red + PR_TAKEN : nullptr);
}

int ipx_cg_prime_state(double *state, const double *red, const int32_t *idx7, double tol_in,
                       double radius, double orth_tol, double norm_A, double cancellation,
                       void *stream) {
  if (!state || !red || !idx7) return IPX_EINVAL;
  ipx_prime_idx ix;
  for (int k = 0; k < 7; ++k) ix.i[k] = idx7[k];
  hipLaunchKernelGGL(k_cg_prime_state, dim3(1), dim3(64), 0, (hipStream_t)stream, state, red, ix,
                     tol_in, radius, orth_tol, norm_A, cancellation * cancellation);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// The whole priming of a projected_cg call (qp_subproblem.py:502-512) behind ONE entry point:
//   x0 = Y(-b) = A'(A A')^-1(-b)            (b == NULL: b = 0, x0 = 0)
//   t  = H x0 + c;   r0 = Z t;   g0 = Z r0  (Z v = v - A'(A A')^-1 A v, each with ||v||^2,
//                                            ||Z v||^2 and ||A Z v||^2 left in `red`)
//   the state block (ipx_cg_prime_state);   p = -g0;   Hp = H p
// into the loop's own buffers (x0 -> a->x, r0 -> a->r, p -> a->p; a->Hp, a->w, a->v, a->t as
// scratch): the kernels and their order are those the Python host enqueued one ctypes call at
// a time (ipsolver/projector.py null_space_enqueue), ~25 calls and three copies.  CSR A and H
// (+ optional diagonal), banded or box-Schur solver; red: >= 14 doubles of device memory, ws:
// the reduction workspace (IPX_WS_DOUBLES).
static int cg_iterate(const ipx_cg_args *a, int32_t it_begin, int32_t it_end, hipStream_t st,
                      hipEvent_t *ev);
static int prime_solve(const ipx_cg_args *a, const double *w, double *v, void *stream) {
  if (a->solver_kind == 1)
    return ipx_boxschur_solve((const ipx_boxschur_args *)a->banded, w, v, nullptr, nullptr, nullptr,
                              stream);
  return ipx_banded_solve(a->banded, w, v, stream);
}
// The priming's products with a norm: per-tile partials into the next free region of the
// workspace, folded later by k_cg_prime_state_folds.
struct PrimeRed {
  double *red, *ws;
  int64_t off;
  PrimeFolds f;
  int spmv(const ipx_csr_view &M, const double *x, double alpha, const double *diag, double beta,
           const double *yin, double *yout, int slot, hipStream_t st) {
    double *part = ws + off;
    off += 2 * (int64_t)M.ntiles;
    f.part[f.n] = part; f.count[f.n] = M.ntiles; f.slot[f.n] = slot; f.single[f.n] = 0; ++f.n;
    return ipx_spmv_launch(M, x, alpha, diag, beta, yin, yout, part, nullptr, st);
  }
  void single(const double *part, int count, int slot) {
    f.part[f.n] = part; f.count[f.n] = count; f.slot[f.n] = slot; f.single[f.n] = 1; ++f.n;
  }
};

// z = sign (x - A'(A A')^-1 A x) with ||z||^2 -> red[base], ||A z||^2 -> red[base + 2] and,
// unless the caller has it already (have_xnorm), ||x||^2 -> red[base + 4].  sign = -1 writes
// the negated projection by the same roundings (alpha, beta = 1, -1 instead of -1, 1: a
// difference and its negation round alike), the norms do not see the sign.
static int prime_project(const ipx_cg_args *a, const ipx_csr_view &A, const ipx_csr_view &At,
                         const double *x, double *z, PrimeRed &R, int base, bool have_xnorm,
                         double sign, hipStream_t st) {
  int rc = have_xnorm ? IPX_OK : ipx_norms(a->n, x, R.red + base + 4, R.ws + R.off, st);
  if (rc) return rc;
  rc = ipx_spmv_launch(A, x, 1.0, nullptr, 0.0, nullptr, a->w, nullptr, nullptr, st);
  if (rc) return rc;
  rc = prime_solve(a, a->w, a->v, st);
  if (rc) return rc;
  rc = R.spmv(At, a->v, -sign, nullptr, sign, x, z, base, st);
  if (rc) return rc;
  return R.spmv(A, z, 1.0, nullptr, 0.0, nullptr, a->t, base + 2, st);
}

// The same projection through the loop's own second launch (tridiagonal A A' on the cyclic-
// reduction / decoupled solve with g = r - A'v as its tail: csrc/banded.hip AtvJob): w = A x by
// the SpMV, then ONE launch for v = (A A')^-1 w, z = sign (x - A'v), the per-workgroup partials
// of ||z||^2 (pz) and of ||w - (A A') v||^2 = ||A z||^2 (pt: the orthogonality measure the
// loop's iterations use, projections.py:52 without a second product by A) -- two launches
// instead of four.  guard: *guard != 0 skips both (a correction step that is not due).
static bool prime_has_tail(const ipx_cg_args *a) {
  return a->solver_kind == 0 && a->At_vown && a->At_qv > 0;
}
static int prime_project_tail(const ipx_cg_args *a, const ipx_csr_view &A, const double *x,
                              double *z, double *pz, double *pt, int *np, double sign,
                              const double *guard, hipStream_t st) {
  int rc = ipx_spmv_launch(A, x, 1.0, nullptr, 0.0, nullptr, a->w, nullptr, guard, st);
  if (rc) return rc;
  return ipx_banded_solve_resid_atv_launch(a->banded, a->w, a->v, pt, np, a->At_rowptr,
                                           a->At_colidx, a->At_val, x, z, a->At_vown,
                                           (int)a->At_qv, pz, guard, st, a->At_ell_row,
                                           a->At_ell_val, a->n, sign);
}

// either way, the partials registered for the state kernel's fold
static int prime_project_any(const ipx_cg_args *a, const ipx_csr_view &A, const ipx_csr_view &At,
                             const double *x, double *z, PrimeRed &R, int base, bool have_xnorm,
                             double sign, hipStream_t st) {
  if (!prime_has_tail(a)) return prime_project(a, A, At, x, z, R, base, have_xnorm, sign, st);
  int rc = have_xnorm ? IPX_OK : ipx_norms(a->n, x, R.red + base + 4, R.ws + R.off, st);
  if (rc) return rc;
  const int cap = (int)((a->m + 255) / 256) + 8;
  double *pz = R.ws + R.off, *pt = pz + 2 * cap;
  R.off += 3 * (int64_t)cap;
  int np = 0;
  rc = prime_project_tail(a, A, x, z, pz, pt, &np, sign, nullptr, st);
  if (rc) return rc;
  R.single(pz, np, base);
  R.single(pt, np, base + 2);
  return IPX_OK;
}

// ... and with its correction step decided and taken on the device (k_prime_decide): the
// projection's own partials are folded by the decide kernel (they leave R.f), the step's wait in
// two more regions for the next decide / the state kernel (`step` out; `prev`: the projection
// before this one, whose step this decide kernel folds first).  z - A'v' with v' = (A A')^-1 A z
// is the same expression for z and for -z (every launch an odd function of its input, bit for
// bit), so the negated second projection needs no sign.
static int prime_project_stepped(const ipx_cg_args *a, const ipx_csr_view &A,
                                 const ipx_csr_view &At, const double *x, double *z, PrimeRed &R,
                                 int base, int xslot, int pj, bool have_xnorm, double sign,
                                 double orth_tol, double norm_A, const double *norm_A2_dev,
                                 double canc2, const PrimeStep *prev, PrimeStep *step,
                                 hipStream_t st) {
  const int n0 = R.f.n;
  int rc;
  const double *skip = R.red + PR_SKIP + pj;
  if (prime_has_tail(a)) {
    if (!have_xnorm) {
      rc = ipx_norms(a->n, x, R.red + base + 4, R.ws + R.off, st);
      if (rc) return rc;
    }
    const int cap = (int)((a->m + 255) / 256) + 8;      // (>= workgroups of the solve, twice)
    double *pz = R.ws + R.off, *pt = pz + 2 * cap, *cz = pt + cap, *ct = cz + 2 * cap;
    R.off += 6 * (int64_t)cap;
    int np = 0, npc = 0;
    rc = prime_project_tail(a, A, x, z, pz, pt, &np, sign, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_prime_decide, dim3(1), dim3(IPX_BLOCK), 0, st, R.red, pz, np, pt, np, 0,
                       base, xslot, pj, orth_tol, norm_A, canc2, norm_A2_dev,
                       prev ? *prev : PrimeStep{}, prev ? 1 : 0);
    IPX_CHECK_LAUNCH();
    // the step: z <- z - A'(A A')^-1 (A z), the same two launches on z itself (in place: the
    // tail reads and writes one variable per lane), guarded
    rc = prime_project_tail(a, A, z, z, cz, ct, &npc, 1.0, skip, st);
    if (rc) return rc;
    *step = PrimeStep{cz, ct, npc, npc, base, pj, 0};
    return IPX_OK;
  }
  rc = prime_project(a, A, At, x, z, R, base, have_xnorm, sign, st);
  if (rc) return rc;
  // (the two regions this projection just registered: folded here, not by the state kernel)
  const double *pz = R.f.part[n0], *pt = R.f.part[n0 + 1];
  const int npz = R.f.count[n0], npt = R.f.count[n0 + 1];
  R.f.n = n0;
  hipLaunchKernelGGL(k_prime_decide, dim3(1), dim3(IPX_BLOCK), 0, st, R.red, pz, npz, pt, npt, 1,
                     base, xslot, pj, orth_tol, norm_A, canc2, norm_A2_dev,
                     prev ? *prev : PrimeStep{}, prev ? 1 : 0);
  IPX_CHECK_LAUNCH();
  if (a->solver_kind == 1)
    rc = ipx_boxschur_solve((const ipx_boxschur_args *)a->banded, a->t, a->v, nullptr, nullptr, skip,
                            st);
  else
    rc = ipx_banded_solve_guarded(a->banded, a->t, a->v, skip, st);
  if (rc) return rc;
  double *cz = R.ws + R.off;
  R.off += 2 * (int64_t)At.ntiles;
  double *ct = R.ws + R.off;
  R.off += 2 * (int64_t)A.ntiles;
  rc = ipx_spmv_launch(At, a->v, -1.0, nullptr, 1.0, z, z, cz, skip, st);
  if (rc) return rc;
  rc = ipx_spmv_launch(A, z, 1.0, nullptr, 0.0, nullptr, a->t, ct, skip, st);
  if (rc) return rc;
  *step = PrimeStep{cz, ct, At.ntiles, A.ntiles, base, pj, 1};
  return IPX_OK;
}

int64_t ipx_cg_prime_ws_doubles(const ipx_cg_args *a, int32_t A_ntiles) {
  if (!a) return -1;
  // (three products by A', two by A, one by H; + the two correction steps' A' and A products)
  return 2 * (5 * a->At_ntiles + 4 * (int64_t)A_ntiles + a->H_ntiles) + 4096;
}

int ipx_cg_prime(const ipx_cg_args *a, const int32_t *A_tiles, int32_t A_ntiles, const double *c,
                 const double *b, double *red, double *ws, double tol_in, double radius,
                 double orth_tol, double norm_A, double cancellation, int32_t first_end,
                 int32_t steps, void *stream) {
  return ipx_cg_prime_dev(a, A_tiles, A_ntiles, c, b, red, ws, tol_in, radius, nullptr, orth_tol,
                          norm_A, nullptr, cancellation, first_end, steps, nullptr, 0, 0,
                          (hipStream_t)stream);
}

}  // extern "C"

int ipx_cg_prime_dev(const ipx_cg_args *a, const int32_t *A_tiles, int32_t A_ntiles,
                     const double *c, const double *b, double *red, double *ws, double tol_in,
                     double radius, const double *radius_dev, double orth_tol, double norm_A,
                     const double *norm_A2_dev, double cancellation, int32_t first_end,
                     int steps, const double *c_part, int32_t c_npart, int x_is_zero,
                     hipStream_t stream) {
  if (!a || !c || !red || !ws || !A_tiles || a->solver_kind > 1 || a->m <= 0 || a->H_operator ||
      !a->H_rowptr || !a->t || first_end < 0)
    return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const ipx_csr_view A{(int)a->m, (int)a->n, a->A_rowptr, a->A_colidx, a->A_val, A_tiles, A_ntiles};
  const ipx_csr_view At{(int)a->n, (int)a->m, a->At_rowptr, a->At_colidx, a->At_val, a->At_tiles,
                        (int)a->At_ntiles};
  const ipx_csr_view Hm{(int)a->n, (int)a->n, a->H_rowptr, a->H_colidx, a->H_val, a->H_tiles,
                        (int)a->H_ntiles};
  // the products' norms come out of their epilogues (no norm kernels); their per-tile partials
  // wait in the workspace for ONE fold in front of the state kernel: six regions (n = 1e6: 17 k
  // of the 64 k doubles) and a 4096-double tail for ipx_norms.  A problem whose regions do not
  // fit (n beyond ~5e6 on the benchmark's band) is primed launch by launch by the caller.
  if (ipx_cg_prime_ws_doubles(a, A_ntiles) > IPX_WS_DOUBLES) return IPX_EUNSUPPORTED;
  PrimeRed R{red, ws, 0, PrimeFolds{}};
  int rc;
  const double *t = c;
  if (b) {
    // x0 = A'(A A')^-1 (-b) = -A' (A A')^-1 b: the solve and the product are odd functions of
    // their input down to the last bit, so the sign is applied by the product's scalar
    rc = prime_solve(a, b, a->v, stream);
    if (rc) return rc;
    rc = R.spmv(At, a->v, -1.0, nullptr, 0.0, nullptr, a->x, 12, st);
    if (rc) return rc;
    rc = R.spmv(Hm, a->x, 1.0, a->H_diag, 1.0, c, a->Hp, 4, st);
    if (rc) return rc;
    t = a->Hp;
  } else if (!x_is_zero &&
             hipMemsetAsync(a->x, 0, (size_t)a->n * sizeof(double), st) != hipSuccess) {
    return IPX_ELAUNCH;
  }
  const double canc2 = cancellation * cancellation;
  PrimeStep s1{}, s2{};
  if (b || !steps) {
    // (b: ||t||^2 = red[4] comes out of the H x0 + c product's partials, folded by the state
    // kernel: the first projection's decision would need it earlier -- no step on this path)
    const bool c_known = !b && c_part && c_npart > 0;
    if (c_known) {                 // (||c||^2 -> red[4]: folded by the state kernel)
      PrimeFolds &f = R.f;
      f.part[f.n] = c_part; f.count[f.n] = c_npart; f.slot[f.n] = 4; f.single[f.n] = 0; ++f.n;
    }
    rc = prime_project_any(a, A, At, t, a->r, R, 0, b != nullptr || c_known, 1.0, st);
    if (rc) return rc;
    rc = prime_project_any(a, A, At, a->r, a->p, R, 6, true, -1.0, st);
    if (rc) return rc;
  } else {
    rc = prime_project_stepped(a, A, At, t, a->r, R, 0, 4, 0, false, 1.0, orth_tol, norm_A,
                               norm_A2_dev, canc2, nullptr, &s1, st);
    if (rc) return rc;
    // g0 = Z r0 lands in p as -g0 (the first direction); its input norm ||r0||^2 is red[0]
    rc = prime_project_stepped(a, A, At, a->r, a->p, R, 6, 0, 1, true, -1.0, orth_tol, norm_A,
                               norm_A2_dev, canc2, &s1, &s2, st);
    if (rc) return rc;
  }
  const int32_t idx_b[7] = {12, 4, 0, 2, 0, 6, 8}, idx_0[7] = {-1, 4, 0, 2, 0, 6, 8};
  ipx_prime_idx ix;
  for (int k = 0; k < 7; ++k) ix.i[k] = (b ? idx_b : idx_0)[k];
  hipLaunchKernelGGL(k_cg_prime_state_folds, dim3(1), dim3(IPX_BLOCK), 0, st, a->state, red, R.f,
                     ix, tol_in, radius, orth_tol, norm_A, canc2, radius_dev, norm_A2_dev, s2,
                     (b || !steps) ? 0 : 1);
  IPX_CHECK_LAUNCH();
  rc = launch_hp(a, nullptr, st);
  if (rc || first_end == 0) return rc;
  // the call's first batch behind the same entry (stop code 9: its launches do nothing)
  return cg_iterate(a, 0, first_end, st, nullptr);
}

extern "C" {

// Hp = H p with p'Hp partials (the tail of an iteration, also used once by
// the host to prime the loop).
int ipx_cg_hp(const ipx_cg_args *a, void *stream) {
  if (!a) return IPX_EINVAL;
  return launch_hp(a, nullptr, (hipStream_t)stream);
}

// The fused step2 + H.p kernel on its own (iteration `it`, step2 mode bits as
// in ipx_cg_resume); IPX_EINVAL when the argument block does not enable it.
int ipx_cg_step2_hp(const ipx_cg_args *a, int32_t it, int32_t mode, void *stream) {
  if (!a || !fused_hp(a)) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // long partial arrays are compacted first, as in the loop (cg_iterate): every workgroup folds
  // all of them, and at n = 1.6e7 the 6000 uncompacted ||g||^2 partials doubled what a
  // workgroup reads -- bench.py's "dominant kernel on its own" measured 495 us where the same
  // kernel takes 226 us inside the loop
  const double *p2 = a->part2, *p3 = a->part3, *p4 = a->part4;
  int np2 = part2_count(a), np3 = part3_count(a), np4 = part4_count(a);
  Compactor cmp(a);
  if (!(mode & 1)) cmp.add(p2, np2, 2, 1024);
  cmp.add(p3, np3, 2, 2048);
  if (!(mode & 2)) cmp.add(p4, np4, 1, 1024);
  int rc = cmp.launch(nullptr, st);
  if (rc) return rc;
  return launch_step2_hp(a, it, mode, p2, np2, p3, np3, p4, np4, st);
}

// Tail of an iteration after the host handled a stop-5 / stop-6 event:
// step2 with the given mode, then Hp = H p.
int ipx_cg_resume(const ipx_cg_args *a, int32_t it, int32_t mode, void *stream) {
  if (!a) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const double *guard = a->state + ST_STOP;
  int np3, np4;
  if (dense_loop(a)) {                        // workgroups of the dense matvecs (rows / 4, capped)
    np3 = (int)std::min<int64_t>((a->n + 3) / 4, 2048);
    np4 = (int)std::min<int64_t>((a->m + 3) / 4, 2048);
  } else {
    np3 = part3_count(a);
    np4 = part4_count(a);
  }
  hipLaunchKernelGGL(k_cg_step2, dim3(ipx_xcd_grid((int)a->vec_grid)), dim3(VB), 0, st, a->n,
                     a->state, it & 1, mode, a->part2, part2_count(a), a->part3, np3, a->part4,
                     np4, a->x, a->p, a->r, (int)a->vec_grid);
  IPX_CHECK_LAUNCH();
  return launch_hp(a, guard, st);
}

// Enqueue iterations it_begin .. it_end-1.  Never synchronises.
static int cg_iterate(const ipx_cg_args *a, int32_t it_begin, int32_t it_end, hipStream_t st,
                      hipEvent_t *ev);

int ipx_cg_iterate(const ipx_cg_args *a, int32_t it_begin, int32_t it_end, void *stream) {
  if (!a || it_end < it_begin) return IPX_EINVAL;
  return cg_iterate(a, it_begin, it_end, (hipStream_t)stream, nullptr);
}

// Instrumented variant for bench.py: HIP events are recorded on `stream`
// around every kernel class of every iteration; after one final stream
// synchronise the per-class totals (ms) are returned in ms_out[0..6]:
// step1, spmv A r, banded solve, spmv r-A'v, spmv A g, step2, spmv H p.
// Slower than ipx_cg_iterate (event records between kernels) -- use it for
// per-kernel attribution, not for the throughput number.
#define IPX_CG_NCLASS 7
int ipx_cg_iterate_timed(const ipx_cg_args *a, int32_t it_begin, int32_t it_end, float *ms_out,
                         void *stream) {
  if (!a || it_end < it_begin || !ms_out) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int per_it = IPX_CG_NCLASS + 1;
  const int nit = it_end - it_begin;
  std::vector<hipEvent_t> ev((size_t)nit * per_it);
  for (auto &e : ev)
    if (hipEventCreate(&e) != hipSuccess) return IPX_ELAUNCH;
  int rc = IPX_OK;
  for (int i = 0; i < nit && rc == IPX_OK; ++i)
    rc = cg_iterate(a, it_begin + i, it_begin + i + 1, st, ev.data() + (size_t)i * per_it);
  if (hipStreamSynchronize(st) != hipSuccess) rc = IPX_ELAUNCH;
  for (int c = 0; c < IPX_CG_NCLASS; ++c) ms_out[c] = 0.f;
  if (rc == IPX_OK) {
    for (int i = 0; i < nit; ++i)
      for (int c = 0; c < IPX_CG_NCLASS; ++c) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[(size_t)i * per_it + c], ev[(size_t)i * per_it + c + 1]);
        ms_out[c] += ms;
      }
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  return rc;
}

// Dense Jacobian (BASELINE config 2; solver_kind 2): the same state machine over the dense
// matvecs of csrc/dense.hip, whose epilogues carry the reductions.  A (m x n) and A' (n x m)
// row major in A_val / At_val, `banded` = G^-1 (M x M, M = m rounded up to 32; w and v are M
// long with a zero tail), H dense (H_rowptr NULL) or CSR.  Seven launches per iteration:
// step1, w = A r, v = G^-1 w, g = r - A'v, t = A g (the reference's orthogonality measure,
// projections.py:52, literally), step2, Hp = H p.
static int cg_iterate_dense(const ipx_cg_args *a, int32_t it_begin, int32_t it_end,
                            hipStream_t st) {
  const double *guard = a->state + ST_STOP;
  const int n = (int)a->n, m = (int)a->m;
  const int M = ((m + 31) / 32) * 32;
  const int grid = (int)a->vec_grid;
  const double *Ginv = (const double *)a->banded;
  for (int it = it_begin; it < it_end; ++it) {
    hipLaunchKernelGGL(k_cg_step1, dim3(ipx_xcd_grid(grid)), dim3(VB), 0, st, a->n, a->state,
                       it & 1, a->part1, part1_count(a), a->x, a->p, a->r, a->Hp, a->lb, a->ub,
                       a->part2, grid, own_all(a->n));
    IPX_CHECK_LAUNCH();
    int np = 0, np3 = 0, np4 = 0;
    int rc = ipx_dense_gemv_launch(m, n, a->A_val, n, a->r, 1.0, nullptr, 0.0, nullptr, a->w,
                                   nullptr, &np, guard, st);
    if (rc) return rc;
    rc = ipx_dense_gemv_launch(M, M, Ginv, M, a->w, 1.0, nullptr, 0.0, nullptr, a->v, nullptr, &np,
                               guard, st);
    if (rc) return rc;
    rc = ipx_dense_gemv_launch(n, m, a->At_val, m, a->v, -1.0, nullptr, 1.0, a->r, a->r, a->part3,
                               &np3, guard, st);                      // g = r - A'v, ||g||^2
    if (rc) return rc;
    rc = ipx_dense_gemv_launch(m, n, a->A_val, n, a->r, 1.0, nullptr, 0.0, nullptr, a->t, a->part4,
                               &np4, guard, st);                      // ||A g||^2
    if (rc) return rc;
    hipLaunchKernelGGL(k_cg_step2, dim3(ipx_xcd_grid(grid)), dim3(VB), 0, st, a->n, a->state,
                       it & 1, 0, a->part2, grid, a->part3, np3, a->part4, np4, a->x, a->p, a->r,
                       grid);
    IPX_CHECK_LAUNCH();
    rc = launch_hp(a, guard, st);
    if (rc) return rc;
  }
  return IPX_OK;
}

static int cg_iterate(const ipx_cg_args *a, int32_t it_begin, int32_t it_end, hipStream_t st,
                      hipEvent_t *ev) {
  if (dense_loop(a)) return ev ? IPX_EINVAL : cg_iterate_dense(a, it_begin, it_end, st);
  if (!ev && it_end > it_begin && ipx_cg_resident_ok(a)) {
    // small banded problems (one CU per workgroup of the solve): the whole batch as ONE
    // resident launch (csrc/resident.hip), which leaves x, p, r, Hp, the state block and the
    // partial arrays as these launches would -- the tile boundaries of p for the fused step2 +
    // H.p kernel (an event handler's resume, a later batch on the separate launches) follow it
    int rc = ipx_cg_resident_launch(a, it_begin, it_end, part1_count(a), part2_count(a),
                                    part3_count(a), part4_count(a), st);
    if (rc) return rc;
    if (fused_hp(a)) {
      const int tot = (int)(a->H_ntiles * 2 * a->H_hmax);
      hipLaunchKernelGGL(k_cg_save_pb, dim3((tot + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0,
                         st, a->p, a->H_tiles, (int)a->H_ntiles, (int)a->H_hmax, a->pb);
      IPX_CHECK_LAUNCH();
    }
    return IPX_OK;
  }
  const double *guard = a->state + ST_STOP;
  ipx_csr_view A{(int)a->m, (int)a->n, a->A_rowptr, a->A_colidx, a->A_val, a->A_tiles, (int)a->A_ntiles};
  ipx_csr_view At{(int)a->n, (int)a->m, a->At_rowptr, a->At_colidx, a->At_val, a->At_tiles, (int)a->At_ntiles};
#define MARK(i) do { if (ev) (void)hipEventRecord(ev[i], st); } while (0)
  int np4 = 1;
  for (int it = it_begin; it < it_end; ++it) {
    MARK(0);
    int rc, np3 = (int)a->At_ntiles;
    const bool fuse1 = fused_ar(a);
    const double *r_in = fuse1 ? a->r_next : a->r;      // what the r - A'v SpMV reads
    Compactor cmp(a);
    const double *p1 = a->part1;
    int np1 = (int)a->H_ntiles;
    cmp.add(p1, np1, 2, 2048);      // beyond what a consumer folds in one or two rounds
    // trust radius +inf, no box (fused step1 implies no box): the radius / box tests of
    // qp_subproblem.py:583,599 cannot trigger; their sums are neither formed nor folded
    const bool no_xn2 = fuse1 && a->no_radius != 0;
    rc = cmp.launch(guard, st);
    if (rc) return rc;
    if (fuse1) {
      MARK(1);
      rc = launch_step1_ar(a, it, p1, np1, st, no_xn2);   // r_next = r + alpha Hp;  w = A r_next
      if (rc) return rc;
      MARK(2);
    } else if (a->m > 0 && box_project(a)) {
      const ipx_boxschur_args *b = (const ipx_boxschur_args *)a->banded;
      const int nblk = step1_box_blocks(a);
      launch_step1_box<false>(a, b, it, p1, np1, nblk, own_all(a->n), ipx_no_peer{}, st);
      IPX_CHECK_LAUNCH();
      MARK(1);
    } else {
      hipLaunchKernelGGL(k_cg_step1, dim3(ipx_xcd_grid((int)a->vec_grid)), dim3(VB), 0, st, a->n,
                         a->state, it & 1, p1, np1, a->x, a->p, a->r,
                         a->Hp, a->lb, a->ub, a->part2, (int)a->vec_grid, own_all(a->n));
      IPX_CHECK_LAUNCH();
      MARK(1);
    }
    if (a->m > 0 && box_project(a)) {
      // simple (box) rows eliminated analytically and never multiplied as matrix rows:
      // g = r - A'(A A')^-1 A r in one call (csrc/boxschur.hip ipx_boxschur_project)
      int32_t n3 = 0, n4 = 0;
      rc = ipx_boxschur_project_from((const ipx_boxschur_args *)a->banded, a->r, a->r, a->part3,
                                     &n3, a->part4, &n4, guard, 1, st);
      if (rc) return rc;
      np3 = n3;
      np4 = n4;
      MARK(2); MARK(3); MARK(4); MARK(5);
    } else if (a->m > 0) {
      if (!fuse1) {
        // w = A r_next
        rc = ipx_spmv_launch(A, a->r, 1.0, nullptr, 0.0, nullptr, a->w, nullptr, guard, st);
        if (rc) return rc;
        MARK(2);
      }
      // v = (AA')^-1 w, and ||A g||^2 for the orthogonality test as the
      // constraint-space residual ||w - (A A') v||^2 from the same launch
      // (see k_correct_oop / k_band_residual); part4 holds ceil(m/256) doubles
      np3 = (int)a->At_ntiles;
      if (a->solver_kind == 0 && a->At_vown && a->At_qv > 0) {
        // decoupled banded solve with g = r - A'v as its tail (one launch for both);
        // ||g||^2 partials are then per workgroup of that kernel
        rc = ipx_banded_solve_resid_atv_launch(a->banded, a->w, a->v, a->part4, &np4,
                                               a->At_rowptr, a->At_colidx, a->At_val, r_in, a->r,
                                               a->At_vown, (int)a->At_qv, a->part3, guard, st,
                                               a->At_ell_row, a->At_ell_val, a->n);
        if (rc) return rc;
        np3 = np4;
        MARK(3);
        MARK(4);
      } else {
        if (a->solver_kind == 1)
          rc = ipx_boxschur_solve((const ipx_boxschur_args *)a->banded, a->w, a->v, a->part4,
                                  &np4, guard, st);
        else
          rc = ipx_banded_solve_resid_launch(a->banded, a->w, a->v, a->part4, &np4, guard, st);
        if (rc) return rc;
        MARK(3);
        // r = r - A'v  (g_next), partials of ||g||^2
        rc = ipx_spmv_launch(At, a->v, -1.0, nullptr, 1.0, r_in, a->r, a->part3, guard, st);
        if (rc) return rc;
        MARK(4);
      }
      MARK(5);
    }
    const double *p2 = a->part2, *p3 = a->part3, *p4 = a->part4;
    int np2 = part2_count(a), n4 = np4;
    if (!no_xn2) cmp.add(p2, np2, 2, 1024);
    if (a->m > 0) {
      cmp.add(p3, np3, 2, 2048);
      cmp.add(p4, n4, 1, 1024);
    }
    rc = cmp.launch(guard, st);
    if (rc) return rc;
    if (fused_hp(a)) {
      MARK(6);
      rc = launch_step2_hp(a, it, (a->m > 0 ? 0 : 2) | (no_xn2 ? 1 : 0), p2, np2,
                           p3, np3, p4, n4, st, a->r);
    } else {
      hipLaunchKernelGGL(k_cg_step2, dim3(ipx_xcd_grid((int)a->vec_grid)), dim3(VB), 0, st, a->n,
                         a->state, it & 1, (a->m > 0 ? 0 : 2) | (no_xn2 ? 1 : 0), p2, np2, p3, np3,
                         p4, n4, a->x, a->p, a->r, (int)a->vec_grid);
      IPX_CHECK_LAUNCH();
      MARK(6);
      rc = launch_hp(a, guard, st);
    }
    if (rc) return rc;
    MARK(7);
  }
#undef MARK
  return IPX_OK;
}

}  // extern "C"
