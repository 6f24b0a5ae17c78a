// Analytic elimination of "simple" rows of A A' (SURVEY.md section 8(f) N3).
//
// In barrier problems the augmented Jacobian has, for every bounded variable
// x_j, rows of the form  alpha * e_j' (+ one private slack entry)  -- the box
// constraints -- next to the general rows R.  In G = A A' those rows only
// couple (a) with the other simple row of the same variable and (b) with the
// general rows through column j.  Eliminating them per variable (1x1 / 2x2
// blocks B_j) leaves the Schur complement
//
//      Sigma = A_R diag(w) A_R',   w_j = 1 - alpha_j' B_j^-1 alpha_j  (1 elsewhere)
//
// which has the sparsity of A_R A_R' (tridiagonal for the banded benchmark)
// instead of a half bandwidth of ~40.  One application of (A A')^-1:
//
//      t   = B^-1 w_S                      (k_pairs_tsolve, also u_j = alpha' t)
//      v_R = Sigma^-1 (w_R - A_R u)        (SpMV + banded solve)
//      v_S = t - B^-1 (alpha * (A_R' v_R)_j)   (SpMV + k_pairs_vsolve)
//
// Reference counterpart: none -- the reference factors the whole augmented
// system with SuperLU (projections.py:93-172).  Same operator, verified against
// a direct solve in tests/test_gpu_qp.py.
#include "ipx_common.h"

namespace {

// group g: rows p = rowp[g] and q = rowq[g] (q < 0: single row).
// B = [[ap^2 + sp^2, ap aq], [ap aq, aq^2 + sq^2]]  (a = shared-column entry,
// s = private entry of the row, 0 if none).  inv = (i11, i12, i22).
__global__ void __launch_bounds__(IPX_BLOCK)
k_pairs_factor(int ng, const int32_t *__restrict__ rowp, const int32_t *__restrict__ rowq,
               const int32_t *__restrict__ pos_a, const int32_t *__restrict__ pos_s,
               const double *__restrict__ val, double *__restrict__ alpha,
               double *__restrict__ inv, double *__restrict__ weight_col,
               const int32_t *__restrict__ col, int *flag, double *__restrict__ grp,
               double *__restrict__ grp2) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const int p = rowp[g], q = rowq[g];
  const double ap = val[pos_a[p]];
  const double sp = pos_s[p] >= 0 ? val[pos_s[p]] : 0.0;
  alpha[p] = ap;
  double aq = 0.0, sq = 0.0;
  if (q >= 0) {
    aq = val[pos_a[q]];
    sq = pos_s[q] >= 0 ? val[pos_s[q]] : 0.0;
    alpha[q] = aq;
  }
  if (grp) {                 // (ap, sp, aq, sq) per group: what the fused projection reads
    grp[4 * g] = ap; grp[4 * g + 1] = sp; grp[4 * g + 2] = aq; grp[4 * g + 3] = sq;
  }
  if (grp2) {                // the compact form (ipx_group_tab::grp2); flag bit 1: not exact
    grp2[2 * g] = copysign(sp, ap); grp2[2 * g + 1] = copysign(sq, aq);
    const bool unit = fabs(ap) == 1.0 && !signbit(sp) &&
                      (q < 0 || (fabs(aq) == 1.0 && !signbit(sq)));
    if (!unit) atomicOr(flag, 2);
  }
  double i11, i12, i22, wgt;
  if (!ipx_group_inverse(q >= 0, ap, sp, aq, sq, i11, i12, i22, wgt)) atomicOr(flag, 1);
  inv[3 * g] = i11; inv[3 * g + 1] = i12; inv[3 * g + 2] = i22;
  weight_col[col[g]] = wgt;
}

// t_S = B^-1 w_S (written at the simple rows of t) and u[col_g] = alpha' t.
__global__ void __launch_bounds__(IPX_BLOCK)
k_pairs_tsolve(int ng, const int32_t *__restrict__ rowp, const int32_t *__restrict__ rowq,
               const double *__restrict__ inv, const double *__restrict__ alpha,
               const double *__restrict__ w, double *__restrict__ t,
               const int32_t *__restrict__ col, double *__restrict__ u) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const int p = rowp[g], q = rowq[g];
  const double wp = w[p];
  if (q < 0) {
    const double tp = inv[3 * g] * wp;
    t[p] = tp;
    u[col[g]] = alpha[p] * tp;
  } else {
    const double wq = w[q];
    const double tp = inv[3 * g] * wp + inv[3 * g + 1] * wq;
    const double tq = inv[3 * g + 1] * wp + inv[3 * g + 2] * wq;
    t[p] = tp; t[q] = tq;
    u[col[g]] = alpha[p] * tp + alpha[q] * tq;
  }
}

// v_S = t_S - B^-1 (alpha * y[col_g])
__global__ void __launch_bounds__(IPX_BLOCK)
k_pairs_vsolve(int ng, const int32_t *__restrict__ rowp, const int32_t *__restrict__ rowq,
               const double *__restrict__ inv, const double *__restrict__ alpha,
               const double *__restrict__ t, const double *__restrict__ y,
               const int32_t *__restrict__ col, double *__restrict__ v) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const int p = rowp[g], q = rowq[g];
  const double yj = y[col[g]];
  const double rp = alpha[p] * yj;
  if (q < 0) {
    v[p] = t[p] - inv[3 * g] * rp;
  } else {
    const double rq = alpha[q] * yj;
    v[p] = t[p] - (inv[3 * g] * rp + inv[3 * g + 1] * rq);
    v[q] = t[q] - (inv[3 * g + 1] * rp + inv[3 * g + 2] * rq);
  }
}

// tsolve for the groups and, in the same launch, the gather of the general
// rows' right-hand side:  threads [0, ng) -> groups, [ng, ng + mR) -> wR[i] = w[general[i]].
__global__ void __launch_bounds__(IPX_BLOCK)
k_pairs_tsolve_gather(int ng, int mR, const int32_t *__restrict__ rowp,
                      const int32_t *__restrict__ rowq, const double *__restrict__ inv,
                      const double *__restrict__ alpha, const double *__restrict__ w,
                      double *__restrict__ t, const int32_t *__restrict__ col,
                      double *__restrict__ u, const int32_t *__restrict__ general,
                      double *__restrict__ wR, const double *__restrict__ guard) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (guard && *guard != 0.0) return;
  if (g >= ng) {
    const int i = g - ng;
    if (i < mR) wR[i] = w[general[i]];
    return;
  }
  const int p = rowp[g], q = rowq[g];
  const double wp = w[p];
  if (q < 0) {
    const double tp = inv[3 * g] * wp;
    t[p] = tp;
    u[col[g]] = alpha[p] * tp;
  } else {
    const double wq = w[q];
    const double tp = inv[3 * g] * wp + inv[3 * g + 1] * wq;
    const double tq = inv[3 * g + 1] * wp + inv[3 * g + 2] * wq;
    t[p] = tp; t[q] = tq;
    u[col[g]] = alpha[p] * tp + alpha[q] * tq;
  }
}

// vsolve for the groups and the scatter of the general rows' solution:
// threads [0, ng) -> groups, [ng, ng + mR) -> v[general[i]] = vR[i].
__global__ void __launch_bounds__(IPX_BLOCK)
k_pairs_vsolve_scatter(int ng, int mR, const int32_t *__restrict__ rowp,
                       const int32_t *__restrict__ rowq, const double *__restrict__ inv,
                       const double *__restrict__ alpha, const double *__restrict__ t,
                       const double *__restrict__ y, const int32_t *__restrict__ col,
                       double *__restrict__ v, const int32_t *__restrict__ general,
                       const double *__restrict__ vR, const double *__restrict__ guard) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (guard && *guard != 0.0) return;
  if (g >= ng) {
    const int i = g - ng;
    if (i < mR) v[general[i]] = vR[i];
    return;
  }
  const int p = rowp[g], q = rowq[g];
  const double yj = y[col[g]];
  const double rp = alpha[p] * yj;
  if (q < 0) {
    v[p] = t[p] - inv[3 * g] * rp;
  } else {
    const double rq = alpha[q] * yj;
    v[p] = t[p] - (inv[3 * g] * rp + inv[3 * g + 1] * rq);
    v[q] = t[q] - (inv[3 * g + 1] * rp + inv[3 * g + 2] * rq);
  }
}

// ---- the whole projection g = r - A'(A A')^-1 A r without the simple rows ever being
// multiplied as matrix rows (the CG loop's use).  The simple rows' part
// of w = A r, of the solve and of A'v is per-group arithmetic on r itself:
//   pre :  w_p, w_q from r (the rows' one or two entries) -> t = B^-1 w_S -> u = r - alpha't
//   ...    rhs = A_R u (= A_R r - A_R alpha't),  v_R = Sigma^-1 rhs,  y = A_R' v_R   (general rows)
//   post:  y_c = (A_R' v_R)_c from the column's few entries, w, t recomputed,
//          v_S = t - B^-1 (alpha y[col]);  g on the group's columns
//          (shared column: r - (y + a_p v_p + a_q v_q); private columns: r - s v), and
//          g = r - y on the columns that belong to no group; ||g||^2 partials.
// The simple rows' values are formed with the operation order of the SpMV-based path (row
// sums left to right, y = alpha * sum + beta * yin; general rows before the simple ones in
// a column of A' -- true of the barrier problem's [nonlinear; lower; upper] row order); the
// Schur right-hand side and the ||g||^2 partial sums are associated differently, so the two
// paths agree to rounding, not bitwise.
// u <- r - (alpha't on the shared columns, 0 elsewhere), on the columns A_R touches
// ([0, ny)): the Schur right-hand side is then ONE product, rhs = A_R u = A_R r - A_R (alpha't).
__global__ void __launch_bounds__(IPX_BLOCK)
k_pairs_pre(int ng, int ngen, ipx_group_tab T, const int32_t *__restrict__ gen_cols, int ny,
            const double *__restrict__ r, double *__restrict__ u,
            const double *__restrict__ guard) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (guard && *guard != 0.0) return;
  if (g >= ng) {
    if (g - ng < ngen) {
      const int c = gen_cols[g - ng];
      if (c < ny) u[c] = r[c];
    }
    return;
  }
  int c, cp, cq;
  double ap, sp, aq, sq, rc, rp, rq, wp, wq;
  ipx_group_w(T, g, r, c, cp, cq, ap, sp, aq, sq, rc, rp, rq, wp, wq);
  double i11, i12, i22, wgt, ut;
  ipx_group_inverse(cq != -2, ap, sp, aq, sq, i11, i12, i22, wgt);
  if (cq == -2) {
    ut = ap * (i11 * wp);
  } else {
    const double tp = i11 * wp + i12 * wq;
    const double tq = i12 * wp + i22 * wq;
    ut = ap * tp + aq * tq;
  }
  if (c < ny) u[c] = rc - ut;
}

constexpr int POST_ITEMS = 4;      // items per thread, or a few more (ipx_balanced_rounds):
                                   // 1024+ per workgroup = one partial
constexpr int POST_BATCH = 1;      // items in flight together, computed columns (2 / 4 measured
                                   // at config 5: 11.3 / 12.0 against 10.7 us)
static int post_rounds(const ipx_boxschur_args *a) {
  return POST_ITEMS;     // (5 / 6 / 9 rounds measured at config 5: 12.0 / 13.4 / 12.4 against 10.6 us)
}

// MODE: the form of the group tables (ipx_group_tab).  YELL: the columns of A_R per item in
// ELL(2) form (yrp = the column planes, yval = the value planes; ycol unused) -- same sums
// (0 + first + second; an absent entry adds a signed zero to a sum that is not -0).
template <int MODE, bool YELL>
__global__ void __launch_bounds__(IPX_BLOCK)
k_pairs_post(int ng, int ngen, ipx_group_tab T, const int32_t *__restrict__ gen_cols,
             const double *r, const int32_t *__restrict__ yrp,
             const int32_t *__restrict__ ycol, const double *__restrict__ yval,
             const double *__restrict__ vR, int ny, double *gout,
             double *__restrict__ part, int npart, const double *__restrict__ guard,
             ipx_own_ranges own, int rounds) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  if (guard && *guard != 0.0) return;
  const int nitems = ng + ngen;
  // y_c = (A_R' v_R)_c: the column's few general-row entries, summed left to right like the
  // SpMV's row sum (columns beyond ny are touched by no general row)
  auto ycolumn = [&](int c, int i) {
    double sum = 0.0;
    if constexpr (YELL) {
      const int c0 = yrp[i], c1 = yrp[nitems + i];
      const double v0 = yval[i], v1 = yval[nitems + i];
      sum += v0 * vR[c0];
      sum += v1 * vR[c1];
    } else if (c < ny) {
      const int b = yrp[c + 1];
      for (int k = yrp[c]; k < b; ++k) sum += yval[k] * vR[ycol[k]];
    }
    return 1.0 * sum;
  };
  double acc = 0.0;
  // one item from its loaded operands (cp / cq < 0: no such column)
  auto item = [&](bool grp, int c, int cp, int cq, double ap, double sp, double aq, double sq,
                  double rc, double rp, double rq, double yj) {
    if (grp) {
      const bool has_q = cq != -2;
      double i11, i12, i22, wgt;
      ipx_group_inverse(has_q, ap, sp, aq, sq, i11, i12, i22, wgt);
      const double wp = cp < 0 ? ap * rc : (cp > c ? ap * rc + sp * rp : sp * rp + ap * rc);
      const double wq = cq < 0 ? aq * rc : (cq > c ? aq * rc + sq * rq : sq * rq + aq * rc);
      const double ep = ap * yj;
      double vp, vq = 0.0;
      if (!has_q) {
        const double tp = i11 * wp;
        vp = tp - i11 * ep;
      } else {
        const double tp = i11 * wp + i12 * wq;
        const double tq = i12 * wp + i22 * wq;
        const double eq = aq * yj;
        vp = tp - (i11 * ep + i12 * eq);
        vq = tq - (i12 * ep + i22 * eq);
      }
      // column c of A': general rows (y), then row p, then row q
      double sum = yj + ap * vp;
      if (has_q) sum += aq * vq;
      double gc = -1.0 * sum;
      gc += 1.0 * rc;
      gout[c] = gc;
      if (own.has(c)) acc += gc * gc;
      if (cp >= 0) {
        double gp = -1.0 * (sp * vp);
        gp += 1.0 * rp;
        gout[cp] = gp;
        if (own.has(cp)) acc += gp * gp;
      }
      if (cq >= 0) {
        double gq = -1.0 * (sq * vq);
        gq += 1.0 * rq;
        gout[cq] = gq;
        if (own.has(cq)) acc += gq * gq;
      }
    } else {
      double gc = -1.0 * yj;
      gc += 1.0 * rc;
      gout[c] = gc;
      if (own.has(c)) acc += gc * gc;
    }
  };
  if constexpr (MODE == IPX_GROUPS_AFFINE && YELL) {
    // computed columns and item-indexed y: the operands of POST_BATCH items are requested
    // together (clamped indices, no branches), then their v_R gathers, then the arithmetic --
    // g may alias r, so the compiler cannot move an item's loads above its predecessor's stores
    for (int k0 = 0; k0 < rounds; k0 += POST_BATCH) {
      constexpr int NB = POST_BATCH;
      int c[NB], y0[NB], y1[NB];
      bool grp[NB], valid[NB];
      double ep[NB], eq[NB], w0[NB], w1[NB];
      double rc[NB], rp[NB], rq[NB], u0[NB], u1[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int i = (blockIdx.x * rounds + k0 + k) * IPX_BLOCK + threadIdx.x;
        const bool on = k0 + k < rounds;                      // (uniform)
        valid[k] = on && i < nitems;
        const int ic = min(i, nitems - 1);
        grp[k] = ic < ng;
        const int gi = min(ic, ng - 1);
        c[k] = grp[k] ? T.c0 + ic : T.gen0 + (ic - ng);
        if (on) {
          y0[k] = yrp[ic]; y1[k] = yrp[nitems + ic];
          w0[k] = yval[ic]; w1[k] = yval[nitems + ic];
          ep[k] = T.grp2[2 * gi]; eq[k] = T.grp2[2 * gi + 1];
          rc[k] = r[c[k]];
          rp[k] = r[grp[k] ? c[k] + T.dp : c[k]];
          rq[k] = r[grp[k] ? c[k] + T.dq : c[k]];
        } else {
          y0[k] = y1[k] = 0;
          w0[k] = w1[k] = ep[k] = eq[k] = rc[k] = rp[k] = rq[k] = 0.0;
        }
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) { u0[k] = vR[y0[k]]; u1[k] = vR[y1[k]]; }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        if (!valid[k]) continue;
        double sum = 0.0;
        sum += w0[k] * u0[k];
        sum += w1[k] * u1[k];
        item(grp[k], c[k], grp[k] ? c[k] + T.dp : -1, grp[k] ? c[k] + T.dq : -1,
             copysign(1.0, ep[k]), fabs(ep[k]), copysign(1.0, eq[k]), fabs(eq[k]), rc[k], rp[k],
             rq[k], 1.0 * sum);
      }
    }
  } else {
#pragma unroll 2
    for (int k = 0; k < rounds; ++k) {
      const int i = (blockIdx.x * rounds + k) * IPX_BLOCK + threadIdx.x;
      if (i >= nitems) continue;
      // all independent loads first: tables, r on the item's columns, the row pointers of y
      const bool grp = i < ng;
      int c, cp = -1, cq = -1;
      double ap = 0.0, sp = 0.0, aq = 0.0, sq = 0.0;
      if (grp) {
        ipx_group_cols<MODE>(T, i, c, cp, cq);
        ipx_group_coeffs<MODE != IPX_GROUPS_FULL>(T, i, ap, sp, aq, sq);
      } else {
        c = MODE == IPX_GROUPS_AFFINE ? T.gen0 + (i - ng) : gen_cols[i - ng];
      }
      const double rc = r[c];
      const double rp = cp >= 0 ? r[cp] : 0.0;
      const double rq = cq >= 0 ? r[cq] : 0.0;
      const double yj = ycolumn(c, i);
      item(grp, c, cp, cq, ap, sp, aq, sq, rc, rp, rq, yj);
    }
  }
  const double tot = ipx_block_reduce<IPX_SUM>(acc, lds);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = tot;
    part[npart + blockIdx.x] = 0.0;
  }
}

}  // namespace

extern "C" int ipx_boxschur_project_count(const ipx_boxschur_args *a);

// have_up != 0: a->up already holds r - alpha't (the CG loop's step1 kernel forms it while it
// updates r: csrc/cg.hip k_cg_step1_box)
int ipx_boxschur_project_from(const ipx_boxschur_args *a, const double *r, double *g,
                              double *part_g, int32_t *npart_g, double *part_res,
                              int32_t *npart_res, const double *guard, int have_up,
                              hipStream_t stream, const ipx_own_ranges *own) {
  if (!a || !r || !g || !part_g || !a->gcol || !a->grp || !a->up ||
      (a->ngen > 0 && !a->gen_cols))
    return IPX_EINVAL;
  hipStream_t st = stream;
  const ipx_group_tab T = ipx_boxschur_tab(a);
  const int ng = (int)a->ng;
  const int64_t items = a->ng + a->ngen;
  if (items > 0 && !have_up) {
    hipLaunchKernelGGL(k_pairs_pre, dim3((unsigned)((items + IPX_BLOCK - 1) / IPX_BLOCK)),
                       dim3(IPX_BLOCK), 0, st, ng, (int)a->ngen, T, a->gen_cols, (int)a->ny, r,
                       a->up, guard);
    IPX_CHECK_LAUNCH();
  }
  ipx_csr_view AR{(int)a->mR, (int)a->n, a->AR_rowptr, a->AR_colidx, a->AR_val, a->AR_tiles,
                  (int)a->AR_ntiles};
  // rhs = A_R (r - alpha't)   (= w_R - A_R u of ipx_boxschur_solve, one product) and the Schur
  // solve: ONE launch when A_R's rows have one power-of-two length and the solve is the cyclic
  // reduction (the product is formed inside it, same bits), else the SpMV and the solve
  int np = 0;
  int rc = IPX_EUNSUPPORTED;
  const int nblk = ipx_boxschur_project_count(a);
  bool post_done = false;
  if (a->AR_rowlen > 0 && part_res) {
    int logL = 0;
    while ((1 << logL) < a->AR_rowlen) ++logL;
    // one GPU, compact tables: the per-item back substitution rides in the solve's kernel
    // (its partials fill the same `nblk` entries per half: workgroup sums, then zeros)
    if (!own && a->post_own_g && a->post_own_e && a->grp2 && a->gaffine && a->yell_col &&
        a->yell_val && nblk > 0) {
      ipx_post_job pj{a->post_own_g, a->post_own_e, ng, (int)items, T, a->yell_col, a->yell_val,
                      r, g, part_g, nblk, (int)a->post_rows_wg, (int)a->post_reach};
      rc = ipx_banded_solve_rows_launch(a->inner, a->AR_colidx, a->AR_val, a->up, logL, a->vR,
                                        part_res, &np, guard, st, &pj);
      post_done = rc == IPX_OK;
    }
    if (!post_done)
      rc = ipx_banded_solve_rows_launch(a->inner, a->AR_colidx, a->AR_val, a->up, logL, a->vR,
                                        part_res, &np, guard, st);
  }
  if (rc == IPX_EUNSUPPORTED) {
    rc = ipx_spmv_launch(AR, a->up, 1.0, nullptr, 0.0, nullptr, a->rhs, nullptr, guard, st);
    if (rc) return rc;
    if (part_res)
      rc = ipx_banded_solve_resid_launch(a->inner, a->rhs, a->vR, part_res, &np, guard, st);
    else
      rc = ipx_banded_solve_guarded(a->inner, a->rhs, a->vR, guard, st);
  }
  if (rc) return rc;
  if (npart_res) *npart_res = np;
  if (npart_g) *npart_g = nblk;
  if (post_done) return IPX_OK;
  ipx_own_ranges all;                                    // every column counts
  for (int k = 0; k < 4; ++k) { all.lo[k] = 0; all.hi[k] = 0; }
  all.hi[0] = a->n;
  if (nblk > 0) {
    const bool yell = a->yell_col && a->yell_val;
    const int32_t *yrp = yell ? a->yell_col : a->ARt_rowptr;
    const double *yval = yell ? a->yell_val : a->ARt_val;
#define IPX_POST(M, Y)                                                                          \
    hipLaunchKernelGGL((k_pairs_post<M, Y>), dim3(nblk), dim3(IPX_BLOCK), 0, st, ng,            \
                       (int)a->ngen, T, a->gen_cols, r, yrp, a->ARt_colidx, yval, a->vR,        \
                       (int)a->ny, g, part_g, nblk, guard, own ? *own : all, post_rounds(a))
    switch (ipx_group_mode(T) * 2 + (yell ? 1 : 0)) {
      case 5: IPX_POST(IPX_GROUPS_AFFINE, true); break;
      case 4: IPX_POST(IPX_GROUPS_AFFINE, false); break;
      case 3: IPX_POST(IPX_GROUPS_UNIT, true); break;
      case 2: IPX_POST(IPX_GROUPS_UNIT, false); break;
      case 1: IPX_POST(IPX_GROUPS_FULL, true); break;
      default: IPX_POST(IPX_GROUPS_FULL, false);
    }
#undef IPX_POST
    IPX_CHECK_LAUNCH();
  }
  return IPX_OK;
}

extern "C" {

// weight_col must be pre-filled with 1.0 (length = number of columns); alpha is
// a scratch vector over the rows; inv holds 3*ng doubles; flag is a device int
// (bit 0 set when a block is not positive definite).
int ipx_pairs_factor(int32_t ng, const int32_t *rowp, const int32_t *rowq, const int32_t *pos_a,
                     const int32_t *pos_s, const double *val, const int32_t *col, double *alpha,
                     double *inv, double *weight_col, int *flag, double *grp, double *grp2,
                     void *stream) {
  if (ng < 0) return IPX_EINVAL;
  if (ng == 0) return IPX_OK;
  if (!rowp || !rowq || !pos_a || !pos_s || !val || !col || !alpha || !inv || !weight_col || !flag)
    return IPX_EINVAL;
  hipLaunchKernelGGL(k_pairs_factor, dim3((ng + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0,
                     (hipStream_t)stream, ng, rowp, rowq, pos_a, pos_s, val, alpha, inv, weight_col,
                     col, flag, grp, grp2);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

int ipx_pairs_tsolve(int32_t ng, const int32_t *rowp, const int32_t *rowq, const double *inv,
                     const double *alpha, const double *w, double *t, const int32_t *col,
                     double *u, void *stream) {
  if (ng < 0) return IPX_EINVAL;
  if (ng == 0) return IPX_OK;
  hipLaunchKernelGGL(k_pairs_tsolve, dim3((ng + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0,
                     (hipStream_t)stream, ng, rowp, rowq, inv, alpha, w, t, col, u);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

int ipx_pairs_vsolve(int32_t ng, const int32_t *rowp, const int32_t *rowq, const double *inv,
                     const double *alpha, const double *t, const double *y, const int32_t *col,
                     double *v, void *stream) {
  if (ng < 0) return IPX_EINVAL;
  if (ng == 0) return IPX_OK;
  hipLaunchKernelGGL(k_pairs_vsolve, dim3((ng + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0,
                     (hipStream_t)stream, ng, rowp, rowq, inv, alpha, t, y, col, v);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// One (A A')^-1 application from a prepared argument block (every pointer is a
// device pointer; see ipsolver/boxschur.py which owns the buffers), plus the
// per-workgroup partials of the normal-equation residual.  The eliminated
// rows satisfy their equations exactly by construction, so the residual of
// the whole system is the residual of the Schur system on the general rows.
int ipx_boxschur_solve(const ipx_boxschur_args *a, const double *w, double *v, double *partial,
                       int32_t *npartial, const double *guard, void *stream) {
  if (!a || !w || !v) return IPX_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int rc = IPX_OK;
  const int tot = (int)(a->ng + a->mR);
  if (tot > 0) {
    hipLaunchKernelGGL(k_pairs_tsolve_gather, dim3((tot + IPX_BLOCK - 1) / IPX_BLOCK),
                       dim3(IPX_BLOCK), 0, st, (int)a->ng, (int)a->mR, a->rowp, a->rowq, a->inv,
                       a->alpha, w, a->t, a->col, a->u, a->general, a->wR, guard);
    IPX_CHECK_LAUNCH();
  }
  ipx_csr_view AR{(int)a->mR, (int)a->n, a->AR_rowptr, a->AR_colidx, a->AR_val, a->AR_tiles,
                  (int)a->AR_ntiles};
  ipx_csr_view ARt{(int)a->n, (int)a->mR, a->ARt_rowptr, a->ARt_colidx, a->ARt_val, a->ARt_tiles,
                   (int)a->ARt_ntiles};
  // rhs = w_R - A_R u
  rc = ipx_spmv_launch(AR, a->u, -1.0, nullptr, 1.0, a->wR, a->rhs, nullptr, guard, st);
  if (rc) return rc;
  int np = 0;
  if (partial)
    rc = ipx_banded_solve_resid_launch(a->inner, a->rhs, a->vR, partial, &np, guard, st);
  else
    rc = ipx_banded_solve_guarded(a->inner, a->rhs, a->vR, guard, st);
  if (rc) return rc;
  if (npartial) *npartial = np;
  // y = A_R' v_R
  rc = ipx_spmv_launch(ARt, a->vR, 1.0, nullptr, 0.0, nullptr, a->y, nullptr, guard, st);
  if (rc) return rc;
  if (tot > 0) {
    hipLaunchKernelGGL(k_pairs_vsolve_scatter, dim3((tot + IPX_BLOCK - 1) / IPX_BLOCK),
                       dim3(IPX_BLOCK), 0, st, (int)a->ng, (int)a->mR, a->rowp, a->rowq, a->inv,
                       a->alpha, a->t, a->y, a->col, v, a->general, a->vR, guard);
    IPX_CHECK_LAUNCH();
  }
  return IPX_OK;
}

// Number of ||g||^2 partials ipx_boxschur_project writes (per half of its partial array).
int ipx_boxschur_project_count(const ipx_boxschur_args *a) {
  if (!a) return 0;
  const int64_t items = a->ng + a->ngen, per = (int64_t)IPX_BLOCK * post_rounds(a);
  return (int)((items + per - 1) / per);
}

// g = r - A'(A A')^-1 A r in one call (g may alias r): the CG loop's projection step.
// part_g (2 x ipx_boxschur_project_count doubles) receives the ||g||^2 partials (second half
// zero), part_res / *npart_res the residual partials of the Schur system (= ||A g||^2, see
// ipx_boxschur_solve).  Needs the group tables of the argument block (gcol, grp, gen_cols);
// IPX_EINVAL without them.
int ipx_boxschur_project(const ipx_boxschur_args *a, const double *r, double *g, double *part_g,
                         int32_t *npart_g, double *part_res, int32_t *npart_res,
                         const double *guard, void *stream) {
  return ipx_boxschur_project_from(a, r, g, part_g, npart_g, part_res, npart_res, guard, 0,
                                   (hipStream_t)stream);
}

}  // extern "C"
