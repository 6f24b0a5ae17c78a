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
               const int32_t *__restrict__ col, int *flag) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const int p = rowp[g], q = rowq[g];
  const double ap = val[pos_a[p]];
  const double sp = pos_s[p] >= 0 ? val[pos_s[p]] : 0.0;
  alpha[p] = ap;
  const double b11 = ap * ap + sp * sp;
  double wgt;      // 1 - alpha' B^-1 alpha, in its cancellation-free form
  if (q < 0) {
    if (!(b11 > 0.0)) atomicOr(flag, 1);
    inv[3 * g] = 1.0 / b11; inv[3 * g + 1] = 0.0; inv[3 * g + 2] = 0.0;
    wgt = sp * sp / b11;
  } else {
    const double aq = val[pos_a[q]];
    const double sq = pos_s[q] >= 0 ? val[pos_s[q]] : 0.0;
    alpha[q] = aq;
    const double b22 = aq * aq + sq * sq, b12 = ap * aq;
    // (ap^2+sp^2)(aq^2+sq^2) - (ap aq)^2 without the cancellation (slacks of active bounds
    // are ~1e-8 next to ap = aq = 1)
    const double det = ap * ap * (sq * sq) + sp * sp * (aq * aq) + sp * sp * (sq * sq);
    if (!(det > 0.0) || !(b11 > 0.0)) atomicOr(flag, 1);
    const double i11 = b22 / det, i12 = -b12 / det, i22 = b11 / det;
    inv[3 * g] = i11; inv[3 * g + 1] = i12; inv[3 * g + 2] = i22;
    wgt = (sp * sp) * (sq * sq) / det;
  }
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

}  // namespace

extern "C" {

// weight_col must be pre-filled with 1.0 (length = number of columns); alpha is
// a scratch vector over the rows; inv holds 3*ng doubles; flag is a device int
// (bit 0 set when a block is not positive definite).
int ipx_pairs_factor(int32_t ng, const int32_t *rowp, const int32_t *rowq, const int32_t *pos_a,
                     const int32_t *pos_s, const double *val, const int32_t *col, double *alpha,
                     double *inv, double *weight_col, int *flag, void *stream) {
  if (ng < 0) return IPX_EINVAL;
  if (ng == 0) return IPX_OK;
  if (!rowp || !rowq || !pos_a || !pos_s || !val || !col || !alpha || !inv || !weight_col || !flag)
    return IPX_EINVAL;
  hipLaunchKernelGGL(k_pairs_factor, dim3((ng + IPX_BLOCK - 1) / IPX_BLOCK), dim3(IPX_BLOCK), 0,
                     (hipStream_t)stream, ng, rowp, rowq, pos_a, pos_s, val, alpha, inv, weight_col,
                     col, flag);
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

}  // extern "C"
