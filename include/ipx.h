/*
 * ipx.h -- C ABI of the MI355X-native trust-region subproblem kernels.
 *
 * Drop-in boundary for the hot path of antonior92/ip-nonlinear-solver
 * (SURVEY.md section 8(b)).  The reference is pure Python: its "FFI" for this
 * path is the duck-typed operator seam
 *
 *     H.dot(v), A.dot(v), A.T.dot(v)        qp_subproblem.py:376,379,410,503,634
 *     Z.dot(v), LS.dot(v), Y.dot(v)         projections.py:402-404
 *     np.dot / norm / elementwise algebra   qp_subproblem.py:99-232,502-634
 *
 * so the entry points below are what a ctypes binding of that seam binds
 * (INTEGRATION.md shows the stub).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (the Python host
 *     keeps them alive as torch tensors); the library allocates nothing the
 *     caller can see except opaque handles with an explicit *_destroy;
 *   - every call takes the hipStream_t to enqueue on (as void*), never
 *     synchronises the device, and returns 0 or a negative IPX_E* code;
 *   - fp64 values, int32 indices (scipy's defaults);
 *   - reductions are two-stage and fixed-order: stage 1 writes one partial
 *     per workgroup into a caller workspace, stage 2 sums them in index
 *     order, so results do not depend on scheduling.
 */
#ifndef IPX_H
#define IPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPX_OK 0
#define IPX_EINVAL (-1)    /* bad argument */
#define IPX_ELAUNCH (-2)   /* HIP launch / runtime error */
#define IPX_ENOTSPD (-3)   /* factorization met a non-positive pivot */
#define IPX_ENOMEM (-4)
#define IPX_EUNSUPPORTED (-5) /* this solver has no path for the matrix at hand: take another */
#define IPX_EILLCOND (-6)  /* ipx_banded_status: factorization complete, but a pivot lost 43 bits
                            * against its diagonal entry (numerically rank-deficient Jacobian) */

/* Number of doubles of reduction workspace any entry point may need. */
#define IPX_WS_DOUBLES 65536
/* Max partials a fused kernel writes per reduced quantity. */
#define IPX_MAX_PARTIALS 2048

const char *ipx_version(void);
/* Kernel launches of the library since it was loaded (a host counter: diagnostics, bench.py). */
long long ipx_launch_count(void);
/* Blocking reads of the library since it was loaded: every wait for device results -- ipx_read_doubles
 * / ipx_read_folded, and the chains of csrc/sqp.hip that hand over their block themselves. */
long long ipx_read_count(void);
/* Text of the last HIP error seen by this thread ("" if none). */
const char *ipx_last_error(void);
int ipx_device_info(int *cu_count, int *lds_bytes, char *arch, int arch_len);
/* Blocking read-back of k <= 512 doubles of device memory behind everything queued on `stream`
 * (pinned staging buffer inside the library; host_out: k doubles of the caller's). */
int ipx_read_doubles(const double *dev, int k, double *host_out, void *stream);
/* ... of nd <= IPX_FOLD_MAX scalars, each folded from a partial array (op: 0 sum, 1 max, 2 min) */
#define IPX_FOLD_MAX 32
typedef struct {
  const double *part;
  int32_t count;
  int32_t op;
} ipx_fold_desc;
int ipx_read_folded(int nd, const ipx_fold_desc *descs, double *host_out, void *stream);
/* ... and their weighted sum left on the device instead (no read): out[0] = ((w[0] r_0 + w[1] r_1)
 * + w[2] r_2) + ... -- an objective value assembled from dot products that only a later kernel
 * consumes (ipx_sqp_judge's f_next_dev; ipsolver.device.ScalarPack.combine). */
int ipx_fold_combine(int nd, const ipx_fold_desc *descs, const double *weights, double *out,
                     void *stream);

/* ---- vectors (np elementwise algebra; qp_subproblem.py:212-216,312,580,622,628)
 * out = a*x + b*y (y may be NULL when b == 0); in-place allowed. */
int ipx_axpby(int64_t n, double a, const double *x, double b, const double *y,
              double *out, void *stream);
/* out = x * y elementwise (diagonal scaling S.dot(d), tr_interior_point.py:111) */
int ipx_mul(int64_t n, const double *x, const double *y, double *out, void *stream);
int ipx_fill(int64_t n, double value, double *out, void *stream);
/* out = min(max(x, lb), ub): reinforce_box_boundaries, qp_subproblem.py:310-317 */
int ipx_clip(int64_t n, const double *x, const double *lb, const double *ub,
             double *out, void *stream);
/* out[i] = a*x[i] + b (scalar shift) */
int ipx_affine(int64_t n, double a, const double *x, double b, double *out, void *stream);

/* out[i] = sign[i] * (x[idx[i]] - shift[i]); sign / shift may be NULL.  Row
 * selection + sign flip of the canonical constraint form
 * (_canonical_constraint.py:240-248) and permutation of constraint-space vectors. */
int ipx_gather(int64_t n, const double *x, const int32_t *idx, const double *sign,
               const double *shift, double *out, void *stream);

/* out[idx[i]] = x[i] (refresh of the slack entries of the augmented Jacobian,
 * tr_interior_point.py:186-191). */
int ipx_scatter(int64_t n, const double *x, const int32_t *idx, double *out, void *stream);
/* out[idx[i]] += x[i], idx without repeats (no atomics). */
int ipx_scatter_add(int64_t n, const double *x, const int32_t *idx, double *out, void *stream);
/* Barrier elementwise ops (tr_interior_point.py:92-93,216-220,294):
 * out = max(x, c);  out = v > 0 ? a : c;  s[mask != 0] = -c[mask != 0]. */
int ipx_max_scalar(int64_t n, const double *x, double c, double *out, void *stream);
int ipx_where_positive(int64_t n, const double *v, const double *a, double c, double *out,
                       void *stream);
int ipx_assign_negated_where(int64_t n, double *s, const double *mask, const double *c,
                             void *stream);
/* out[0] = sum log(s_i) over s_i > 0, out[1] = #{s_i <= 0} (the reference's
 * log-barrier term is -inf when out[1] > 0). */
int ipx_sum_log(int64_t n, const double *s, double *out, double *ws, void *stream);

/* ---- reductions.  `out` is a device array; `ws` >= IPX_WS_DOUBLES doubles.
 * ipx_dot:      out[0] = sum x*y                      (np.dot)
 * ipx_norms:    out[0] = sum x^2, out[1] = max |x|    (norm(.), norm(., inf))
 * ipx_box_inside: out[0] = #{i: x<lb or x>ub}         (inside_box_boundaries :306-308)
 * ipx_box_sphere_reduce (box_intersections :198-216 + sphere_intersections :112-114):
 *   out[0]=d.d out[1]=z.d out[2]=z.z
 *   out[3]=max_i min(t_lb,t_ub) out[4]=min_i max(t_lb,t_ub) over d_i != 0
 *   out[5]=#{i: d_i==0 and (z_i<lb_i or z_i>ub_i)}   out[6]=#{i: d_i != 0}
 *   with z' = z0 + zs*z? no: z is used as given; d is scaled by `dscale`
 *   (the reference passes alpha*p, :585). lb/ub may be NULL (= -inf/+inf). */
int ipx_dot(int64_t n, const double *x, const double *y, double *out,
            double *ws, void *stream);
int ipx_norms(int64_t n, const double *x, double *out, double *ws, void *stream);
/* The first stage of ipx_dot / ipx_norms alone: ipx_reduce_grid(n) partials per quantity at
 * part[q * grid + workgroup] (norms: q = 0 the sum of squares, q = 1 max |x|), for results only
 * the host wants -- ipx_read_folded folds them inside the read-back, in the order of the
 * second launch of ipx_dot / ipx_norms (same bits), so such a reduction is ONE launch. */
int ipx_reduce_grid(int64_t n);
int ipx_dot_partials(int64_t n, const double *x, const double *y, double *part, void *stream);
int ipx_norms_partials(int64_t n, const double *x, double *part, void *stream);
int ipx_box_inside(int64_t n, const double *x, const double *lb, const double *ub,
                   double *out, double *ws, void *stream);
int ipx_box_sphere_reduce(int64_t n, const double *z, const double *d, double dscale,
                          const double *lb, const double *ub, double *out,
                          double *ws, void *stream);

/* ---- CSR SpMV (scipy _sparsetools csr_matvec; every A.dot / A.T.dot / H.dot).
 * Row tiles: `tiles` holds 2*(ntiles+1) entries -- ntiles+1 row indices (tile t
 * = rows [tiles[t], tiles[t+1])) followed by rowptr at those rows -- with at
 * most IPX_SPMV_TILE_NNZ nonzeros per tile unless a single row is longer;
 * build it with ipx_csr_tiles_host (cap >= 2*(nrows+2) always suffices).
 *
 * y_out = alpha * (A x)_i  [+ diag_i * x_i]  [+ beta * yin_i]
 * and, when red != NULL, red[0] = sum_i y_out_i^2, red[1] = sum_i xrow_i * y_out_i
 * (xrow = x when the matrix is square, used for p'Hp; pass square=0 otherwise).
 */
#define IPX_FOLD_WS_DOUBLES 16384
#define IPX_SPMV_TILE_NNZ 2048
#define IPX_SPMV_TILE_ROWS 1024   /* max rows per tile the fast path takes */
int ipx_csr_tiles_host(int64_t nrows, const int32_t *rowptr_host, int32_t tile_nnz,
                       int32_t max_rows, int32_t *tiles_out, int64_t cap);
int ipx_csr_spmv(int64_t nrows, int64_t ncols, const int32_t *rowptr,
                 const int32_t *colidx, const double *val,
                 const int32_t *tiles, int32_t ntiles,
                 const double *x, double alpha,
                 const double *diag, double beta, const double *yin,
                 double *yout, int square, double *red, double *ws, void *stream);

/* Extended SpMV for the row-sharded CG: explicit row vector (x may carry halo
 * entries), device stop flag, unfolded per-tile partials; and the fold. */
int ipx_csr_spmv_ex(int64_t nrows, int64_t ncols, const int32_t *rowptr, const int32_t *colidx,
                    const double *val, const int32_t *tiles, int32_t ntiles, const double *x,
                    double alpha, const double *diag, double beta, const double *yin,
                    double *yout, const double *xrow, double *partial, const double *guard,
                    void *stream);
int ipx_fold2(const double *partial, int32_t count, double *red, const double *guard,
              void *stream);

/* ---- dense Jacobian path (projections.py:175-233 QR; here Gram + Cholesky).
 * A is row-major with leading dimension lda.  ipx_dense_gemv mirrors
 * ipx_csr_spmv's fused epilogue.  G / X are M x M, M = ipx_dense_padded(m). */
int ipx_dense_gemv(int64_t m, int64_t n, const double *A, int64_t lda, const double *x,
                   double alpha, const double *diag, double beta, const double *yin,
                   double *yout, double *red, double *ws, void *stream);
int64_t ipx_dense_padded(int64_t m);
int ipx_gram_f64_mfma(int64_t m, int64_t n, const double *A, int64_t lda, double *G,
                      void *stream);
/* The same with the sum over the columns cut into `splits` parts (ipx_gram_splits picks the
 * count that evens the tiles out over the CUs; partial tiles in ws, ipx_gram_ws_doubles(m,
 * splits) doubles, added in a fixed order: deterministic). */
int ipx_gram_splits(int64_t m, int64_t n);
int64_t ipx_gram_ws_doubles(int64_t m, int32_t splits);
int ipx_gram_f64_mfma_split(int64_t m, int64_t n, const double *A, int64_t lda, double *G,
                            double *ws, int32_t splits, void *stream);
/* Same G from a CSR A (sparse Jacobian whose A A' is not narrow-banded). */
int ipx_aat_dense(int64_t m, const int32_t *rowptr, const int32_t *colidx, const double *val,
                  double *G, void *stream);
/* Blocked Cholesky G = L L' in place (64 x 64 tiles, the trailing updates on the fp64 matrix
 * cores) and X = G^-1 from it (in-place triangular inverse by recursive doubling in G's
 * storage -- G is consumed --, then X = L^-T L^-1 as MFMA tiles): what a dense NONLINEAR
 * constraint pays per accepted step (reference: a pivoted QR, projections.py:175-233).
 * M = ipx_dense_padded(m), a multiple of 64.
 * work: M + 1 doubles; work[M] <- min pivot / original diagonal (~1/cond(G)). */
int ipx_chol_factor(int64_t M, double *G, int *flag, double *work, void *stream);
int ipx_chol_inverse(int64_t M, double *G, double *X, void *stream);

/* ---- banded SPD solve with S = A A' (normal equations, projections.py:58-90;
 * replaces SuperLU solve :102,120 / CHOLMOD :62).  Partitioned (SPIKE-style)
 * LDL': see csrc/banded.hip.  Half bandwidth <= ipx_banded_kmax(). */
int ipx_banded_kmax(void);
void *ipx_banded_create(int64_t m, int32_t k, int32_t chunk);
void ipx_banded_destroy(void *handle);
int ipx_banded_levels(void *handle);
/* band[d*m+i] = S[i][i-d], d = 0..k; must outlive the solves. */
int ipx_banded_factor(void *handle, const double *band, void *stream);
/* Blocking: IPX_OK; IPX_ENOTSPD when a pivot was <= 0 (rank-deficient A); IPX_EILLCOND when
 * every pivot is positive but one lost 43 bits (solves are available; the Python host takes the
 * reference's SVD exit, projections.py:101-108, when the matrix is small enough for it). */
int ipx_banded_status(void *handle, void *stream);
/* After ipx_banded_status: 1 when the separator system was found diagonal to
 * working precision (|off-diagonal| <= 2^-56 |diagonal|) so solves skip the
 * middle kernel; ipx_banded_set_decoupling(h, 0) forces the full path. */
int ipx_banded_decoupled(void *handle);
int ipx_banded_set_decoupling(void *handle, int allow);
/* k = 1: level (1..7) at which the cyclic reduction of A A' has decoupled -- the single-launch
 * solve is then a parallel cyclic reduction over windows of rows; 0 = chunk recurrences.
 * ipx_banded_set_decoupling(h, 2) switches back to the chunk form (cross-checks);
 * (h, 16 + L) forces the reduction to stop at level L (tests: an inexact solve). */
int ipx_banded_pcr_level(void *handle);
/* After ipx_banded_status: correction steps per solve when the factorization runs defect
 * correction on the single-launch solve (separator blocks coupled and the separator level
 * long or -- half bandwidth 5..8 -- not compiled), else 0; *eta (may be NULL) = measured
 * contraction bound.  ipx_banded_status returns IPX_EUNSUPPORTED for half bandwidth 5..8
 * when the bound is >= 0.5: the caller takes another solver. */
int ipx_banded_refine_steps(void *handle, double *eta);
int ipx_banded_solve(void *handle, const double *w, double *x, void *stream);
/* Same, skipped on the device when *guard != 0 (stop flag of the CG loops). */
int ipx_banded_solve_guarded_c(void *handle, const double *w, double *x, const double *guard,
                               void *stream);
/* One-kernel-per-level variant of the same solve (cross-check / LDS fallback). */
/* After ipx_banded_status: 1 when solves take the single-launch decoupled path; then
 * out[0] = constraint rows per workgroup of that kernel, out[1] = its workgroups. */
int ipx_banded_decoupled_geometry(void *handle, int32_t *out);
/* solve + per-workgroup partials of ||w - (A A') x||^2 (ceil(m/256) doubles). */
int ipx_banded_solve_resid(void *handle, const double *w, double *x, double *partial,
                           int32_t *npartial, const double *guard, void *stream);
int ipx_banded_solve_multilaunch(void *handle, const double *w, double *x, void *stream);
/* band of (P A)(P A)' for CSR A with row order perm (NULL = identity). */
int ipx_aat_band(int64_t m, int32_t k, const int32_t *rowptr, const int32_t *colidx,
                 const double *val, const int32_t *perm, double *band, void *stream);

/* ---- device-resident projected CG (qp_subproblem.py:549-634): see csrc/cg.hip.
 * The argument block holds device pointers only; every member is 8 bytes. */
typedef struct ipx_cg_args {
  int64_t n, m;
  const int32_t *A_rowptr, *A_colidx; const double *A_val; const int32_t *A_tiles; int64_t A_ntiles;
  const int32_t *At_rowptr, *At_colidx; const double *At_val; const int32_t *At_tiles; int64_t At_ntiles;
  const int32_t *H_rowptr, *H_colidx; const double *H_val; const int32_t *H_tiles; int64_t H_ntiles;
  const double *H_diag;
  void *banded;
  double *x, *p, *r, *Hp;
  double *w, *v, *t;
  const double *lb, *ub;
  double *state;
  double *part1, *part2, *part3, *part4;
  int64_t vec_grid;
  int64_t solver_kind;   /* 0: `banded` is an ipx_banded handle; 1: an ipx_boxschur_args*;
                          * 2: dense Jacobian -- A_val / At_val are row-major m x n / n x m matrices
                          * (the index arrays NULL), `banded` is G^-1 (M x M doubles, M = m rounded up
                          * to 32; w and v M long, zero tail), H is CSR or, with H_rowptr NULL, a
                          * row-major n x n matrix in H_val; part1/3/4 hold 2 x 2048 doubles */
  /* step2 fused into the H.p SpMV (banded H): H_hmax > 0 = widest distance of a row tile's
   * columns from its own row range (<= 64, <= rows of every tile), pb = 2 x H_ntiles x
   * 2*H_hmax doubles of scratch (tile-boundary copies of p, by iteration parity).
   * pb == NULL or H_hmax == 0: separate step2 and H.p launches. */
  double *pb;
  int64_t H_hmax;
  int64_t H_tile_rows;   /* rows of H's longest row tile (0 = unknown) */
  /* step1 fused into the A.r SpMV (banded A, no box): A_own = 2*(A_ntiles+1) ints -- the
   * first column each row tile owns (A_ntiles+1 entries, a partition of [0, n)), then one
   * past the last column it touches; A_span = longest own-start-to-last-touched distance
   * (<= 2048); r_next = n doubles of scratch.  Any of them 0 / NULL, or lb given: separate
   * launches. */
  double *r_next;
  const int32_t *A_own;
  int64_t A_span;
  /* IPX_FOLD_WS_DOUBLES doubles of scratch: partial-sum arrays longer than ~1000 entries
   * (n beyond ~1e6) are compacted into it before the consumers fold them; NULL = never. */
  double *fold_ws;
  /* g = r - A'v as the tail of the single-launch decoupled banded solve (tridiagonal A A'):
   * At_vown = workgroups+1 ints, the variables each workgroup of that kernel owns (those
   * whose first constraint lies in its rows; see ipx_banded_decoupled_geometry); At_qv =
   * ceil(most variables of one workgroup / 256) <= 16.  NULL / 0: separate SpMV. */
  const int32_t *At_vown;
  int64_t At_qv;
  int64_t A_tile_nnz;    /* must be 0 (a row-tile table of its own for the fused step1 was an
                          * experiment of round 2; the field keeps the layout) */
  /* the rows of A' once more in ELL(2) form for that tail, indexed by the variable alone (no
   * row-pointer round trip): At_ell_val = 2n doubles, entry t of variable j at [t*n + j], t = 0
   * its FIRST constraint, t = 1 the next row (tridiagonal A A': a variable sees at most two
   * constraints and they are adjacent), an absent entry 0; At_ell_row = n uint16 (n even), the
   * first constraint's row as an offset from the first row of the workgroup that owns the
   * variable (At_vown; < rows per workgroup).  18 bytes per variable.  NULL: CSR. */
  const uint16_t *At_ell_row;
  const double *At_ell_val;
  /* Compact index form of H for the fused step2 + H.p kernel (H_hmax > 0), or NULL:
   * H_col16 = one uint16 per nonzero, its column as an offset into the row tile's span
   * (col - max(tile's first row - H_hmax, 0)); H_rowlen = one int per row tile, the common
   * length of its rows or -1 (then H_rowptr is read for that tile).  Same arithmetic; 2 B
   * instead of 4 per nonzero and no row pointers on uniform tiles. */
  const void *H_col16;
  const int32_t *H_rowlen;
  /* The same for the fused step1 + A.r kernel (A_span > 0, standard tiles):
   * one uint16 per nonzero of A, col - A_own[its row tile]; NULL: A_colidx is read. */
  const void *A_col16;
  /* != 0: the trust radius in the state block is +inf (and, fused step1 implying no box, the
   * tests of qp_subproblem.py:583,599 can never trigger): ||x + alpha p||^2 is not formed --
   * the fused step1 + A.r kernel then does not read x and p.  Only read when step1 is fused. */
  int64_t no_radius;
  /* Tables of the RESIDENT form of the loop (csrc/resident.hip: a whole batch of iterations in
   * one launch, one workgroup per 260 rows of a tridiagonal A A', all of them co-resident; the
   * per-rank sizes of a multi-GPU run).  For a Jacobian whose rows all have A_rl entries, no
   * box: A_off16 = one uint16 per entry of A (column - first column of its row), A_rowfirst =
   * one int per row, P_win = 2 ints per workgroup of the solve (first column / one past the
   * last column of the span it needs: the columns of its window's rows and its own variables),
   * P_nspan = the longest span (<= 4096), P_navn below; needs At_vown. */
  const void *A_off16;
  const int32_t *A_rowfirst;
  int64_t A_rl;
  const int32_t *P_win;
  int64_t P_nspan;
  int64_t P_navn;        /* most own variables of one workgroup (At_vown differences) */
  /* != 0: the Hessian is an operator the CALLER applies between two iterations (reference
   * _canonical_constraint.py:119-139 allows LinearOperator terms: finite differences, user
   * callbacks): the loop's launches leave out H.p; after every iteration the caller writes
   * Hp = H p and the scalar p'Hp into part1[1] (H_ntiles = 1, the H_* arrays unused).  The
   * branches and step lengths stay on the device. */
  int64_t H_operator;
  /* != 0 with the tables above: ipx_cg_iterate runs a batch as one resident launch when
   * ipx_cg_resident_ok says the sizes fit (every row of H at most 4 entries -- the caller's
   * check --, <= 224 workgroups, spans within the kernel's budgets).  R_ll = the hand-off buffer
   * (ipx_cg_resident_ll_words(workgroups, R_hw) 8-byte words, zeroed once), R_hw = the longest
   * halo (columns of a workgroup's span beyond its own variables, either side), R_seq = a HOST
   * counter owned by the caller (starts at 0; the library advances it by the tags a launch
   * uses).  A launch in which a hand-off timed out records stop code 8 and leaves x, p, r, Hp
   * and the state block untouched: clear the code, set resident = 0 and repeat the batch. */
  int64_t resident;
  void *R_ll;
  int64_t R_hw;
  int64_t *R_seq;
} ipx_cg_args;
int ipx_cg_resident_ok(const ipx_cg_args *a);
int64_t ipx_cg_resident_ll_words(int32_t nwg, int32_t hw);
/* the kernel's budgets for the host code that builds its tables: workgroups per launch, threads,
 * span columns / own variables / window rows per workgroup, entries per row of A / of H, halo
 * entries; and the workgroups of all ranks of a sharded launch together */
void ipx_cg_resident_limits(int32_t *out8);
int32_t ipx_cg_resident_max_global(void);
int ipx_cg_state_size(void);
int ipx_cg_vec_grid(int64_t n);
/* Hp = H p (+ diag*p) with p'Hp partials: primes the loop. */
int ipx_cg_hp(const ipx_cg_args *a, void *stream);
/* Enqueue iterations [it_begin, it_end); never synchronises. */
int ipx_cg_iterate(const ipx_cg_args *a, int32_t it_begin, int32_t it_end, void *stream);
/* The state block of a new call from reductions left in device memory (red[idx7[k]], idx < 0:
 * zero): ||x0||^2; ||t||^2, ||r0||^2, ||A r0||^2; ||r0||^2, ||g0||^2, ||A g0||^2 -- rt_g, the
 * tolerance (tol_in, or the rule of qp_subproblem.py:529-530 when tol_in is NaN), radius and
 * orthogonality threshold are written by a one-thread kernel, stop code 9 when a projection
 * needs the host (refinement, cancellation step) or the start is at the trust-region boundary:
 * no host read between a call's priming and its first batch. */
/* The whole priming of a call (qp_subproblem.py:502-512: x0 = Y(-b), r0 = Z(H x0 + c), g0 = Z r0,
 * the state block, p = -g0, Hp = H p) enqueued by ONE call into the loop's own buffers; CSR A
 * (A_tiles / A_ntiles: its standard SpMV row tiles) and H, solver_kind 0 or 1; b NULL = 0;
 * red: 18 doubles, ws: IPX_WS_DOUBLES doubles of device memory.  first_end > 0: iterations
 * [0, first_end) are enqueued behind the priming by the same call (as ipx_cg_iterate would).
 * steps != 0 (b == NULL only): the two projections may each take ONE correction step on the
 * device -- the refinement of projections.py:72-78 / the cancellation step of
 * ipsolver/projector.py, decided from the same norms by a kernel in between (red[14 + j] = 0
 * when projection j takes it, red[16 + j] = 1 once it has; state block: ST_PRIME_STEPS); six more
 * launches, no-ops when none is due.  steps == 0: a projection that needs one ends the priming
 * with stop code 9.  With a tridiagonal A A' on the single-launch solve a projection is the
 * product A x + that solve with z = x - A'v, ||z||^2 and ||A z||^2 (as the residual
 * ||A x - (A A') v||^2: the loop's own measure) from its tail -- two launches.
 * Stop code 9 in the state block afterwards: the host must prime (ipx_cg_prime_state); the
 * iterations enqueued with it did nothing. */
/* doubles of reduction workspace ipx_cg_prime needs for this argument block (the per-tile
 * partials of its six products wait there for one fold); more than IPX_WS_DOUBLES: the entry
 * point returns IPX_EUNSUPPORTED and the caller primes launch by launch. */
int64_t ipx_cg_prime_ws_doubles(const ipx_cg_args *a, int32_t A_ntiles);
int ipx_cg_prime(const ipx_cg_args *a, const int32_t *A_tiles, int32_t A_ntiles, const double *c,
                 const double *b, double *red, double *ws, double tol_in, double radius,
                 double orth_tol, double norm_A, double cancellation, int32_t first_end,
                 int32_t steps, void *stream);
int ipx_cg_prime_state(double *state, const double *red, const int32_t *idx7, double tol_in,
                       double radius, double orth_tol, double norm_A, double cancellation,
                       void *stream);
/* Same launches with HIP events around each kernel class; synchronises once at
 * the end and returns per-class totals in ms_out[0..6] = {step1, A r, banded,
 * r-A'v, A g, step2, H p}.  For per-kernel attribution in bench.py. */
int ipx_cg_iterate_timed(const ipx_cg_args *a, int32_t it_begin, int32_t it_end,
                         float *ms_out, void *stream);
/* The two vector kernels on their own (explicit partial buffers / counts). */
int ipx_cg_step1(int64_t n, double *state, int32_t it, const double *p1, int32_t np1,
                 const double *x, const double *p, double *r, const double *Hp, const double *lb,
                 const double *ub, double *part2, int32_t grid, void *stream);
int ipx_cg_step2(int64_t n, double *state, int32_t it, int32_t mode, const double *p2,
                 int32_t np2, const double *p3, int32_t np3, const double *p4, int32_t np4,
                 double *x, double *p, const double *g, int32_t grid, void *stream);
/* ---- matrix-free (A A')^-1 w: Jacobi-preconditioned CG on A (A' v) = w, device resident
 * (csrc/pcg.hip; replaces the sparse LU of projections.py:93-172 for Jacobians beyond both
 * device factorizations).  State block (doubles): [0],[1] r'z by iteration parity, [2],[3]
 * smallest ||r||, [4],[5] stall counter, [6] done (0 running, 1 converged, 2 fp64 floor,
 * 3 not positive definite), [7] iterations, [8] ||w||, [9] rtol, [10] last ||r||. */
typedef struct ipx_pcg_args {
  int64_t m, n;
  const int32_t *A_rowptr, *A_colidx; const double *A_val; const int32_t *A_tiles; int64_t A_ntiles;
  const int32_t *At_rowptr, *At_colidx; const double *At_val; const int32_t *At_tiles; int64_t At_ntiles;
  const double *dinv;       /* 1 / diag(A A') */
  double *v, *r, *p, *Sp;   /* m-vectors */
  double *t;                /* n-vector: A' p */
  double *state;            /* ipx_pcg_state_size() doubles */
  double *part1, *part2;    /* 2 * A_ntiles and 2 * grid doubles */
  int64_t grid;             /* ipx_cg_vec_grid(m) */
  /* block-Jacobi preconditioner (ipx_blockjacobi_build) or NULL / 0 for the diagonal one:
   * binv = nblk x 32 x 32 inverse blocks, border = 32 nblk row indices (the rows of block b;
   * -1 = padding), z = m doubles (M^-1 r), part3 = nblk / 8 + 1 doubles; dinv is then all zero. */
  const double *binv; const int32_t *border; int64_t nblk;
  double *z, *part3;
} ipx_pcg_args;
int ipx_pcg_state_size(void);
int ipx_pcg_iterate(const ipx_pcg_args *a, int32_t it_begin, int32_t it_end, void *stream);
int ipx_blockjacobi_build(int64_t nblk, const int32_t *rowptr, const int32_t *colidx,
                          const double *val, const int32_t *order, double *binv, int *flag,
                          void *stream);
int ipx_blockjacobi_apply(int64_t m, int64_t nblk, const int32_t *order, const double *binv,
                          const double *r, double *z, double *ws, const double *state,
                          void *stream);
/* ---- partitioned row-sharded loop (ipsolver/sharded.py FusedShardedCG; replaces the
 * per-iteration body of qp_subproblem.py:549-634 on one rank of a node).  `a` describes the
 * rank's extended local problem (own rows / variables + halo copies; solver_kind 0 or 1); the
 * scalars travel through two all-reduced device buffers.  Ranges are in units of the partial
 * arrays' entries (row tiles / solve workgroups) and select the rank's OWN part. */
typedef struct ipx_shard2_ext {
  double *s1;                    /* [2]  s1[1] = p'Hp (own sum, then all-reduced) */
  double *pack;                  /* [4]  ||x+ap||^2, #violations, ||g||^2, ||A g||^2 */
  int64_t nseg;                  /* segments of the local vector space: 1 (x-space) .. 4 (the
                                  * barrier problem's z = [x; s_nl; s_lb; s_ub]), each laid out
                                  * [left halo | own | right halo] */
  int64_t own_lo[4], own_hi[4];  /* own elements per segment, local indices (step1's reductions) */
  int64_t p1_lo[4], p1_hi[4];    /* own row tiles of H per segment                 (part1) */
  int64_t p3_lo[4], p3_hi[4];    /* own entries of part3 (||g||^2 partials) per segment; with
                                  * g = r - A'v fused into the banded solve: its own workgroups,
                                  * segment 0 only */
  int64_t p2_lo, p2_hi;          /* own entries of part2 (row tiles of A with the fused step1;
                                  * every vector chunk otherwise: step1 masks by element) */
  int64_t p4_lo, p4_hi;          /* own workgroups of the (inner) banded solve (part4) */
  /* Peer mailbox (ipx_peer_create) or NULL.  With it the scalars are all-reduced and the halo
   * of g exchanged INSIDE the loop's own launches (the kernels write into the peers' HBM over
   * xGMI); without it the caller all-reduces s1 / pack and exchanges the halo between the
   * phases itself. */
  void *peer;
  int64_t seg_lo[4], seg_hi[4];  /* the segments' local extents [left halo | own | right halo] */
  int64_t send_left[4], send_right[4];   /* own entries the left / right neighbour keeps as halo */
  /* != 0: ipx_cg_shard2_iterate does the two all-reduces and the halo exchange in the PROLOGUES
   * of the kernels that consume them -- 3 launches per iteration instead of 5; the reduced
   * sums are also left in pack.  The GROUP's decision: set it only when ipx_cg_shard2_fusable
   * returned 1 on EVERY rank (the two forms order an iteration's collectives differently);
   * a rank whose own argument block does not allow it gets IPX_EINVAL. */
  int64_t fuse_comm;
  /* RESIDENT form of the sharded loop (csrc/resident.hip, PEER; ipx_cg_shard2_resident): this
   * rank's own blocks [res_wg0, res_wg0 + res_nwg) of its local banded solve are the global
   * workgroups res_gwg0 .. of res_gnwg (all ranks' own blocks in rank order).  Needs the
   * resident tables of the argument block and ipx_peer_attach_resident on e->peer; the GROUP's
   * decision, like fuse_comm. */
  int64_t res_wg0, res_nwg, res_gwg0, res_gnwg;
} ipx_shard2_ext;
/* 1 when this rank's argument block allows fuse_comm (peer set, the fused 16-bit-index kernels
 * and g = r - A'v as the solve's tail for x-space problems, or the box-Schur projection). */
int ipx_cg_shard2_fusable(const ipx_cg_args *a, const ipx_shard2_ext *e);
int ipx_cg_shard2_segment(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t phase,
                          int32_t it, int32_t mode, void *stream);
/* Iterations [it_begin, it_end), both phases, communication included (needs e->peer): one
 * call per batch, nothing between the iterations on the host. */
int ipx_cg_shard2_iterate(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t it_begin,
                          int32_t it_end, void *stream);
/* The same batch as ONE resident launch per rank (reference loop: qp_subproblem.py:549-634; the
 * workgroups of all ranks hand their scalars and halos to each other directly, two hops per
 * iteration).  Writes a rank's OWN entries of x, p, r, Hp only: before the host uses the local
 * vectors (an event, the end of the loop, a batch on the separate launches) it synchronises
 * their halos and calls ipx_cg_save_pb.  A wait that times out records stop code 7. */
int ipx_cg_shard2_resident_ok(const ipx_cg_args *a, const ipx_shard2_ext *e);
int ipx_cg_shard2_resident(const ipx_cg_args *a, const ipx_shard2_ext *e, int32_t it_begin,
                           int32_t it_end, void *stream);
/* p at the row-tile boundaries of H for the fused step2 + H.p kernel, from a->p (after p
 * changed behind the loop's back: a halo synchronisation). */
int ipx_cg_save_pb(const ipx_cg_args *a, void *stream);

/* ---- peer mailboxes (csrc/peer.hip): the transport of the sharded loop's two all-reduces
 * (torch.distributed all_reduce in round 2; qp_subproblem.py:556,583,626 are the reduction
 * points) and of its halo exchange, rank to rank through hipIpc-mapped device memory.
 * create -> export (ipx_peer_handle_bytes() bytes) -> hand the blobs around -> import every
 * other rank's -> ipx_peer_ready.  halo_cap: doubles per side a rank may receive. */
int ipx_peer_handle_bytes(void);
void *ipx_peer_create(int32_t rank, int32_t world, int64_t halo_cap);
int ipx_peer_export(void *peer, void *handle_out);
int ipx_peer_import(void *peer, int32_t rank, const void *handle_in);
int ipx_peer_ready(void *peer);
int64_t ipx_peer_halo_capacity(void *peer);
/* Deadline of a kernel's wait for a peer's word (default 10 s); past it the loop records stop
 * code 7 and returns -- never a hung GPU. */
int ipx_peer_set_timeout(void *peer, double seconds);
int ipx_peer_sequence(void *peer, int64_t *out2);
int64_t ipx_peer_fused_launches(void *peer);   /* loop kernels that did a collective in their prologue */
void ipx_peer_destroy(void *peer);
/* `reps` all-reduces (sum) of nq <= 8 doubles back to back: the mailbox path's latency probe. */
int ipx_peer_allreduce(void *peer, int32_t nq, const double *in, double *out, int *failed,
                       int32_t reps, void *stream);
/* The outer loops' collectives through the mailboxes (no torch.distributed call, no upload):
 * all-gather of nq <= 7 scalars per rank, the caller's own in HOST memory (passed to the kernel
 * by value) -> out (device): world * nq doubles in rank order + 2 (an earlier halo exchange
 * timed out on some rank / this all-gather did);
 * halo exchange of nseg <= 4 segments of a local vector (geom: seg_lo, own_lo, own_hi, seg_hi,
 * send_left, send_right per segment, local indices).  Both collective (every rank, same
 * arguments' shapes), neither synchronises; failed: device int, set when a wait timed out. */
int ipx_peer_allgather(void *peer, int32_t nq, const double *vals, double *out, int *failed,
                       void *stream);
int ipx_peer_exchange(void *peer, double *v, int32_t nseg, const int64_t *geom, int *failed,
                      void *stream);
/* `reps` round trips of one tagged word with `partner` (-1: sit the round out; every rank of
 * the group calls it once per round, the sequence numbers advance alike): ticks2[0] = 100 MHz
 * wall-clock ticks of the exchange, ticks2[1] = 1 when a wait timed out (device int64[2]).
 * The cost of one cross-GPU hand-off, measured before a multi-GPU run (bench.py preflight). */
int ipx_peer_pingpong(void *peer, int32_t partner, int32_t reps, long long *ticks2, void *stream);
/* Hand-off buffers of the resident loop kernel between the ranks: attach (allocates `words`
 * 8-byte words -- ipx_cg_resident_ll_words of the largest launch of the group -- uncached,
 * zeroed) -> export_resident -> hand the blobs around -> import_resident every other rank's
 * -> resident_ready (builds the device-side pointer table; 1 when complete). */
int ipx_peer_attach_resident(void *peer, int64_t words);
int ipx_peer_export_resident(void *peer, void *handle_out);
int ipx_peer_import_resident(void *peer, int32_t rank, const void *handle_in);
int ipx_peer_resident_ready(void *peer);
int64_t ipx_peer_resident_launches(void *peer);
int ipx_cg_shard2_fold_hp(const ipx_cg_args *a, const ipx_shard2_ext *e, void *stream);
/* The fused step2 + H.p launch alone (needs pb / H_hmax in the argument block). */
int ipx_cg_step2_hp(const ipx_cg_args *a, int32_t it, int32_t mode, void *stream);
/* Finish iteration `it` after the host handled a stop-5/6 event. */
int ipx_cg_resume(const ipx_cg_args *a, int32_t it, int32_t mode, void *stream);

/* band of (P A) diag(wcol) (P A)' (column weights; wcol NULL = ones). */
int ipx_aat_band_w(int64_t m, int32_t k, const int32_t *rowptr, const int32_t *colidx,
                   const double *val, const int32_t *perm, const double *wcol, double *band,
                   void *stream);

/* ---- analytic elimination of box-like rows of A A' (csrc/boxschur.hip): groups
 * of one or two rows that touch the same shared column (plus a private entry
 * each).  factor: 1x1 / 2x2 inverses, the rows' shared-column entries (alpha)
 * and the Schur column weights 1 - alpha'B^-1 alpha; tsolve: t = B^-1 w on those
 * rows and u[col] = alpha't; vsolve: v = t - B^-1 (alpha * y[col]).  grp (may be NULL):
 * 4 doubles per group, the rows' entries (a_p, s_p, a_q, s_q) for ipx_boxschur_project.
 * grp2 (may be NULL): 2 doubles per group, (s_p, s_q) carrying the signs of (a_p, a_q) -- the
 * same table in half the bytes when every a is +-1 and no s is negative (box rows); bit 1 of
 * *flag is set when that does not hold (bit 0: a block is not positive definite). */
int ipx_pairs_factor(int32_t ng, const int32_t *rowp, const int32_t *rowq, const int32_t *pos_a,
                     const int32_t *pos_s, const double *val, const int32_t *col, double *alpha,
                     double *inv, double *weight_col, int *flag, double *grp, double *grp2,
                     void *stream);
int ipx_pairs_tsolve(int32_t ng, const int32_t *rowp, const int32_t *rowq, const double *inv,
                     const double *alpha, const double *w, double *t, const int32_t *col,
                     double *u, void *stream);
int ipx_pairs_vsolve(int32_t ng, const int32_t *rowp, const int32_t *rowq, const double *inv,
                     const double *alpha, const double *t, const double *y, const int32_t *col,
                     double *v, void *stream);

/* Prepared argument block of one box-Schur (A A')^-1 (all members 8 bytes). */
typedef struct ipx_boxschur_args {
  int64_t m, n, ng, mR;
  const int32_t *rowp, *rowq, *col, *general;
  const double *inv, *alpha;
  const int32_t *AR_rowptr, *AR_colidx; const double *AR_val; const int32_t *AR_tiles; int64_t AR_ntiles;
  const int32_t *ARt_rowptr, *ARt_colidx; const double *ARt_val; const int32_t *ARt_tiles; int64_t ARt_ntiles;
  void *inner;                       /* ipx_banded handle of the Schur complement */
  double *t, *u, *wR, *rhs, *vR, *y; /* scratch: m, n, mR, mR, mR, n doubles */
  /* group tables for ipx_boxschur_project (NULL: only ipx_boxschur_solve is available):
   * gcol = 3 ints per group (shared column, private column of row p, of row q; -1: the row
   * has none, -2 in the q slot: single-row group),
   * grp = 4 doubles per group (a_p, s_p, a_q, s_q; written by ipx_pairs_factor),
   * gen_cols = the ngen columns that belong to no group, ny = 1 + last column A_R touches,
   * up = n doubles of scratch (zero beyond ny). */
  const int32_t *gcol; const double *grp; const int32_t *gen_cols; int64_t ngen;
  int64_t ny;
  double *up;
  /* optional compact tables for the CG loop's two group kernels (NULL: the forms above):
   * grp2 = ipx_pairs_factor's compact group table (only when its flag bit 1 stayed clear);
   * yell_col / yell_val = the columns of A_R once more, per ITEM of the projection (the ng
   * groups by their shared column, then the ngen other columns) in ELL(2) form: entry t of
   * item i at [t * (ng + ngen) + i], an absent entry = a valid row with value 0.  Indexed by
   * the item alone, the loads need no column -> row pointer -> entries round trip.  Only when
   * no such column holds more than two entries. */
  const double *grp2;
  const int32_t *yell_col; const double *yell_val;
  /* gaffine != 0 (with grp2): group g has the columns (gc0 + g, gc0 + g + gdp, gc0 + g + gdq)
   * and gen_cols[k] = gen0 + k -- every variable bounded on both sides, the usual
   * BoxConstraint; the two kernels compute the columns instead of reading gcol / gen_cols. */
  int64_t gaffine, gc0, gdp, gdq, gen0;
  /* AR_rowlen = 2, 4, 8 or 16 when EVERY row of A_R has that many entries (entry j of row i at
   * AR_colidx / AR_val [i * AR_rowlen + j]), else 0: ipx_boxschur_project then forms A_R u
   * inside the Schur solve's kernel (cyclic-reduction path) instead of by an SpMV launch. */
  int64_t AR_rowlen;
  /* optional (with AR_rowlen, grp2, gaffine, yell_*): ipx_boxschur_project does the per-item
   * back substitution as the TAIL of the Schur solve's kernel instead of in a launch of its own
   * (4 -> 3 launches per CG iteration of the barrier problem).  post_own_g / post_own_e =
   * (workgroups of the solve + 1) ints each: the range of groups / of other columns whose
   * FIRST general row lies in a workgroup's post_rows_wg rows; post_reach = the largest
   * distance from an item's first general row to its second.  NULL: the separate launch. */
  const int32_t *post_own_g, *post_own_e;
  int64_t post_rows_wg, post_reach;
} ipx_boxschur_args;
/* v = (A A')^-1 w; partial (optional, ceil(mR/256) doubles) receives the residual partials. */
int ipx_boxschur_solve(const ipx_boxschur_args *a, const double *w, double *v, double *partial,
                       int32_t *npartial, const double *guard, void *stream);
/* g = r - A'(A A')^-1 A r (the CG loop's projection; g may alias r) with the simple rows
 * handled per group on r itself instead of as matrix rows: 6 launches and ~1/3 of the traffic
 * of  A r -> ipx_boxschur_solve -> r - A'v.  part_g: 2 x ipx_boxschur_project_count(a) doubles
 * (||g||^2 partials; second half zero), part_res / npart_res as ipx_boxschur_solve's. */
int ipx_boxschur_project_count(const ipx_boxschur_args *a);
int ipx_boxschur_project(const ipx_boxschur_args *a, const double *r, double *g, double *part_g,
                         int32_t *npart_g, double *part_res, int32_t *npart_res,
                         const double *guard, void *stream);

/* Deferred form of ipx_banded_status (no blocking read): the verdict of the previous, clean
 * factorization on this handle is ASSUMED for the one just enqueued; a one-thread kernel writes
 * verdict[0] = 0 when the flags agree, else 1 (then: ipx_banded_status, and repeat the solves).
 * IPX_EUNSUPPORTED: nothing to assume on this handle. */
int ipx_banded_status_deferred(void *handle, double *verdict, void *stream);
/* A numeric refresh in THREE launches on a handle that qualifies for the deferred verdict
 * (tridiagonal A A', rows in their own order): the band of A A' (wcol: column weights or NULL),
 * the cyclic reduction's check of the matrix alone, the verdict kernel.  The chunked LDL' that the
 * cyclic-reduction solves never read is left out; a solve that needs it runs it first, with
 * ipx_banded_status' blocking verdict.  Returns 1: done, read `verdict` as above; 0: the handle
 * does not qualify and NOTHING was enqueued (ipx_aat_band_w + ipx_banded_factor +
 * ipx_banded_status); < 0: error. */
int ipx_banded_refactor(void *handle, int64_t m, int32_t k, const int32_t *rowptr,
                        const int32_t *colidx, const double *val, const double *wcol,
                        double *band, double *verdict, void *stream);

/* ---- one outer iteration of the trust-region SQP method as three chains of launches whose
 * decisions are taken on the device (csrc/sqp.hip; reference equality_constrained_sqp.py:102-250,
 * the Newton point of modified_dogleg qp_subproblem.py:366-373, projected_cg :416-643).
 * Every chain ends in a reduction whose last workgroup to arrive folds the partial sums in their
 * fixed order, decides, and writes the scalar block q (ipx_sqp_block_size() doubles, layout in
 * csrc/sqp.hip SQ_*, mirrored by ipsolver/sqp_chain.py); the host reads the block once per chain
 * (host_block: the kernel hands it over itself).  CSR Jacobian and Hessian, banded or box-Schur solver (solver_kind 0 / 1
 * of the CG loop's argument block), constraint rows in the projector's own order. */
typedef struct ipx_sqp_args {
  int64_t n, m;
  const ipx_cg_args *cg;            /* the CG loop's block: A, A', H, the solver, its buffers;
                                     * cg->lb / cg->ub must be lbt / ubt below (or NULL) */
  const int32_t *A_tiles;           /* standard SpMV row tiles of A */
  int64_t A_ntiles;
  double *q;                        /* the block */
  const double *x, *c, *b;          /* iterate (n), gradient (n), constraint value (m) */
  const double *lb, *ub;            /* trust_lb / trust_ub of the step (n) or NULL */
  const double *scale;              /* diagonal of S (n) or NULL */
  double *dn, *ct, *lbt, *ubt, *d, *Hd, *x_next;   /* n each (lbt / ubt NULL with lb / ub) */
  double *Ad;                       /* m */
  double *v_out;                    /* m: the multipliers ipx_sqp_refresh writes */
  double *part;                     /* ipx_sqp_part_doubles() doubles of partial sums */
  double *red, *ws;                 /* 16 doubles / IPX_WS_DOUBLES of scratch (ipx_cg_prime) */
  double orth_tol, cancellation;    /* of the projections (ipx_cg_prime) */
  const double *verdict;            /* device: ipx_banded_status_deferred's word, or NULL */
  const double *A_norm_part;        /* ipx_norms_partials over A's values (ipx_sqp_refresh folds
                                     * them into the block's ||A||_F^2), or NULL */
  int64_t A_norm_grid;
  double *host_block;               /* HOST memory, ipx_sqp_block_size() doubles, or NULL.  Non-NULL:
                                     * ipx_sqp_front / _model / _judge / _refresh return when the
                                     * block has arrived there -- the workgroup that completes the
                                     * block publishes it (no read-back launch behind the chain).
                                     * NULL: the entry returns behind its last launch; the block
                                     * stays in q -- a later chain's block carries all of it (the
                                     * caller enqueues the user's callbacks and ipx_sqp_judge
                                     * behind ipx_sqp_front and reads once) */
} ipx_sqp_args;
int ipx_sqp_block_size(void);
int64_t ipx_sqp_part_doubles(const ipx_sqp_args *s);
/* normal step (have_dn == 0: the Newton point, accepted on the device -- block entry
 * NORMAL_KIND 1 -- or not; then, with_dogleg != 0, the dogleg proper of qp_subproblem.py:375-413
 * behind it on the device -- NORMAL_KIND 2 --, else 0: the caller computes the dogleg step into
 * s->dn and calls again with have_dn = 1), c_t = H dn + c, shifted bounds, the projected CG's
 * priming with the tangential radius of the block and iterations [0, first_end), then
 * ipx_sqp_model.  (with_dogleg costs nine launches that do nothing when the Newton point stands:
 * the caller sets it when the last normal step was not the Newton point.)  with_steps != 0: the
 * priming's two projections may each take one correction step on the device (eight more
 * launches, no-ops when none is due; without them such a priming ends in stop code 9). */
int ipx_sqp_front(const ipx_sqp_args *s, int have_dn, int with_dogleg, int with_steps,
                  double radius, double penalty, double f, double norm_b, double tr_factor,
                  double box_factor, double tol_in, double norm_A, int32_t first_end,
                  void *stream);
/* the CG loop's trust-region / negative-curvature exits (:565-576, :585-596), d = dn + dt,
 * x_next = x + S d, the five sums and the model / penalty / predicted reduction (:135-153);
 * host_cg != 0: the caller finished the CG loop itself (exits included), dt = cg->x as it is */
int ipx_sqp_model(const ipx_sqp_args *s, double penalty, double f, double norm_b, int host_cg,
                  void *stream);
/* ||b_next||, actual / predicted, second-order-correction test, trust-radius ladder, accept
 * (:156-242); f_next by value or, f_next_dev non-NULL, a device scalar */
int ipx_sqp_judge(const ipx_sqp_args *s, const double *b_next, double f_next,
                  const double *f_next_dev, void *stream);
/* v = -(A A')^-1 A c, ||c + A'v||_inf, ||b||_inf, ||b|| (:83-87, 226-239), ||A||_F^2 */
int ipx_sqp_refresh(const ipx_sqp_args *s, void *stream);
/* measurement aid: on != 0 starts bracketing the projected CG's priming + first batch inside
 * ipx_sqp_front with HIP events; on == 0 stops, synchronises and returns their GPU time (ms) and
 * count since the start (the in-solve projected-CG rate of the benchmark) */
int ipx_sqp_cg_timing(int on, double *ms_total, int *calls);
/* the decisions' scalar arithmetic on a HOST copy of the block (the same code the kernels run) */
void ipx_sqp_model_host(double *q);
void ipx_sqp_ratio_host(double *q);
void ipx_sqp_radius_host(double *q);
/* box_sphere_intersections' scalar tail (qp_subproblem.py:99-149,194-234,286-296) from the seven
 * sums of ipx_box_sphere_reduce: out3 = (ta, tb, intersect) */
void ipx_sqp_box_sphere_host(const double *sums7, double radius, int entire_line, double *out3);

#ifdef __cplusplus
}
#endif
#endif /* IPX_H */
