/*
 * CPU oracle for the GraphVAE graph-matching block.  TEST INFRASTRUCTURE ONLY:
 * it is linked/loaded by tests/, by __graft_entry__.smoke() and by bench.py's
 * cpu_baseline leg as the checker; the product path never calls it.
 *
 * It restates, in plain C with every floating-point reduction order written out,
 * what the reference computes on its CPU path:
 *   - edge-similarity tensor      reference graphvae/models/model.py:171-216
 *   - max-pool matching (MPM)     reference graphvae/models/model.py:247-276
 *   - assignment                  reference graphvae/models/model.py:319-323, which calls
 *                                 scipy.optimize.linear_sum_assignment (SciPy 1.15.3,
 *                                 third-party, not under the reference tree; its published
 *                                 shortest-augmenting-path algorithm is restated below)
 * The reference's tensor expressions run through PyTorch's CPU reduction kernels
 * (third-party as well); the three summation orders those kernels use on fp32
 * (`sum(dim=1)` of a contiguous (R,R,R) tensor, `sum(1)` of a contiguous (R,R)
 * matrix, and the 2-norm of a contiguous matrix) are restated here so that the
 * bits of X, and therefore the permutation, do not depend on the machine.
 *
 * Pinned by: tests/golden/g1_forward_test.npz (reference model.py:447-463 known
 * answer), g2_matching.npz, g4_lsap.npz, g7_reduction_orders.npz, all produced by
 * tools/make_golden.py from the imported reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: the only fused multiply-adds are the explicit fmaf()
 * calls in ora_norm2().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORA_LEVELS 4

/* ---- cascade summation: one column of PyTorch's multi_row_sum ------------------
 * Sums x[0], x[stride], ... (n terms).  Terms are accumulated sequentially into
 * level 0; every `step` terms level 0 is spilled upward (level j is spilled into
 * j+1 when the running index is a multiple of step^(j+1)); at the end the tail in
 * level 0 receives levels 1..3 in that order.                                   */
static int ceil_log2_i64(int64_t x) {
  int l = 0;
  while (((int64_t)1 << l) < x) ++l;
  return l;
}

static float cascade_sum(const float *x, int64_t stride, int64_t n) {
  int lp = ceil_log2_i64(n) / ORA_LEVELS;
  if (lp < 4) lp = 4;
  const int64_t step = (int64_t)1 << lp, mask = step - 1;
  float acc[ORA_LEVELS] = {0.f, 0.f, 0.f, 0.f};
  int64_t i = 0;
  while (i + step <= n) {
    for (int64_t j = 0; j < step; ++j, ++i) acc[0] += x[i * stride];
    for (int j = 1; j < ORA_LEVELS; ++j) {
      acc[j] += acc[j - 1];
      acc[j - 1] = 0.f;
      if ((i & (mask << (j * lp))) != 0) break;
    }
  }
  for (; i < n; ++i) acc[0] += x[i * stride];
  for (int j = 1; j < ORA_LEVELS; ++j) acc[0] += acc[j];
  return acc[0];
}

/* PyTorch's row_sum: four interleaved cascade sums (elements k, k+4, k+8, ...),
 * the tail beyond 4*floor(n/4) added to partial 0, then partials 1..3 folded in. */
static float ilp4_sum(const float *x, int64_t stride, int64_t n) {
  const int64_t n4 = n / 4;
  float p[4];
  for (int k = 0; k < 4; ++k) p[k] = cascade_sum(x + k * stride, 4 * stride, n4);
  for (int64_t i = n4 * 4; i < n; ++i) p[0] += x[i * stride];
  p[0] += p[1];
  p[0] += p[2];
  p[0] += p[3];
  return p[0];
}

/* Which of the two orders an output column of the (R rows x R cols) outer
 * reduction uses: columns inside a full group of 32 (or, for R < 8, of 4) take the
 * plain cascade; the remaining columns take the 4-way interleaved form.          */
static int col_uses_cascade(int a, int R) {
  const int grp = (R >= 8) ? 32 : 4;
  return a < (R / grp) * grp;
}

/* out[a] = sum_j T[j*R + a]  for one (R,R) slab, canonical order */
static void outer_sum_slab(const float *T, int R, float *out) {
  for (int a = 0; a < R; ++a)
    out[a] = col_uses_cascade(a, R) ? cascade_sum(T + a, R, R) : ilp4_sum(T + a, R, R);
}

/* (R,R,R) contiguous -> (R,R): out[i,a] = sum_j T[i,j,a] */
void ora_sum_dim1(const float *T, int R, float *out) {
  for (int i = 0; i < R; ++i) outer_sum_slab(T + (size_t)i * R * R, R, out + (size_t)i * R);
}

/* Row sum of a contiguous row of n floats (inner reduction).  n < 8: ilp4 over the
 * scalars.  Otherwise the row is cut into floor(n/8) vectors of 8 lanes, each lane
 * ilp4-summed over the vector index; then fin = 0 + scalar tail (in order) +
 * lane 0 + ... + lane 7.                                                         */
static float inner_row_sum(const float *x, int n) {
  if (n < 8) return ilp4_sum(x, 1, n);
  const int cnt = n / 8;
  float lane[8];
  for (int l = 0; l < 8; ++l) lane[l] = ilp4_sum(x + l, 8, cnt);
  float fin = 0.f;
  for (int k = cnt * 8; k < n; ++k) fin += x[k];
  for (int l = 0; l < 8; ++l) fin += lane[l];
  return fin;
}

void ora_rowsum(const float *M, int R, int ld, float *out) {
  for (int i = 0; i < R; ++i) out[i] = inner_row_sum(M + (size_t)i * ld, R);
}

/* 2-norm of n contiguous floats: 8 lane accumulators with fused multiply-add over
 * consecutive chunks of 8, lanes folded 0+1+...+7, then the tail of n % 8 elements:
 * when it has 4 or more elements its first four squares are rounded on their own and
 * added one by one (the PyTorch CPU build computes them as one 4-wide product), every
 * other tail element is fused in; finally sqrt.                                    */
float ora_norm2(const float *x, int64_t n) {
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t d = 0;
  for (; d + 8 <= n; d += 8)
    for (int l = 0; l < 8; ++l) acc[l] = fmaf(x[d + l], x[d + l], acc[l]);
  float s = acc[0];
  for (int l = 1; l < 8; ++l) s = s + acc[l];
  if (n - d >= 4) {
    for (int k = 0; k < 4; ++k, ++d) {
      const float sq = x[d] * x[d];
      s = s + sq;
    }
  }
  for (; d < n; ++d) s = fmaf(x[d], x[d], s);
  return sqrtf(s);
}

/* ---- edge-similarity tensor (reference model.py:171-216) ----------------------
 * A, B: leading R x R block of row-major matrices with leading dimensions lda/ldb.
 * Both diagonals are forced to 1 first (model.py:184-185); degrees include it.
 * S is contiguous (R,R,R,R), index order (i,j,a,b).
 *   S[i,j,a,b] = ((((A[i,j]*A[j,j])*A[i,i])*B[a,b])*B[a,a])*B[b,b]   (model.py:205)
 *   zeroed where exactly one of i==j, a==b holds                     (model.py:208-209)
 *   S[i,i,a,a] = A[i,i]*B[a,a]*1/(|degA[i]-degB[a]|+1)               (model.py:212-215,54-56)
 * node_sim (R,R) may be NULL.                                                     */
void ora_edge_sim(const float *A, int lda, const float *B, int ldb, int R, float *S,
                  float *node_sim) {
  float *a = (float *)malloc(sizeof(float) * R * R), *b = (float *)malloc(sizeof(float) * R * R);
  float *dA = (float *)malloc(sizeof(float) * R), *dB = (float *)malloc(sizeof(float) * R);
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < R; ++j) {
      a[i * R + j] = (i == j) ? 1.f : A[(size_t)i * lda + j];
      b[i * R + j] = (i == j) ? 1.f : B[(size_t)i * ldb + j];
    }
  ora_rowsum(a, R, R, dA);
  ora_rowsum(b, R, R, dB);
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < R; ++j)
      for (int p = 0; p < R; ++p)
        for (int q = 0; q < R; ++q) {
          float v;
          if (i == j && p == q) {
            float ns = 1.0f / (fabsf(dA[i] - dB[p]) + 1.0f);
            v = (a[i * R + i] * b[p * R + p]) * ns;
            if (node_sim) node_sim[i * R + p] = v;
          } else if (i == j || p == q) {
            v = 0.f;
          } else {
            v = a[i * R + j] * a[j * R + j];
            v = v * a[i * R + i];
            v = v * b[p * R + q];
            v = v * b[p * R + p];
            v = v * b[q * R + q];
          }
          S[(((size_t)i * R + j) * R + p) * R + q] = v;
        }
  free(a); free(b); free(dA); free(dB);
}

/* ---- MPM on a materialised S (reference model.py:247-276) ---------------------
 * Returns the number of iterations executed.  tol is compared in fp32, as the
 * reference's `tensor < python_float` does.                                       */
int ora_mpm(const float *S, const float *X0, int R, int iters, float tol, int early_stop,
            float *X) {
  const size_t R2 = (size_t)R * R;
  float *ns = (float *)malloc(sizeof(float) * R2), *T = (float *)malloc(sizeof(float) * R2);
  float *Xn = (float *)malloc(sizeof(float) * R2), *et = (float *)malloc(sizeof(float) * R);
  memcpy(X, X0, sizeof(float) * R2);
  for (int i = 0; i < R; ++i)
    for (int p = 0; p < R; ++p) ns[i * R + p] = S[(((size_t)i * R + i) * R + p) * R + p];
  int done = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = 0; i < R; ++i) {
      /* T[j,a] = mask[i,j] * max_b S[i,j,a,b]*X[j,b]   (model.py:266-267) */
      for (int j = 0; j < R; ++j)
        for (int p = 0; p < R; ++p) {
          const float *s = S + (((size_t)i * R + j) * R + p) * R;
          float m = s[0] * X[j * R + 0];
          for (int q = 1; q < R; ++q) {
            float v = s[q] * X[j * R + q];
            if (v > m || v != v) m = v;
          }
          T[j * R + p] = (j == i) ? m * 0.f : m;
        }
      outer_sum_slab(T, R, et);
      for (int p = 0; p < R; ++p) Xn[i * R + p] = X[i * R + p] * ns[i * R + p] + et[p];
    }
    const float nrm = ora_norm2(Xn, (int64_t)R2);
    float dmax = 0.f;
    for (size_t k = 0; k < R2; ++k) {
      float v = Xn[k] / nrm;
      float d = fabsf(v - X[k]);
      if (d > dmax || d != d) dmax = d;
      X[k] = v;
    }
    ++done;
    if (early_stop && dmax < tol) break;
  }
  free(ns); free(T); free(Xn); free(et);
  return done;
}

/* Same recurrence without materialising S, valid when S is the tensor produced by
 * ora_edge_sim from a 0/1 matrix A (unit diagonal) and B in (0,1), and X >= 0:
 *   max_b S[i,j,a,b]*X[j,b] = A[i,j] * max_{b != a} B[a,b]*X[j,b]   for j != i.
 * Derived form (not in the reference); tests prove it bit-equal to ora_mpm.       */
int ora_mpm_sfree(const float *A, int lda, const float *B, int ldb, const float *X0, int R,
                  int iters, float tol, int early_stop, float *X) {
  const size_t R2 = (size_t)R * R;
  float *a = (float *)malloc(sizeof(float) * R2), *b = (float *)malloc(sizeof(float) * R2);
  float *dA = (float *)malloc(sizeof(float) * R), *dB = (float *)malloc(sizeof(float) * R);
  float *ns = (float *)malloc(sizeof(float) * R2), *M = (float *)malloc(sizeof(float) * R2);
  float *T = (float *)malloc(sizeof(float) * R2), *Xn = (float *)malloc(sizeof(float) * R2);
  float *et = (float *)malloc(sizeof(float) * R);
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < R; ++j) {
      a[i * R + j] = (i == j) ? 1.f : A[(size_t)i * lda + j];
      b[i * R + j] = (i == j) ? 1.f : B[(size_t)i * ldb + j];
    }
  ora_rowsum(a, R, R, dA);
  ora_rowsum(b, R, R, dB);
  for (int i = 0; i < R; ++i)
    for (int p = 0; p < R; ++p) ns[i * R + p] = 1.0f / (fabsf(dA[i] - dB[p]) + 1.0f);
  memcpy(X, X0, sizeof(float) * R2);
  int done = 0;
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < R; ++j)
      for (int p = 0; p < R; ++p) {
        float m = 0.f; /* the a==b entry of S is 0 and X >= 0, so 0 is always a candidate */
        for (int q = 0; q < R; ++q) {
          if (q == p) continue;
          float v = b[p * R + q] * X[j * R + q];
          if (v > m) m = v;
        }
        M[j * R + p] = m;
      }
    for (int i = 0; i < R; ++i) {
      for (int j = 0; j < R; ++j)
        for (int p = 0; p < R; ++p)
          T[j * R + p] = (j != i && a[i * R + j] != 0.f) ? M[j * R + p] : 0.f;
      outer_sum_slab(T, R, et);
      for (int p = 0; p < R; ++p) Xn[i * R + p] = X[i * R + p] * ns[i * R + p] + et[p];
    }
    const float nrm = ora_norm2(Xn, (int64_t)R2);
    float dmax = 0.f;
    for (size_t k = 0; k < R2; ++k) {
      float v = Xn[k] / nrm;
      float d = fabsf(v - X[k]);
      if (d > dmax || d != d) dmax = d;
      X[k] = v;
    }
    ++done;
    if (early_stop && dmax < tol) break;
  }
  free(a); free(b); free(dA); free(dB); free(ns); free(M); free(T); free(Xn); free(et);
  return done;
}

/* ---- linear sum assignment: SciPy 1.15.3 rectangular_lsap (square case) -------
 * Shortest augmenting path (Crouse 2016).  Rows are augmented in order 0..n-1; the
 * list of unscanned columns starts as n-1, n-2, ..., 0 and shrinks by swap-remove;
 * a column's tentative cost is lowered only on strict '<'; among columns at the
 * current minimum the scan keeps the first one seen unless a later one at the same
 * value is unassigned (then the last such one wins).  All arithmetic in double.
 * Returns 0, or -1 if infeasible / non-finite cost.  col4row[i] = column of row i. */
int ora_lsap(const double *cost, int n, int64_t *col4row_out) {
  if (n == 0) return 0;
  for (size_t k = 0; k < (size_t)n * n; ++k)
    if (cost[k] != cost[k] || cost[k] == -INFINITY) return -1;
  double *u = (double *)calloc(n, sizeof(double)), *v = (double *)calloc(n, sizeof(double));
  double *spc = (double *)malloc(sizeof(double) * n);
  int *path = (int *)malloc(sizeof(int) * n), *col4row = (int *)malloc(sizeof(int) * n);
  int *row4col = (int *)malloc(sizeof(int) * n), *remaining = (int *)malloc(sizeof(int) * n);
  char *SR = (char *)malloc(n), *SC = (char *)malloc(n);
  for (int k = 0; k < n; ++k) { path[k] = -1; col4row[k] = -1; row4col[k] = -1; }
  int rc = 0;
  for (int cur = 0; cur < n && rc == 0; ++cur) {
    double minVal = 0.0;
    int i = cur, num_remaining = n, sink = -1;
    for (int it = 0; it < n; ++it) remaining[it] = n - it - 1;
    memset(SR, 0, n);
    memset(SC, 0, n);
    for (int k = 0; k < n; ++k) spc[k] = INFINITY;
    while (sink == -1) {
      int index = -1;
      double lowest = INFINITY;
      SR[i] = 1;
      for (int it = 0; it < num_remaining; ++it) {
        int j = remaining[it];
        double r = minVal + cost[(size_t)i * n + j] - u[i] - v[j];
        if (r < spc[j]) { path[j] = i; spc[j] = r; }
        if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
          lowest = spc[j];
          index = it;
        }
      }
      minVal = lowest;
      if (minVal == INFINITY) { rc = -1; break; }
      int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--num_remaining];
    }
    if (rc) break;
    u[cur] += minVal;
    for (int r = 0; r < n; ++r)
      if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
    for (int c = 0; c < n; ++c)
      if (SC[c]) v[c] -= minVal - spc[c];
    int j = sink;
    for (;;) {
      int r = path[j];
      row4col[j] = r;
      int t = col4row[r]; col4row[r] = j; j = t;
      if (r == cur) break;
    }
  }
  if (rc == 0) for (int k = 0; k < n; ++k) col4row_out[k] = col4row[k];
  free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(remaining);
  free(SR); free(SC);
  return rc;
}

/* Reference model.py:319-323: cost = -X (fp32 -> double), solve, perm[col[i]] = i. */
int ora_assign_perm(const float *X, int R, int64_t *perm) {
  double *c = (double *)malloc(sizeof(double) * (size_t)R * R);
  int64_t *col = (int64_t *)malloc(sizeof(int64_t) * R);
  for (size_t k = 0; k < (size_t)R * R; ++k) c[k] = (double)(-X[k]);
  int rc = ora_lsap(c, R, col);
  if (rc == 0) for (int i = 0; i < R; ++i) perm[col[i]] = i;
  free(c); free(col);
  return rc;
}
