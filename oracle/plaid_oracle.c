/* plaid_oracle.c -- plain-C, single-threaded CPU restatement of the scoring hot path.
 *
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Used by tests/ as a second, independent checker
 * of oracle/plaid_oracle.py and by bench.py's `cpu_baseline` leg (kind "port": the reference
 * is R and cannot run here or on the GPU box).  plaid_amd/ never links or loads it.
 *
 * Each function cites the reference lines it follows (/root/reference/R/plaid.R).  The
 * reference is single-threaded (one R process calling Matrix/matrixStats C code), so is this;
 * the *_mt entry points at the end run the SAME per-column code over columns with OpenMP
 * (bench.py's "all cores" baseline: the honest best a CPU box does, not what the reference does).
 * Parity status: see the header of oracle/plaid_oracle.py (rank path unpinned by the reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* t(G) %*% X with G = 1*(matG != 0), column-scaled by 1/(1e-8 + colSums(G)) for stat "mean"
 * (R/plaid.R:73-80, Matrix::crossprod at :107).  X: g x n column-major; G: CSC pattern in X's
 * row space; S: m x n column-major.  Each term is multiplied by the scaled entry, as the
 * reference's crossprod of the colScale'd matrix does. */
void oracle_crossprod_dense(const double* X, int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi,
                            int32_t m, int stat_sum, double* S) {
  for (int32_t c = 0; c < n; ++c) {
    const double* xc = X + (size_t)c * g;
    double* sc = S + (size_t)c * m;
    for (int32_t j = 0; j < m; ++j) {
      const int32_t p0 = Gp[j], p1 = Gp[j + 1];
      const double w = stat_sum ? 1.0 : 1.0 / (1e-8 + (double)(p1 - p0));
      double s = 0.0;
      for (int32_t p = p0; p < p1; ++p) s += w * xc[Gi[p]];
      sc[j] = s;
    }
  }
}

static int cmp_double(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

/* normalize_medians(x, ignore.zero), R/plaid.R:554-575.  ignore_zero: -1 = NULL (auto). */
void oracle_normalize_medians(double* S, int32_t m, int32_t n, int ignore_zero, double* med_out) {
  const size_t tot = (size_t)m * n;
  if (ignore_zero < 0) { /* :556-557 */
    double mn = INFINITY;
    for (size_t i = 0; i < tot; ++i)
      if (S[i] == S[i] && S[i] < mn) mn = S[i];
    ignore_zero = (mn == 0.0);
  }
  double* buf = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
  double* med = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  double msum = 0.0;
  size_t mcnt = 0;
  for (int32_t c = 0; c < n; ++c) {
    const double* sc = S + (size_t)c * m;
    int32_t k = 0;
    for (int32_t i = 0; i < m; ++i) {
      const double v = sc[i];
      if (v != v) continue;                    /* na.rm = TRUE */
      if (ignore_zero && v == 0.0) continue;   /* :562-563 */
      buf[k++] = v;
    }
    double md;
    if (k == 0) {
      md = ignore_zero ? 0.0 : NAN;            /* :566 */
    } else {
      qsort(buf, (size_t)k, sizeof(double), cmp_double);
      md = (k & 1) ? buf[k / 2] : 0.5 * (buf[k / 2 - 1] + buf[k / 2]);
    }
    med[c] = md;
    if (md == md) { msum += md; ++mcnt; }
  }
  const double mean = mcnt ? msum / (double)mcnt : NAN;   /* mean(medx, na.rm=TRUE), :572 */
  for (int32_t c = 0; c < n; ++c) {
    double* sc = S + (size_t)c * m;
    for (int32_t i = 0; i < m; ++i) sc[i] = (sc[i] - med[c]) + mean;
  }
  if (med_out) memcpy(med_out, med, sizeof(double) * (size_t)n);
  free(buf);
  free(med);
}

typedef struct { double v; int32_t i; } kv_t;
static int cmp_kv(const void* a, const void* b) {
  const double x = ((const kv_t*)a)->v, y = ((const kv_t*)b)->v;
  return (x > y) - (x < y);
}

/* rank(x, ties.method) for one NaN-free vector; ties: 0 average, 1 min, 2 max.
 * is_signed: sign(x) * rank(|x|) (R/plaid.R:603-606, 612-615, 637-640). */
static void rank_vector(const double* x, int32_t len, int ties, int is_signed, kv_t* tmp, double* out) {
  for (int32_t i = 0; i < len; ++i) {
    tmp[i].v = is_signed ? fabs(x[i]) : x[i];
    if (tmp[i].v == 0.0) tmp[i].v = 0.0;
    tmp[i].i = i;
  }
  qsort(tmp, (size_t)len, sizeof(kv_t), cmp_kv);
  int32_t a = 0;
  while (a < len) {
    int32_t b = a + 1;
    while (b < len && tmp[b].v == tmp[a].v) ++b;
    double r;
    if (ties == 1) r = (double)(a + 1);
    else if (ties == 2) r = (double)b;
    else r = 0.5 * (double)(a + 1 + b);
    for (int32_t k = a; k < b; ++k) {
      const int32_t i = tmp[k].i;
      double s = 1.0;
      if (is_signed) s = (x[i] > 0) - (x[i] < 0);
      out[i] = s * r;
    }
    a = b;
  }
}

/* colranks(), dense branch: t(matrixStats::colRanks(as.matrix(X), ties.method)) (R/plaid.R:611-619) */
void oracle_colranks_dense(const double* X, int32_t g, int32_t n, int ties, int is_signed, double* R) {
  kv_t* tmp = (kv_t*)malloc(sizeof(kv_t) * (size_t)(g > 0 ? g : 1));
  for (int32_t c = 0; c < n; ++c) rank_vector(X + (size_t)c * g, g, ties, is_signed, tmp, R + (size_t)c * g);
  free(tmp);
}

/* sparse_colranks(), R/plaid.R:631-650: rank() over the stored non-zeros of each column */
void oracle_sparse_colranks(const int32_t* Xp, const double* Xx, int32_t n, int ties, int is_signed, double* Rx) {
  int32_t mx = 1;
  for (int32_t c = 0; c < n; ++c)
    if (Xp[c + 1] - Xp[c] > mx) mx = Xp[c + 1] - Xp[c];
  kv_t* tmp = (kv_t*)malloc(sizeof(kv_t) * (size_t)mx);
  for (int32_t c = 0; c < n; ++c) rank_vector(Xx + Xp[c], Xp[c + 1] - Xp[c], ties, is_signed, tmp, Rx + Xp[c]);
  free(tmp);
}

/* plaid(X, matG, stats, normalize) body after alignment (R/plaid.R:73-85), dense X */
void oracle_plaid_dense(const double* X, int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi,
                        int32_t m, int stat_sum, int normalize, double* S) {
  oracle_crossprod_dense(X, g, n, Gp, Gi, m, stat_sum, S);
  if (normalize) oracle_normalize_medians(S, m, n, -1, NULL);
}

/* ---------------------------------------------------------------------------------------------
 * Sparse X: t(G) %*% X for a dgCMatrix X (Matrix::crossprod, sparse x sparse branch, R/plaid.R:107).
 * Gustavson / CHOLMOD-ssmult order: for every stored x[i, c], add w_j * x to every set j that contains
 * gene i -- work = sum over stored values of (sets per gene), not the dense z per column.
 * Gt: the membership gene-major (CSR of G = CSC of t(G)): Gtp[g + 1], Gtj[z] set ids. */
static void crossprod_csc_column(const int32_t* Xi, const double* Xx, int32_t q0, int32_t q1, const int32_t* Gtp,
                                 const int32_t* Gtj, const double* w, int32_t m, double* sc) {
  for (int32_t j = 0; j < m; ++j) sc[j] = 0.0;
  for (int32_t q = q0; q < q1; ++q) {
    const int32_t i = Xi[q];
    const double x = Xx[q];
    for (int32_t p = Gtp[i]; p < Gtp[i + 1]; ++p) sc[Gtj[p]] += w[Gtj[p]] * x;
  }
}

static void transpose_pattern(int32_t g, int32_t m, const int32_t* Gp, const int32_t* Gi, int32_t* Gtp, int32_t* Gtj) {
  memset(Gtp, 0, sizeof(int32_t) * ((size_t)g + 1));
  for (int32_t p = 0; p < Gp[m]; ++p) Gtp[Gi[p] + 1]++;
  for (int32_t i = 0; i < g; ++i) Gtp[i + 1] += Gtp[i];
  int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * ((size_t)g + 1));
  memcpy(fill, Gtp, sizeof(int32_t) * ((size_t)g + 1));
  for (int32_t j = 0; j < m; ++j)
    for (int32_t p = Gp[j]; p < Gp[j + 1]; ++p) Gtj[fill[Gi[p]]++] = j;
  free(fill);
}

void oracle_crossprod_csc_mt(const int32_t* Xp, const int32_t* Xi, const double* Xx, int32_t g, int32_t n,
                             const int32_t* Gp, const int32_t* Gi, int32_t m, int stat_sum, double* S, int threads) {
  int32_t* Gtp = (int32_t*)malloc(sizeof(int32_t) * ((size_t)g + 1));
  int32_t* Gtj = (int32_t*)malloc(sizeof(int32_t) * (size_t)(Gp[m] > 0 ? Gp[m] : 1));
  double* w = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
  transpose_pattern(g, m, Gp, Gi, Gtp, Gtj);
  for (int32_t j = 0; j < m; ++j) w[j] = stat_sum ? 1.0 : 1.0 / (1e-8 + (double)(Gp[j + 1] - Gp[j]));
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
  for (int32_t c = 0; c < n; ++c)
    crossprod_csc_column(Xi, Xx, Xp[c], Xp[c + 1], Gtp, Gtj, w, m, S + (size_t)c * m);
  free(Gtp);
  free(Gtj);
  free(w);
}

void oracle_crossprod_dense_mt(const double* X, int32_t g, int32_t n, const int32_t* Gp, const int32_t* Gi,
                               int32_t m, int stat_sum, double* S, int threads) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
  for (int32_t c = 0; c < n; ++c)
    oracle_crossprod_dense(X + (size_t)c * g, g, 1, Gp, Gi, m, stat_sum, S + (size_t)c * m);
}

void oracle_colranks_dense_mt(const double* X, int32_t g, int32_t n, int ties, int is_signed, double* R, int threads) {
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    kv_t* tmp = (kv_t*)malloc(sizeof(kv_t) * (size_t)(g > 0 ? g : 1));
#pragma omp for schedule(dynamic, 4)
    for (int32_t c = 0; c < n; ++c) rank_vector(X + (size_t)c * g, g, ties, is_signed, tmp, R + (size_t)c * g);
    free(tmp);
  }
}

void oracle_sparse_colranks_mt(const int32_t* Xp, const double* Xx, int32_t n, int ties, int is_signed, double* Rx,
                               int threads) {
  int32_t mx = 1;
  for (int32_t c = 0; c < n; ++c)
    if (Xp[c + 1] - Xp[c] > mx) mx = Xp[c + 1] - Xp[c];
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    kv_t* tmp = (kv_t*)malloc(sizeof(kv_t) * (size_t)mx);
#pragma omp for schedule(dynamic, 64)
    for (int32_t c = 0; c < n; ++c) rank_vector(Xx + Xp[c], Xp[c + 1] - Xp[c], ties, is_signed, tmp, Rx + Xp[c]);
    free(tmp);
  }
}

/* column medians of normalize_medians (R/plaid.R:561-566) over columns in parallel; the caller (Python) does the
 * global parts (ignore.zero rule, mean of medians, shift) exactly as oracle_normalize_medians */
void oracle_normalize_medians_mt(double* S, int32_t m, int32_t n, int ignore_zero, double* med_out, int threads) {
  const size_t tot = (size_t)m * n;
  if (ignore_zero < 0) {
    double mn = INFINITY;
    for (size_t i = 0; i < tot; ++i)
      if (S[i] == S[i] && S[i] < mn) mn = S[i];
    ignore_zero = (mn == 0.0);
  }
  double* med = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    double* buf = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
#pragma omp for schedule(dynamic, 16)
    for (int32_t c = 0; c < n; ++c) {
      const double* sc = S + (size_t)c * m;
      int32_t k = 0;
      for (int32_t i = 0; i < m; ++i) {
        const double v = sc[i];
        if (v != v) continue;
        if (ignore_zero && v == 0.0) continue;
        buf[k++] = v;
      }
      double md;
      if (k == 0) md = ignore_zero ? 0.0 : NAN;
      else {
        qsort(buf, (size_t)k, sizeof(double), cmp_double);
        md = (k & 1) ? buf[k / 2] : 0.5 * (buf[k / 2 - 1] + buf[k / 2]);
      }
      med[c] = md;
    }
    free(buf);
  }
  double msum = 0.0;
  size_t mcnt = 0;
  for (int32_t c = 0; c < n; ++c)
    if (med[c] == med[c]) { msum += med[c]; ++mcnt; }
  const double mean = mcnt ? msum / (double)mcnt : NAN;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (int32_t c = 0; c < n; ++c) {
    double* sc = S + (size_t)c * m;
    for (int32_t i = 0; i < m; ++i) sc[i] = (sc[i] - med[c]) + mean;
  }
  if (med_out) memcpy(med_out, med, sizeof(double) * (size_t)n);
  free(med);
}
