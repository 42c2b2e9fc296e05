/* .Call() shim between R and the C ABI of include/plaidhip.h.
 *
 * Argument unpacking only: every number is produced by libplaidhip.so.  Not compiled in the
 * build container (R is absent there); it needs only R's headers and -lplaidhip.
 * Error contract: the C ABI never throws or longjmps; a non-zero status is turned into
 * Rf_error() here, after the library call has returned (no C++ frames to unwind).
 * Threading: .Call arrives on R's main thread; the library synchronises before returning.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include <string.h>

#include "plaidhip.h"

static plaidhip_ctx* g_ctx = NULL;
static int g_device = 0;

static plaidhip_ctx* ctx(void) {
  if (g_ctx == NULL) {
    int rc = plaidhip_init(g_device, NULL, &g_ctx);
    if (rc != PLAIDHIP_OK) Rf_error("plaidhip: %s", plaidhip_last_error_string());
  }
  return g_ctx;
}

static void check(int rc) {
  if (rc != PLAIDHIP_OK) Rf_error("plaidhip: %s", plaidhip_last_error_string());
}

/* options(plaidhip.device, plaidhip.precision): called by every R wrapper before its .Call.  A changed device
 * closes the session context (the next call opens one on the new device). */
SEXP R_plaidhip_session(SEXP device, SEXP precision) {
  const int d = Rf_asInteger(device);
  if (d != g_device && g_ctx != NULL) { plaidhip_finalize(g_ctx); g_ctx = NULL; }
  g_device = d;
  check(plaidhip_set_precision(ctx(), Rf_asInteger(precision)));
  check(plaidhip_multi_set_precision(Rf_asInteger(precision)));   /* options(plaidhip.devices) with several GPUs */
  return R_NilValue;
}

/* plaid(): X numeric matrix g x n; Gp/Gi integer vectors = aligned membership pattern in
 * X's row space (built in R/plaid-hip.R); returns an m x n numeric matrix. */
SEXP R_plaidhip_plaid_dense(SEXP X, SEXP Gp, SEXP Gi, SEXP stat, SEXP normalize) {
  const int g = Rf_nrows(X), n = Rf_ncols(X), m = LENGTH(Gp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_plaid_dense(ctx(), REAL(X), g, n, INTEGER(Gp), INTEGER(Gi), m, Rf_asInteger(stat),
                             Rf_asLogical(normalize), REAL(S)));
  UNPROTECT(1);
  return S;
}

/* same for a dgCMatrix: slots @p, @i, @x and nrow */
SEXP R_plaidhip_plaid_csc(SEXP Xp, SEXP Xi, SEXP Xx, SEXP g, SEXP Gp, SEXP Gi, SEXP stat, SEXP normalize) {
  const int n = LENGTH(Xp) - 1, m = LENGTH(Gp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_plaid_csc(ctx(), INTEGER(Xp), INTEGER(Xi), REAL(Xx), Rf_asInteger(g), n, INTEGER(Gp),
                           INTEGER(Gi), m, Rf_asInteger(stat), Rf_asLogical(normalize), REAL(S)));
  UNPROTECT(1);
  return S;
}

/* chunked_crossprod(x, y) with a general sparse x (stored values differ inside a column): x as its dgCMatrix slots,
 * y a numeric matrix (R/plaid.R:107, :117: Matrix::crossprod(x, y[, jj])) */
SEXP R_plaidhip_crossprod_weighted_dense(SEXP Wp, SEXP Wi, SEXP Wx, SEXP Y) {
  const int g = Rf_nrows(Y), n = Rf_ncols(Y), m = LENGTH(Wp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_crossprod_weighted_dense(ctx(), INTEGER(Wp), INTEGER(Wi), REAL(Wx), g, m, REAL(Y), n, REAL(S)));
  UNPROTECT(1);
  return S;
}

/* same for a dgCMatrix y: slots @p, @i, @x and nrow */
SEXP R_plaidhip_crossprod_weighted_csc(SEXP Wp, SEXP Wi, SEXP Wx, SEXP Yp, SEXP Yi, SEXP Yx, SEXP g) {
  const int n = LENGTH(Yp) - 1, m = LENGTH(Wp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_crossprod_weighted_csc(ctx(), INTEGER(Wp), INTEGER(Wi), REAL(Wx), Rf_asInteger(g), m, INTEGER(Yp),
                                        INTEGER(Yi), REAL(Yx), n, REAL(S)));
  UNPROTECT(1);
  return S;
}

/* replaid.sing for a dgCMatrix: slots @p, @i, @x and nrow (no as.matrix(X) on the host) */
SEXP R_plaidhip_sing_csc(SEXP Xp, SEXP Xi, SEXP Xx, SEXP g, SEXP Gp, SEXP Gi) {
  const int n = LENGTH(Xp) - 1, m = LENGTH(Gp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_sing_csc(ctx(), INTEGER(Xp), INTEGER(Xi), REAL(Xx), Rf_asInteger(g), n, INTEGER(Gp), INTEGER(Gi), m,
                          REAL(S)));
  UNPROTECT(1);
  return S;
}

/* normalize_medians(x, ignore.zero): ignore_zero = NA (NULL in R) / FALSE / TRUE */
SEXP R_plaidhip_normalize_medians(SEXP x, SEXP ignore_zero) {
  const int m = Rf_nrows(x), n = Rf_ncols(x);
  SEXP S = PROTECT(Rf_duplicate(x));
  const int iz = Rf_asLogical(ignore_zero);
  check(plaidhip_normalize_medians(ctx(), REAL(S), m, n, iz == NA_LOGICAL ? PLAIDHIP_IGNORE_ZERO_AUTO : iz, NULL));
  UNPROTECT(1);
  return S;
}

SEXP R_plaidhip_colranks_dense(SEXP X, SEXP ties, SEXP is_signed) {
  const int g = Rf_nrows(X), n = Rf_ncols(X);
  SEXP R = PROTECT(Rf_allocMatrix(REALSXP, g, n));
  check(plaidhip_colranks_dense(ctx(), REAL(X), g, n, Rf_asInteger(ties), Rf_asLogical(is_signed), REAL(R)));
  UNPROTECT(1);
  return R;
}

/* sparse_colranks(): returns the new @x vector; the caller keeps @i/@p (R/plaid.R:645-646) */
SEXP R_plaidhip_colranks_csc(SEXP Xp, SEXP Xx, SEXP ties, SEXP is_signed) {
  const int n = LENGTH(Xp) - 1;
  SEXP R = PROTECT(Rf_allocVector(REALSXP, XLENGTH(Xx)));
  check(plaidhip_colranks_csc(ctx(), INTEGER(Xp), REAL(Xx), n, Rf_asInteger(ties), Rf_asLogical(is_signed), REAL(R)));
  UNPROTECT(1);
  return R;
}

/* colranks(sparse X, keep.zero = FALSE): dense g x n ranks from the CSC slots (R/plaid.R:602-609) */
SEXP R_plaidhip_colranks_csc_dense(SEXP Xp, SEXP Xi, SEXP Xx, SEXP g, SEXP ties, SEXP is_signed) {
  const int n = LENGTH(Xp) - 1, gg = Rf_asInteger(g);
  SEXP R = PROTECT(Rf_allocMatrix(REALSXP, gg, n));
  check(plaidhip_colranks_csc_dense(ctx(), INTEGER(Xp), INTEGER(Xi), REAL(Xx), gg, n, Rf_asInteger(ties),
                                    Rf_asLogical(is_signed), REAL(R)));
  UNPROTECT(1);
  return R;
}

SEXP R_plaidhip_sing_dense(SEXP X, SEXP Gp, SEXP Gi) {
  const int g = Rf_nrows(X), n = Rf_ncols(X), m = LENGTH(Gp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_sing_dense(ctx(), REAL(X), g, n, INTEGER(Gp), INTEGER(Gi), m, REAL(S)));
  UNPROTECT(1);
  return S;
}

SEXP R_plaidhip_ssgsea_dense(SEXP X, SEXP Gp, SEXP Gi, SEXP alpha) {
  const int g = Rf_nrows(X), n = Rf_ncols(X), m = LENGTH(Gp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_ssgsea_dense(ctx(), REAL(X), g, n, INTEGER(Gp), INTEGER(Gi), m, Rf_asReal(alpha), REAL(S)));
  UNPROTECT(1);
  return S;
}

SEXP R_plaidhip_ssgsea_csc(SEXP Xp, SEXP Xi, SEXP Xx, SEXP g, SEXP Gp, SEXP Gi, SEXP alpha) {
  const int n = LENGTH(Xp) - 1, m = LENGTH(Gp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_ssgsea_csc(ctx(), INTEGER(Xp), INTEGER(Xi), REAL(Xx), Rf_asInteger(g), n, INTEGER(Gp),
                            INTEGER(Gi), m, Rf_asReal(alpha), REAL(S)));
  UNPROTECT(1);
  return S;
}

/* X: numeric matrix or NULL; for a dgCMatrix pass Xp/Xi/Xx (else R_NilValue) */
static const int* int_or_null(SEXP x) { return Rf_isNull(x) ? NULL : INTEGER(x); }

/* several GPUs from the one R process: `devices` integer vector of ordinals (plaidhip_*_multi, a host thread per
 * device inside the library) */
SEXP R_plaidhip_plaid_multi(SEXP devices, SEXP Xp, SEXP Xi, SEXP Xv, SEXP g, SEXP n, SEXP Gp, SEXP Gi, SEXP stat,
                            SEXP normalize) {
  const int m = LENGTH(Gp) - 1, nn = Rf_asInteger(n);
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, nn));
  check(plaidhip_plaid_multi(INTEGER(devices), LENGTH(devices), int_or_null(Xp), int_or_null(Xi), REAL(Xv), Rf_asInteger(g),
                             nn, INTEGER(Gp), INTEGER(Gi), m, Rf_asInteger(stat), Rf_asLogical(normalize), REAL(S)));
  UNPROTECT(1);
  return S;
}

SEXP R_plaidhip_sing_csc_multi(SEXP devices, SEXP Xp, SEXP Xi, SEXP Xx, SEXP g, SEXP Gp, SEXP Gi) {
  const int n = LENGTH(Xp) - 1, m = LENGTH(Gp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_sing_csc_multi(INTEGER(devices), LENGTH(devices), INTEGER(Xp), INTEGER(Xi), REAL(Xx), Rf_asInteger(g), n,
                                INTEGER(Gp), INTEGER(Gi), m, REAL(S)));
  UNPROTECT(1);
  return S;
}

SEXP R_plaidhip_sing_multi(SEXP devices, SEXP X, SEXP Gp, SEXP Gi) {
  const int g = Rf_nrows(X), n = Rf_ncols(X), m = LENGTH(Gp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  check(plaidhip_sing_multi(INTEGER(devices), LENGTH(devices), REAL(X), g, n, INTEGER(Gp), INTEGER(Gi), m, REAL(S)));
  UNPROTECT(1);
  return S;
}

SEXP R_plaidhip_ssgsea_multi(SEXP devices, SEXP Xp, SEXP Xi, SEXP Xv, SEXP g, SEXP n, SEXP Gp, SEXP Gi, SEXP alpha) {
  const int m = LENGTH(Gp) - 1, nn = Rf_asInteger(n);
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, nn));
  check(plaidhip_ssgsea_multi(INTEGER(devices), LENGTH(devices), int_or_null(Xp), int_or_null(Xi), REAL(Xv),
                              Rf_asInteger(g), nn, INTEGER(Gp), INTEGER(Gi), m, Rf_asReal(alpha), REAL(S)));
  UNPROTECT(1);
  return S;
}

SEXP R_plaidhip_ucell(SEXP Xp, SEXP Xi, SEXP Xv, SEXP g, SEXP n, SEXP Gp, SEXP Gi, SEXP kfull, SEXP rmax) {
  const int m = LENGTH(Gp) - 1, nn = Rf_asInteger(n);
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, nn));
  check(plaidhip_ucell(ctx(), int_or_null(Xp), int_or_null(Xi), REAL(Xv), Rf_asInteger(g), nn, INTEGER(Gp),
                       INTEGER(Gi), m, REAL(kfull), Rf_asReal(rmax), REAL(S)));
  UNPROTECT(1);
  return S;
}

SEXP R_plaidhip_aucell(SEXP Xp, SEXP Xi, SEXP Xv, SEXP g, SEXP n, SEXP Gp, SEXP Gi, SEXP auc_max_rank) {
  const int m = LENGTH(Gp) - 1, nn = Rf_asInteger(n);
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, nn));
  check(plaidhip_aucell(ctx(), int_or_null(Xp), int_or_null(Xi), REAL(Xv), Rf_asInteger(g), nn, INTEGER(Gp),
                        INTEGER(Gi), m, Rf_asReal(auc_max_rank), REAL(S)));
  UNPROTECT(1);
  return S;
}

SEXP R_plaidhip_scse(SEXP Xp, SEXP Xi, SEXP Xv, SEXP g, SEXP n, SEXP Gp, SEXP Gi, SEXP remove_log2, SEXP score_mean) {
  const int m = LENGTH(Gp) - 1, nn = Rf_asInteger(n);
  const int rl = Rf_asLogical(remove_log2);
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, nn));
  int removed = 0;
  check(plaidhip_scse(ctx(), int_or_null(Xp), int_or_null(Xi), REAL(Xv), Rf_asInteger(g), nn, INTEGER(Gp),
                      INTEGER(Gi), m, rl == NA_LOGICAL ? -1 : rl, Rf_asLogical(score_mean), REAL(S), &removed));
  /* whether 2**x ran (decided on the device when removeLog2 = NULL): the R wrapper prints R/plaid.R:164's message */
  Rf_setAttrib(S, Rf_install("removedLog2"), Rf_ScalarLogical(removed));
  UNPROTECT(1);
  return S;
}


/* gmt2mat(read.gmt(file)) in one native call: returns list(p, i, Dim, rownames, colnames); the R side
 * wraps it with new("dgCMatrix", ...).  (R/gmt-utils.R:19-66, 99-125)                             */
static SEXP split_lines(const char* txt, int64_t nbytes, int64_t count) {
  SEXP out = PROTECT(Rf_allocVector(STRSXP, (R_xlen_t)count));
  int64_t b = 0, k = 0;
  for (int64_t e = 0; e <= nbytes && k < count; ++e)
    if (e == nbytes || txt[e] == '\n') {
      SET_STRING_ELT(out, (R_xlen_t)k++, Rf_mkCharLenCE(txt + b, (int)(e - b), CE_UTF8));
      b = e + 1;
    }
  UNPROTECT(1);
  return out;
}

SEXP R_plaidhip_gmt2mat_file(SEXP path, SEXP add_source, SEXP nrows, SEXP max_genes, SEXP ntop, SEXP bg) {
  plaidhip_gmt* gmt = NULL;
  plaidhip_gmtmat* mat = NULL;
  if (plaidhip_gmt_read(CHAR(STRING_ELT(path, 0)), Rf_asLogical(add_source), (int64_t)Rf_asReal(nrows), &gmt) != PLAIDHIP_OK)
    Rf_error("plaidhip: %s", plaidhip_last_error_string());
  const int64_t nbg = Rf_isNull(bg) ? 0 : (int64_t)XLENGTH(bg);
  const char** bgp = nbg ? (const char**)R_alloc((size_t)nbg, sizeof(char*)) : NULL;
  for (int64_t k = 0; k < nbg; ++k) bgp[k] = Rf_translateCharUTF8(STRING_ELT(bg, (R_xlen_t)k));
  const int rc = plaidhip_gmt2mat(gmt, (int64_t)Rf_asReal(max_genes), (int64_t)Rf_asReal(ntop), bgp, nbg, &mat);
  plaidhip_gmt_destroy(gmt);
  if (rc != PLAIDHIP_OK) Rf_error("plaidhip: %s", plaidhip_last_error_string());
  /* everything R needs is copied into R_alloc() memory (released by R itself, also on error) and `mat` is freed
   * BEFORE the first R allocation that could longjmp: an allocation failure cannot leak the native object */
  int64_t dims[3], nb_r = 0, nb_c = 0;
  plaidhip_gmtmat_dims(mat, dims);
  int* cp = (int*)R_alloc((size_t)dims[1] + 1, sizeof(int));
  int* ci = (int*)R_alloc((size_t)(dims[2] > 0 ? dims[2] : 1), sizeof(int));
  memcpy(cp, plaidhip_gmtmat_p(mat), sizeof(int) * (size_t)(dims[1] + 1));
  if (dims[2]) memcpy(ci, plaidhip_gmtmat_i(mat), sizeof(int) * (size_t)dims[2]);
  const char* rn0 = plaidhip_gmtmat_names(mat, 0, &nb_r);
  char* rn = (char*)R_alloc((size_t)nb_r + 1, 1);
  memcpy(rn, rn0, (size_t)nb_r);
  const char* cn0 = plaidhip_gmtmat_names(mat, 1, &nb_c);
  char* cn = (char*)R_alloc((size_t)nb_c + 1, 1);
  memcpy(cn, cn0, (size_t)nb_c);
  plaidhip_gmtmat_destroy(mat);
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 5));
  SEXP p = PROTECT(Rf_allocVector(INTSXP, (R_xlen_t)dims[1] + 1));
  SEXP i = PROTECT(Rf_allocVector(INTSXP, (R_xlen_t)dims[2]));
  memcpy(INTEGER(p), cp, sizeof(int) * (size_t)(dims[1] + 1));
  if (dims[2]) memcpy(INTEGER(i), ci, sizeof(int) * (size_t)dims[2]);
  SEXP dim = PROTECT(Rf_allocVector(INTSXP, 2));
  INTEGER(dim)[0] = (int)dims[0];
  INTEGER(dim)[1] = (int)dims[1];
  SEXP rnames = PROTECT(split_lines(rn, nb_r, dims[0]));
  SEXP cnames = PROTECT(split_lines(cn, nb_c, dims[1]));
  SET_VECTOR_ELT(out, 0, p);
  SET_VECTOR_ELT(out, 1, i);
  SET_VECTOR_ELT(out, 2, dim);
  SET_VECTOR_ELT(out, 3, rnames);
  SET_VECTOR_ELT(out, 4, cnames);
  UNPROTECT(6);
  return out;
}


/* plaid.test(): sets x 6 matrix (gsetFC, p.one, p.two, p.lm, p.meta, q.meta) in G's column order */
SEXP R_plaidhip_plaid_test(SEXP X, SEXP y, SEXP Gp, SEXP Gi, SEXP gsetX, SEXP tests, SEXP metap) {
  const int g = Rf_nrows(X), n = Rf_ncols(X), m = LENGTH(Gp) - 1;
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, m, 6));
  int rc = plaidhip_plaid_test(ctx(), REAL(X), g, n, INTEGER(y), INTEGER(Gp), INTEGER(Gi), m,
                               Rf_isNull(gsetX) ? NULL : REAL(gsetX), Rf_asInteger(tests), Rf_asInteger(metap), REAL(out));
  if (rc != PLAIDHIP_OK) Rf_error("%s", plaidhip_last_error_string());
  UNPROTECT(1);
  return out;
}


SEXP R_plaidhip_gsva(SEXP X, SEXP Gp, SEXP Gi, SEXP tau, SEXP rowtf) {
  const int g = Rf_nrows(X), n = Rf_ncols(X), m = LENGTH(Gp) - 1;
  SEXP S = PROTECT(Rf_allocMatrix(REALSXP, m, n));
  int rc = plaidhip_gsva(ctx(), REAL(X), g, n, INTEGER(Gp), INTEGER(Gi), m, Rf_asReal(tau), Rf_asInteger(rowtf), REAL(S));
  if (rc != PLAIDHIP_OK) Rf_error("%s", plaidhip_last_error_string());
  UNPROTECT(1);
  return S;
}

static const R_CallMethodDef call_methods[] = {
    {"R_plaidhip_session", (DL_FUNC)&R_plaidhip_session, 2},
    {"R_plaidhip_plaid_dense", (DL_FUNC)&R_plaidhip_plaid_dense, 5},
    {"R_plaidhip_plaid_csc", (DL_FUNC)&R_plaidhip_plaid_csc, 8},
    {"R_plaidhip_crossprod_weighted_dense", (DL_FUNC)&R_plaidhip_crossprod_weighted_dense, 4},
    {"R_plaidhip_crossprod_weighted_csc", (DL_FUNC)&R_plaidhip_crossprod_weighted_csc, 7},
    {"R_plaidhip_sing_csc", (DL_FUNC)&R_plaidhip_sing_csc, 6},
    {"R_plaidhip_sing_csc_multi", (DL_FUNC)&R_plaidhip_sing_csc_multi, 7},
    {"R_plaidhip_normalize_medians", (DL_FUNC)&R_plaidhip_normalize_medians, 2},
    {"R_plaidhip_colranks_dense", (DL_FUNC)&R_plaidhip_colranks_dense, 3},
    {"R_plaidhip_colranks_csc", (DL_FUNC)&R_plaidhip_colranks_csc, 4},
    {"R_plaidhip_colranks_csc_dense", (DL_FUNC)&R_plaidhip_colranks_csc_dense, 6},
    {"R_plaidhip_plaid_multi", (DL_FUNC)&R_plaidhip_plaid_multi, 10},
    {"R_plaidhip_sing_multi", (DL_FUNC)&R_plaidhip_sing_multi, 4},
    {"R_plaidhip_ssgsea_multi", (DL_FUNC)&R_plaidhip_ssgsea_multi, 9},
    {"R_plaidhip_sing_dense", (DL_FUNC)&R_plaidhip_sing_dense, 3},
    {"R_plaidhip_ssgsea_dense", (DL_FUNC)&R_plaidhip_ssgsea_dense, 4},
    {"R_plaidhip_ssgsea_csc", (DL_FUNC)&R_plaidhip_ssgsea_csc, 7},
    {"R_plaidhip_ucell", (DL_FUNC)&R_plaidhip_ucell, 9},
    {"R_plaidhip_aucell", (DL_FUNC)&R_plaidhip_aucell, 8},
    {"R_plaidhip_scse", (DL_FUNC)&R_plaidhip_scse, 9},
    {"R_plaidhip_gmt2mat_file", (DL_FUNC)&R_plaidhip_gmt2mat_file, 6},
    {"R_plaidhip_plaid_test", (DL_FUNC)&R_plaidhip_plaid_test, 7},
    {"R_plaidhip_gsva", (DL_FUNC)&R_plaidhip_gsva, 5},
    {NULL, NULL, 0}};

void R_init_plaidhip(DllInfo* dll) {
  R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}

void R_unload_plaidhip(DllInfo* dll) {
  (void)dll;
  if (g_ctx) { plaidhip_finalize(g_ctx); g_ctx = NULL; }
  plaidhip_multi_finalize();
}
