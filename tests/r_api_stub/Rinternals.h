/* TEST-ONLY declarations of the part of R's C API that r-pkg/src/plaidhip_R.c uses, written from R's documented
 * interface ("Writing R Extensions", sections 5.9 and 5.4) so that the shim can be put through a C compiler
 * (gcc -fsyntax-only -Wall -Wextra) in a container without R.  Signatures only: nothing here is linked or run, and a
 * real build uses R's own headers. */
#ifndef PLAIDHIP_TEST_RINTERNALS_H
#define PLAIDHIP_TEST_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int Rboolean;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
typedef unsigned int SEXPTYPE;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
typedef enum { CE_NATIVE = 0, CE_UTF8 = 1, CE_LATIN1 = 2, CE_BYTES = 3 } cetype_t;
extern SEXP R_NilValue;
extern int R_NaInt;
#define NA_LOGICAL R_NaInt
#define NA_INTEGER R_NaInt
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
int* INTEGER(SEXP);
double* REAL(SEXP);
int* LOGICAL(SEXP);
int LENGTH(SEXP);
R_xlen_t XLENGTH(SEXP);
const char* CHAR(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
double Rf_asReal(SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
Rboolean Rf_isNull(SEXP);
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_duplicate(SEXP);
SEXP Rf_ScalarLogical(int);
SEXP Rf_install(const char*);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_mkCharLenCE(const char*, int, cetype_t);
const char* Rf_translateCharUTF8(SEXP);
#if defined(__GNUC__)
void Rf_error(const char*, ...) __attribute__((noreturn, format(printf, 1, 2)));
#else
void Rf_error(const char*, ...);
#endif
char* R_alloc(size_t, int);
#endif
