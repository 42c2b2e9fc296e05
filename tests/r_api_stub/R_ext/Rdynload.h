/* TEST-ONLY stand-in for <R_ext/Rdynload.h> (see ../Rinternals.h): the registration types of
 * "Writing R Extensions" 5.4 */
#ifndef PLAIDHIP_TEST_RDYNLOAD_H
#define PLAIDHIP_TEST_RDYNLOAD_H
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct { const char* name; DL_FUNC fun; int numArgs; void* types; } R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo*, const R_CMethodDef*, const R_CallMethodDef*, const R_FortranMethodDef*, const R_ExternalMethodDef*);
int R_useDynamicSymbols(DllInfo*, int);
#endif
