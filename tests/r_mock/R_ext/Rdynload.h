/* tests/r_mock/R_ext/Rdynload.h — NOT R: see tests/r_mock/R.h. */
#ifndef GFICF_R_MOCK_RDYNLOAD_H
#define GFICF_R_MOCK_RDYNLOAD_H
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
typedef struct { const char* name; DL_FUNC fun; int numArgs; void* types; } R_CMethodDef;
typedef R_CallMethodDef R_FortranMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
int R_registerRoutines(DllInfo*, const R_CMethodDef*, const R_CallMethodDef*, const R_FortranMethodDef*, const R_ExternalMethodDef*);
int R_useDynamicSymbols(DllInfo*, int);
#endif
