/* tests/r_mock/Rinternals.h — NOT R: see tests/r_mock/R.h. */
#ifndef GFICF_R_MOCK_RINTERNALS_H
#define GFICF_R_MOCK_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned char Rbyte;
typedef enum { FALSE = 0, TRUE } Rboolean;
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define RAWSXP 24
#define R_XLEN_T_MAX 4503599627370496
extern SEXP R_NilValue, R_DimSymbol, R_NamesSymbol;
int TYPEOF(SEXP);
int* INTEGER(SEXP);
double* REAL(SEXP);
Rbyte* RAW(SEXP);
const char* CHAR(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
R_xlen_t XLENGTH(SEXP);
int Rf_length(SEXP);
Rboolean Rf_isMatrix(SEXP);
Rboolean Rf_isNull(SEXP);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_allocMatrix(unsigned int, int, int);
SEXP Rf_allocVector(unsigned int, R_xlen_t);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
int Rf_asLogical(SEXP);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
SEXP Rf_ScalarReal(double);
SEXP Rf_install(const char*);
SEXP Rf_mkChar(const char*);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP R_do_slot(SEXP, SEXP);
void Rf_error(const char*, ...) __attribute__((noreturn));
#endif
