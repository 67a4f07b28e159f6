/* tests/r_mock/R.h — NOT R.  A minimal stand-in for the declarations of R's C API that
 * integration/gficf_hip_glue.c uses, so that the glue can be syntax- and type-checked in an image without R
 * (tests/test_glue_compile.py: gcc -fsyntax-only -Wall -Wextra -Werror).  It pins nothing about R's behaviour
 * and is never linked or shipped: boundary hygiene only.  Signatures follow R >= 4.0's Rinternals.h. */
#ifndef GFICF_R_MOCK_R_H
#define GFICF_R_MOCK_R_H
#include <stddef.h>
#include <stdlib.h>
void Rprintf(const char*, ...);
char* R_alloc(size_t n, int size);
#endif
