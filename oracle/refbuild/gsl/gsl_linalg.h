/* Test-infrastructure stub (NOT GSL): symbols named at main.cpp:13015-13029. */
#ifndef CUP3D_ORACLE_GSL_LINALG_STUB_H
#define CUP3D_ORACLE_GSL_LINALG_STUB_H
#include "gsl_bspline.h"
typedef struct { size_t size1, size2; double *data; } gsl_matrix;
typedef struct { gsl_matrix matrix; } gsl_matrix_view;
typedef struct { gsl_vector vector; } gsl_vector_view;
typedef struct { size_t size; size_t *data; } gsl_permutation;
static inline gsl_matrix_view gsl_matrix_view_array(double *, size_t, size_t) { CUP3D_GSL_STUB_DIE("gsl_matrix_view_array"); gsl_matrix_view v = {}; return v; }
static inline gsl_vector_view gsl_vector_view_array(double *, size_t) { CUP3D_GSL_STUB_DIE("gsl_vector_view_array"); gsl_vector_view v = {}; return v; }
static inline gsl_permutation *gsl_permutation_alloc(size_t) { CUP3D_GSL_STUB_DIE("gsl_permutation_alloc"); return 0; }
static inline void gsl_permutation_free(gsl_permutation *) { CUP3D_GSL_STUB_DIE("gsl_permutation_free"); }
static inline int gsl_linalg_LU_decomp(gsl_matrix *, gsl_permutation *, int *) { CUP3D_GSL_STUB_DIE("gsl_linalg_LU_decomp"); return 0; }
static inline int gsl_linalg_LU_solve(const gsl_matrix *, const gsl_permutation *, const gsl_vector *, gsl_vector *) { CUP3D_GSL_STUB_DIE("gsl_linalg_LU_solve"); return 0; }
#endif
