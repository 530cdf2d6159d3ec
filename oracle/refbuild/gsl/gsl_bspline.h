/* Test-infrastructure stub (NOT GSL): declares only the symbols the reference
 * translation unit names at main.cpp:11936-11963 so that it compiles for
 * obstacle-free configurations.  The B-spline entry points abort when called; gsl_vector_alloc/get/free are functional
 * (plain malloc-backed) because the 6x6 solve of Obstacle::computeVelocities (13015-13029) uses them, see gsl_linalg.h. */
#ifndef CUP3D_ORACLE_GSL_BSPLINE_STUB_H
#define CUP3D_ORACLE_GSL_BSPLINE_STUB_H
#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>
#define CUP3D_GSL_STUB_DIE(name)                                               \
  do {                                                                         \
    fprintf(stderr, "oracle/_ref: GSL stub '%s' called (fish configs are out "  \
                    "of scope)\n", name);                                      \
    abort();                                                                   \
  } while (0)
typedef struct { size_t size; double *data; } gsl_vector;
typedef struct { int unused; } gsl_bspline_workspace;
static inline gsl_bspline_workspace *gsl_bspline_alloc(size_t, size_t) { CUP3D_GSL_STUB_DIE("gsl_bspline_alloc"); return 0; }
static inline gsl_vector *gsl_vector_alloc(size_t n) {
  gsl_vector *v = (gsl_vector *)malloc(sizeof(gsl_vector));
  v->size = n;
  v->data = (double *)calloc(n, sizeof(double));
  return v;
}
static inline int gsl_bspline_knots_uniform(double, double, gsl_bspline_workspace *) { CUP3D_GSL_STUB_DIE("gsl_bspline_knots_uniform"); return 0; }
static inline int gsl_bspline_eval(double, gsl_vector *, gsl_bspline_workspace *) { CUP3D_GSL_STUB_DIE("gsl_bspline_eval"); return 0; }
static inline double gsl_vector_get(const gsl_vector *v, size_t i) { return v->data[i]; }
static inline void gsl_bspline_free(gsl_bspline_workspace *) { CUP3D_GSL_STUB_DIE("gsl_bspline_free"); }
static inline void gsl_vector_free(gsl_vector *v) { if (v) { free(v->data); free(v); } }
#endif
