/* Test-infrastructure stand-in (NOT GSL): the symbols named at main.cpp:13015-13029.  A plain dense LU with partial pivoting
 * (not GSL's code; same mathematics) so that Obstacle::computeVelocities can run for the SYNTHETIC obstacle of the drop-in tests
 * (harness command `obstacle`): the reference binary and the reference-with-HIP-operators binary both link this file, so the
 * comparison between them is unaffected by it.  Nothing here is pinned against GSL ("parity unpinned" at that boundary). */
#ifndef CUP3D_ORACLE_GSL_LINALG_STUB_H
#define CUP3D_ORACLE_GSL_LINALG_STUB_H
#include <math.h>
#include "gsl_bspline.h"
typedef struct { size_t size1, size2; double *data; } gsl_matrix;
typedef struct { gsl_matrix matrix; } gsl_matrix_view;
typedef struct { gsl_vector vector; } gsl_vector_view;
typedef struct { size_t size; size_t *data; } gsl_permutation;
static inline gsl_matrix_view gsl_matrix_view_array(double *a, size_t n1, size_t n2) { gsl_matrix_view v = {{n1, n2, a}}; return v; }
static inline gsl_vector_view gsl_vector_view_array(double *a, size_t n) { gsl_vector_view v = {{n, a}}; return v; }
static inline gsl_permutation *gsl_permutation_alloc(size_t n) {
  gsl_permutation *p = (gsl_permutation *)malloc(sizeof(gsl_permutation));
  p->size = n;
  p->data = (size_t *)malloc(n * sizeof(size_t));
  for (size_t i = 0; i < n; i++) p->data[i] = i;
  return p;
}
static inline void gsl_permutation_free(gsl_permutation *p) { if (p) { free(p->data); free(p); } }
/* in place: A <- L\U of P A, row-major n x n; p->data[i] = source row of row i */
static inline int gsl_linalg_LU_decomp(gsl_matrix *A, gsl_permutation *p, int *signum) {
  const size_t n = A->size1;
  double *a = A->data;
  *signum = 1;
  for (size_t j = 0; j + 1 < n; j++) {
    size_t piv = j;
    double big = fabs(a[j * n + j]);
    for (size_t i = j + 1; i < n; i++)
      if (fabs(a[i * n + j]) > big) { big = fabs(a[i * n + j]); piv = i; }
    if (piv != j) {
      for (size_t k = 0; k < n; k++) { const double t = a[j * n + k]; a[j * n + k] = a[piv * n + k]; a[piv * n + k] = t; }
      const size_t t = p->data[j]; p->data[j] = p->data[piv]; p->data[piv] = t;
      *signum = -*signum;
    }
    if (a[j * n + j] != 0.0)
      for (size_t i = j + 1; i < n; i++) {
        const double l = a[i * n + j] / a[j * n + j];
        a[i * n + j] = l;
        for (size_t k = j + 1; k < n; k++) a[i * n + k] -= l * a[j * n + k];
      }
  }
  return 0;
}
static inline int gsl_linalg_LU_solve(const gsl_matrix *LU, const gsl_permutation *p, const gsl_vector *b, gsl_vector *x) {
  const size_t n = LU->size1;
  const double *a = LU->data;
  for (size_t i = 0; i < n; i++) x->data[i] = b->data[p->data[i]];
  for (size_t i = 0; i < n; i++)
    for (size_t k = 0; k < i; k++) x->data[i] -= a[i * n + k] * x->data[k];
  for (size_t ii = n; ii-- > 0;) {
    for (size_t k = ii + 1; k < n; k++) x->data[ii] -= a[ii * n + k] * x->data[k];
    x->data[ii] /= a[ii * n + ii];
  }
  return 0;
}
#endif
