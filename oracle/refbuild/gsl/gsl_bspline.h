/* TEST INFRASTRUCTURE -- a stand-in for <gsl/gsl_bspline.h>, NOT GSL.
 *
 * The reference translation unit needs the GNU Scientific Library (module GSL, version unpinned: Makefile:7-8 takes whatever
 * `pkg-config gsl` finds) at two places: MidlineShapes::integrateBSpline (main.cpp:11936-11963: uniform cubic B-spline basis,
 * gsl_bspline_alloc(4, n - 2) -> n basis functions) and Obstacle::computeVelocities (13015-13029, see gsl_linalg.h).  GSL's source is
 * not under /root/reference and is not installed here, so this header restates the PUBLISHED algorithm behind the five B-spline
 * entry points the TU names: the Cox - de Boor recurrence on a clamped uniform knot vector (de Boor, "A Practical Guide to
 * Splines", BSPLVB), with GSL's documented conventions --
 *   gsl_bspline_alloc(k, nbreak)      order k (4 = cubic), nbreak breakpoints, n = nbreak + k - 2 basis functions, n + k knots;
 *   gsl_bspline_knots_uniform(a, b)   knots: a (k times), a + i (b - a) / (nbreak - 1) for i = 1 .. nbreak - 2, b (k times);
 *   gsl_bspline_eval(x, B)            B[0..n) = the n basis functions at x (at most k of them non-zero, summing to 1); x == b
 *                                     belongs to the last interval (B[n-1] = 1).
 * Nothing here is pinned against a real GSL ("parity unpinned" at this boundary): both binaries of a fish comparison
 * (ref_tool and ref_tool_hip) link this same header, so the fish geometry is the same on both sides by construction and only the
 * hot-path operators differ.
 */
#ifndef CUP3D_ORACLE_GSL_BSPLINE_STUB_H
#define CUP3D_ORACLE_GSL_BSPLINE_STUB_H
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
typedef struct { size_t size; double *data; } gsl_vector;
typedef struct {
  size_t k;       /* order */
  size_t nbreak;  /* breakpoints */
  size_t n;       /* basis functions */
  double *knots;  /* n + k */
  double *dl, *dr, *b; /* k each: BSPLVB scratch */
} gsl_bspline_workspace;
static inline gsl_bspline_workspace *gsl_bspline_alloc(size_t k, size_t nbreak) {
  if (k < 1 || nbreak < 2) { fprintf(stderr, "gsl_bspline_alloc stand-in: bad order / breakpoints\n"); abort(); }
  gsl_bspline_workspace *w = (gsl_bspline_workspace *)malloc(sizeof(gsl_bspline_workspace));
  w->k = k;
  w->nbreak = nbreak;
  w->n = nbreak + k - 2;
  w->knots = (double *)calloc(w->n + k, sizeof(double));
  w->dl = (double *)calloc(k, sizeof(double));
  w->dr = (double *)calloc(k, sizeof(double));
  w->b = (double *)calloc(k, sizeof(double));
  return w;
}
static inline gsl_vector *gsl_vector_alloc(size_t n) {
  gsl_vector *v = (gsl_vector *)malloc(sizeof(gsl_vector));
  v->size = n;
  v->data = (double *)calloc(n, sizeof(double));
  return v;
}
static inline int gsl_bspline_knots_uniform(double a, double b, gsl_bspline_workspace *w) {
  const size_t k = w->k, l = w->nbreak - 1;
  const double delta = (b - a) / (double)l;
  size_t i;
  for (i = 0; i < k; i++) w->knots[i] = a;
  for (i = 0; i + 1 < l; i++) w->knots[k + i] = a + (double)(i + 1) * delta;
  for (i = w->n; i < w->n + k; i++) w->knots[i] = b;
  return 0;
}
static inline int gsl_bspline_eval(double x, gsl_vector *B, gsl_bspline_workspace *w) {
  const size_t k = w->k, n = w->n;
  const double *t = w->knots;
  size_t i, j, r;
  if (B->size != n) { fprintf(stderr, "gsl_bspline_eval stand-in: vector of length %zu, %zu basis functions\n", B->size, n); abort(); }
  /* interval i with t[i] <= x < t[i+1], k - 1 <= i <= n - 1; the right end point belongs to the last interval; x outside [a, b]
     is clamped to the nearest interval (GSL raises an error there; the reference never leaves [0, len]) */
  i = k - 1;
  while (i + 1 < n && x >= t[i + 1]) i++;
  /* BSPLVB: the k basis functions that are non-zero on interval i */
  w->b[0] = 1.0;
  for (j = 0; j + 1 < k; j++) {
    double saved = 0.0;
    w->dr[j] = t[i + j + 1] - x;
    w->dl[j] = x - t[i - j];
    for (r = 0; r <= j; r++) {
      const double term = w->b[r] / (w->dr[r] + w->dl[j - r]);
      w->b[r] = saved + w->dr[r] * term;
      saved = w->dl[j - r] * term;
    }
    w->b[j + 1] = saved;
  }
  for (j = 0; j < n; j++) B->data[j] = 0.0;
  for (j = 0; j < k; j++) B->data[i - (k - 1) + j] = w->b[j];
  return 0;
}
static inline double gsl_vector_get(const gsl_vector *v, size_t i) { return v->data[i]; }
static inline void gsl_bspline_free(gsl_bspline_workspace *w) {
  if (w) { free(w->knots); free(w->dl); free(w->dr); free(w->b); free(w); }
}
static inline void gsl_vector_free(gsl_vector *v) { if (v) { free(v->data); free(v); } }
#endif
