/* TEST INFRASTRUCTURE — CPU restatement of the reference's ghost reconstruction on
 * multi-level (AMR) block meshes: BlockLab::load / post_load with SameLevelExchange,
 * FineToCoarseExchange (restrict: 8-cell AverageDown), CoarseFineExchange + FillCoarseVersion
 * (coarse shadow tile) and CoarseFineInterpolation (prolong: TestInterp and the
 * finite-difference mode), slitvinov/CUP3D main.cpp:3623-4614, plus the domain-face rules on
 * both tiles (5929-6004, 6107-6503).  Pinned against the reference's own tiles dumped by
 * oracle/_ref/ref_tool (`lab` command) in tests/test_oracle_amr.py.  The product never links it.
 *
 * Conventions: a mesh is a list of leaf blocks (level, Z) sorted by blockID_2; fields are the
 * reference's block memory [nb][8][8][8][nc].  The fine tile F covers fine-cell coordinates
 * [s, 8+e-1)^3 relative to the block; the coarse shadow tile C covers coarse-cell coordinates
 * [off, 4+eC-1)^3 (coarse cell X spans fine cells 2X, 2X+1), off = (s-1)/2 - 1, eC = e/2 + 2.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cup3d_oracle.h"

#define BS 8
#define HB 4
#define BS3 512

struct orc_mesh {
  orc_sfc *sfc;
  int bpd[3], level_max, bc[3];
  double maxextent;
  long nblocks;
  int *level;
  long long *Z, *id2;
  int *index;
  int **slot_at; /* [level][(k*ny+j)*nx+i] -> slot or -1 */
};

typedef struct { long long id2; int level; long long Z; } mrec;
static int cmp_mrec(const void *a, const void *b) {
  const mrec *x = (const mrec *)a, *y = (const mrec *)b;
  return (x->id2 > y->id2) - (x->id2 < y->id2);
}

orc_mesh *orc_mesh_create(int bx, int by, int bz, int level_max, double maxextent, const int bc[3], long nblocks,
                          const int *levels, const long long *Zs) {
  orc_mesh *m = (orc_mesh *)calloc(1, sizeof *m);
  m->sfc = orc_sfc_create(bx, by, bz, level_max);
  m->bpd[0] = bx; m->bpd[1] = by; m->bpd[2] = bz;
  m->level_max = level_max; m->maxextent = maxextent; m->nblocks = nblocks;
  for (int d = 0; d < 3; d++) m->bc[d] = bc[d];
  m->level = (int *)malloc(nblocks * sizeof(int));
  m->Z = (long long *)malloc(nblocks * sizeof(long long));
  m->id2 = (long long *)malloc(nblocks * sizeof(long long));
  m->index = (int *)malloc(nblocks * 3 * sizeof(int));
  mrec *r = (mrec *)malloc(nblocks * sizeof(mrec));
  for (long i = 0; i < nblocks; i++) {
    int c[3];
    orc_sfc_inverse(m->sfc, Zs[i], levels[i], c);
    r[i].level = levels[i]; r[i].Z = Zs[i]; r[i].id2 = orc_sfc_encode(m->sfc, levels[i], c);
  }
  qsort(r, nblocks, sizeof(mrec), cmp_mrec);
  m->slot_at = (int **)calloc(level_max, sizeof(int *));
  for (int l = 0; l < level_max; l++) {
    const long n = (long)(bx << l) * (by << l) * (bz << l);
    m->slot_at[l] = (int *)malloc(n * sizeof(int));
    for (long i = 0; i < n; i++) m->slot_at[l][i] = -1;
  }
  for (long s = 0; s < nblocks; s++) {
    m->level[s] = r[s].level; m->Z[s] = r[s].Z; m->id2[s] = r[s].id2;
    orc_sfc_inverse(m->sfc, r[s].Z, r[s].level, &m->index[3 * s]);
    const int l = r[s].level, *c = &m->index[3 * s];
    m->slot_at[l][((long)c[2] * (by << l) + c[1]) * (bx << l) + c[0]] = (int)s;
  }
  free(r);
  return m;
}
void orc_mesh_destroy(orc_mesh *m) {
  if (!m) return;
  for (int l = 0; l < m->level_max; l++) free(m->slot_at[l]);
  free(m->slot_at); orc_sfc_destroy(m->sfc); free(m->level); free(m->Z); free(m->id2); free(m->index); free(m);
}
long orc_mesh_nblocks(const orc_mesh *m) { return m->nblocks; }
void orc_mesh_tables(const orc_mesh *m, long long *o) {
  for (long s = 0; s < m->nblocks; s++) {
    o[6 * s] = m->level[s]; o[6 * s + 1] = m->Z[s];
    for (int d = 0; d < 3; d++) o[6 * s + 2 + d] = m->index[3 * s + d];
    o[6 * s + 5] = m->id2[s];
  }
}
double orc_mesh_h(const orc_mesh *m, long b) {
  int mb = m->bpd[0] > m->bpd[1] ? m->bpd[0] : m->bpd[1];
  if (m->bpd[2] > mb) mb = m->bpd[2];
  return m->maxextent / (double)(mb * BS) / (double)(1 << m->level[b]);
}

static int nblk(const orc_mesh *m, int l, int d) { return m->bpd[d] << l; }
#ifdef _OPENMP
#include <omp.h>
static int mesh_threads(const orc_mesh *m) { /* small meshes: a team per 8 blocks at most (big hosts have 256 cores) */
  long t = m->nblocks / 8, mx = omp_get_max_threads();
  if (t < 1) t = 1;
  return (int)(t < mx ? t : mx);
}
#else
static int mesh_threads(const orc_mesh *m) { (void)m; return 1; }
#endif
/* leaf slot at (level l, block index c) with periodic wrap; -1 if not a leaf at that level */
static int leaf_at(const orc_mesh *m, int l, const int c[3]) {
  if (l < 0 || l >= m->level_max) return -1;
  int w[3];
  for (int d = 0; d < 3; d++) { const int n = nblk(m, l, d); w[d] = ((c[d] % n) + n) % n; }
  return m->slot_at[l][((long)w[2] * nblk(m, l, 1) + w[1]) * nblk(m, l, 0) + w[0]];
}
/* TreePosition of (l, c): 1 Exists, -2 CheckCoarser, -1 CheckFiner (main.cpp:321-330) */
static int tree_state(const orc_mesh *m, int l, const int c[3]) {
  if (leaf_at(m, l, c) >= 0) return 1;
  int w[3];
  for (int d = 0; d < 3; d++) { const int n = nblk(m, l, d); w[d] = ((c[d] % n) + n) % n; }
  for (int k = 1; k <= l; k++) {
    const int p[3] = {w[0] >> k, w[1] >> k, w[2] >> k};
    if (leaf_at(m, l - k, p) >= 0) return -2;
  }
  return -1;
}

/* ------------------------------------------------------------------ tiles */
typedef struct {
  int nc, is_vector, s, e, tens, use_avg; /* stencil [s,e)^3 */
  int L, off, eC, Lc;
  double *F, *C;
} tile_t;
#define FT(t, x, y, z, c) (t)->F[((((long)(z) - (t)->s) * (t)->L + ((y) - (t)->s)) * (t)->L + ((x) - (t)->s)) * (t)->nc + (c)]
#define CT(t, x, y, z, c) (t)->C[((((long)(z) - (t)->off) * (t)->Lc + ((y) - (t)->off)) * (t)->Lc + ((x) - (t)->off)) * (t)->nc + (c)]

static void tile_init(tile_t *t, int nc, int is_vector, int s, int e, int tens) {
  t->nc = nc; t->is_vector = is_vector; t->s = s; t->e = e; t->tens = tens;
  t->L = BS + e - s - 1;
  t->off = (s - 1) / 2 + (-1);        /* main.cpp:3596 */
  t->eC = e / 2 + 1 + 2 - 1;          /* main.cpp:3599 */
  t->Lc = HB + t->eC - t->off - 1;
  t->use_avg = tens || s < -2 || e > 3; /* main.cpp:3618-3621 (FiniteDifferences is always true) */
  t->F = (double *)calloc((size_t)t->L * t->L * t->L * nc, sizeof(double));
  t->C = (double *)calloc((size_t)t->Lc * t->Lc * t->Lc * nc, sizeof(double));
}
static void tile_free(tile_t *t) { free(t->F); free(t->C); }

static inline double avg_down(double e0, double e1, double e2, double e3, double e4, double e5, double e6, double e7) {
  return 0.125 * (e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7); /* AverageDown, main.cpp:3877-3882 */
}
/* value of cell (gx,gy,gz) (global cell coordinates of level l, periodic wrap) if a level-l leaf holds it */
static int cell_value(const orc_mesh *m, const double *f, int nc, int l, const long g[3], int c, double *out) {
  int b[3], q[3];
  for (int d = 0; d < 3; d++) {
    const long n = (long)nblk(m, l, d) * BS;
    const long w = ((g[d] % n) + n) % n;
    b[d] = (int)(w / BS); q[d] = (int)(w % BS);
  }
  const int slot = leaf_at(m, l, b);
  if (slot < 0) return 0;
  *out = f[((long)slot * BS3 + (q[2] * BS + q[1]) * BS + q[0]) * nc + c];
  return 1;
}

/* domain-face passes on the fine (coarse=0) or coarse shadow (coarse=1) tile, in the reference's order;
 * each fills the whole ghost slab behind the face from the face cell with equal transverse coordinates */
static void apply_bc(const orc_mesh *m, long b, tile_t *t, int coarse) {
  const int l = m->level[b], *idx = &m->index[3 * b];
  const int lo = coarse ? t->off : t->s, hi = coarse ? HB + t->eC - 1 : BS + t->e - 1, bs = coarse ? HB : BS;
  for (int d = 0; d < 3; d++) {
    if (m->bc[d] == ORC_BC_PERIODIC) continue;
    for (int side = 0; side < 2; side++) {
      if (side == 0 ? idx[d] != 0 : idx[d] != nblk(m, l, d) - 1) continue;
      const int d1 = (d + 1) % 3, d2 = (d + 2) % 3, face = side ? bs - 1 : 0;
      const int g0 = side ? bs : lo, g1 = side ? hi : 0;
      for (int a2 = lo; a2 < hi; a2++)
        for (int a1 = lo; a1 < hi; a1++)
          for (int gg = g0; gg < g1; gg++) {
            int p[3], q[3];
            p[d] = gg; q[d] = face; p[d1] = q[d1] = a1; p[d2] = q[d2] = a2;
            for (int c = 0; c < t->nc; c++) {
              double v = coarse ? CT(t, q[0], q[1], q[2], c) : FT(t, q[0], q[1], q[2], c);
              if (t->is_vector) { /* 1: vector element; 2 + k: scalar element of BlockLabBC<.., direction = k> (6120, 6384-6394) */
                const int ceff = t->is_vector == 1 ? c : t->is_vector - 2;
                if (m->bc[d] == ORC_BC_WALL) v = (-1.0) * v;
                else if (ceff == d) v = (-1.) * v;
              }
              if (coarse) CT(t, p[0], p[1], p[2], c) = v; else FT(t, p[0], p[1], p[2], c) = v;
            }
          }
    }
  }
}

/* TestInterp, main.cpp:3883-3906: 8 fine values R[x+2y+4z] from the 3x3x3 coarse neighbourhood around (X,Y,Z) */
static void test_interp(const tile_t *t, int X, int Y, int Z, int c, double R[8]) {
#define Cc(i, j, k) CT(t, X - 1 + (i), Y - 1 + (j), Z - 1 + (k), c)
  const double dudx = 0.125 * (Cc(2, 1, 1) - Cc(0, 1, 1));
  const double dudy = 0.125 * (Cc(1, 2, 1) - Cc(1, 0, 1));
  const double dudz = 0.125 * (Cc(1, 1, 2) - Cc(1, 1, 0));
  const double dudxdy = 0.015625 * (Cc(0, 0, 1) + Cc(2, 2, 1) - Cc(2, 0, 1) - Cc(0, 2, 1));
  const double dudxdz = 0.015625 * (Cc(0, 1, 0) + Cc(2, 1, 2) - Cc(2, 1, 0) - Cc(0, 1, 2));
  const double dudydz = 0.015625 * (Cc(1, 0, 0) + Cc(1, 2, 2) - Cc(1, 2, 0) - Cc(1, 0, 2));
  const double lap = Cc(1, 1, 1) + 0.03125 * (Cc(0, 1, 1) + Cc(2, 1, 1) + Cc(1, 0, 1) + Cc(1, 2, 1) + Cc(1, 1, 0) + Cc(1, 1, 2) + (-6.0) * Cc(1, 1, 1));
#undef Cc
  R[0] = lap - dudx - dudy - dudz + dudxdy + dudxdz + dudydz;
  R[1] = lap + dudx - dudy - dudz - dudxdy - dudxdz + dudydz;
  R[2] = lap - dudx + dudy - dudz - dudxdy + dudxdz - dudydz;
  R[3] = lap + dudx + dudy - dudz + dudxdy - dudxdz - dudydz;
  R[4] = lap - dudx - dudy + dudz + dudxdy - dudxdz - dudydz;
  R[5] = lap + dudx - dudy + dudz - dudxdy + dudxdz - dudydz;
  R[6] = lap - dudx + dudy + dudz - dudxdy - dudxdz + dudydz;
  R[7] = lap + dudx + dudy + dudz + dudxdy + dudxdz + dudydz;
}

static const double d_coef_plus[9] = {-0.09375, 0.4375, 0.15625, 0.15625, -0.5625, 0.90625, -0.09375, 0.4375, 0.15625};  /* 3485-3488 */
static const double d_coef_minus[9] = {0.15625, -0.5625, 0.90625, -0.09375, 0.4375, 0.15625, 0.15625, 0.4375, -0.09375};

/* 1-D quadratic interpolation along axis `ax` at coarse position P (coordinates in C), main.cpp:4419-4441 */
static double interp1d(const tile_t *t, const int P[3], int ax, const double *coef, int inner, int start, int c, int *pm) {
  int A[3] = {P[0], P[1], P[2]}, B[3] = {P[0], P[1], P[2]};
  double r;
  if (inner) {
    A[ax] = P[ax] - 1; B[ax] = P[ax] + 1;
    r = (coef[6] * CT(t, A[0], A[1], A[2], c) + coef[8] * CT(t, B[0], B[1], B[2], c)) + coef[7] * CT(t, P[0], P[1], P[2], c);
    pm[0] = P[ax] + 1; pm[1] = P[ax] - 1;
  } else if (start) {
    A[ax] = P[ax] + 2; B[ax] = P[ax] + 1;
    r = (coef[0] * CT(t, A[0], A[1], A[2], c) + coef[1] * CT(t, B[0], B[1], B[2], c)) + coef[2] * CT(t, P[0], P[1], P[2], c);
    pm[0] = P[ax] + 1; pm[1] = P[ax];
  } else {
    A[ax] = P[ax] - 2; B[ax] = P[ax] - 1;
    r = (coef[3] * CT(t, A[0], A[1], A[2], c) + coef[4] * CT(t, B[0], B[1], B[2], c)) + coef[5] * CT(t, P[0], P[1], P[2], c);
    pm[0] = P[ax]; pm[1] = P[ax] - 1;
  }
  return r;
}

/* BlockLab::load + post_load for block b: fills t->F (and t->C) */
void orc_mesh_lab(const orc_mesh *m, const double *f, long b, tile_t *t) {
  const int nc = t->nc, l = m->level[b], *idx = &m->index[3 * b];
  const int s = t->s, e = t->e;
  int N[3], skin[3], skip[3];
  for (int d = 0; d < 3; d++) {
    N[d] = nblk(m, l, d);
    skin[d] = idx[d] == 0 || idx[d] == N[d] - 1;
    skip[d] = idx[d] == 0 ? -1 : 1;
  }
  const double *own = f + (long)b * BS3 * nc;
  for (int z = 0; z < BS; z++)
    for (int y = 0; y < BS; y++)
      for (int x = 0; x < BS; x++)
        for (int c = 0; c < nc; c++) FT(t, x, y, z, c) = own[((z * BS + y) * BS + x) * nc + c];
  int same_codes[26], nsame = 0, coarse_codes[27], ncoarse = 0, coarsened = 0;
  for (int icode = 0; icode < 27; icode++) {
    if (icode == 13) continue;
    const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, icode / 9 - 1};
    int skipped = 0;
    for (int d = 0; d < 3; d++)
      if (m->bc[d] != ORC_BC_PERIODIC && code[d] == skip[d] && skin[d]) skipped = 1;
    if (skipped) continue;
    const int nei[3] = {idx[0] + code[0], idx[1] + code[1], idx[2] + code[2]};
    const int st = tree_state(m, l, nei);
    if (st == 1) same_codes[nsame++] = icode;
    else if (st == -2) {
      coarse_codes[ncoarse++] = icode;
      /* CoarseFineExchange, main.cpp:4066-4170: the coarse shadow region behind this code, from the coarser leaf */
      int sC[3], eCc[3];
      for (int d = 0; d < 3; d++) {
        sC[d] = code[d] < 1 ? (code[d] < 0 ? t->off : 0) : HB;
        eCc[d] = code[d] < 1 ? (code[d] < 0 ? 0 : HB) : HB + e / 2 + 2 - 1;
      }
      for (int Zc = sC[2]; Zc < eCc[2]; Zc++)
        for (int Yc = sC[1]; Yc < eCc[1]; Yc++)
          for (int Xc = sC[0]; Xc < eCc[0]; Xc++) {
            const int P[3] = {Xc, Yc, Zc};
            long g[3];
            for (int d = 0; d < 3; d++) g[d] = (long)(idx[d] >> 1) * BS + (idx[d] & 1) * HB + P[d];
            /* negative block indices: floor semantics of idx>>1 are fine since idx >= 0 */
            for (int c = 0; c < nc; c++) {
              double v;
              if (cell_value(m, f, nc, l - 1, g, c, &v)) CT(t, Xc, Yc, Zc, c) = v;
            }
          }
    }
    if (!t->tens && !t->use_avg && abs(code[0]) + abs(code[1]) + abs(code[2]) > 1) continue;
    int sF[3], eF[3];
    for (int d = 0; d < 3; d++) {
      sF[d] = code[d] < 1 ? (code[d] < 0 ? s : 0) : BS;
      eF[d] = code[d] < 1 ? (code[d] < 0 ? 0 : BS) : BS + e - 1;
    }
    if (st == 1) { /* SameLevelExchange, 3823-3876 */
      const double *nb = f + (long)leaf_at(m, l, nei) * BS3 * nc;
      for (int z = sF[2]; z < eF[2]; z++)
        for (int y = sF[1]; y < eF[1]; y++)
          for (int x = sF[0]; x < eF[0]; x++)
            for (int c = 0; c < nc; c++)
              FT(t, x, y, z, c) = nb[(((z - code[2] * BS) * BS + (y - code[1] * BS)) * BS + (x - code[0] * BS)) * nc + c];
    } else if (st == -1) { /* FineToCoarseExchange, 3907-4065: 8-cell mean of the finer leaves */
      for (int z = sF[2]; z < eF[2]; z++)
        for (int y = sF[1]; y < eF[1]; y++)
          for (int x = sF[0]; x < eF[0]; x++)
            for (int c = 0; c < nc; c++) {
              double v[8];
              int ok = 1;
              for (int q = 0; q < 8 && ok; q++) { /* e0..e7: (dx,dy,dz) = (q>>2, (q>>1)&1, q&1) */
                const long g[3] = {2 * ((long)idx[0] * BS + x) + (q >> 2), 2 * ((long)idx[1] * BS + y) + ((q >> 1) & 1), 2 * ((long)idx[2] * BS + z) + (q & 1)};
                ok = cell_value(m, f, nc, l + 1, g, c, &v[q]);
              }
              if (ok) FT(t, x, y, z, c) = avg_down(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            }
    }
  }
  /* FillCoarseVersion for the same-level neighbours that touch a coarsened region, 3726-3738, 4171-4235 */
  if (ncoarse > 0)
    for (int i = 0; i < nsame; i++) {
      const int icode = same_codes[i];
      const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, icode / 9 - 1};
      const int bidx[3] = {(idx[0] + code[0] + N[0]) % N[0], (idx[1] + code[1] + N[1]) % N[1], (idx[2] + code[2] + N[2]) % N[2]};
      /* UseCoarseStencil, 3788-3822 */
      int use = 0;
      if (l != 0 && t->use_avg) {
        int imin[3], imax[3];
        for (int d = 0; d < 3; d++) {
          imin[d] = idx[d] < bidx[d] ? 0 : -1;
          imax[d] = idx[d] > bidx[d] ? 0 : +1;
          if (m->bc[d] == ORC_BC_PERIODIC) {
            if (idx[d] == 0 && bidx[d] == N[d] - 1) imin[d] = -1;
            if (bidx[d] == 0 && idx[d] == N[d] - 1) imax[d] = +1;
          } else {
            if (idx[d] == 0 && bidx[d] == 0) imin[d] = 0;
            if (idx[d] == N[d] - 1 && bidx[d] == N[d] - 1) imax[d] = 0;
          }
        }
        for (int it = 0; it < ncoarse && !use; it++)
          for (int i2 = imin[2]; i2 <= imax[2] && !use; i2++)
            for (int i1 = imin[1]; i1 <= imax[1] && !use; i1++)
              for (int i0 = imin[0]; i0 <= imax[0] && !use; i0++)
                if (coarse_codes[it] == (i0 + 1) + 3 * (i1 + 1) + 9 * (i2 + 1)) use = 1;
      }
      if (!use) continue;
      coarsened = 1;
      /* only performed when SameLevelExchange loaded that neighbour (myblocks[icode] != nullptr): with use_avg all 26 are */
      const double *nb = f + (long)leaf_at(m, l, bidx) * BS3 * nc;
      int sC[3], eCc[3], start[3];
      for (int d = 0; d < 3; d++) {
        sC[d] = code[d] < 1 ? (code[d] < 0 ? t->off : 0) : HB;
        eCc[d] = code[d] < 1 ? (code[d] < 0 ? 0 : HB) : HB + (e / 2 + 2) - 1;
        start[d] = sC[d] + (code[d] > 0 ? code[d] : 0) * HB - code[d] * BS + (code[d] < 0 ? code[d] : 0) * (eCc[d] - sC[d]);
      }
      for (int Zc = sC[2]; Zc < eCc[2]; Zc++)
        for (int Yc = sC[1]; Yc < eCc[1]; Yc++) {
          if (code[1] == 0 && code[2] == 0 && Yc > 1 && Yc < HB - 2 && Zc > 1 && Zc < HB - 2) continue; /* never true for 4-cell blocks */
          for (int Xc = sC[0]; Xc < eCc[0]; Xc++) {
            const int XX = start[0] + 2 * (Xc - sC[0]), YY = start[1] + 2 * (Yc - sC[1]), ZZ = start[2] + 2 * (Zc - sC[2]);
#define NB(a, bb, cc, c) nb[((((ZZ) + (cc)) * BS + ((YY) + (bb))) * BS + ((XX) + (a))) * nc + (c)]
            for (int c = 0; c < nc; c++)
              CT(t, Xc, Yc, Zc, c) = avg_down(NB(0, 0, 0, c), NB(0, 0, 1, c), NB(0, 1, 0, c), NB(0, 1, 1, c), NB(1, 0, 0, c), NB(1, 0, 1, c), NB(1, 1, 0, c), NB(1, 1, 1, c));
#undef NB
          }
        }
    }
  /* post_load, 3746-3787 */
  if (coarsened)
    for (int k = 0; k < HB; k++)
      for (int j = 0; j < HB; j++)
        for (int i = 0; i < HB; i++) {
          if (i > 1 && i < HB - 2 && j > 1 && j < HB - 2 && k > 1 && k < HB - 2) continue;
          for (int c = 0; c < nc; c++)
            CT(t, i, j, k, c) = avg_down(FT(t, 2 * i, 2 * j, 2 * k, c), FT(t, 2 * i + 1, 2 * j, 2 * k, c), FT(t, 2 * i, 2 * j + 1, 2 * k, c), FT(t, 2 * i + 1, 2 * j + 1, 2 * k, c),
                                         FT(t, 2 * i, 2 * j, 2 * k + 1, c), FT(t, 2 * i + 1, 2 * j, 2 * k + 1, c), FT(t, 2 * i, 2 * j + 1, 2 * k + 1, c), FT(t, 2 * i + 1, 2 * j + 1, 2 * k + 1, c));
        }
  apply_bc(m, b, t, 1);
  /* CoarseFineInterpolation, 4236-4614 */
  for (int ii = 0; ii < ncoarse; ii++) {
    const int icode = coarse_codes[ii];
    const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, (icode / 9) % 3 - 1};
    if (!t->tens && !t->use_avg && abs(code[0]) + abs(code[1]) + abs(code[2]) > 1) continue;
    int sF[3], eF[3], sC[3];
    for (int d = 0; d < 3; d++) {
      sF[d] = code[d] < 1 ? (code[d] < 0 ? s : 0) : BS;
      eF[d] = code[d] < 1 ? (code[d] < 0 ? 0 : BS) : BS + e - 1;
      sC[d] = code[d] < 1 ? (code[d] < 0 ? ((s - 1) / 2) : 0) : HB;
    }
    if (t->use_avg)
      for (int iz = sF[2]; iz < eF[2]; iz += 2) {
        const int ZZ = (iz - sF[2] - (code[2] < 0 ? code[2] : 0) * ((eF[2] - sF[2]) % 2)) / 2 + sC[2];
        const int izp = (abs(iz) % 2 == 1) ? -1 : 1, rzp = izp == 1 ? 1 : 0, rz = izp == 1 ? 0 : 1;
        for (int iy = sF[1]; iy < eF[1]; iy += 2) {
          const int YY = (iy - sF[1] - (code[1] < 0 ? code[1] : 0) * ((eF[1] - sF[1]) % 2)) / 2 + sC[1];
          const int iyp = (abs(iy) % 2 == 1) ? -1 : 1, ryp = iyp == 1 ? 1 : 0, ry = iyp == 1 ? 0 : 1;
          for (int ix = sF[0]; ix < eF[0]; ix += 2) {
            const int XX = (ix - sF[0] - (code[0] < 0 ? code[0] : 0) * ((eF[0] - sF[0]) % 2)) / 2 + sC[0];
            const int ixp = (abs(ix) % 2 == 1) ? -1 : 1, rxp = ixp == 1 ? 1 : 0, rx = ixp == 1 ? 0 : 1;
            for (int c = 0; c < nc; c++) {
              double R[8];
              test_interp(t, XX, YY, ZZ, c, R);
              for (int q = 0; q < 8; q++) {
                const int px = (q & 1) ? ix + ixp : ix, py = (q & 2) ? iy + iyp : iy, pz = (q & 4) ? iz + izp : iz;
                if (px < sF[0] || px >= eF[0] || py < sF[1] || py >= eF[1] || pz < sF[2] || pz >= eF[2]) continue;
                FT(t, px, py, pz, c) = R[((q & 1) ? rxp : rx) + 2 * ((q & 2) ? ryp : ry) + 4 * ((q & 4) ? rzp : rz)];
              }
            }
          }
        }
      }
    if (abs(code[0]) + abs(code[1]) + abs(code[2]) == 1) {
      int cf[3], mn[3], mx[3];
      for (int d = 0; d < 3; d++) {
        cf[d] = (code[d] < 0 ? code[d] : 0) * ((eF[d] - sF[d]) % 2);
        mn[d] = sF[d] > -2 ? sF[d] : -2;
        mx[d] = eF[d] < BS + 2 ? eF[d] : BS + 2;
      }
      const int ax = code[0] != 0 ? 0 : (code[1] != 0 ? 1 : 2);
      const int a1 = ax == 0 ? 1 : 0, a2 = ax == 2 ? 1 : 2; /* the two tangential axes in the reference's order (y,z) / (x,z) / (x,y) */
      for (int iz = mn[2]; iz < mx[2]; iz++)
        for (int iy = mn[1]; iy < mx[1]; iy++)
          for (int ix = mn[0]; ix < mx[0]; ix++) {
            const int ip[3] = {ix, iy, iz};
            int P[3], par[3], inner[3], start[3];
            double dd[3];
            const double *coef[3];
            for (int d = 0; d < 3; d++) {
              P[d] = (ip[d] - sF[d] - cf[d]) / 2 + sC[d]; /* absolute coarse coordinate (the reference subtracts offset for storage) */
              par[d] = abs(ip[d] - sF[d] - cf[d]) % 2;
              dd[d] = 0.25 * (2 * par[d] - 1);
              coef[d] = dd[d] > 0 ? d_coef_plus : d_coef_minus;
              inner[d] = P[d] != 0 && P[d] != HB - 1;
              start[d] = P[d] == 0;
            }
            for (int c = 0; c < nc; c++) {
              int pm1[2], pm2[2];
              const double x1D = interp1d(t, P, a1, coef[a1], inner[a1], start[a1], c, pm1);
              const double x2D = interp1d(t, P, a2, coef[a2], inner[a2], start[a2], c, pm2);
              double mixed_coef = 1.0;
              if (inner[a1]) mixed_coef *= 0.5;
              if (inner[a2]) mixed_coef *= 0.5;
              int Q[3] = {P[0], P[1], P[2]};
              double vmm, vpp, vpm, vmp;
              Q[a1] = pm1[1]; Q[a2] = pm2[1]; vmm = CT(t, Q[0], Q[1], Q[2], c);
              Q[a1] = pm1[0]; Q[a2] = pm2[0]; vpp = CT(t, Q[0], Q[1], Q[2], c);
              Q[a1] = pm1[0]; Q[a2] = pm2[1]; vpm = CT(t, Q[0], Q[1], Q[2], c);
              Q[a1] = pm1[1]; Q[a2] = pm2[0]; vmp = CT(t, Q[0], Q[1], Q[2], c);
              const double mixed = mixed_coef * dd[a1] * dd[a2] * ((vmm + vpp) - (vpm + vmp));
              double a = (x1D + x2D) + mixed;
              int pb[3], pc[3];
              for (int d = 0; d < 3; d++) {
                pb[d] = ip[d] + (-3 * code[d] + 1) / 2 - par[d] * abs(code[d]);
                pc[d] = ip[d] + (-5 * code[d] + 1) / 2 - par[d] * abs(code[d]);
              }
              const double bv = FT(t, pb[0], pb[1], pb[2], c), cv = FT(t, pc[0], pc[1], pc[2], c);
              const int ccc = code[0] + code[1] + code[2];
              const int xyz = abs(code[0]) * par[0] + abs(code[1]) * par[1] + abs(code[2]) * par[2];
              if (ccc == 1) a = (xyz == 0) ? (1.0 / 15.0) * (8.0 * a + (10.0 * bv - 3.0 * cv)) : (1.0 / 15.0) * (24.0 * a + (-15.0 * bv + 6 * cv));
              else a = (xyz == 1) ? (1.0 / 15.0) * (8.0 * a + (10.0 * bv - 3.0 * cv)) : (1.0 / 15.0) * (24.0 * a + (-15.0 * bv + 6 * cv));
              FT(t, ix, iy, iz, c) = a;
            }
          }
    }
  }
  apply_bc(m, b, t, 0);
}

/* all tiles of a field: out [nb][L][L][L][nc] (x fastest), like ref_tool's `lab` command */
void orc_mesh_labs(const orc_mesh *m, const double *f, int nc, int is_vector, int s, int e, int tens, double *out) {
  tile_t t;
  tile_init(&t, nc, is_vector, s, e, tens);
  const size_t per = (size_t)t.L * t.L * t.L * nc;
  for (long b = 0; b < m->nblocks; b++) {
    /* the reference reuses one lab object: cells it does not fill keep the previous block's values; here they keep ours */
    orc_mesh_lab(m, f, b, &t);
    memcpy(out + b * per, t.F, per * sizeof(double));
  }
  tile_free(&t);
}

/* ======================= operators on multi-level meshes ======================= */
/* compute<Lab>(kernel, g, g_corr) with flux correction (main.cpp:5584-5644, 588-802): every kernel below
 * fills `out` per block and, for the faces whose same-level neighbour does not exist (FluxCorrection::prepare,
 * 676-711), the face flux arrays face[b][f][64][fc]; fix_fluxes() then adds, on the coarse side of every
 * coarse/fine face, the coarse face flux plus the four fine ones to the boundary cells (FillBlockCases 729-801). */
typedef struct { int nf; double *v; char *stored; } faces_t; /* v: [nb][6][64][fc], stored: [nb][6] */

static void faces_init(const orc_mesh *m, faces_t *F, int fc) {
  F->nf = fc;
  F->v = (double *)calloc((size_t)m->nblocks * 6 * 64 * fc, sizeof(double));
  F->stored = (char *)calloc((size_t)m->nblocks * 6, 1);
  for (long b = 0; b < m->nblocks; b++) {
    const int l = m->level[b], *idx = &m->index[3 * b];
    for (int f = 0; f < 6; f++) {
      const int d = f >> 1, side = f & 1;
      const int skin = idx[d] == 0 || idx[d] == nblk(m, l, d) - 1, skip = idx[d] == 0 ? -1 : 1;
      if (m->bc[d] != ORC_BC_PERIODIC && (side ? 1 : -1) == skip && skin) continue;
      int nei[3] = {idx[0], idx[1], idx[2]};
      nei[d] += side ? 1 : -1;
      if (tree_state(m, l, nei) != 1) F->stored[b * 6 + f] = 1;
    }
  }
}
static void faces_free(faces_t *F) { free(F->v); free(F->stored); }
#define FACE(F, b, f, i, c) (F)->v[((((long)(b) * 6 + (f)) * 64) + (i)) * (F)->nf + (c)]

/* out: block field with oc components; the face flux has fc (= oc) components.
 * The grid is always a GridMPI, so the corrector that runs is FluxCorrectionMPI::FillBlockCases (main.cpp:2825-2935):
 * every fine face is reduced 2x2 as (f00+f10)+(f01+f11) into the message (2848-2851), the coarse face accumulates
 * it (FillCase 2579-2629), and the coarse boundary cells are then updated direction by direction - all x faces,
 * then y, then z (FillCase_2, 2918-2926; called once per fine block, so three of the four calls add +0.0). */
static void fix_fluxes(const orc_mesh *m, faces_t *F, double *out, int oc) {
  for (int d = 0; d < 3; d++)
    for (long b = 0; b < m->nblocks; b++) {
      const int l = m->level[b], *idx = &m->index[3 * b];
      for (int side = 0; side < 2; side++) {
        const int f = 2 * d + side;
        if (!F->stored[b * 6 + f]) continue;
        int nei[3] = {idx[0], idx[1], idx[2]};
        nei[d] += side ? 1 : -1;
        if (tree_state(m, l, nei) != -1) continue; /* only the coarse side of a coarse/fine face is corrected */
        const int of = f ^ 1;
        const int dfast = d == 0 ? 1 : 0, dslow = d == 2 ? 1 : 2; /* face arrays: index = fast + 8*slow */
        for (int B = 0; B < 4; B++) {
          int fi[3];
          for (int k = 0; k < 3; k++) fi[k] = 2 * idx[k];
          fi[d] = 2 * idx[d] + (side ? 2 : -1);
          fi[dfast] += B % 2;
          fi[dslow] += B / 2;
          const int fb = leaf_at(m, l + 1, fi);
          if (fb < 0) continue;
          const int base = (B % 2) * 4 + (B / 2) * 4 * 8;
          for (int i1 = 0; i1 < 8; i1 += 2)
            for (int i2 = 0; i2 < 8; i2 += 2)
              for (int c = 0; c < oc; c++) {
                const double avg = (FACE(F, fb, of, i2 + i1 * 8, c) + FACE(F, fb, of, i2 + 1 + i1 * 8, c)) +
                                   (FACE(F, fb, of, i2 + (i1 + 1) * 8, c) + FACE(F, fb, of, i2 + 1 + (i1 + 1) * 8, c));
                FACE(F, b, f, base + (i2 / 2) + (i1 / 2) * 8, c) += avg;
              }
        }
        double *blk = out + (long)b * BS3 * oc;
        const int j = side ? BS - 1 : 0;
        for (int i1 = 0; i1 < 8; i1++)
          for (int i2 = 0; i2 < 8; i2++) {
            const int x = d == 0 ? j : i2, y = d == 1 ? j : (d == 0 ? i2 : i1), z = d == 2 ? j : i1;
            for (int c = 0; c < oc; c++) {
              double *v = &blk[((z * BS + y) * BS + x) * oc + c];
              *v += FACE(F, b, f, i2 + i1 * 8, c);
              *v += 0.0; /* the repeated FillCase_2 calls on the cleared face (turns -0.0 into +0.0) */
            }
          }
      }
    }
}

static inline double upwind5m(double U, double um3, double um2, double um1, double u, double up1, double up2, double up3) {
  if (U > 0) return (-2 * um3 + 15 * um2 - 60 * um1 + 20 * u + 30 * up1 - 3 * up2) / 60.;
  else return (2 * up3 - 15 * up2 + 60 * up1 - 20 * u - 30 * um1 + 3 * um2) / 60.;
}

/* KernelAdvectDiffuse incl. face fluxes (main.cpp:9484-9637) + flux correction on tmpV */
void orc_mesh_advdiff_stage_rhs(const orc_mesh *m, const double *vel, double *tmpV, double dt, double nu, const double uinf[3]) {
  faces_t F;
  faces_init(m, &F, 3);
#define V(x, y, z, c) FT(&t, x, y, z, c)
#pragma omp parallel num_threads(mesh_threads(m))
  {
  tile_t t;
  tile_init(&t, 3, 1, -3, 4, 0);
#pragma omp for schedule(dynamic, 1)
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, vel, b, &t);
    const double h = orc_mesh_h(m, b), h3 = h * h * h;
    const double facA = -dt / h * h3 * 1.0, facD = (nu / h) * (dt / h) * h3 * 1.0;
    double *o = tmpV + b * BS3 * 3;
    for (int z = 0; z < BS; z++)
      for (int y = 0; y < BS; y++)
        for (int x = 0; x < BS; x++) {
          const double uAbs[3] = {V(x, y, z, 0) + uinf[0], V(x, y, z, 1) + uinf[1], V(x, y, z, 2) + uinf[2]};
          double dx[3], dy[3], dz[3], lap[3];
          for (int c = 0; c < 3; c++) {
            dx[c] = upwind5m(uAbs[0], V(x - 3, y, z, c), V(x - 2, y, z, c), V(x - 1, y, z, c), V(x, y, z, c), V(x + 1, y, z, c), V(x + 2, y, z, c), V(x + 3, y, z, c));
            dy[c] = upwind5m(uAbs[1], V(x, y - 3, z, c), V(x, y - 2, z, c), V(x, y - 1, z, c), V(x, y, z, c), V(x, y + 1, z, c), V(x, y + 2, z, c), V(x, y + 3, z, c));
            dz[c] = upwind5m(uAbs[2], V(x, y, z - 3, c), V(x, y, z - 2, c), V(x, y, z - 1, c), V(x, y, z, c), V(x, y, z + 1, c), V(x, y, z + 2, c), V(x, y, z + 3, c));
          }
          lap[0] = ((V(x + 1, y, z, 0) + V(x - 1, y, z, 0)) + ((V(x, y + 1, z, 0) + V(x, y - 1, z, 0)) + (V(x, y, z + 1, 0) + V(x, y, z - 1, 0)))) - 6 * V(x, y, z, 0);
          lap[1] = ((V(x, y + 1, z, 1) + V(x, y - 1, z, 1)) + ((V(x, y, z + 1, 1) + V(x, y, z - 1, 1)) + (V(x + 1, y, z, 1) + V(x - 1, y, z, 1)))) - 6 * V(x, y, z, 1);
          lap[2] = ((V(x, y, z + 1, 2) + V(x, y, z - 1, 2)) + ((V(x + 1, y, z, 2) + V(x - 1, y, z, 2)) + (V(x, y + 1, z, 2) + V(x, y - 1, z, 2)))) - 6 * V(x, y, z, 2);
          const double duA = uAbs[0] * dx[0] + (uAbs[1] * dy[0] + uAbs[2] * dz[0]);
          const double dvA = uAbs[1] * dy[1] + (uAbs[2] * dz[1] + uAbs[0] * dx[1]);
          const double dwA = uAbs[2] * dz[2] + (uAbs[0] * dx[2] + uAbs[1] * dy[2]);
          double *oc = o + ((z * BS + y) * BS + x) * 3;
          oc[0] += facA * duA + facD * lap[0];
          oc[1] += facA * dvA + facD * lap[1];
          oc[2] += facA * dwA + facD * lap[2];
        }
    for (int f = 0; f < 6; f++) {
      if (!F.stored[b * 6 + f]) continue;
      const int d = f >> 1, side = f & 1;
      for (int i1 = 0; i1 < 8; i1++)
        for (int i2 = 0; i2 < 8; i2++) {
          int p[3], q[3];
          const int dfast = d == 0 ? 1 : 0, dslow = d == 2 ? 1 : 2;
          p[dfast] = q[dfast] = i2; p[dslow] = q[dslow] = i1;
          p[d] = side ? BS - 1 : 0; q[d] = side ? BS : -1;
          for (int c = 0; c < 3; c++) FACE(&F, b, f, i2 + i1 * 8, c) = facD * (V(p[0], p[1], p[2], c) - V(q[0], q[1], q[2], c));
        }
    }
  }
  tile_free(&t);
  }
#undef V
  fix_fluxes(m, &F, tmpV, 3);
  faces_free(&F);
}

void orc_mesh_advect_diffuse(const orc_mesh *m, double *vel, double *tmpV, double dt, double nu, const double uinf[3]) {
  const double alpha[3] = {1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0}, beta[3] = {-5.0 / 9.0, -153.0 / 128.0, 0.0};
  memset(tmpV, 0, (size_t)m->nblocks * BS3 * 3 * sizeof(double));
  for (int rk = 0; rk < 3; rk++) {
    orc_mesh_advdiff_stage_rhs(m, vel, tmpV, dt, nu, uinf);
    for (long b = 0; b < m->nblocks; b++) {
      const double h = orc_mesh_h(m, b), ih3 = alpha[rk] / (h * h * h);
      for (long i = b * BS3 * 3; i < (b + 1) * BS3 * 3; i++) { vel[i] += tmpV[i] * ih3; tmpV[i] *= beta[rk]; }
    }
  }
}

/* 7-point scalar kernels with their face fluxes */
static void lhs_kernel_mesh(const orc_mesh *m, const double *pres, double *lhs) { /* KernelLHSPoisson, 9205-9268 */
  faces_t F;
  faces_init(m, &F, 1);
#define P(x, y, z) FT(&t, x, y, z, 0)
#pragma omp parallel num_threads(mesh_threads(m))
  {
  tile_t t;
  tile_init(&t, 1, 0, -1, 2, 0);
#pragma omp for schedule(dynamic, 1)
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, pres, b, &t);
    const double h = orc_mesh_h(m, b);
    double *o = lhs + b * BS3;
    for (int z = 0; z < BS; z++)
      for (int y = 0; y < BS; y++)
        for (int x = 0; x < BS; x++)
          o[(z * BS + y) * BS + x] = h * (P(x - 1, y, z) + P(x + 1, y, z) + P(x, y - 1, z) + P(x, y + 1, z) + P(x, y, z - 1) + P(x, y, z + 1) - 6.0 * P(x, y, z));
    for (int f = 0; f < 6; f++) {
      if (!F.stored[b * 6 + f]) continue;
      const int d = f >> 1, side = f & 1, dfast = d == 0 ? 1 : 0, dslow = d == 2 ? 1 : 2;
      for (int i1 = 0; i1 < 8; i1++)
        for (int i2 = 0; i2 < 8; i2++) {
          int p[3], q[3];
          p[dfast] = q[dfast] = i2; p[dslow] = q[dslow] = i1;
          p[d] = side ? BS - 1 : 0; q[d] = side ? BS : -1;
          FACE(&F, b, f, i2 + i1 * 8, 0) = h * (P(p[0], p[1], p[2]) - P(q[0], q[1], q[2]));
        }
    }
  }
  tile_free(&t);
  }
#undef P
  fix_fluxes(m, &F, lhs, 1);
  faces_free(&F);
}

static long corner_block_mesh(const orc_mesh *m) {
  for (long b = 0; b < m->nblocks; b++)
    if (m->index[3 * b] == 0 && m->index[3 * b + 1] == 0 && m->index[3 * b + 2] == 0) return b;
  return -1;
}

void orc_mesh_lhs(const orc_mesh *m, const double *pres, double *lhs, int mc) { /* ComputeLHS, 9273-9327 */
  double avgP = 0;
  if (mc <= 2 && mc > 0)
    for (long b = 0; b < m->nblocks; b++) {
      const double h = orc_mesh_h(m, b), h3 = h * h * h;
      for (int i = 0; i < BS3; i++) avgP += pres[b * BS3 + i] * h3;
    }
  lhs_kernel_mesh(m, pres, lhs);
  if (mc == 0) return;
  /* `index` in the reference is the LAST block (in m_vInfo order) whose index is (0,0,0), 9287-9289 */
  long corner = -1;
  for (long b = 0; b < m->nblocks; b++)
    if (m->index[3 * b] == 0 && m->index[3 * b + 1] == 0 && m->index[3 * b + 2] == 0) corner = b;
  if (mc <= 2 && mc > 0) {
    if (mc == 1 && corner >= 0) lhs[corner * BS3] = avgP;
    else if (mc == 2)
      for (long b = 0; b < m->nblocks; b++) {
        const double h = orc_mesh_h(m, b), h3 = h * h * h;
        for (int i = 0; i < BS3; i++) lhs[b * BS3 + i] += avgP * h3;
      }
  } else if (corner >= 0)
    lhs[corner * BS3] = pres[corner * BS3];
  (void)corner_block_mesh;
}

void orc_mesh_precond(const orc_mesh *m, double *pres) { /* getZImplParallel, main.cpp:14704-14745: invh of each block */
#pragma omp parallel for num_threads(mesh_threads(m))
  for (long b = 0; b < m->nblocks; b++) orc_precond_block(pres + b * BS3, orc_mesh_h(m, b));
}

static void mesh_lhs_cb(void *m, const double *in, double *out, int mc) { orc_mesh_lhs((const orc_mesh *)m, in, out, mc); }
static void mesh_precond_cb(void *m, double *io) { orc_mesh_precond((const orc_mesh *)m, io); }
void orc_mesh_solve(const orc_mesh *m, double *lhs, double *pres, orc_solve_info *info) {
  long corner = -1;
  for (long b = 0; b < m->nblocks; b++)
    if (m->index[3 * b] == 0 && m->index[3 * b + 1] == 0 && m->index[3 * b + 2] == 0) corner = b;
  orc_solve_generic((void *)m, m->nblocks * BS3, corner * BS3, mesh_lhs_cb, mesh_precond_cb, lhs, pres, info);
}

/* generic face loop helper: (p) inside face cell, (q) ghost cell behind face f at face index (i2,i1) */
#define FACE_CELLS(f, i2, i1, p, q)                                        \
  const int d_ = (f) >> 1, side_ = (f) & 1, df_ = d_ == 0 ? 1 : 0, ds_ = d_ == 2 ? 1 : 2; \
  int p[3], q[3];                                                          \
  p[df_] = q[df_] = (i2); p[ds_] = q[ds_] = (i1);                          \
  p[d_] = side_ ? BS - 1 : 0; q[d_] = side_ ? BS : -1;

/* KernelPressureRHS incl. faces, main.cpp:14849-14950; flux-corrected into lhs */
void orc_mesh_pressure_rhs(const orc_mesh *m, const double *vel, const double *udef, const double *chi, double *lhs, double dt) {
  faces_t F;
  faces_init(m, &F, 1);
#define U(x, y, z, c) FT(&t, x, y, z, c)
#define D(x, y, z, c) FT(&t2, x, y, z, c)
#pragma omp parallel num_threads(mesh_threads(m))
  {
  tile_t t, t2;
  tile_init(&t, 3, 1, -1, 2, 0);
  tile_init(&t2, 3, 1, -1, 2, 0);
#pragma omp for schedule(dynamic, 1)
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, vel, b, &t);
    orc_mesh_lab(m, udef, b, &t2);
    const double h = orc_mesh_h(m, b), fac = 0.5 * h * h / dt;
    for (int z = 0; z < BS; z++)
      for (int y = 0; y < BS; y++)
        for (int x = 0; x < BS; x++) {
          const long i = b * BS3 + (z * BS + y) * BS + x;
          double p = fac * (U(x + 1, y, z, 0) - U(x - 1, y, z, 0) + U(x, y + 1, z, 1) - U(x, y - 1, z, 1) + U(x, y, z + 1, 2) - U(x, y, z - 1, 2));
          const double divUs = D(x + 1, y, z, 0) - D(x - 1, y, z, 0) + D(x, y + 1, z, 1) - D(x, y - 1, z, 1) + D(x, y, z + 1, 2) - D(x, y, z - 1, 2);
          p += -chi[i] * fac * divUs;
          lhs[i] = p;
        }
    for (int f = 0; f < 6; f++) {
      if (!F.stored[b * 6 + f]) continue;
      for (int i1 = 0; i1 < 8; i1++)
        for (int i2 = 0; i2 < 8; i2++) {
          FACE_CELLS(f, i2, i1, p, q)
          const double c = chi[b * BS3 + (p[2] * BS + p[1]) * BS + p[0]];
          const int comp = d_;
          double v;
          if (!side_) v = fac * (U(q[0], q[1], q[2], comp) + U(p[0], p[1], p[2], comp)) - c * fac * (D(q[0], q[1], q[2], comp) + D(p[0], p[1], p[2], comp));
          else v = -fac * (U(q[0], q[1], q[2], comp) + U(p[0], p[1], p[2], comp)) + c * fac * (D(q[0], q[1], q[2], comp) + D(p[0], p[1], p[2], comp));
          FACE(&F, b, f, i2 + i1 * 8, 0) = v;
        }
    }
  }
  tile_free(&t); tile_free(&t2);
  }
#undef U
#undef D
  fix_fluxes(m, &F, lhs, 1);
  faces_free(&F);
}

/* KernelDivPressure incl. faces, main.cpp:14769-14834: writes tmpV.u[0]; correction on tmpV */
void orc_mesh_div_pressure(const orc_mesh *m, const double *pres, double *tmpV) {
  faces_t F;
  faces_init(m, &F, 3);
#define P(x, y, z) FT(&t, x, y, z, 0)
#pragma omp parallel num_threads(mesh_threads(m))
  {
  tile_t t;
  tile_init(&t, 1, 0, -1, 2, 0);
#pragma omp for schedule(dynamic, 1)
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, pres, b, &t);
    const double fac = orc_mesh_h(m, b);
    for (int z = 0; z < BS; z++)
      for (int y = 0; y < BS; y++)
        for (int x = 0; x < BS; x++)
          tmpV[(b * BS3 + (z * BS + y) * BS + x) * 3] =
              fac * (P(x + 1, y, z) + P(x - 1, y, z) + P(x, y + 1, z) + P(x, y - 1, z) + P(x, y, z + 1) + P(x, y, z - 1) - 6.0 * P(x, y, z));
    for (int f = 0; f < 6; f++) {
      if (!F.stored[b * 6 + f]) continue;
      for (int i1 = 0; i1 < 8; i1++)
        for (int i2 = 0; i2 < 8; i2++) {
          FACE_CELLS(f, i2, i1, p, q)
          FACE(&F, b, f, i2 + i1 * 8, 0) = !side_ ? fac * (P(p[0], p[1], p[2]) - P(q[0], q[1], q[2])) : -fac * (P(q[0], q[1], q[2]) - P(p[0], p[1], p[2]));
        }
    }
  }
  tile_free(&t);
  }
#undef P
  fix_fluxes(m, &F, tmpV, 3);
  faces_free(&F);
}

/* KernelGradP incl. faces, main.cpp:14990-15055 */
void orc_mesh_grad_p(const orc_mesh *m, const double *pres, double *tmpV, double dt) {
  faces_t F;
  faces_init(m, &F, 3);
#define P(x, y, z) FT(&t, x, y, z, 0)
#pragma omp parallel num_threads(mesh_threads(m))
  {
  tile_t t;
  tile_init(&t, 1, 0, -1, 2, 0);
#pragma omp for schedule(dynamic, 1)
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, pres, b, &t);
    const double h = orc_mesh_h(m, b), fac = -0.5 * dt * h * h;
    for (int z = 0; z < BS; z++)
      for (int y = 0; y < BS; y++)
        for (int x = 0; x < BS; x++) {
          double *o = tmpV + (b * BS3 + (z * BS + y) * BS + x) * 3;
          o[0] = fac * (P(x + 1, y, z) - P(x - 1, y, z));
          o[1] = fac * (P(x, y + 1, z) - P(x, y - 1, z));
          o[2] = fac * (P(x, y, z + 1) - P(x, y, z - 1));
        }
    for (int f = 0; f < 6; f++) {
      if (!F.stored[b * 6 + f]) continue;
      for (int i1 = 0; i1 < 8; i1++)
        for (int i2 = 0; i2 < 8; i2++) {
          FACE_CELLS(f, i2, i1, p, q)
          FACE(&F, b, f, i2 + i1 * 8, d_) = !side_ ? fac * (P(q[0], q[1], q[2]) + P(p[0], p[1], p[2])) : -fac * (P(q[0], q[1], q[2]) + P(p[0], p[1], p[2]));
        }
    }
  }
  tile_free(&t);
  }
#undef P
  fix_fluxes(m, &F, tmpV, 3);
  faces_free(&F);
}

/* PressureProjection::operator(), main.cpp:15061-15160, on a multi-level mesh; with an obstacle (n > 0) tmpV receives its
 * deformation velocity first (kernelUpdateTmpV, 15081-15082) and chi enters the right-hand side */
void orc_mesh_update_tmpv(const orc_mesh *m, double *tmpV, const double *chi_field, long n, const long long *ids, const double *chi,
                          const double *udef);
void orc_mesh_project_obst(const orc_mesh *m, double *vel, double *pres, double *tmpV, double *lhs, const double *chi, double dt, int step,
                           orc_solve_info *info, long n, const long long *ids, const double *ochi, const double *oudef) {
  const long N = m->nblocks * BS3;
  double *pOld = (double *)malloc(N * sizeof(double));
  memcpy(pOld, pres, N * sizeof(double));
  memset(tmpV, 0, 3 * N * sizeof(double));
  if (n > 0) orc_mesh_update_tmpv(m, tmpV, chi, n, ids, ochi, oudef);
  orc_mesh_pressure_rhs(m, vel, tmpV, chi, lhs, dt);
  if (step > 2) {
    orc_mesh_div_pressure(m, pres, tmpV);
    for (long i = 0; i < N; i++) { lhs[i] -= tmpV[3 * i]; pres[i] = 0; }
  } else
    memset(pres, 0, N * sizeof(double));
  orc_mesh_solve(m, lhs, pres, info);
  double avg = 0, avg1 = 0;
  for (long b = 0; b < m->nblocks; b++) {
    const double h = orc_mesh_h(m, b), vv = h * h * h;
    for (long i = b * BS3; i < (b + 1) * BS3; i++) { avg += pres[i] * vv; avg1 += vv; }
  }
  avg = avg / avg1;
  for (long i = 0; i < N; i++) pres[i] -= avg;
  if (step > 2)
    for (long i = 0; i < N; i++) pres[i] += pOld[i];
  orc_mesh_grad_p(m, pres, tmpV, dt);
  for (long b = 0; b < m->nblocks; b++) {
    const double h = orc_mesh_h(m, b), fac = 1.0 / (h * h * h);
    for (long i = b * BS3 * 3; i < (b + 1) * BS3 * 3; i++) vel[i] += fac * tmpV[i];
  }
  free(pOld);
}
void orc_mesh_project(const orc_mesh *m, double *vel, double *pres, double *tmpV, double *lhs, const double *chi, double dt, int step,
                      orc_solve_info *info) {
  orc_mesh_project_obst(m, vel, pres, tmpV, lhs, chi, dt, step, info, 0, NULL, NULL, NULL);
}


/* ------------------------------------------------------------------ implicit diffusion (AdvectionDiffusionImplicit)
 * KernelAdvect (main.cpp:9849-10029): tmpV = facD*lap(u) (+ face fluxes, flux-corrected), vel += facA*(u.grad)u/h^3 IN PLACE.
 * The reference updates vel while later blocks still build their ghosted tiles from it (the lab is loaded per block inside
 * the same loop, 5598-5602), so its result depends on the block order and, with more than one thread, on timing.
 *   sequential = 1: the reference's behaviour with ONE thread (blocks in m_vInfo order, each tile loaded from the partially
 *                   updated field) -- what oracle/_ref reproduces bit for bit with OMP_NUM_THREADS=1;
 *   sequential = 0: every tile is loaded from the field as it was on entry (what the operator means, and the only
 *                   order-independent reading): the device path implements this one. */
void orc_mesh_advect_implicit(const orc_mesh *m, double *vel, double *tmpV, double dt, double nu, const double uinf[3], int sequential) {
  faces_t F;
  faces_init(m, &F, 3);
  double *copy = NULL;
  const double *src = vel;
  if (!sequential) {
    copy = (double *)malloc((size_t)m->nblocks * BS3 * 3 * sizeof(double));
    memcpy(copy, vel, (size_t)m->nblocks * BS3 * 3 * sizeof(double));
    src = copy;
  }
#define V(x, y, z, c) FT(&t, x, y, z, c)
  tile_t t;
  tile_init(&t, 3, 1, -3, 4, 0);
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, src, b, &t);
    const double h = orc_mesh_h(m, b), h3 = h * h * h;
    const double facA = -dt / h * h3, facD = (nu / h) * (dt / h) * h3;
    double *o = tmpV + b * BS3 * 3, *v = vel + b * BS3 * 3;
    for (int z = 0; z < BS; z++)
      for (int y = 0; y < BS; y++)
        for (int x = 0; x < BS; x++) {
          const double uAbs[3] = {V(x, y, z, 0) + uinf[0], V(x, y, z, 1) + uinf[1], V(x, y, z, 2) + uinf[2]};
          double dx[3], dy[3], dz[3], lap[3];
          for (int c = 0; c < 3; c++) {
            dx[c] = upwind5m(uAbs[0], V(x - 3, y, z, c), V(x - 2, y, z, c), V(x - 1, y, z, c), V(x, y, z, c), V(x + 1, y, z, c), V(x + 2, y, z, c), V(x + 3, y, z, c));
            dy[c] = upwind5m(uAbs[1], V(x, y - 3, z, c), V(x, y - 2, z, c), V(x, y - 1, z, c), V(x, y, z, c), V(x, y + 1, z, c), V(x, y + 2, z, c), V(x, y + 3, z, c));
            dz[c] = upwind5m(uAbs[2], V(x, y, z - 3, c), V(x, y, z - 2, c), V(x, y, z - 1, c), V(x, y, z, c), V(x, y, z + 1, c), V(x, y, z + 2, c), V(x, y, z + 3, c));
          }
          lap[0] = ((V(x + 1, y, z, 0) + V(x - 1, y, z, 0)) + ((V(x, y + 1, z, 0) + V(x, y - 1, z, 0)) + (V(x, y, z + 1, 0) + V(x, y, z - 1, 0)))) - 6 * V(x, y, z, 0);
          lap[1] = ((V(x, y + 1, z, 1) + V(x, y - 1, z, 1)) + ((V(x, y, z + 1, 1) + V(x, y, z - 1, 1)) + (V(x + 1, y, z, 1) + V(x - 1, y, z, 1)))) - 6 * V(x, y, z, 1);
          lap[2] = ((V(x, y, z + 1, 2) + V(x, y, z - 1, 2)) + ((V(x + 1, y, z, 2) + V(x - 1, y, z, 2)) + (V(x, y + 1, z, 2) + V(x, y - 1, z, 2)))) - 6 * V(x, y, z, 2);
          const double duA = uAbs[0] * dx[0] + (uAbs[1] * dy[0] + uAbs[2] * dz[0]);
          const double dvA = uAbs[1] * dy[1] + (uAbs[2] * dz[1] + uAbs[0] * dx[1]);
          const double dwA = uAbs[2] * dz[2] + (uAbs[0] * dx[2] + uAbs[1] * dy[2]);
          const long i = ((z * BS + y) * BS + x) * 3;
          o[i + 0] = facD * lap[0];
          o[i + 1] = facD * lap[1];
          o[i + 2] = facD * lap[2];
          v[i + 0] += facA * duA / h3; /* 9939-9941 */
          v[i + 1] += facA * dvA / h3;
          v[i + 2] += facA * dwA / h3;
        }
    for (int f = 0; f < 6; f++) {
      if (!F.stored[b * 6 + f]) continue;
      for (int i1 = 0; i1 < 8; i1++)
        for (int i2 = 0; i2 < 8; i2++) {
          FACE_CELLS(f, i2, i1, p, q)
          for (int c = 0; c < 3; c++) FACE(&F, b, f, i2 + i1 * 8, c) = facD * (V(p[0], p[1], p[2], c) - V(q[0], q[1], q[2], c));
        }
    }
  }
  tile_free(&t);
#undef V
  free(copy);
  fix_fluxes(m, &F, tmpV, 3);
  faces_free(&F);
}

/* KernelDiffusionRHS (main.cpp:9729-9848): tmpV = h*lap(vel) with the component-wise associations, flux-corrected */
void orc_mesh_diffusion_rhs(const orc_mesh *m, const double *vel, double *tmpV) {
  faces_t F;
  faces_init(m, &F, 3);
#define V(x, y, z, c) FT(&t, x, y, z, c)
#pragma omp parallel num_threads(mesh_threads(m))
  {
  tile_t t;
  tile_init(&t, 3, 1, -1, 2, 0);
#pragma omp for schedule(dynamic, 1)
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, vel, b, &t);
    const double facD = orc_mesh_h(m, b);
    double *o = tmpV + b * BS3 * 3;
    for (int z = 0; z < BS; z++)
      for (int y = 0; y < BS; y++)
        for (int x = 0; x < BS; x++) {
          const double duD = ((V(x + 1, y, z, 0) + V(x - 1, y, z, 0)) + ((V(x, y + 1, z, 0) + V(x, y - 1, z, 0)) + (V(x, y, z + 1, 0) + V(x, y, z - 1, 0)))) - 6 * V(x, y, z, 0);
          const double dvD = ((V(x, y + 1, z, 1) + V(x, y - 1, z, 1)) + ((V(x, y, z + 1, 1) + V(x, y, z - 1, 1)) + (V(x + 1, y, z, 1) + V(x - 1, y, z, 1)))) - 6 * V(x, y, z, 1);
          const double dwD = ((V(x, y, z + 1, 2) + V(x, y, z - 1, 2)) + ((V(x + 1, y, z, 2) + V(x - 1, y, z, 2)) + (V(x, y + 1, z, 2) + V(x, y - 1, z, 2)))) - 6 * V(x, y, z, 2);
          const long i = ((z * BS + y) * BS + x) * 3;
          o[i + 0] = facD * duD;
          o[i + 1] = facD * dvD;
          o[i + 2] = facD * dwD;
        }
    for (int f = 0; f < 6; f++) {
      if (!F.stored[b * 6 + f]) continue;
      for (int i1 = 0; i1 < 8; i1++)
        for (int i2 = 0; i2 < 8; i2++) {
          FACE_CELLS(f, i2, i1, p, q)
          for (int c = 0; c < 3; c++) FACE(&F, b, f, i2 + i1 * 8, c) = facD * (V(p[0], p[1], p[2], c) - V(q[0], q[1], q[2], c));
        }
    }
  }
  tile_free(&t);
  }
#undef V
  fix_fluxes(m, &F, tmpV, 3);
  faces_free(&F);
}

/* DiffusionSolver::_lhs -> KernelLHSDiffusion (main.cpp:6726-6803) on the BlockLabBC<ScalarGrid, .., direction> tile:
 * lhs = h*(sum6 - 6p) + coef*p, coef = -h^3/(dt nu); face fluxes h*(p_in - p_ghost); flux-corrected */
void orc_mesh_diff_lhs(const orc_mesh *m, const double *pres, double *lhs, int direction, double dt, double nu) {
  faces_t F;
  faces_init(m, &F, 1);
#define P(x, y, z) FT(&t, x, y, z, 0)
#pragma omp parallel num_threads(mesh_threads(m))
  {
  tile_t t;
  tile_init(&t, 1, 2 + direction, -1, 2, 0);
#pragma omp for schedule(dynamic, 1)
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, pres, b, &t);
    const double h = orc_mesh_h(m, b), coef = -1.0 / (dt * nu) * h * h * h;
    double *o = lhs + b * BS3;
    for (int z = 0; z < BS; z++)
      for (int y = 0; y < BS; y++)
        for (int x = 0; x < BS; x++)
          o[(z * BS + y) * BS + x] =
              h * (P(x - 1, y, z) + P(x + 1, y, z) + P(x, y - 1, z) + P(x, y + 1, z) + P(x, y, z - 1) + P(x, y, z + 1) - 6.0 * P(x, y, z)) + coef * P(x, y, z);
    for (int f = 0; f < 6; f++) {
      if (!F.stored[b * 6 + f]) continue;
      for (int i1 = 0; i1 < 8; i1++)
        for (int i2 = 0; i2 < 8; i2++) {
          FACE_CELLS(f, i2, i1, p, q)
          FACE(&F, b, f, i2 + i1 * 8, 0) = h * (P(p[0], p[1], p[2]) - P(q[0], q[1], q[2]));
        }
    }
  }
  tile_free(&t);
  }
#undef P
  fix_fluxes(m, &F, lhs, 1);
  faces_free(&F);
}

/* DiffusionSolver::_preconditioner -> diffusion_kernels::getZImplParallel (main.cpp:10534-10579), in place */
void orc_mesh_diff_precond(const orc_mesh *m, double *pres, double dt, double nu) {
#pragma omp parallel for num_threads(mesh_threads(m))
  for (long b = 0; b < m->nblocks; b++) {
    const double h = orc_mesh_h(m, b);
    orc_precond_block_coef(pres + b * BS3, h, -6.0 - h * h / nu / dt); /* 10570 */
  }
}

typedef struct { const orc_mesh *m; int direction; double dt, nu; } diff_ctx;
static void diff_lhs_cb(void *c, const double *in, double *out, int mc) {
  (void)mc;
  const diff_ctx *d = (const diff_ctx *)c;
  orc_mesh_diff_lhs(d->m, in, out, d->direction, d->dt, d->nu);
}
static void diff_precond_cb(void *c, double *io) {
  const diff_ctx *d = (const diff_ctx *)c;
  orc_mesh_diff_precond(d->m, io, d->dt, d->nu);
}
/* DiffusionSolver::solve (main.cpp:6896-7146): rhs in lhs, initial guess and result in pres */
void orc_mesh_diff_solve(const orc_mesh *m, double *lhs, double *pres, int direction, double dt, double nu, orc_solve_info *info) {
  diff_ctx c = {m, direction, dt, nu};
  info->mean_constraint = 0;
  orc_solve_generic2(&c, m->nblocks * BS3, 0, diff_lhs_cb, diff_precond_cb, lhs, pres, info, 0x7fffffff);
}

/* AdvectionDiffusionImplicit::euler (main.cpp:10030-10118).  On return: vel advanced, pres unchanged, tmpV = the right-hand
 * side vector of the three Helmholtz solves, lhs = scratch of the last solve; iters[3] = iterations per component */
void orc_mesh_advdiff_implicit(const orc_mesh *m, double *vel, double *pres, double *tmpV, double *lhs, double dt, double nu,
                               const double uinf[3], double tol, double tol_rel, int sequential, int iters[3]) {
  const long N = m->nblocks * BS3;
  double *pressure = (double *)malloc(N * sizeof(double)), *velocity = (double *)malloc(3 * N * sizeof(double));
  orc_mesh_advect_implicit(m, vel, tmpV, dt, nu, uinf, sequential);
  for (long b = 0; b < m->nblocks; b++) {
    const double h = orc_mesh_h(m, b), ih3 = 1.0 / (h * h * h);
    for (long i = b * BS3; i < (b + 1) * BS3; i++) {
      pressure[i] = pres[i];
      for (int c = 0; c < 3; c++) {
        velocity[3 * i + c] = vel[3 * i + c];
        vel[3 * i + c] = tmpV[3 * i + c] * ih3 + vel[3 * i + c]; /* 10051-10053 */
      }
    }
  }
  orc_mesh_diffusion_rhs(m, vel, tmpV);
  for (long b = 0; b < m->nblocks; b++) {
    const double h = orc_mesh_h(m, b), ih3 = 1.0 / (h * h * h);
    for (long i = b * BS3 * 3; i < (b + 1) * BS3 * 3; i++)
      tmpV[i] = -tmpV[i] * ih3 + (vel[i] - velocity[i]) / (dt * nu); /* 10066-10074 */
  }
  for (int index = 0; index < 3; index++) {
    for (long b = 0; b < m->nblocks; b++) {
      const double h = orc_mesh_h(m, b), h3 = h * h * h;
      for (long i = b * BS3; i < (b + 1) * BS3; i++) { pres[i] = 0; lhs[i] = h3 * tmpV[3 * i + index]; }
    }
    orc_solve_info info = {tol, tol_rel, 0, 0, 0, 0, 0};
    orc_mesh_diff_solve(m, lhs, pres, index, dt, nu, &info);
    if (iters) iters[index] = info.iters;
    for (long i = 0; i < N; i++) vel[3 * i + index] += pres[i];
  }
  memcpy(pres, pressure, N * sizeof(double));
  free(pressure); free(velocity);
}

double orc_mesh_max_u(const orc_mesh *m, const double *vel, const double uinf[3]) { /* findMaxU, main.cpp:8603-8623 */
  double mx = 0;
  for (long i = 0; i < m->nblocks * BS3; i++)
    for (int c = 0; c < 3; c++) { const double a = fabs(vel[3 * i + c] + uinf[c]); if (a > mx) mx = a; }
  return mx;
}

/* neighbour states of all 27 codes of every block as BlockLab::load sees them (3690-3712): out[b][code] = slot of the
 * same-level leaf (Exists), -100 - slot of the coarser leaf (CheckCoarser), -3 (CheckFiner), -1 skipped (domain face) */
void orc_mesh_states(const orc_mesh *m, int *out) {
  for (long b = 0; b < m->nblocks; b++) {
    const int l = m->level[b], *idx = &m->index[3 * b];
    for (int icode = 0; icode < 27; icode++) {
      const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, icode / 9 - 1};
      int skipped = 0, nei[3];
      for (int d = 0; d < 3; d++) {
        const int n = nblk(m, l, d), skin = idx[d] == 0 || idx[d] == n - 1, skip = idx[d] == 0 ? -1 : 1;
        if (m->bc[d] != ORC_BC_PERIODIC && code[d] == skip && skin) skipped = 1;
        nei[d] = idx[d] + code[d];
      }
      int v = -1;
      if (!skipped) {
        const int st = tree_state(m, l, nei);
        if (st == 1) v = leaf_at(m, l, nei);
        else if (st == -1) v = -3;
        else {
          int w[3];
          for (int d = 0; d < 3; d++) { const int n = nblk(m, l, d); w[d] = (((nei[d] % n) + n) % n) >> 1; }
          v = -100 - leaf_at(m, l - 1, w);
        }
      }
      out[b * 27 + icode] = v;
    }
  }
}

/* ComputeVorticity::operator(), main.cpp:8726-8746: KernelVorticity (8624-8724) followed by tmpV *= 1/h^3.
 * The kernel's face-flux branch (8646-8722) is dead code in the reference: it looks for the BlockCase in `info.auxiliary` of
 * the VELOCITY grid's Info, which is always nullptr (Info::setup 395; only flux-corrected grids get one, and vel never is
 * one), so the faces stay zero and FluxCorrectionMPI adds 0 to the coarse cells: the result is the plain curl of the ghosted
 * tile.  (Confirmed against the compiled reference on two- and three-level meshes.)  A uniform grid is the single-level
 * special case of a mesh. */
void orc_mesh_vorticity(const orc_mesh *m, const double *vel, double *tmpV) {
#define L(x, y, z, c) FT(&t, x, y, z, c)
#pragma omp parallel num_threads(mesh_threads(m))
  {
  tile_t t;
  tile_init(&t, 3, 1, -1, 2, 0);
#pragma omp for schedule(dynamic, 1)
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, vel, b, &t);
    const double h = orc_mesh_h(m, b), inv2h = .5 * h * h, fac = 1.0 / (h * h * h);
    for (int z = 0; z < BS; z++)
      for (int y = 0; y < BS; y++)
        for (int x = 0; x < BS; x++) {
          double *o = tmpV + (b * BS3 + (z * BS + y) * BS + x) * 3;
          o[0] = inv2h * ((L(x, y + 1, z, 2) - L(x, y - 1, z, 2)) - (L(x, y, z + 1, 1) - L(x, y, z - 1, 1)));
          o[1] = inv2h * ((L(x, y, z + 1, 0) - L(x, y, z - 1, 0)) - (L(x + 1, y, z, 2) - L(x - 1, y, z, 2)));
          o[2] = inv2h * ((L(x + 1, y, z, 1) - L(x - 1, y, z, 1)) - (L(x, y + 1, z, 0) - L(x, y - 1, z, 0)));
          for (int c = 0; c < 3; c++) o[c] *= fac;
        }
  }
  tile_free(&t);
  }
#undef L
}

/* compute<ScalarLab>(GradChiOnTmp(sim), sim.chi), main.cpp:8540-8600 -- the chi-driven half of adaptMesh's tagging input (15180-15182):
 * on the tensorial [-2,3) tile of chi (zero-gradient domain faces), scanned in z, y, x order over the block grown by `offset` cells
 * (2 on the finest level, else 1): values are clamped to [0,1]; the FIRST cell with 1e-5 < chi < 0.9 (an obstacle surface within
 * reach) writes 1e10 into tmpV.u[0] of the centre cells the reference lists (six distinct ones of its eight assignments) and ends
 * the scan; interior cells with chi > 0.9 met BEFORE that are cleared (deep inside the body: no vorticity-driven refinement).
 * Before the scan, on level levelMaxVorticity - 1 (when that is below levelMax) vorticity magnitudes >= Rtol are capped to
 * (Rtol + Ctol) / 2 so that such blocks are neither refined nor compressed. */
void orc_mesh_grad_chi_on_tmp(const orc_mesh *m, const double *chi, double *tmpV, double Rtol, double Ctol, int level_max_vorticity) {
  tile_t t;
  tile_init(&t, 1, 0, -2, 3, 1);
  for (long b = 0; b < m->nblocks; b++) {
    orc_mesh_lab(m, chi, b, &t);
    double *T = tmpV + b * BS3 * 3;
    if (m->level[b] == level_max_vorticity - 1 && level_max_vorticity < m->level_max)
      for (int i = 0; i < BS3; i++) {
        double *e = T + 3 * i;
        if (sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) >= Rtol) { e[0] = 0.5 * (Rtol + Ctol); e[1] = 0.0; e[2] = 0.0; }
      }
    const int offset = m->level[b] == m->level_max - 1 ? 2 : 1;
    int done = 0;
    for (int z = -offset; z < BS + offset && !done; z++)
      for (int y = -offset; y < BS + offset && !done; y++)
        for (int x = -offset; x < BS + offset; x++) {
          double v = FT(&t, x, y, z, 0);
          v = v < 1.0 ? v : 1.0;   /* std::min(lab, 1.0), 8564 */
          v = v > 0.0 ? v : 0.0;   /* std::max(lab, 0.0), 8565 */
          if (v > 0.00001 && v < 0.9) {
            static const int cc[6][3] = {{3, 3, 3}, {4, 3, 3}, {3, 4, 3}, {3, 3, 4}, {4, 4, 4}, {4, 3, 4}}; /* 8567-8590, duplicates dropped */
            for (int q = 0; q < 6; q++) T[((cc[q][2] * BS + cc[q][1]) * BS + cc[q][0]) * 3] = 1e10;
            done = 1;
            break;
          } else if (v > 0.9 && z >= 0 && z < BS && y >= 0 && y < BS && x >= 0 && x < BS) {
            double *e = T + ((z * BS + y) * BS + x) * 3;
            e[0] = 0.0; e[1] = 0.0; e[2] = 0.0;
          }
        }
  }
  tile_free(&t);
}

/* TagLoadedBlock (5566-5582) with the level clamps of TagBlocksVector (5207-5211) on every block of a mesh */
void orc_mesh_tag(const orc_mesh *m, const double *f, int nc, double rtol, double ctol, signed char *states) {
  for (long b = 0; b < m->nblocks; b++) {
    double Linf = 0.0;
    for (int i = 0; i < BS3; i++) {
      const double *e = f + ((long)b * BS3 + i) * nc;
      const double mag = nc == 3 ? sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) : e[0]; /* magnitude(), 5783 / 5873-5879 */
      Linf = fmax(Linf, fabs(mag));
    }
    signed char s = 0;
    if (Linf > rtol) s = 1;
    else if (Linf < ctol) s = -1;
    if (s == 1 && m->level[b] == m->level_max - 1) s = 0;
    if (s == -1 && m->level[b] == 0) s = 0;
    states[b] = s;
  }
}

/* ======================= mesh adaptation on multi-level meshes ======================= */
/* MeshAdaptation::ValidStates, main.cpp:5330-5492, one rank (UpdateBoundary is a no-op then): in/out states[nb] in
 * {-1 Compress, 0 Leave, 1 Refine}, m_vInfo order.  Refinement propagates to coarser neighbours level by level (finest
 * first), a block next to finer blocks or to a refining same-level block may not compress, and an octet compresses only
 * if all eight siblings exist and agree. */
void orc_mesh_valid_states(const orc_mesh *m, signed char *st) {
  const int lmax = m->level_max;
  for (long b = 0; b < m->nblocks; b++)
    if ((st[b] == 1 && m->level[b] == lmax - 1) || (st[b] == -1 && m->level[b] == 0)) st[b] = 0;
  for (int lv = lmax - 1; lv >= 0; lv--) {
    for (long b = 0; b < m->nblocks; b++) {
      if (!(m->level[b] == lv && st[b] != 1 && m->level[b] != lmax - 1)) continue;
      const int *idx = &m->index[3 * b];
      for (int icode = 0; icode < 27; icode++) {
        if (st[b] == 1) break;
        if (icode == 13) continue;
        const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, (icode / 9) % 3 - 1};
        int skipped = 0, nei[3];
        for (int d = 0; d < 3; d++) {
          const int n = nblk(m, lv, d), skin = idx[d] == 0 || idx[d] == n - 1, skip = idx[d] == 0 ? -1 : 1;
          if (m->bc[d] != ORC_BC_PERIODIC && code[d] == skip && skin) skipped = 1;
          nei[d] = idx[d] + code[d];
        }
        if (skipped) continue;
        if (tree_state(m, lv, nei) != -1) continue; /* CheckFiner */
        if (st[b] == -1) st[b] = 0;
        const int tmp = abs(code[0]) + abs(code[1]) + abs(code[2]);
        const int Bstep = tmp == 2 ? 3 : (tmp == 3 ? 4 : 1);
        for (int B = 0; B <= 3; B += Bstep) {
          const int aux = (abs(code[0]) == 1) ? (B % 2) : (B / 2);
          const int fi[3] = {2 * idx[0] + (code[0] > 0 ? code[0] : 0) + code[0] + (B % 2) * (1 - abs(code[0]) > 0 ? 1 - abs(code[0]) : 0),
                             2 * idx[1] + (code[1] > 0 ? code[1] : 0) + code[1] + aux * (1 - abs(code[1]) > 0 ? 1 - abs(code[1]) : 0),
                             2 * idx[2] + (code[2] > 0 ? code[2] : 0) + code[2] + (B / 2) * (1 - abs(code[2]) > 0 ? 1 - abs(code[2]) : 0)};
          const int fb = leaf_at(m, lv + 1, fi);
          if (fb >= 0 && st[fb] == 1) { st[b] = 1; break; }
        }
      }
    }
    if (lv == 0) break;
    for (long b = 0; b < m->nblocks; b++) {
      if (!(m->level[b] == lv && st[b] == -1)) continue;
      const int *idx = &m->index[3 * b];
      for (int icode = 0; icode < 27; icode++) {
        if (icode == 13) continue;
        const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, (icode / 9) % 3 - 1};
        int skipped = 0, nei[3];
        for (int d = 0; d < 3; d++) {
          const int n = nblk(m, lv, d), skin = idx[d] == 0 || idx[d] == n - 1, skip = idx[d] == 0 ? -1 : 1;
          if (m->bc[d] != ORC_BC_PERIODIC && code[d] == skip && skin) skipped = 1;
          nei[d] = idx[d] + code[d];
        }
        if (skipped) continue;
        const int nb_ = leaf_at(m, lv, nei);
        if (nb_ >= 0 && st[nb_] == 1) { st[b] = 0; break; }
      }
    }
  }
  /* sibling agreement, 5451-5491 (the reference walks the blocks sequentially and clears the octet through the
   * all-blocks table; the net effect is: an octet keeps Compress only if all eight siblings exist and are Compress) */
  for (long b = 0; b < m->nblocks; b++) {
    if (st[b] != -1) continue;
    const int l = m->level[b], *idx = &m->index[3 * b];
    int all = 1;
    for (int q = 0; q < 8 && all; q++) {
      const int s[3] = {2 * (idx[0] / 2) + (q & 1), 2 * (idx[1] / 2) + ((q >> 1) & 1), 2 * (idx[2] / 2) + (q >> 2)};
      const int sb = leaf_at(m, l, s);
      if (sb < 0 || st[sb] != -1) all = 0;
    }
    if (!all) st[b] = -2; /* marked; cleared below so that the test above keeps seeing the original states */
  }
  for (long b = 0; b < m->nblocks; b++)
    if (st[b] == -2) st[b] = 0;
}

/* the leaf set after MeshAdaptation::Adapt (5086-5159) applied the (valid) states: Refine -> eight children, an octet of
 * Compress -> its parent, Leave -> unchanged.  levels/Zs must hold nblocks + 7*nrefine entries; returns the new count. */
long orc_mesh_adapted_leaves(const orc_mesh *m, const signed char *st, int *levels, long long *Zs) {
  long n = 0;
  for (long b = 0; b < m->nblocks; b++) {
    const int l = m->level[b], *idx = &m->index[3 * b];
    if (st[b] == 1) {
      for (int q = 0; q < 8; q++) {
        levels[n] = l + 1;
        Zs[n++] = orc_sfc_forward(m->sfc, l + 1, 2 * idx[0] + (q & 1), 2 * idx[1] + ((q >> 1) & 1), 2 * idx[2] + (q >> 2));
      }
    } else if (st[b] == -1) {
      if (idx[0] % 2 == 0 && idx[1] % 2 == 0 && idx[2] % 2 == 0) {
        levels[n] = l - 1;
        Zs[n++] = orc_sfc_forward(m->sfc, l - 1, idx[0] / 2, idx[1] / 2, idx[2] / 2);
      }
    } else {
      levels[n] = l;
      Zs[n++] = m->Z[b];
    }
  }
  return n;
}

/* Block ownership after MeshAdaptation::Adapt on several ranks (main.cpp:5086-5159 + LoadBalancer 4660-5022).
 * `mo` = the old mesh (all leaves of all ranks, m_vInfo order), owner[b] = rank of leaf b (ranks own contiguous runs of the
 * order), st = valid states, `mn` = the adapted mesh; new_owner[slot of mn] receives the rank after the adaptation:
 *   - the children of a refined block are created on the parent's rank (refine_1/refine_2, 5227-5271);
 *   - the eight siblings of a compressed octet are first gathered on the rank of the base block, the one with even indices
 *     (PrepareCompression 4729-4804), so the parent appears there;
 *   - Balance_Diffusion (4805-4905) with the block counts after the adaptation: if max/min > 1.01 (or a rank is empty) ->
 *     Balance_Global (4906-5021): the rank-major concatenation of the sorted per-rank lists is cut into `size` pieces, the first
 *     total % size one block longer; else each rank passes (my - neighbour)/4 blocks (C integer division) from the start of
 *     its sorted list to the left neighbour / from the end to the right one.
 * Pinned against the reference run under a real MPI (tests/golden/adapt_mpi.npz). */
static int cmp_long(const void *a, const void *b) { const long x = *(const long *)a, y = *(const long *)b; return (x > y) - (x < y); }
void orc_mesh_adapted_owners(const orc_mesh *mo, const int *owner, const signed char *st, int nranks, const orc_mesh *mn, int *new_owner) {
  long **list = (long **)calloc(nranks, sizeof(long *));
  long *cnt = (long *)calloc(nranks, sizeof(long));
  for (int r = 0; r < nranks; r++) list[r] = (long *)malloc(sizeof(long) * (mn->nblocks + 1));
  for (long b = 0; b < mo->nblocks; b++) {
    const int l = mo->level[b], *idx = &mo->index[3 * b], r = owner[b];
    if (st[b] == 1) {
      for (int q = 0; q < 8; q++) {
        const int c[3] = {2 * idx[0] + (q & 1), 2 * idx[1] + ((q >> 1) & 1), 2 * idx[2] + (q >> 2)};
        list[r][cnt[r]++] = leaf_at(mn, l + 1, c);
      }
    } else if (st[b] == -1) {
      if (idx[0] % 2 == 0 && idx[1] % 2 == 0 && idx[2] % 2 == 0) {
        const int c[3] = {idx[0] / 2, idx[1] / 2, idx[2] / 2};
        list[r][cnt[r]++] = leaf_at(mn, l - 1, c);
      }
    } else
      list[r][cnt[r]++] = leaf_at(mn, l, idx);
  }
  long mx = cnt[0], mi = cnt[0], total = 0;
  for (int r = 0; r < nranks; r++) {
    qsort(list[r], cnt[r], sizeof(long), cmp_long); /* slots of mn are in blockID_2 order */
    if (cnt[r] > mx) mx = cnt[r];
    if (cnt[r] < mi) mi = cnt[r];
    total += cnt[r];
  }
  if (mi == 0 || (double)mx / mi > 1.01) { /* 4817-4820 */
    long pos = 0;
    int r = 0;
    long left = total / nranks + (0 < total % nranks ? 1 : 0);
    for (int q = 0; q < nranks; q++)
      for (long i = 0; i < cnt[q]; i++) {
        while (left == 0) { r++; left = total / nranks + (r < total % nranks ? 1 : 0); }
        new_owner[list[q][i]] = r;
        left--;
        pos++;
      }
  } else {
    for (int r = 0; r < nranks; r++)
      for (long i = 0; i < cnt[r]; i++) new_owner[list[r][i]] = r;
    for (int r = 0; r < nranks; r++) {
      const long fl = r == 0 ? 0 : (cnt[r] - cnt[r - 1]) / 4, fr = r == nranks - 1 ? 0 : (cnt[r] - cnt[r + 1]) / 4; /* 4836-4839 */
      for (long i = 0; i < fl; i++) new_owner[list[r][i]] = r - 1;
      for (long i = 0; i < fr; i++) new_owner[list[r][cnt[r] - 1 - i]] = r + 1;
    }
  }
  for (int r = 0; r < nranks; r++) free(list[r]);
  free(list); free(cnt);
}

/* field data on the adapted mesh (refine_1 + RefineBlocks 5227-5249, 5493-5565 from the parent's tensorial [-1,2) tile on
 * the OLD mesh; compress 5272-5329; unchanged blocks copied) */
void orc_mesh_transfer(const orc_mesh *mo, const orc_mesh *mn, const double *fo, double *fn, int nc, int is_vector) {
  tile_t t;
  tile_init(&t, nc, is_vector, -1, 2, 1);
#define Lb(x, y, z) FT(&t, x, y, z, c)
  for (long b = 0; b < mn->nblocks; b++) {
    const int l = mn->level[b], *idx = &mn->index[3 * b];
    double *out = fn + b * BS3 * nc;
    const int same = leaf_at(mo, l, idx);
    if (same >= 0) { memcpy(out, fo + (long)same * BS3 * nc, BS3 * nc * sizeof(double)); continue; }
    const int pidx[3] = {idx[0] >> 1, idx[1] >> 1, idx[2] >> 1};
    const int par = l > 0 ? leaf_at(mo, l - 1, pidx) : -1;
    if (par >= 0) { /* this block is child (I,J,K) of a refined parent */
      const int I = idx[0] & 1, J = idx[1] & 1, K = idx[2] & 1;
      orc_mesh_lab(mo, fo, par, &t);
#define B(i, j, k) out[(((k) * BS + (j)) * BS + (i)) * nc + c]
      for (int k = 0; k < BS; k += 2)
        for (int j = 0; j < BS; j += 2)
          for (int i = 0; i < BS; i += 2)
            for (int c = 0; c < nc; c++) {
              const int x = i / 2 + 4 * I, y = j / 2 + 4 * J, z = k / 2 + 4 * K;
              const double dudx = 0.5 * (Lb(x + 1, y, z) - Lb(x - 1, y, z));
              const double dudy = 0.5 * (Lb(x, y + 1, z) - Lb(x, y - 1, z));
              const double dudz = 0.5 * (Lb(x, y, z + 1) - Lb(x, y, z - 1));
              const double dudx2 = (Lb(x + 1, y, z) + Lb(x - 1, y, z)) - 2.0 * Lb(x, y, z);
              const double dudy2 = (Lb(x, y + 1, z) + Lb(x, y - 1, z)) - 2.0 * Lb(x, y, z);
              const double dudz2 = (Lb(x, y, z + 1) + Lb(x, y, z - 1)) - 2.0 * Lb(x, y, z);
              const double dudxdy = 0.25 * ((Lb(x + 1, y + 1, z) + Lb(x - 1, y - 1, z)) - (Lb(x + 1, y - 1, z) + Lb(x - 1, y + 1, z)));
              const double dudxdz = 0.25 * ((Lb(x + 1, y, z + 1) + Lb(x - 1, y, z - 1)) - (Lb(x + 1, y, z - 1) + Lb(x - 1, y, z + 1)));
              const double dudydz = 0.25 * ((Lb(x, y + 1, z + 1) + Lb(x, y - 1, z - 1)) - (Lb(x, y + 1, z - 1) + Lb(x, y - 1, z + 1)));
              const double u = Lb(x, y, z), q2 = 0.03125 * (dudx2 + dudy2 + dudz2);
              B(i, j, k) = u + 0.25 * (-(1.0) * dudx - dudy - dudz) + q2 + 0.0625 * (dudxdy + dudxdz + dudydz);
              B(i + 1, j, k) = u + 0.25 * (dudx - dudy - dudz) + q2 + 0.0625 * (-(1.0) * dudxdy - dudxdz + dudydz);
              B(i, j + 1, k) = u + 0.25 * (-(1.0) * dudx + dudy - dudz) + q2 + 0.0625 * (-(1.0) * dudxdy + dudxdz - dudydz);
              B(i + 1, j + 1, k) = u + 0.25 * (dudx + dudy - dudz) + q2 + 0.0625 * (dudxdy - dudxdz - dudydz);
              B(i, j, k + 1) = u + 0.25 * (-(1.0) * dudx - dudy + dudz) + q2 + 0.0625 * (dudxdy - dudxdz - dudydz);
              B(i + 1, j, k + 1) = u + 0.25 * (dudx - dudy + dudz) + q2 + 0.0625 * (-(1.0) * dudxdy + dudxdz - dudydz);
              B(i, j + 1, k + 1) = u + 0.25 * (-(1.0) * dudx + dudy + dudz) + q2 + 0.0625 * (-(1.0) * dudxdy - dudxdz + dudydz);
              B(i + 1, j + 1, k + 1) = u + 0.25 * (dudx + dudy + dudz) + q2 + 0.0625 * (dudxdy + dudxdz + dudydz);
            }
#undef B
      continue;
    }
    /* parent of a compressed octet */
    for (int q = 0; q < 8; q++) {
      const int I = q & 1, J = (q >> 1) & 1, K = q >> 2;
      const int ci[3] = {2 * idx[0] + I, 2 * idx[1] + J, 2 * idx[2] + K};
      const int cb = leaf_at(mo, l + 1, ci);
      if (cb < 0) { fprintf(stderr, "orc_mesh_transfer: block without a source\n"); abort(); }
      const double *src = fo + (long)cb * BS3 * nc;
#define S(i, j, k, c) src[(((k) * BS + (j)) * BS + (i)) * nc + (c)]
      for (int k = 0; k < BS; k += 2)
        for (int j = 0; j < BS; j += 2)
          for (int i = 0; i < BS; i += 2)
            for (int c = 0; c < nc; c++)
              out[(((k / 2 + 4 * K) * BS + (j / 2 + 4 * J)) * BS + (i / 2 + 4 * I)) * nc + c] =
                  0.125 * ((S(i, j, k, c) + S(i + 1, j + 1, k + 1, c)) + (S(i + 1, j, k, c) + S(i, j + 1, k + 1, c)) +
                           (S(i, j + 1, k, c) + S(i + 1, j, k + 1, c)) + (S(i + 1, j + 1, k, c) + S(i, j, k + 1, c)));
#undef S
    }
  }
#undef Lb
  tile_free(&t);
}

/* ======================= obstacle operators ======================= */
/* One obstacle = the ObstacleBlocks it owns (7256-7263): block ids, chi[n][512], udef[n][512][3] (AoS), and its rigid motion.
 * KernelPenalization::visit + kernelFinalizePenalizationForce, main.cpp:13853-13938: vel is penalised in place towards the
 * obstacle velocity, force6 = force[3], torque[3] summed over the obstacle's blocks in block order. */
void orc_mesh_penalize(const orc_mesh *m, double *vel, const double *chi_field, long n, const long long *ids, const double *chi,
                       const double *udef, const double rigid[9], double dt, double lambda, int implicit, double force6[6]) {
  const double invdt = 1.0 / dt, lambdaFac = implicit ? lambda : invdt;
  const double *CM = rigid, *vt = rigid + 3, *om = rigid + 6;
  double M[6] = {0, 0, 0, 0, 0, 0};
  /* blocks are visited in m_vInfo order (the obstacleBlocks vector is indexed by blockID) */
  long *order = (long *)malloc(n * sizeof(long));
  for (long i = 0; i < n; i++) order[i] = i;
  for (long i = 1; i < n; i++) { long k = order[i], j = i; while (j > 0 && ids[order[j - 1]] > ids[k]) { order[j] = order[j - 1]; j--; } order[j] = k; }
  for (long oi = 0; oi < n; oi++) {
    const long i = order[oi], b = (long)ids[i];
    const double h = orc_mesh_h(m, b), dv = pow(h, 3);
    const double org[3] = {m->index[3 * b] * BS * h, m->index[3 * b + 1] * BS * h, m->index[3 * b + 2] * BS * h};
    double F[6] = {0, 0, 0, 0, 0, 0};
    for (int iz = 0; iz < BS; iz++)
      for (int iy = 0; iy < BS; iy++)
        for (int ix = 0; ix < BS; ix++) {
          const int c = (iz * BS + iy) * BS + ix;
          const double CHI = chi[i * BS3 + c];
          if (chi_field[b * BS3 + c] > CHI) continue;
          if (CHI <= 0) continue;
          double p[3] = {org[0] + h * (ix + 0.5), org[1] + h * (iy + 0.5), org[2] + h * (iz + 0.5)};
          p[0] -= CM[0]; p[1] -= CM[1]; p[2] -= CM[2];
          const double *U = udef + (i * BS3 + c) * 3;
          const double UT[3] = {vt[0] + om[1] * p[2] - om[2] * p[1] + U[0], vt[1] + om[2] * p[0] - om[0] * p[2] + U[1],
                                vt[2] + om[0] * p[1] - om[1] * p[0] + U[2]};
          const double X = implicit ? (CHI > 0.5 ? 1.0 : 0.0) : CHI;
          const double penalFac = implicit ? X * lambdaFac / (1 + X * lambdaFac * dt) : X * lambdaFac;
          double *v = vel + (b * BS3 + c) * 3;
          const double FPX = penalFac * (UT[0] - v[0]), FPY = penalFac * (UT[1] - v[1]), FPZ = penalFac * (UT[2] - v[2]);
          v[0] = v[0] + dt * FPX; v[1] = v[1] + dt * FPY; v[2] = v[2] + dt * FPZ;
          F[0] += dv * FPX; F[1] += dv * FPY; F[2] += dv * FPZ;
          F[3] += dv * (p[1] * FPZ - p[2] * FPY);
          F[4] += dv * (p[2] * FPX - p[0] * FPZ);
          F[5] += dv * (p[0] * FPY - p[1] * FPX);
        }
    for (int k = 0; k < 6; k++) M[k] += F[k];
  }
  for (int k = 0; k < 6; k++) force6[k] = M[k];
  free(order);
}

/* kernelUpdateTmpV, main.cpp:14948-14979: tmpV += udef where the block's chi does not exceed the obstacle's */
void orc_mesh_update_tmpv(const orc_mesh *m, double *tmpV, const double *chi_field, long n, const long long *ids, const double *chi,
                          const double *udef) {
  (void)m;
  for (long i = 0; i < n; i++) {
    const long b = (long)ids[i];
    for (int c = 0; c < BS3; c++) {
      if (chi_field[b * BS3 + c] > chi[i * BS3 + c]) continue;
      for (int k = 0; k < 3; k++) tmpV[(b * BS3 + c) * 3 + k] += udef[(i * BS3 + c) * 3 + k];
    }
  }
}
