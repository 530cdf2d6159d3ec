/* TEST INFRASTRUCTURE — see cup3d_oracle.h.  CPU restatement of the reference
 * hot path (slitvinov/CUP3D main.cpp); the product never links this file. */
#include "cup3d_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BS 8
#define BS3 512

/* ======================= Hilbert space-filling curve ======================= */
/* Skilling's transpose form, restating main.cpp:107-192. */
struct orc_sfc {
  int bx, by, bz, level_max, regular, base_level;
  long long *zsave; /* [bz][by][bx] -> compacted level-0 index        (main.cpp:234) */
  int *inv;         /* compacted level-0 index -> (i,j,k)             (main.cpp:231-233) */
};

static long long axes_to_transpose(const int xin[3], int b) { /* main.cpp:107-152 */
  if (b == 0) return 0;
  int X[3] = {xin[0], xin[1], xin[2]};
  const int M = 1 << (b - 1);
  for (int Q = M; Q > 1; Q >>= 1) {
    const int P = Q - 1;
    for (int i = 0; i < 3; i++) {
      if (X[i] & Q) X[0] ^= P;
      else { const int t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
    }
  }
  for (int i = 1; i < 3; i++) X[i] ^= X[i - 1];
  int t = 0;
  for (int Q = M; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  for (int i = 0; i < 3; i++) X[i] ^= t;
  long long r = 0;
  for (int lev = 0; lev < b; lev++) {
    r += ((long long)((X[2] >> lev) & 1)) << (3 * lev);
    r += ((long long)((X[1] >> lev) & 1)) << (3 * lev + 1);
    r += ((long long)((X[0] >> lev) & 1)) << (3 * lev + 2);
  }
  return r;
}

static void transpose_to_axes(long long index, long long X[3], int b) { /* main.cpp:153-192 */
  X[0] = X[1] = X[2] = 0;
  if (b == 0 && index == 0) return;
  for (int bit = 0; index > 0; bit++) {
    X[2] += (index & 1) << bit; index >>= 1;
    X[1] += (index & 1) << bit; index >>= 1;
    X[0] += (index & 1) << bit; index >>= 1;
  }
  const int N = 2 << (b - 1);
  long long t = X[2] >> 1;
  for (int i = 2; i >= 1; i--) X[i] ^= X[i - 1];
  X[0] ^= t;
  for (int Q = 2; Q != N; Q <<= 1) {
    const int P = Q - 1;
    for (int i = 2; i >= 0; i--) {
      if (X[i] & Q) X[0] ^= P;
      else { t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
    }
  }
}

orc_sfc *orc_sfc_create(int bx, int by, int bz, int level_max) { /* main.cpp:196-236 */
  orc_sfc *s = (orc_sfc *)calloc(1, sizeof *s);
  s->bx = bx; s->by = by; s->bz = bz; s->level_max = level_max;
  int nmax = bx > by ? bx : by; if (bz > nmax) nmax = bz;
  s->base_level = (int)(log((double)nmax) / log(2.0));
  if ((double)s->base_level < log((double)nmax) / log(2.0)) s->base_level++;
  const long n = (long)bx * by * bz;
  s->zsave = (long long *)malloc(n * sizeof(long long));
  s->inv = (int *)malloc(n * 3 * sizeof(int));
  s->regular = 1;
  for (int k = 0; k < bz; k++)
    for (int j = 0; j < by; j++)
      for (int i = 0; i < bx; i++) {
        const int c[3] = {i, j, k};
        long long index = axes_to_transpose(c, s->base_level);
        long long sub = 0;
        for (long long h = 0; h < index; h++) {
          long long X[3];
          transpose_to_axes(h, X, s->base_level);
          if (X[0] >= bx || X[1] >= by || X[2] >= bz) sub++;
        }
        index -= sub;
        if (sub > 0) s->regular = 0;
        s->inv[3 * index + 0] = i; s->inv[3 * index + 1] = j; s->inv[3 * index + 2] = k;
        s->zsave[((long)k * by + j) * bx + i] = index;
      }
  return s;
}
void orc_sfc_destroy(orc_sfc *s) { if (s) { free(s->zsave); free(s->inv); free(s); } }

long long orc_sfc_forward(const orc_sfc *s, int l, int i, int j, int k) { /* main.cpp:237-255 */
  const int aux = 1 << l;
  if (l >= s->level_max) return 0;
  if (!s->regular) {
    const int I = i / aux, J = j / aux, K = k / aux;
    const int c[3] = {i - I * aux, j - J * aux, k - K * aux};
    long long r = axes_to_transpose(c, l);
    r += s->zsave[((long)J + (long)K * s->by) * s->bx + I] * aux * aux * aux;
    return r;
  }
  const int c[3] = {i, j, k};
  return axes_to_transpose(c, l + s->base_level);
}

void orc_sfc_inverse(const orc_sfc *s, long long Z, int l, int ijk[3]) { /* main.cpp:256-276 */
  long long X[3];
  if (s->regular) {
    transpose_to_axes(Z, X, l + s->base_level);
    ijk[0] = (int)X[0]; ijk[1] = (int)X[1]; ijk[2] = (int)X[2];
  } else {
    const long long aux = 1 << l;
    transpose_to_axes(Z % (aux * aux * aux), X, l);
    const long long idx = Z / (aux * aux * aux);
    ijk[0] = (int)(X[0] + s->inv[3 * idx + 0] * aux);
    ijk[1] = (int)(X[1] + s->inv[3 * idx + 1] * aux);
    ijk[2] = (int)(X[2] + s->inv[3 * idx + 2] * aux);
  }
}

long long orc_sfc_encode(const orc_sfc *s, int level, const int index[3]) { /* main.cpp:287-318 */
  long long r = 0;
  int ix = index[0], iy = index[1], iz = index[2];
  for (int l = level; l >= 0; l--) {
    r += orc_sfc_forward(s, l, ix, iy, iz);
    ix /= 2; iy /= 2; iz /= 2;
  }
  ix = 2 * index[0]; iy = 2 * index[1]; iz = 2 * index[2];
  for (int l = level + 1; l < s->level_max; l++) {
    long long Zc = orc_sfc_forward(s, l, ix, iy, iz);
    Zc -= Zc % 8;
    r += Zc;
    int c[3];
    orc_sfc_inverse(s, Zc, l, c);
    ix = 2 * c[0]; iy = 2 * c[1]; iz = 2 * c[2];
  }
  return r + level;
}

void orc_info_tables(const orc_sfc *s, const int bpd[3], int level, const int index[3],
                     long long nei[27], long long child[8], long long *parent) { /* main.cpp:396-417 */
  const int two = 1 << level;
  const int B[3] = {bpd[0] * two, bpd[1] * two, bpd[2] * two};
  for (int i = -1; i < 2; i++)
    for (int j = -1; j < 2; j++)
      for (int k = -1; k < 2; k++)
        nei[((i + 1) * 3 + (j + 1)) * 3 + (k + 1)] =
            orc_sfc_forward(s, level, (index[0] + i + B[0]) % B[0], (index[1] + j + B[1]) % B[1], (index[2] + k + B[2]) % B[2]);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++)
        child[(i * 2 + j) * 2 + k] = orc_sfc_forward(s, level + 1, 2 * index[0] + i, 2 * index[1] + j, 2 * index[2] + k);
  *parent = (level == 0) ? 0
                         : orc_sfc_forward(s, level - 1, (index[0] / 2 + B[0]) % B[0], (index[1] / 2 + B[1]) % B[1], (index[2] / 2 + B[2]) % B[2]);
}

/* =============================== uniform grid =============================== */
struct orc_grid {
  orc_sfc *sfc;
  int bpd[3], level_max, level, bc[3];
  int nb[3]; /* blocks per dimension at `level` */
  long nblocks;
  double h, maxextent;
  long long *Z, *id2;
  int *index;   /* [nblocks][3] */
  long *slot_of; /* [nb2][nb1][nb0] -> slot */
};

typedef struct { long long id2, Z; } sort_rec;
static int cmp_rec(const void *a, const void *b) {
  const sort_rec *x = (const sort_rec *)a, *y = (const sort_rec *)b;
  return (x->id2 > y->id2) - (x->id2 < y->id2);
}

orc_grid *orc_grid_create(int bx, int by, int bz, int level_max, int level, double maxextent, const int bc[3]) {
  orc_grid *g = (orc_grid *)calloc(1, sizeof *g);
  g->sfc = orc_sfc_create(bx, by, bz, level_max);
  g->bpd[0] = bx; g->bpd[1] = by; g->bpd[2] = bz;
  g->level_max = level_max; g->level = level; g->maxextent = maxextent;
  for (int d = 0; d < 3; d++) { g->bc[d] = bc[d]; g->nb[d] = g->bpd[d] << level; }
  g->nblocks = (long)g->nb[0] * g->nb[1] * g->nb[2];
  int mb = bx > by ? bx : by; if (bz > mb) mb = bz;
  const double h0 = maxextent / (double)(mb * BS); /* main.cpp:1059-1062 */
  g->h = h0 / (double)(1 << level);
  g->Z = (long long *)malloc(g->nblocks * sizeof(long long));
  g->id2 = (long long *)malloc(g->nblocks * sizeof(long long));
  g->index = (int *)malloc(g->nblocks * 3 * sizeof(int));
  g->slot_of = (long *)malloc(g->nblocks * sizeof(long));
  sort_rec *rec = (sort_rec *)malloc(g->nblocks * sizeof(sort_rec));
  for (long long z = 0; z < g->nblocks; z++) { /* every Z of the level exists (main.cpp:2970-2986, 1 rank) */
    int c[3];
    orc_sfc_inverse(g->sfc, z, level, c);
    rec[z].Z = z;
    rec[z].id2 = orc_sfc_encode(g->sfc, level, c);
  }
  qsort(rec, g->nblocks, sizeof(sort_rec), cmp_rec); /* FillPos: sort by blockID_2, main.cpp:944 */
  for (long s = 0; s < g->nblocks; s++) {
    g->Z[s] = rec[s].Z; g->id2[s] = rec[s].id2;
    orc_sfc_inverse(g->sfc, rec[s].Z, level, &g->index[3 * s]);
    const int *c = &g->index[3 * s];
    g->slot_of[((long)c[2] * g->nb[1] + c[1]) * g->nb[0] + c[0]] = s;
  }
  free(rec);
  return g;
}
void orc_grid_destroy(orc_grid *g) {
  if (!g) return;
  orc_sfc_destroy(g->sfc); free(g->Z); free(g->id2); free(g->index); free(g->slot_of); free(g);
}
long orc_grid_nblocks(const orc_grid *g) { return g->nblocks; }
double orc_grid_h(const orc_grid *g) { return g->h; }
void orc_grid_tables(const orc_grid *g, long long *o, double *geom) {
  for (long s = 0; s < g->nblocks; s++) {
    o[6 * s] = g->level; o[6 * s + 1] = g->Z[s];
    o[6 * s + 2] = g->index[3 * s]; o[6 * s + 3] = g->index[3 * s + 1]; o[6 * s + 4] = g->index[3 * s + 2];
    o[6 * s + 5] = g->id2[s];
    geom[4 * s] = g->h;
    for (int d = 0; d < 3; d++) geom[4 * s + 1 + d] = g->index[3 * s + d] * BS * g->h; /* main.cpp:1066-1068 */
  }
}
void orc_partition(long long total, int rank, int size, long long *z_start, long long *count) { /* main.cpp:2970-2980 */
  long long my = total / size;
  if ((long long)rank < total % size) my++;
  long long n0 = rank * (total / size);
  if (total % size > 0) {
    if ((long long)rank < total % size) n0 += rank;
    else n0 += total % size;
  }
  *z_start = n0; *count = my;
}

/* ============================ ghosted tile ("lab") ============================ */
/* BlockLab::load + SameLevelExchange + _apply_bc restricted to what the star-shaped
 * kernels read: the six face slabs of width w (main.cpp:3623-3743, 3823-3876).
 * Domain faces: vector fields follow BlockLabBC (wall: all components negated
 * 6369-6394; freespace: copy, normal component negated 6133-6154), scalar fields
 * BlockLabNeumann3D (copy of the face cell, 5929-5978).  The ghost value is the
 * face cell for every ghost layer. */
static long nbr_slot(const orc_grid *g, long b, int d, int side) { /* -1 = domain face with BC */
  int c[3] = {g->index[3 * b], g->index[3 * b + 1], g->index[3 * b + 2]};
  const int at_face = side ? (c[d] == g->nb[d] - 1) : (c[d] == 0);
  if (at_face && g->bc[d] != ORC_BC_PERIODIC) return -1;
  c[d] = (c[d] + (side ? 1 : -1) + g->nb[d]) % g->nb[d];
  return g->slot_of[((long)c[2] * g->nb[1] + c[1]) * g->nb[0] + c[0]];
}

/* lab is [(8+2w)]^3 x nc, x fastest, origin shifted by w; only centre + faces filled */
static void load_lab(const orc_grid *g, const double *f, int nc, int is_vector, long b, int w, double *lab) {
  const int L = BS + 2 * w;
#define LAB(x, y, z, c) lab[((((long)(z) + w) * L + ((y) + w)) * L + ((x) + w)) * nc + (c)]
  const double *blk = f + (long)b * BS3 * nc;
  for (int z = 0; z < BS; z++)
    for (int y = 0; y < BS; y++)
      for (int x = 0; x < BS; x++)
        for (int c = 0; c < nc; c++) LAB(x, y, z, c) = blk[((z * BS + y) * BS + x) * nc + c];
  for (int d = 0; d < 3; d++)
    for (int side = 0; side < 2; side++) {
      const long n = nbr_slot(g, b, d, side);
      const double *nb = n >= 0 ? f + n * BS3 * nc : 0;
      for (int a2 = 0; a2 < BS; a2++)
        for (int a1 = 0; a1 < BS; a1++)
          for (int gl = 0; gl < w; gl++) {
            int p[3], q[3]; /* p: ghost coordinate in the lab; q: source cell */
            const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
            p[d1] = a1; p[d2] = a2; q[d1] = a1; q[d2] = a2;
            p[d] = side ? BS + gl : -1 - gl;
            if (nb) q[d] = side ? gl : BS - 1 - gl;          /* neighbour's cells */
            else q[d] = side ? BS - 1 : 0;                   /* own face cell */
            const double *src = (nb ? nb : blk) + ((q[2] * BS + q[1]) * BS + q[0]) * nc;
            for (int c = 0; c < nc; c++) {
              double v = src[c];
              if (!nb && is_vector) {
                if (g->bc[d] == ORC_BC_WALL) v = (-1.0) * v;
                else if (c == d) v = (-1.) * v;
              }
              LAB(p[0], p[1], p[2], c) = v;
            }
          }
    }
#undef LAB
}

/* =============================== operators =============================== */
void orc_ic_taylor_green(const orc_grid *g, double *vel, const double ext[3], double umax) { /* main.cpp:12516-12539 */
  const double a = 2 * M_PI / ext[0], b = 2 * M_PI / ext[1], c = 2 * M_PI / ext[2];
  const double A = umax, B = -umax * ext[1] / ext[0];
  for (long s = 0; s < g->nblocks; s++) {
    const double o[3] = {g->index[3 * s] * BS * g->h, g->index[3 * s + 1] * BS * g->h, g->index[3 * s + 2] * BS * g->h};
    for (int iz = 0; iz < BS; iz++)
      for (int iy = 0; iy < BS; iy++)
        for (int ix = 0; ix < BS; ix++) {
          const double p[3] = {o[0] + g->h * (ix + 0.5), o[1] + g->h * (iy + 0.5), o[2] + g->h * (iz + 0.5)}; /* Info::pos 369-373 */
          double *u = vel + (s * BS3 + (iz * BS + iy) * BS + ix) * 3;
          u[0] = A * cos(a * p[0]) * sin(b * p[1]) * sin(c * p[2]);
          u[1] = B * sin(a * p[0]) * cos(b * p[1]) * sin(c * p[2]);
          u[2] = 0;
        }
  }
}

double orc_max_u(const orc_grid *g, const double *vel, const double uinf[3]) { /* main.cpp:8603-8623 */
  double m = 0;
  for (long i = 0; i < g->nblocks * BS3; i++)
    for (int c = 0; c < 3; c++) {
      const double a = fabs(vel[3 * i + c] + uinf[c]);
      if (a > m) m = a;
    }
  return m;
}

double orc_calc_dt(double hmin, double umax, double nu, double cfl, int step, int rampup, double dt_old, double coefU[3]) {
  return orc_calc_dt2(hmin, umax, nu, cfl, step, rampup, dt_old, coefU, 0);
}
double orc_calc_dt2(double hmin, double umax, double nu, double cfl, int step, int rampup, double dt_old, double coefU[3], int implicitDiffusion) {
  /* main.cpp:15254-15305, CFL > 0 */
  double dt;
  const double dtDiffusion = (implicitDiffusion && step > 10) ? 0.1 : (1.0 / 6.0) * hmin * hmin / (nu + (1.0 / 6.0) * hmin * umax);
  const double dtAdvection = hmin / (umax + 1e-8);
  if (step < rampup) {
    const double x = step / (double)rampup;
    const double rampCFL = exp(log(1e-3) * (1 - x) + log(cfl) * x);
    dt = fmin(dtDiffusion, rampCFL * dtAdvection);
  } else
    dt = fmin(dtDiffusion, cfl * dtAdvection);
  if (step > 2) { /* step_2nd_start = 2, main.cpp:15355 */
    const double a = dt_old, b = dt;
    const double c1 = -(a + b) / (a * b);
    const double c2 = b / (a + b) / a;
    coefU[0] = -b * (c1 + c2);
    coefU[1] = b * c1;
    coefU[2] = b * c2;
  }
  return dt;
}

void orc_external_forcing(const orc_grid *g, double *vel, double umax_forced, double nu, double H, double dt) { /* 10581-10596 */
  const double gradPdt = 8 * umax_forced * nu / H / H * dt;
  for (long i = 0; i < g->nblocks * BS3; i++) vel[3 * i] += gradPdt;
}

static inline double upwind5(double U, double um3, double um2, double um1, double u, double up1, double up2, double up3) {
  /* KernelAdvectDiffuse::derivative, main.cpp:9474-9483 */
  if (U > 0) return (-2 * um3 + 15 * um2 - 60 * um1 + 20 * u + 30 * up1 - 3 * up2) / 60.;
  else return (2 * up3 - 15 * up2 + 60 * up1 - 20 * u - 30 * um1 + 3 * um2) / 60.;
}

void orc_advdiff_stage_rhs(const orc_grid *g, const double *vel, double *tmpV, double dt, double nu, const double uinf[3]) {
  /* KernelAdvectDiffuse::operator(), main.cpp:9484-9549, coef = 1 */
  const int w = 3, L = BS + 2 * w;
#pragma omp parallel
  {
    double *lab = (double *)calloc((size_t)L * L * L * 3, sizeof(double));
#define V(x, y, z, c) lab[((((long)(z) + w) * L + ((y) + w)) * L + ((x) + w)) * 3 + (c)]
#pragma omp for
    for (long b = 0; b < g->nblocks; b++) {
      load_lab(g, vel, 3, 1, b, w, lab);
      const double h = g->h, h3 = h * h * h;
      const double facA = -dt / h * h3 * 1.0;
      const double facD = (nu / h) * (dt / h) * h3 * 1.0;
      double *o = tmpV + b * BS3 * 3;
      for (int z = 0; z < BS; z++)
        for (int y = 0; y < BS; y++)
          for (int x = 0; x < BS; x++) {
            const double uAbs[3] = {V(x, y, z, 0) + uinf[0], V(x, y, z, 1) + uinf[1], V(x, y, z, 2) + uinf[2]};
            double dx[3], dy[3], dz[3], lap[3];
            for (int c = 0; c < 3; c++) {
              dx[c] = upwind5(uAbs[0], V(x - 3, y, z, c), V(x - 2, y, z, c), V(x - 1, y, z, c), V(x, y, z, c), V(x + 1, y, z, c), V(x + 2, y, z, c), V(x + 3, y, z, c));
              dy[c] = upwind5(uAbs[1], V(x, y - 3, z, c), V(x, y - 2, z, c), V(x, y - 1, z, c), V(x, y, z, c), V(x, y + 1, z, c), V(x, y + 2, z, c), V(x, y + 3, z, c));
              dz[c] = upwind5(uAbs[2], V(x, y, z - 3, c), V(x, y, z - 2, c), V(x, y, z - 1, c), V(x, y, z, c), V(x, y, z + 1, c), V(x, y, z + 2, c), V(x, y, z + 3, c));
            }
            /* the three Laplacians use three different association orders, main.cpp:9531-9542 */
            lap[0] = ((V(x + 1, y, z, 0) + V(x - 1, y, z, 0)) + ((V(x, y + 1, z, 0) + V(x, y - 1, z, 0)) + (V(x, y, z + 1, 0) + V(x, y, z - 1, 0)))) - 6 * V(x, y, z, 0);
            lap[1] = ((V(x, y + 1, z, 1) + V(x, y - 1, z, 1)) + ((V(x, y, z + 1, 1) + V(x, y, z - 1, 1)) + (V(x + 1, y, z, 1) + V(x - 1, y, z, 1)))) - 6 * V(x, y, z, 1);
            lap[2] = ((V(x, y, z + 1, 2) + V(x, y, z - 1, 2)) + ((V(x + 1, y, z, 2) + V(x - 1, y, z, 2)) + (V(x, y + 1, z, 2) + V(x, y - 1, z, 2)))) - 6 * V(x, y, z, 2);
            const double duA = uAbs[0] * dx[0] + (uAbs[1] * dy[0] + uAbs[2] * dz[0]); /* 9543 */
            const double dvA = uAbs[1] * dy[1] + (uAbs[2] * dz[1] + uAbs[0] * dx[1]); /* 9544 */
            const double dwA = uAbs[2] * dz[2] + (uAbs[0] * dx[2] + uAbs[1] * dy[2]); /* 9545 */
            double *oc = o + ((z * BS + y) * BS + x) * 3;
            oc[0] += facA * duA + facD * lap[0];
            oc[1] += facA * dvA + facD * lap[1];
            oc[2] += facA * dwA + facD * lap[2];
          }
    }
#undef V
    free(lab);
  }
}

void orc_advect_diffuse(const orc_grid *g, double *vel, double *tmpV, double dt, double nu, const double uinf[3]) {
  /* AdvectionDiffusion::operator(), main.cpp:9699-9726 */
  const double alpha[3] = {1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0};
  const double beta[3] = {-5.0 / 9.0, -153.0 / 128.0, 0.0};
  const long n = g->nblocks * BS3 * 3;
  memset(tmpV, 0, n * sizeof(double));
  for (int rk = 0; rk < 3; rk++) {
    orc_advdiff_stage_rhs(g, vel, tmpV, dt, nu, uinf);
    const double ih3 = alpha[rk] / (g->h * g->h * g->h);
#pragma omp parallel for
    for (long i = 0; i < n; i++) {
      vel[i] += tmpV[i] * ih3;
      tmpV[i] *= beta[rk];
    }
  }
}

static long corner_block(const orc_grid *g) { /* block with index (0,0,0), main.cpp:9287-9289 */
  return g->slot_of[0];
}

static void lhs_kernel(const orc_grid *g, const double *pres, double *lhs) { /* KernelLHSPoisson, main.cpp:9205-9215 */
  const int w = 1, L = BS + 2;
#pragma omp parallel
  {
    double *lab = (double *)calloc((size_t)L * L * L, sizeof(double));
#define P(x, y, z) lab[(((long)(z) + w) * L + ((y) + w)) * L + ((x) + w)]
#pragma omp for
    for (long b = 0; b < g->nblocks; b++) {
      load_lab(g, pres, 1, 0, b, w, lab);
      double *o = lhs + b * BS3;
      for (int z = 0; z < BS; z++)
        for (int y = 0; y < BS; y++)
          for (int x = 0; x < BS; x++)
            o[(z * BS + y) * BS + x] =
                g->h * (P(x - 1, y, z) + P(x + 1, y, z) + P(x, y - 1, z) + P(x, y + 1, z) + P(x, y, z - 1) + P(x, y, z + 1) - 6.0 * P(x, y, z));
    }
#undef P
    free(lab);
  }
}

void orc_lhs(const orc_grid *g, const double *pres, double *lhs, int mc) { /* ComputeLHS::operator(), main.cpp:9273-9327 */
  double avgP = 0;
  if (mc <= 2 && mc > 0) {
    const double h3 = g->h * g->h * g->h;
    for (long i = 0; i < g->nblocks * BS3; i++) avgP += pres[i] * h3; /* sequential = 1 OpenMP thread */
  }
  lhs_kernel(g, pres, lhs);
  if (mc == 0) return;
  if (mc <= 2 && mc > 0) {
    if (mc == 1) lhs[corner_block(g) * BS3] = avgP;
    else {
      const double h3 = g->h * g->h * g->h;
      for (long i = 0; i < g->nblocks * BS3; i++) lhs[i] += avgP * h3;
    }
  } else {
    const long c = corner_block(g) * BS3;
    lhs[c] = pres[c];
  }
}

/* poisson_kernels, main.cpp:14617-14745 */
static double precond_inner(double p[BS + 2][BS + 2][BS + 2], double Ax[BS][BS][BS], double r[BS][BS][BS],
                            double blk[BS][BS][BS], double sqrNorm0, double rr, double coefficient) {
  /* kernelPoissonGetZInner 14651-14703 (coefficient = -6: `- 6 * p` and `+ (-6) * p` round identically) and
   * kernelDiffusionGetZInner 10482-10533 (coefficient = -6 - h^2/(nu dt)): the same routine otherwise */
  double a2Partial[BS] = {0};
  for (int iz = 0; iz < BS; iz++)
    for (int iy = 0; iy < BS; iy++) {
      double t[BS];
      for (int ix = 0; ix < BS; ix++) t[ix] = p[iz + 1][iy + 1][ix] + p[iz + 1][iy + 1][ix + 2] + coefficient * p[iz + 1][iy + 1][ix + 1];
      for (int ix = 0; ix < BS; ix++) t[ix] += p[iz + 1][iy][ix + 1];
      for (int ix = 0; ix < BS; ix++) t[ix] += p[iz + 1][iy + 2][ix + 1];
      for (int ix = 0; ix < BS; ix++) t[ix] += p[iz][iy + 1][ix + 1];
      for (int ix = 0; ix < BS; ix++) t[ix] += p[iz + 2][iy + 1][ix + 1];
      for (int ix = 0; ix < BS; ix++) Ax[iz][iy][ix] = t[ix];
      for (int ix = 0; ix < BS; ix++) a2Partial[ix] += p[iz + 1][iy + 1][ix + 1] * t[ix];
    }
  double a2 = 0;
  for (int ix = 0; ix < BS; ix++) a2 += a2Partial[ix];
  const double a = rr / (a2 + 1e-55);
  for (int iz = 0; iz < BS; iz++)
    for (int iy = 0; iy < BS; iy++)
      for (int ix = 0; ix < BS; ix++) blk[iz][iy][ix] += a * p[iz + 1][iy + 1][ix + 1];
  /* subAndSumSqr, 14625-14641: 16 interleaved partial sums over the flattened block */
  double s16[16] = {0};
  double *rf = &r[0][0][0], *af = &Ax[0][0][0];
  for (int jy = 0; jy < BS3 / 16; jy++) {
    for (int jx = 0; jx < 16; jx++) rf[jy * 16 + jx] -= a * af[jy * 16 + jx];
    for (int jx = 0; jx < 16; jx++) s16[jx] += rf[jy * 16 + jx] * rf[jy * 16 + jx];
  }
  double sqrSum = 0;
  for (int jx = 0; jx < 16; jx++) sqrSum += s16[jx];
  const double beta = sqrSum / (rr + 1e-55);
  const double sqrNorm = (double)1 / (BS3 * BS3) * sqrSum;
  if (sqrNorm < 1e-7 * 1e-7 * sqrNorm0 || sqrNorm < 1e-16 * 1e-16) return -1.0;
  for (int iz = 0; iz < BS; iz++)
    for (int iy = 0; iy < BS; iy++)
      for (int ix = 0; ix < BS; ix++) p[iz + 1][iy + 1][ix + 1] = r[iz][iy][ix] + beta * p[iz + 1][iy + 1][ix + 1];
  return sqrSum;
}

long orc_precond_total_iters = 0; /* diagnostic: block-CG iterations summed over blocks by the last orc_precond call */
/* one block of getZImplParallel (main.cpp:14704-14745): blk <- approximate solve of the 8^3 Dirichlet problem; returns CG iterations */
long orc_precond_block_coef(double *pblk, double h, double coefficient);
long orc_precond_block(double *pblk, double h) { return orc_precond_block_coef(pblk, h, -6.0); }
/* the same with the centre coefficient of the block operator as a parameter: diffusion_kernels::getZImplParallel,
 * main.cpp:10534-10579, passes -6 - h^2/nu/dt (10570) */
long orc_precond_block_coef(double *pblk, double h, double coefficient) {
  long total = 0;
  double p[BS + 2][BS + 2][BS + 2], Ax[BS][BS][BS], r[BS][BS][BS];
  memset(p, 0, sizeof p);
  double(*blk)[BS][BS] = (double(*)[BS][BS])pblk;
  const double invh = 1 / h;
  double rrPartial[BS] = {0};
  for (int iz = 0; iz < BS; iz++)
    for (int iy = 0; iy < BS; iy++)
      for (int ix = 0; ix < BS; ix++) {
        r[iz][iy][ix] = invh * blk[iz][iy][ix];
        rrPartial[ix] += r[iz][iy][ix] * r[iz][iy][ix];
        p[iz + 1][iy + 1][ix + 1] = r[iz][iy][ix];
        blk[iz][iy][ix] = 0;
      }
  double rr = 0;
  for (int ix = 0; ix < BS; ix++) rr += rrPartial[ix];
  const double sqrNorm0 = (double)1 / (BS3 * BS3) * rr;
  if (sqrNorm0 < 1e-32) return 0;
  for (int k = 0; k < 100; k++) {
    rr = precond_inner(p, Ax, r, blk, sqrNorm0, rr, coefficient);
    total++;
    if (rr <= 0) break;
  }
  return total;
}
void orc_precond(const orc_grid *g, double *pres) { /* getZImplParallel, main.cpp:14704-14745 */
  long total = 0;
#pragma omp parallel for reduction(+ : total)
  for (long b = 0; b < g->nblocks; b++) total += orc_precond_block(pres + b * BS3, g->h);
  orc_precond_total_iters = total;
}

/* PoissonSolverAMR::solve, main.cpp:14363-14616, over any mesh: `op_lhs` is _lhs (9365-9393), `op_precond` is
 * _preconditioner (9334-9364, in place), `corner` the first cell of the block whose index is (0,0,0) */
void orc_solve_generic(void *g, long N, long corner, void (*op_lhs)(void *, const double *, double *, int),
                       void (*op_precond)(void *, double *), double *lhs, double *pres, orc_solve_info *info) {
  orc_solve_generic2(g, N, corner, op_lhs, op_precond, lhs, pres, info, 100);
}
/* max_restarts: 100 in PoissonSolverAMR::solve (14567); DiffusionSolver::solve (6896-7146) is the same routine without a
 * restart cap and without mean constraint (pass mean_constraint 0 and INT_MAX) */
void orc_solve_generic2(void *g, long N, long corner, void (*op_lhs)(void *, const double *, double *, int),
                        void (*op_precond)(void *, double *), double *lhs, double *pres, orc_solve_info *info, int max_restarts) {
#define solver_lhs(g, in, out, mc) op_lhs(g, in, out, mc)
#define solver_precond(g, in, out, N) (memcpy(out, in, (N) * sizeof(double)), op_precond(g, out))
  const int mc = info->mean_constraint;
  const double eps = 1e-100, max_error = info->tol, max_rel_error = info->tol_rel;
  int serious_breakdown = 0, useXopt = 0, restarts = 0;
  double min_norm = 1e50, norm_1 = 0.0, norm_2 = 0.0;
  double *buf = (double *)calloc((size_t)18 * N, sizeof(double));
  double *phat = buf, *rhat = buf + N, *shat = buf + 2 * N, *what = buf + 3 * N, *zhat = buf + 4 * N, *qhat = buf + 5 * N,
         *s = buf + 6 * N, *w = buf + 7 * N, *z = buf + 8 * N, *t = buf + 9 * N, *v = buf + 10 * N, *q = buf + 11 * N,
         *r = buf + 12 * N, *y = buf + 13 * N, *x = buf + 14 * N, *r0 = buf + 15 * N, *b = buf + 16 * N, *x_opt = buf + 17 * N;
  if (mc == 1 || mc > 2) lhs[corner] = 0.0; /* 14404-14407 */
  for (long j = 0; j < N; j++) { b[j] = lhs[j]; r[j] = lhs[j]; x[j] = pres[j]; }
  solver_lhs(g, x, r0, mc);
  for (long i = 0; i < N; i++) { r0[i] = r[i] - r0[i]; r[i] = r0[i]; }
  solver_precond(g, r0, rhat, N);
  solver_lhs(g, rhat, w, mc);
  solver_precond(g, w, what, N);
  solver_lhs(g, what, t, mc);
  double alpha = 0.0, norm = 0.0, beta = 0.0, omega = 0.0, r0r_prev;
  {
    double temp0 = 0.0, temp1 = 0.0;
    for (long j = 0; j < N; j++) { temp0 += r0[j] * r0[j]; temp1 += r0[j] * w[j]; norm += r0[j] * r0[j]; }
    alpha = temp0 / (temp1 + eps);
    r0r_prev = temp0;
    norm = sqrt(norm);
  }
  const double init_norm = norm;
  int k;
  for (k = 0; k < 1000; k++) {
    double qy = 0.0, yy = 0.0;
    if (k % 50 != 0) {
      for (long j = 0; j < N; j++) {
        phat[j] = rhat[j] + beta * (phat[j] - omega * shat[j]);
        s[j] = w[j] + beta * (s[j] - omega * z[j]);
        shat[j] = what[j] + beta * (shat[j] - omega * zhat[j]);
        z[j] = t[j] + beta * (z[j] - omega * v[j]);
        q[j] = r[j] - alpha * s[j];
        qhat[j] = rhat[j] - alpha * shat[j];
        y[j] = w[j] - alpha * z[j];
        qy += q[j] * y[j];
        yy += y[j] * y[j];
      }
    } else {
      for (long j = 0; j < N; j++) phat[j] = rhat[j] + beta * (phat[j] - omega * shat[j]);
      solver_lhs(g, phat, s, mc);
      solver_precond(g, s, shat, N);
      solver_lhs(g, shat, z, mc);
      for (long j = 0; j < N; j++) {
        q[j] = r[j] - alpha * s[j];
        qhat[j] = rhat[j] - alpha * shat[j];
        y[j] = w[j] - alpha * z[j];
        qy += q[j] * y[j];
        yy += y[j] * y[j];
      }
    }
    solver_precond(g, z, zhat, N);
    solver_lhs(g, zhat, v, mc);
    omega = qy / (yy + eps);
    double r0r = 0.0, r0w = 0.0, r0s = 0.0, r0z = 0.0;
    norm = 0.0; norm_1 = 0.0; norm_2 = 0.0;
    if (k % 50 != 0) {
      for (long j = 0; j < N; j++) {
        x[j] = x[j] + alpha * phat[j] + omega * qhat[j];
        r[j] = q[j] - omega * y[j];
        rhat[j] = qhat[j] - omega * (what[j] - alpha * zhat[j]);
        w[j] = y[j] - omega * (t[j] - alpha * v[j]);
        r0r += r0[j] * r[j];
        r0w += r0[j] * w[j];
        r0s += r0[j] * s[j];
        r0z += r0[j] * z[j];
        norm += r[j] * r[j];
        norm_1 += r[j] * r[j];
        norm_2 += r0[j] * r0[j];
      }
    } else {
      for (long j = 0; j < N; j++) x[j] = x[j] + alpha * phat[j] + omega * qhat[j];
      solver_lhs(g, x, r, mc);
      for (long j = 0; j < N; j++) r[j] = b[j] - r[j];
      solver_precond(g, r, rhat, N);
      solver_lhs(g, rhat, w, mc);
      for (long j = 0; j < N; j++) {
        r0r += r0[j] * r[j];
        r0w += r0[j] * w[j];
        r0s += r0[j] * s[j];
        r0z += r0[j] * z[j];
        norm += r[j] * r[j];
        norm_1 += r[j] * r[j];
        norm_2 += r0[j] * r0[j];
      }
    }
    solver_precond(g, w, what, N);
    solver_lhs(g, what, t, mc);
    norm = sqrt(norm);
    beta = alpha / (omega + eps) * r0r / (r0r_prev + eps);
    alpha = r0r / (r0w + beta * r0s - beta * omega * r0z);
    double alphat = 1.0 / (omega + eps) + r0w / (r0r + eps) - beta * omega * r0z / (r0r + eps);
    alphat = 1.0 / (alphat + eps);
    if (fabs(alphat) < 10 * fabs(alpha)) alpha = alphat;
    r0r_prev = r0r;
    serious_breakdown = r0r * r0r < 1e-16 * norm_1 * norm_2;
    if (serious_breakdown && restarts < max_restarts) {
      restarts++;
      for (long i = 0; i < N; i++) r0[i] = r[i];
      solver_precond(g, r0, rhat, N);
      solver_lhs(g, rhat, w, mc);
      alpha = 0.0;
      double temp0 = 0.0, temp1 = 0.0;
      for (long j = 0; j < N; j++) { temp0 += r0[j] * r0[j]; temp1 += r0[j] * w[j]; }
      solver_precond(g, w, what, N);
      solver_lhs(g, what, t, mc);
      alpha = temp0 / (temp1 + eps);
      r0r_prev = temp0;
      beta = 0.0;
      omega = 0.0;
    }
    if (norm < min_norm) {
      useXopt = 1;
      min_norm = norm;
      memcpy(x_opt, x, N * sizeof(double));
    }
    if (norm < max_error || norm / (init_norm + eps) < max_rel_error) break;
  }
  memcpy(pres, useXopt ? x_opt : x, N * sizeof(double));
  /* the reference leaves scratch in sim.lhs (last _lhs output): mirror it, 9383-9392 */
  memcpy(lhs, t, N * sizeof(double));
  info->iters = k < 1000 ? k + 1 : 1000; /* number of 7-double reductions issued (14546) */
  info->restarts = restarts;
  info->norm0 = init_norm;
  info->norm = norm;
  free(buf);
#undef solver_lhs
#undef solver_precond
}
static void uni_lhs(void *g, const double *in, double *out, int mc) { orc_lhs((const orc_grid *)g, in, out, mc); }
static void uni_precond(void *g, double *io) { orc_precond((const orc_grid *)g, io); }
void orc_solve(const orc_grid *g, double *lhs, double *pres, orc_solve_info *info) {
  orc_solve_generic((void *)g, g->nblocks * BS3, corner_block(g) * BS3, uni_lhs, uni_precond, lhs, pres, info);
}

void orc_pressure_rhs(const orc_grid *g, const double *vel, const double *udef, const double *chi, double *lhs, double dt) {
  /* KernelPressureRHS::operator(), main.cpp:14849-14875 */
  const int w = 1, L = BS + 2;
#pragma omp parallel
  {
    double *lab = (double *)calloc((size_t)L * L * L * 3, sizeof(double));
    double *lab2 = (double *)calloc((size_t)L * L * L * 3, sizeof(double));
#define U(x, y, z, c) lab[((((long)(z) + w) * L + ((y) + w)) * L + ((x) + w)) * 3 + (c)]
#define D(x, y, z, c) lab2[((((long)(z) + w) * L + ((y) + w)) * L + ((x) + w)) * 3 + (c)]
#pragma omp for
    for (long b = 0; b < g->nblocks; b++) {
      load_lab(g, vel, 3, 1, b, w, lab);
      load_lab(g, udef, 3, 1, b, w, lab2);
      const double h = g->h, fac = 0.5 * h * h / dt;
      for (int z = 0; z < BS; z++)
        for (int y = 0; y < BS; y++)
          for (int x = 0; x < BS; x++) {
            const long i = b * BS3 + (z * BS + y) * BS + x;
            double p = fac * (U(x + 1, y, z, 0) - U(x - 1, y, z, 0) + U(x, y + 1, z, 1) - U(x, y - 1, z, 1) + U(x, y, z + 1, 2) - U(x, y, z - 1, 2));
            const double divUs = D(x + 1, y, z, 0) - D(x - 1, y, z, 0) + D(x, y + 1, z, 1) - D(x, y - 1, z, 1) + D(x, y, z + 1, 2) - D(x, y, z - 1, 2);
            p += -chi[i] * fac * divUs;
            lhs[i] = p;
          }
    }
#undef U
#undef D
    free(lab); free(lab2);
  }
}

void orc_div_pressure(const orc_grid *g, const double *pres, double *tmpV) { /* KernelDivPressure, main.cpp:14769-14778 */
  const int w = 1, L = BS + 2;
#pragma omp parallel
  {
    double *lab = (double *)calloc((size_t)L * L * L, sizeof(double));
#define P(x, y, z) lab[(((long)(z) + w) * L + ((y) + w)) * L + ((x) + w)]
#pragma omp for
    for (long b = 0; b < g->nblocks; b++) {
      load_lab(g, pres, 1, 0, b, w, lab);
      const double fac = g->h;
      for (int z = 0; z < BS; z++)
        for (int y = 0; y < BS; y++)
          for (int x = 0; x < BS; x++)
            tmpV[(b * BS3 + (z * BS + y) * BS + x) * 3] =
                fac * (P(x + 1, y, z) + P(x - 1, y, z) + P(x, y + 1, z) + P(x, y - 1, z) + P(x, y, z + 1) + P(x, y, z - 1) - 6.0 * P(x, y, z));
    }
#undef P
    free(lab);
  }
}

void orc_grad_p(const orc_grid *g, const double *pres, double *tmpV, double dt) { /* KernelGradP, main.cpp:14990-14999 */
  const int w = 1, L = BS + 2;
#pragma omp parallel
  {
    double *lab = (double *)calloc((size_t)L * L * L, sizeof(double));
#define P(x, y, z) lab[(((long)(z) + w) * L + ((y) + w)) * L + ((x) + w)]
#pragma omp for
    for (long b = 0; b < g->nblocks; b++) {
      load_lab(g, pres, 1, 0, b, w, lab);
      const double fac = -0.5 * dt * g->h * g->h;
      for (int z = 0; z < BS; z++)
        for (int y = 0; y < BS; y++)
          for (int x = 0; x < BS; x++) {
            double *o = tmpV + (b * BS3 + (z * BS + y) * BS + x) * 3;
            o[0] = fac * (P(x + 1, y, z) - P(x - 1, y, z));
            o[1] = fac * (P(x, y + 1, z) - P(x, y - 1, z));
            o[2] = fac * (P(x, y, z + 1) - P(x, y, z - 1));
          }
    }
#undef P
    free(lab);
  }
}

void orc_project(const orc_grid *g, double *vel, double *pres, double *tmpV, double *lhs, const double *chi,
                 double dt, int step, orc_solve_info *info) { /* PressureProjection::operator(), main.cpp:15061-15160 */
  const long N = g->nblocks * BS3;
  double *pOld = (double *)malloc(N * sizeof(double));
  memcpy(pOld, pres, N * sizeof(double));
  memset(tmpV, 0, 3 * N * sizeof(double));
  orc_pressure_rhs(g, vel, tmpV, chi, lhs, dt);
  if (step > 2) {
    orc_div_pressure(g, pres, tmpV);
    for (long i = 0; i < N; i++) { lhs[i] -= tmpV[3 * i]; pres[i] = 0; }
  } else
    memset(pres, 0, N * sizeof(double));
  orc_solve(g, lhs, pres, info);
  double avg = 0, avg1 = 0;
  const double vv = g->h * g->h * g->h;
  for (long i = 0; i < N; i++) { avg += pres[i] * vv; avg1 += vv; }
  avg = avg / avg1;
  for (long i = 0; i < N; i++) pres[i] -= avg;
  if (step > 2)
    for (long i = 0; i < N; i++) pres[i] += pOld[i];
  orc_grad_p(g, pres, tmpV, dt);
  const double fac = 1.0 / (g->h * g->h * g->h);
  for (long i = 0; i < 3 * N; i++) vel[i] += fac * tmpV[i];
  free(pOld);
}

/* ===================== mesh-adaptation block operators ===================== */
/* Full tensorial [-1,2) tile of block b: centre, all 26 same-level neighbours, then the
 * domain-face passes in the reference's order x-,x+,y-,y+,z-,z+ (BlockLab::load 3623-3743;
 * BlockLabBC::_apply_bc 6513-6551 with applyBCfaceWall 6369-6429 / applyBCfaceOpen 6107-6231;
 * BlockLabNeumann3D::_apply_bc 6561-6581 with Neumann3D 5929-6004).  Each pass fills the whole
 * ghost slab behind the face -- interior part first, then the edge/corner strips -- from the
 * face cell with the same transverse coordinates, which may themselves be ghosts filled earlier. */
static void load_lab_full(const orc_grid *g, const double *f, int nc, int is_vector, long b, double *lab) {
  const int w = 1, L = BS + 2;
#define LAB(x, y, z, c) lab[((((long)(z) + w) * L + ((y) + w)) * L + ((x) + w)) * nc + (c)]
  const int *idx = &g->index[3 * b];
  for (long i = 0; i < (long)L * L * L * nc; i++) lab[i] = 0.0;
  int skip[3], skin[3];
  for (int d = 0; d < 3; d++) {
    skin[d] = idx[d] == 0 || idx[d] == g->nb[d] - 1;
    skip[d] = idx[d] == 0 ? -1 : 1; /* main.cpp:3681-3686 */
  }
  for (int cz = -1; cz <= 1; cz++)
    for (int cy = -1; cy <= 1; cy++)
      for (int cx = -1; cx <= 1; cx++) {
        const int code[3] = {cx, cy, cz};
        int skipped = 0;
        for (int d = 0; d < 3; d++)
          if (g->bc[d] != ORC_BC_PERIODIC && code[d] == skip[d] && skin[d]) skipped = 1; /* 3696-3701 */
        if (skipped) continue;
        int c[3];
        for (int d = 0; d < 3; d++) c[d] = (idx[d] + code[d] + g->nb[d]) % g->nb[d];
        const double *nb = f + g->slot_of[((long)c[2] * g->nb[1] + c[1]) * g->nb[0] + c[0]] * BS3 * nc;
        const int s[3] = {cx < 0 ? -1 : (cx == 0 ? 0 : BS), cy < 0 ? -1 : (cy == 0 ? 0 : BS), cz < 0 ? -1 : (cz == 0 ? 0 : BS)};
        const int e[3] = {cx < 0 ? 0 : (cx == 0 ? BS : BS + 1), cy < 0 ? 0 : (cy == 0 ? BS : BS + 1), cz < 0 ? 0 : (cz == 0 ? BS : BS + 1)};
        for (int z = s[2]; z < e[2]; z++)
          for (int y = s[1]; y < e[1]; y++)
            for (int x = s[0]; x < e[0]; x++)
              for (int k = 0; k < nc; k++)
                LAB(x, y, z, k) = nb[(((z - cz * BS) * BS + (y - cy * BS)) * BS + (x - cx * BS)) * nc + k];
      }
  for (int d = 0; d < 3; d++) {
    if (g->bc[d] == ORC_BC_PERIODIC) continue;
    for (int side = 0; side < 2; side++) {
      if (side == 0 ? idx[d] != 0 : idx[d] != g->nb[d] - 1) continue;
      const int ghost = side ? BS : -1, face = side ? BS - 1 : 0, d1 = (d + 1) % 3, d2 = (d + 2) % 3;
      for (int a2 = -1; a2 <= BS; a2++)
        for (int a1 = -1; a1 <= BS; a1++) {
          int p[3], q[3];
          p[d] = ghost; q[d] = face; p[d1] = q[d1] = a1; p[d2] = q[d2] = a2;
          for (int k = 0; k < nc; k++) {
            double v = LAB(q[0], q[1], q[2], k);
            if (is_vector) {
              if (g->bc[d] == ORC_BC_WALL) v = (-1.0) * v;
              else if (k == d) v = (-1.) * v;
            }
            LAB(p[0], p[1], p[2], k) = v;
          }
        }
    }
  }
#undef LAB
}

void orc_restrict(const orc_grid *fine, const orc_grid *coarse, const double *ff, double *cf, int nc) {
  for (long pb = 0; pb < coarse->nblocks; pb++) {
    const int *pi = &coarse->index[3 * pb];
    double *dst = cf + pb * BS3 * nc;
    for (int K = 0; K < 2; K++)
      for (int J = 0; J < 2; J++)
        for (int I = 0; I < 2; I++) {
          const long cb = fine->slot_of[((long)(2 * pi[2] + K) * fine->nb[1] + (2 * pi[1] + J)) * fine->nb[0] + (2 * pi[0] + I)];
          const double *b = ff + cb * BS3 * nc;
#define B(i, j, k, c) b[(((k) * BS + (j)) * BS + (i)) * nc + (c)]
          for (int k = 0; k < BS; k += 2)
            for (int j = 0; j < BS; j += 2)
              for (int i = 0; i < BS; i += 2)
                for (int c = 0; c < nc; c++)
                  dst[(((k / 2 + 4 * K) * BS + (j / 2 + 4 * J)) * BS + (i / 2 + 4 * I)) * nc + c] =
                      0.125 * ((B(i, j, k, c) + B(i + 1, j + 1, k + 1, c)) + (B(i + 1, j, k, c) + B(i, j + 1, k + 1, c)) +
                               (B(i, j + 1, k, c) + B(i + 1, j, k + 1, c)) + (B(i + 1, j + 1, k, c) + B(i, j, k + 1, c)));
#undef B
        }
  }
}

void orc_prolong(const orc_grid *coarse, const orc_grid *fine, const double *cf, double *ff, int nc, int is_vector) {
  const int L = BS + 2;
  double *lab = (double *)malloc((size_t)L * L * L * nc * sizeof(double));
#define Lb(x, y, z) lab[((((long)(z) + 1) * L + ((y) + 1)) * L + ((x) + 1)) * nc + c]
  for (long pb = 0; pb < coarse->nblocks; pb++) {
    load_lab_full(coarse, cf, nc, is_vector, pb, lab);
    const int *pi = &coarse->index[3 * pb];
    for (int K = 0; K < 2; K++)
      for (int J = 0; J < 2; J++)
        for (int I = 0; I < 2; I++) {
          const long cb = fine->slot_of[((long)(2 * pi[2] + K) * fine->nb[1] + (2 * pi[1] + J)) * fine->nb[0] + (2 * pi[0] + I)];
          double *b = ff + cb * BS3 * nc;
#define B(i, j, k) b[(((k) * BS + (j)) * BS + (i)) * nc + c]
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
        }
  }
#undef Lb
  free(lab);
}

void orc_tag(const orc_grid *g, const double *f, int nc, double rtol, double ctol, signed char *states) {
  for (long b = 0; b < g->nblocks; b++) {
    double Linf = 0.0;
    for (int i = 0; i < BS3; i++) {
      double m;
      if (nc == 1) m = f[b * BS3 + i]; /* ScalarElement::magnitude = s, main.cpp:5783 */
      else {
        double s1 = 0.0;
        for (int c = 0; c < nc; c++) s1 += f[(b * BS3 + i) * nc + c] * f[(b * BS3 + i) * nc + c];
        m = sqrt(s1); /* VectorElement::magnitude, 5873-5879 */
      }
      Linf = fmax(Linf, fabs(m));
    }
    signed char st = Linf > rtol ? 1 : (Linf < ctol ? -1 : 0);
    if (st == 1 && g->level == g->level_max - 1) st = 0; /* 5207-5211 */
    if (st == -1 && g->level == 0) st = 0;
    states[b] = st;
  }
}
