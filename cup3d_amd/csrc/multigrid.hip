// Geometric multigrid as the preconditioner of the pressure solver -- an ALTERNATIVE to the reference's block-local CG
// (cup3d_poisson_params.block_solver = 5), NOT a restatement of anything in main.cpp: the reference has no multigrid,
// its preconditioner is getZImplParallel (14704-14745).  BASELINE.json's north_star words the path as "geometric-multigrid
// smoother / restrict / prolong ... red-black Gauss-Seidel smoother"; this file is that wording taken literally, behind the same
// BiCGSTAB driver, the same operator A = h (sum6 - 6 p) (KernelLHSPoisson, 9205-9215), the same mean constraint and the same
// stopping rule -- so the CONVERGED pressure is the reference's to solver tolerance (tests), while the iteration count drops
// from O(150) to O(10) at 512^3.  bench.py reports it under `alt_multigrid`, never as `value`.  Uniform grids; over several ranks each
// rank cycles on its own blocks (additive Schwarz, see mg_setup).
//
// One application M^-1 r = one V(2,2)-cycle from a zero guess on the hierarchy of uniform block grids, level L (the solver's grid)
// down to level 0 (the bpd[0] x bpd[1] x bpd[2] box of 8^3 blocks):
//   * operator on every level: rediscretisation, A_l = h_l (sum6 - 6 .), zero-gradient domain faces (the pressure tile, 6561-6581);
//   * smoother: one workgroup per block loads the ghosted 10^3 tile into LDS and runs `sweeps` red-black Gauss-Seidel sweeps on
//     the block with the ghosts frozen (block-Jacobi across blocks: the iterate is double-buffered, so the cycle is a FIXED
//     linear operator, as BiCGSTAB needs, and deterministic);
//   * restriction of the residual: A is in finite-volume form (h^3 x the Laplacian), so a coarse right-hand side is the SUM of
//     its eight children; computed in the kernel that forms the residual (the fine residual never goes to HBM);
//   * prolongation: piecewise constant (cell-centred, order 1 + order 2 of the restriction > 2, enough for a preconditioner);
//   * level 0: the right-hand side's mean is removed (the all-Neumann operator is singular), then 64 sweeps.
#include <memory>
#include <stdexcept>

#include "sim.hpp"
#include "tile.hpp"
#include "tile7.hpp"

namespace cup3d {

struct MGLevel {
  std::unique_ptr<Grid> grid;  // nullptr on the finest level (the solver's own grid)
  int64_t nb = 0;
  double h = 0;
  int32_t *d_nbr = nullptr;     // [nb][6]
  int32_t *d_parent = nullptr;  // [nb][2]: slot of the parent block on the next coarser level, octant (x + 2y + 4z)
  double *x = nullptr, *x2 = nullptr, *b = nullptr;  // coarse levels; the finest level uses the caller's vectors + x2
};
struct Multigrid {
  std::vector<MGLevel> lev;  // [0] coarsest ... [L] finest
  double *zeros = nullptr;   // ghost values behind faces owned by other ranks (see mg_setup); the x pointer itself on one rank
  bool local = false;        // a rank-local hierarchy (several ranks)
  ~Multigrid() {
    if (zeros) hipFree(zeros);
    for (size_t i = 0; i < lev.size(); ++i) {
      MGLevel &l = lev[i];
      if (l.grid && l.d_nbr) hipFree(l.d_nbr);
      if (l.d_parent) hipFree(l.d_parent);
      if (l.x) hipFree(l.x);
      if (l.x2) hipFree(l.x2);
      if (l.b) hipFree(l.b);
    }
  }
};

// `sweeps` red-black Gauss-Seidel sweeps of h (sum6 - 6 x) = b on one block, ghosts from the neighbours' xin (frozen);
// ZERO: the incoming iterate is zero everywhere (first smoothing of a cycle): nothing is loaded but b
template <bool ZERO>
__global__ void __launch_bounds__(256) k_mg_smooth(GridDev g, const double *__restrict__ xin, const double *__restrict__ halo, const double *__restrict__ b,
                                                   double *__restrict__ xout, int sweeps) {
  __shared__ double tile[kT];
  const int slot = block_slot(g);
  if (slot < 0) return;
  const int t = threadIdx.x;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  if (ZERO) {
    for (int i = t; i < kT; i += 256) tile[i] = 0.0;
  } else {
    double c[2];
    load_scalar_tile(g, slot, xin, halo, tile, c);  // halo: zeros behind the faces other ranks own (never a literal nullptr: clang 22's inliner crashes on it)
  }
  const double invh = 1.0 / g.h;
  const double r0 = invh * b[(size_t)slot * 512 + cell0], r1 = invh * b[(size_t)slot * 512 + 256 + cell0];
  const int i0 = tix(x, y, z0), i1 = tix(x, y, z0 + 4);
  const bool red = ((x + y + z0) & 1) == 0;  // both cells of a thread have the same colour (z0 and z0 + 4)
  __syncthreads();
  for (int s = 0; s < sweeps; ++s) {
#pragma unroll
    for (int colour = 0; colour < 2; ++colour) {
      if (red == (colour == 0)) {
        tile[i0] = (1.0 / 6.0) * ((tile[i0 - 1] + tile[i0 + 1]) + (tile[i0 - 10] + tile[i0 + 10]) + (tile[i0 - kTP] + tile[i0 + kTP]) - r0);
        tile[i1] = (1.0 / 6.0) * ((tile[i1 - 1] + tile[i1 + 1]) + (tile[i1 - 10] + tile[i1 + 10]) + (tile[i1 - kTP] + tile[i1 + kTP]) - r1);
      }
      __syncthreads();
    }
  }
  xout[(size_t)slot * 512 + cell0] = tile[i0];
  xout[(size_t)slot * 512 + 256 + cell0] = tile[i1];
}

// coarse b (parent block, this block's octant) <- sum over 2x2x2 of the fine residual b - A x
__global__ void __launch_bounds__(256) k_mg_residual_restrict(GridDev g, const double *__restrict__ x, const double *__restrict__ halo, const double *__restrict__ b,
                                                              const int32_t *__restrict__ parent, double *__restrict__ bc) {
  __shared__ double tile[kT];
  __shared__ double r[512];
  const int slot = block_slot(g);
  if (slot < 0) return;
  const int t = threadIdx.x;
  double c[2];
  load_scalar_tile(g, slot, x, halo, tile, c);
  __syncthreads();
  int x_, y, z0, cell0;
  thread_cells(t, x_, y, z0, cell0);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tix(x_, y, z0 + 4 * k);
    r[k * 256 + cell0] = b[(size_t)slot * 512 + k * 256 + cell0] -
                         g.h * ((tile[i - 1] + tile[i + 1]) + (tile[i - 10] + tile[i + 10]) + (tile[i - kTP] + tile[i + kTP]) - 6.0 * c[k]);
  }
  __syncthreads();
  if (t < 64) {
    const int X = t & 3, Y = (t >> 2) & 3, Z = t >> 4;
    const int o = (2 * Z) * 64 + (2 * Y) * 8 + 2 * X;
    const double s = ((r[o] + r[o + 1]) + (r[o + 8] + r[o + 9])) + ((r[o + 64] + r[o + 65]) + (r[o + 72] + r[o + 73]));
    const int ps = parent[2 * slot], oct = parent[2 * slot + 1];
    const int cx = 4 * (oct & 1) + X, cy = 4 * ((oct >> 1) & 1) + Y, cz = 4 * (oct >> 2) + Z;
    bc[(size_t)ps * 512 + cz * 64 + cy * 8 + cx] = s;
  }
}

// x (fine) += the value of the coarse cell that contains the fine cell
__global__ void __launch_bounds__(256) k_mg_prolong_add(int nb, double *__restrict__ x, const int32_t *__restrict__ parent, const double *__restrict__ xc) {
  const int slot = blockIdx.x, t = threadIdx.x;
  if (slot >= nb) return;
  const int ps = parent[2 * slot], oct = parent[2 * slot + 1];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int cell = k * 256 + t, cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
    const int X = 4 * (oct & 1) + (cx >> 1), Y = 4 * ((oct >> 1) & 1) + (cy >> 1), Z = 4 * (oct >> 2) + (cz >> 1);
    x[(size_t)slot * 512 + cell] += xc[(size_t)ps * 512 + Z * 64 + Y * 8 + X];
  }
}

// level 0: b -= mean(b) (one workgroup; a handful of blocks)
__global__ void __launch_bounds__(256) k_mg_remove_mean(double *__restrict__ b, long n) {
  __shared__ double red[4];
  double s = 0;
  for (long i = threadIdx.x; i < n; i += 256) s += b[i];
  s = group_sum<4>(s, red);
  const double m = s / (double)n;
  for (long i = threadIdx.x; i < n; i += 256) b[i] -= m;
}

static GridDev level_gdev(const MGLevel &L) {
  GridDev g;
  g.nbr = L.d_nbr;
  g.list = nullptr;
  g.nblocks = (int)L.nb;
  g.chunk = (g.nblocks + 7) / 8;
  g.h = L.h;
  g.hb = nullptr;
  g.flux = nullptr;
  g.raw = nullptr;
  return g;
}

static int mg_setup(Sim *s) {
  if (s->mg) return CUP3D_OK;
  const Grid *g = s->grid;
  if (g->multilevel) { set_error("the multigrid preconditioner (block_solver 5) runs on uniform grids"); return CUP3D_EINVAL; }
  std::unique_ptr<Multigrid> mg(new Multigrid());
  const int L = g->level, N = g->nranks;
  // Several ranks: every rank runs the V-cycle on ITS blocks only, with zero ghost values behind the faces other ranks own -- an
  // additive-Schwarz preconditioner whose subdomain solves are multigrid cycles, the reference's block-local preconditioner
  // (zero ghosts around every 8^3 block) scaled up from a block to a rank.  No message is exchanged inside M^-1; BiCGSTAB's own
  // LHS applications and dot products couple the ranks.  The hierarchy goes down as far as the Hilbert-range partition of the
  // coarser grid still nests in this one (every rank's share a whole number of parent blocks): level 1 for 2, 4 or 8 ranks of a
  // cubic base grid of one block.
  mg->local = N > 1;
  auto nblocks_at = [&](int l) { return (int64_t)g->bpd[0] * g->bpd[1] * g->bpd[2] << (3 * l); };
  int lmin = L;
  while (lmin > 0 && (N == 1 || (nblocks_at(lmin - 1) % N == 0 && nblocks_at(lmin) % N == 0))) --lmin;
  const int nlev = L - lmin + 1;
  mg->lev.resize(nlev);
  auto up = [&](int32_t **d, const std::vector<int32_t> &v) -> int {
    CUP3D_HIP(hipMalloc((void **)d, std::max<size_t>(v.size(), 1) * sizeof(int32_t)));
    CUP3D_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return CUP3D_OK;
  };
  int rc;
  try {
    int64_t max_halo_faces = 0;
    for (int i = nlev - 1; i >= 0; --i) {
      MGLevel &M = mg->lev[i];
      const Grid *gl = g;
      if (i < nlev - 1) {
        M.grid.reset(new Grid(g->bpd, g->level_max, lmin + i, g->maxextent, g->bc, g->rank, N));
        gl = M.grid.get();
        if ((rc = up(&M.d_nbr, gl->nbr))) return rc;
        const size_t bytes = (size_t)gl->nblocks() * 512 * sizeof(double);
        CUP3D_HIP(hipMalloc((void **)&M.x, bytes));
        CUP3D_HIP(hipMalloc((void **)&M.x2, bytes));
        CUP3D_HIP(hipMalloc((void **)&M.b, bytes));
        s->bytes += 3 * bytes;
      } else {
        M.d_nbr = s->d_nbr;
        CUP3D_HIP(hipMalloc((void **)&M.x2, (size_t)s->nb * 512 * sizeof(double)));
        s->bytes += (size_t)s->nb * 512 * sizeof(double);
      }
      M.nb = gl->nblocks();
      M.h = gl->h;
      max_halo_faces = std::max<int64_t>(max_halo_faces, gl->n_recv_faces);
    }
    if (N > 1) {
      const size_t bytes = (size_t)std::max<int64_t>(max_halo_faces, 1) * 64 * sizeof(double);
      CUP3D_HIP(hipMalloc((void **)&mg->zeros, bytes));
      CUP3D_HIP(hipMemset(mg->zeros, 0, bytes));
    }
    for (int i = nlev - 1; i >= 1; --i) {  // parent tables
      const Grid *gf = i == nlev - 1 ? g : mg->lev[i].grid.get(), *gc = mg->lev[i - 1].grid.get();
      std::vector<int32_t> par(2 * (size_t)gf->nblocks());
      for (int64_t b = 0; b < gf->nblocks(); ++b) {
        const int ii = gf->index[3 * b], j = gf->index[3 * b + 1], k = gf->index[3 * b + 2];
        const int32_t ps = gc->slot_of_index(ii >> 1, j >> 1, k >> 1);
        if (ps < 0) throw std::logic_error("multigrid: the parent of a local block is not local (partition does not nest)");
        par[2 * b] = ps;
        par[2 * b + 1] = (ii & 1) + 2 * (j & 1) + 4 * (k & 1);
      }
      if ((rc = up(&mg->lev[i].d_parent, par))) return rc;
    }
  } catch (const std::exception &e) {
    set_error("multigrid setup: %s", e.what());
    return CUP3D_EINVAL;
  }
  s->mg = mg.release();
  return CUP3D_OK;
}

void mg_destroy(Sim *s) {
  delete reinterpret_cast<Multigrid *>(s->mg);
  s->mg = nullptr;
}

// `launches` smoothing launches of `sweeps` sweeps each on one level; the iterate alternates between *xa and *xb and ends in *xa
static void mg_smooth(const MGLevel &M, const double *zeros, double **xa, double **xb, const double *rhs, int launches, int sweeps, bool from_zero) {
  const GridDev g = level_gdev(M);
  const dim3 G(launch_groups(g)), B(256);
  ProfileScope ps("mg_smooth");
  for (int i = 0; i < launches; ++i) {
    const double *halo = zeros ? zeros : (const double *)*xa;
    if (from_zero && i == 0) hipLaunchKernelGGL(k_mg_smooth<true>, G, B, 0, stream(), g, (const double *)nullptr, halo, rhs, *xb, sweeps);
    else hipLaunchKernelGGL(k_mg_smooth<false>, G, B, 0, stream(), g, (const double *)*xa, halo, rhs, *xb, sweeps);
    double *t = *xa;
    *xa = *xb;
    *xb = t;
  }
}

// out = V-cycle(in) from a zero guess; `in` is not modified
int mg_vcycle(Sim *s, const double *in, double *out) {
  int rc = mg_setup(s);
  if (rc) return rc;
  Multigrid &mg = *reinterpret_cast<Multigrid *>(s->mg);
  const int L = (int)mg.lev.size() - 1;
  // smoothing launches before / after the coarse-grid correction, sweeps per launch (ghosts are frozen within a launch);
  // cup3d_debug_set_option("mg_launches" / "mg_sweeps") for tuning scans
  const int nu = debug_option("mg_launches") > 0 ? debug_option("mg_launches") : 2, sw = debug_option("mg_sweeps") > 0 ? debug_option("mg_sweeps") : 2;
  std::vector<double *> xa(L + 1), xb(L + 1);  // xa[l]: where the level's iterate currently lives
  std::vector<const double *> rhs(L + 1);
  for (int l = 0; l <= L; ++l) {
    xa[l] = l == L ? out : mg.lev[l].x;
    xb[l] = mg.lev[l].x2;
    rhs[l] = l == L ? in : mg.lev[l].b;
  }

  for (int l = L; l >= 1; --l) {  // downward leg
    mg_smooth(mg.lev[l], mg.zeros, &xa[l], &xb[l], rhs[l], nu, sw, true);
    const GridDev g = level_gdev(mg.lev[l]);
    ProfileScope ps("mg_residual_restrict");
    hipLaunchKernelGGL(k_mg_residual_restrict, dim3(launch_groups(g)), dim3(256), 0, stream(), g, (const double *)xa[l], mg.zeros ? (const double *)mg.zeros : (const double *)xa[l], rhs[l], (const int32_t *)mg.lev[l].d_parent,
                       mg.lev[l - 1].b);
  }
  // coarsest level.  Its right-hand side loses its mean (the all-Neumann operator is singular) -- except in a ONE-level hierarchy, where
  // that would make the whole M^-1 singular (it would annihilate the constant component of every input and BiCGSTAB could never
  // reduce the residual along it); there the sweeps just carry a multiple of mean(b) along, a fixed linear map like the rest.
  if (L > 0 && !mg.local) hipLaunchKernelGGL(k_mg_remove_mean, dim3(1), dim3(256), 0, stream(), mg.lev[0].b, (long)mg.lev[0].nb * 512);
  if (mg.lev[0].nb == 1) mg_smooth(mg.lev[0], mg.zeros, &xa[0], &xb[0], rhs[0], 1, 64, true);
  else mg_smooth(mg.lev[0], mg.zeros, &xa[0], &xb[0], rhs[0], 16, 4, true);
  for (int l = 1; l <= L; ++l) {  // upward leg
    {
      ProfileScope ps("mg_prolong_add");
      hipLaunchKernelGGL(k_mg_prolong_add, dim3((unsigned)mg.lev[l].nb), dim3(256), 0, stream(), (int)mg.lev[l].nb, xa[l], (const int32_t *)mg.lev[l].d_parent,
                         (const double *)xa[l - 1]);
    }
    mg_smooth(mg.lev[l], mg.zeros, &xa[l], &xb[l], rhs[l], nu, sw, false);
  }
  if (xa[L] != out)  // an odd number of buffer swaps on the finest level cannot happen with nu + nu launches, but stay safe
    CUP3D_HIP(hipMemcpyAsync(out, xa[L], (size_t)s->nb * 512 * sizeof(double), hipMemcpyDeviceToDevice, stream()));
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

}  // namespace cup3d
