// Geometric multigrid as the preconditioner of the pressure solver -- an ALTERNATIVE to the reference's block-local CG
// (cup3d_poisson_params.block_solver = 5), NOT a restatement of anything in main.cpp: the reference has no multigrid,
// its preconditioner is getZImplParallel (14704-14745).  BASELINE.json's north_star words the path as "geometric-multigrid
// smoother / restrict / prolong ... red-black Gauss-Seidel smoother"; this file is that wording taken literally, behind the same
// BiCGSTAB driver, the same operator A = h (sum6 - 6 p) (KernelLHSPoisson, 9205-9215), the same mean constraint and the same
// stopping rule -- so the CONVERGED pressure is the reference's to solver tolerance (tests), while the iteration count drops
// from O(150) to O(10) at 512^3.  bench.py reports it under `alt_multigrid`, never as `value`.  Uniform grids (over several ranks the
// levels exchange face slabs before every launch, see mg_setup) and multi-level meshes (mg_setup_amr: the octree's own levels are the
// hierarchy; restriction / prolongation across the refinement levels, AverageDown / TestInterp's place, main.cpp:3877-3906) -- on one rank
// and spread over ranks (rank views: every rank holds its owned nodes of every level + ghost nodes, Grid::mg_hierarchy; mg_ghosts /
// mg_restrict_exchange carry the iterate and the restricted octants between ranks).
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
  int index = 0;               // position in Multigrid::lev (0 = coarsest): the per-level entries of the profile
  int64_t nb = 0;
  double h = 0;
  int32_t *d_nbr = nullptr;     // [nb][6]
  int32_t *d_parent = nullptr;  // [nb][2]: slot of the parent block on the next coarser level, octant (x + 2y + 4z)
  double *x = nullptr, *x2 = nullptr, *b = nullptr;  // coarse levels; the finest level uses the caller's vectors + x2
  bool own_nbr = false;
  // multi-level meshes (mg_setup_amr): the nodes of octree level l = its leaves + the ancestors of finer leaves
  int32_t *d_leaf = nullptr;    // [nb] slot of the leaf in the solver's mesh, -1 for an ancestor
  int32_t *d_cf = nullptr;      // [ncf][4] faces without a same-level neighbour: coarse node slot (level l-1), direction, side, tangential parities
  int64_t ncf = 0;
  double *slabs = nullptr;      // [ncf][64] ghost values behind those faces, injected from the coarse iterate
  Sim *xch = nullptr;           // several ranks: what halo_exchange() needs for this level's iterate (the solver's Sim on the finest level)
  bool own_xch = false;
  // multi-level meshes over ranks (Grid::mg_hierarchy): the arrays hold nb owned nodes + nghost ghost nodes; the exchange plans
  const MGLevelPlan *plan = nullptr;
  int64_t nghost = 0;
  int32_t *d_send_slots = nullptr, *d_rsend = nullptr, *d_rrecv = nullptr;
  double *pack = nullptr, *rpack = nullptr, *runpack = nullptr;  // send buffer of the ghost exchange; octant buffers of the restriction (64 doubles per item)
};
// profile entry of one kernel on one level, "mg_smooth@L6": the levels differ by a factor 8 in work, and the roofline of the V-cycle is a
// statement about the fine ones (bench.py: alt_multigrid.kernels); interned strings, ProfileScope keeps the pointer's text
// The table is built ONCE, inside the initialisation of a function-local static (thread-safe since C++11: the ranks of the in-process
// test communicator are host threads that make their first call for the same (kind, level) together -- ADVICE r5), and is never
// destroyed (a leaked heap object: no destructor runs at exit while another thread may still hold a pointer into it).
static const char *level_name(int kind, int level) {
  static const char *const base[4] = {"mg_smooth", "mg_smooth_from_zero", "mg_residual_restrict", "mg_prolong_add"};
  struct Names {
    std::string n[4][24];
    Names() {
      for (int k = 0; k < 4; ++k)
        for (int l = 0; l < 24; ++l) n[k][l] = std::string(base[k]) + "@L" + std::to_string(l);
    }
  };
  static const Names *const names = new Names;
  if (level < 0 || level >= 24) return base[kind];
  return names->n[kind][level].c_str();
}
struct Multigrid {
  std::vector<MGLevel> lev;  // [0] coarsest ... [L] finest
  double *zeros = nullptr;   // ghost values behind faces owned by other ranks (see mg_setup); the x pointer itself on one rank
  bool local = false;        // a rank-local hierarchy (several ranks)
  bool amr = false;          // hierarchy of a multi-level mesh (mg_setup_amr)
  std::shared_ptr<const MGHierarchy> plan;  // ... its tables (kept alive: the exchange plans are read at every cycle)
  ~Multigrid() {
    if (zeros) hipFree(zeros);
    for (size_t i = 0; i < lev.size(); ++i) {
      MGLevel &l = lev[i];
      if ((l.grid || l.own_nbr) && l.d_nbr) hipFree(l.d_nbr);
      if (l.own_xch) sim_comm_only_destroy(l.xch);
      void *p[] = {l.d_parent, l.x, l.x2, l.b, l.d_leaf, l.d_cf, l.slabs, l.d_send_slots, l.d_rsend, l.d_rrecv, l.pack, l.rpack, l.runpack};
      for (void *q : p) if (q) hipFree(q);
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

// The same smoother by ONE WAVEFRONT per block, the form BASELINE.json's north_star names ("wavefront shuffles for the red-black
// Gauss-Seidel smoother").  Lane l owns the z-column of cell (x = l & 7, y = l >> 3) in registers, split by colour: with p = (x + y) & 1,
// R[k] is the cell at z = 2k + p ((x + y + z) even: red) and B[k] the one at z = 2k + 1 - p.  A red cell's x / y neighbours are the
// BLACK cells of the adjacent lanes at the same z -- and since those lanes have the other parity, that is their B[k] for the very same
// k: one wavefront shuffle (ds_bpermute: the LDS crossbar, no LDS memory) per neighbour, no index arithmetic.  Its z neighbours are the
// lane's own B registers.  No LDS tile, no barrier: the four wavefronts of the workgroup form met at 2 x sweeps barriers per launch.
// Ghost values (neighbour blocks' face cells, the zero-gradient domain face, coarse/fine or other ranks' slabs) live in registers of
// the lanes on the block's faces and stay frozen during the launch, as in k_mg_smooth.  Same expression, same association: BIT-IDENTICAL
// iterates (tests/test_gpu_parity.py::test_multigrid_smoother_forms_agree).
__device__ __forceinline__ double shfl_f64(double v, int src_lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)b), hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <bool ZERO>
__global__ void __launch_bounds__(64) k_mg_smooth_wave(GridDev g, const double *__restrict__ xin, const double *__restrict__ halo, const double *__restrict__ b,
                                                       double *__restrict__ xout, int sweeps) {
  const int slot = block_slot(g);
  if (slot < 0) return;
  const int l = threadIdx.x, x = l & 7, y = l >> 3;
  const bool p = ((x + y) & 1) != 0;
  const size_t bo = (size_t)slot * 512;
  const double invh = 1.0 / g.h;
  double R[4], B[4], rR[4], rB[4];       // iterate and right-hand side / h, by colour
  double gxR[4], gxB[4], gyR[4], gyB[4];  // lateral ghosts of the face lanes at the z of R[k] / B[k]
  double gzm = 0.0, gzp = 0.0;            // ghosts below z = 0 and above z = 7
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double b0 = invh * b[bo + (2 * k) * 64 + l], b1 = invh * b[bo + (2 * k + 1) * 64 + l];
    rR[k] = p ? b1 : b0;
    rB[k] = p ? b0 : b1;
    R[k] = B[k] = gxR[k] = gxB[k] = gyR[k] = gyB[k] = 0.0;
  }
  if constexpr (!ZERO) {
    const double *own = xin + bo;
    double v[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) v[z] = own[z * 64 + l];
    // ghosts: the lanes on a face fetch their column of the neighbour's face layer (strided for the x and y faces: 8 lanes x 8 loads)
    const int fx = x == 0 ? 0 : 1, fy = y == 0 ? 2 : 3;
    const bool onx = x == 0 || x == 7, ony = y == 0 || y == 7;
    const int nx = onx ? g.nbr[slot * 6 + fx] : -1, ny = ony ? g.nbr[slot * 6 + fy] : -1, nzm = g.nbr[slot * 6 + 4], nzp = g.nbr[slot * 6 + 5];
    // every ghost is ONE load from a selected global address -- the neighbour's face cell, the slab element, or (zero-gradient domain
    // face) the own face cell once more -- never "a register or a load": the compiler turns that into a select of addresses with the
    // register parked in scratch
    double gx[8], gy[8];
    const double *px = nx >= kNbrHalo ? halo + (size_t)(nx - kNbrHalo) * 64 + y           // slab order: face1(), (a1, a2) = (y, z): + z * 8
                                      : (nx >= 0 ? xin + (size_t)nx * 512 + y * 8 + (x == 0 ? 7 : 0) : own + l);
    const double *py = ny >= kNbrHalo ? halo + (size_t)(ny - kNbrHalo) * 64 + x           // (a1, a2) = (x, z)
                                      : (ny >= 0 ? xin + (size_t)ny * 512 + (y == 0 ? 7 : 0) * 8 + x : own + l);
    const int sx = nx >= kNbrHalo ? 8 : 64, sy = ny >= kNbrHalo ? 8 : 64;
#pragma unroll
    for (int z = 0; z < 8; ++z) {
      gx[z] = px[z * sx];
      gy[z] = py[z * sy];
    }
    gzm = *(nzm >= kNbrHalo ? halo + (size_t)(nzm - kNbrHalo) * 64 + l : (nzm >= 0 ? xin + (size_t)nzm * 512 + 7 * 64 + l : own + l));  // (a1, a2) = (x, y)
    gzp = *(nzp >= kNbrHalo ? halo + (size_t)(nzp - kNbrHalo) * 64 + l : (nzp >= 0 ? xin + (size_t)nzp * 512 + l : own + 7 * 64 + l));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      R[k] = p ? v[2 * k + 1] : v[2 * k];
      B[k] = p ? v[2 * k] : v[2 * k + 1];
      gxR[k] = p ? gx[2 * k + 1] : gx[2 * k];
      gxB[k] = p ? gx[2 * k] : gx[2 * k + 1];
      gyR[k] = p ? gy[2 * k + 1] : gy[2 * k];
      gyB[k] = p ? gy[2 * k] : gy[2 * k + 1];
    }
  }
  const int lxm = l - 1, lxp = l + 1, lym = l - 8, lyp = l + 8;  // (the face lanes' out-of-block sources are replaced by their ghosts)
  for (int s = 0; s < sweeps; ++s) {
    // red: R[k] at z = 2k + p; lateral neighbours = the adjacent lanes' B[k]; z neighbours = own B[k - 1 + p], B[k + p]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double sxm = shfl_f64(B[k], lxm), sxp = shfl_f64(B[k], lxp), sym = shfl_f64(B[k], lym), syp = shfl_f64(B[k], lyp);
      const double xm = x == 0 ? gxR[k] : sxm, xp = x == 7 ? gxR[k] : sxp, ym = y == 0 ? gyR[k] : sym, yp = y == 7 ? gyR[k] : syp;
      const double below = k > 0 ? B[k > 0 ? k - 1 : 0] : gzm, above = k < 3 ? B[k < 3 ? k + 1 : 3] : gzp;
      const double zm = p ? B[k] : below, zp = p ? above : B[k];
      R[k] = (1.0 / 6.0) * ((xm + xp) + (ym + yp) + (zm + zp) - rR[k]);
    }
    // black: B[k] at z = 2k + 1 - p; lateral neighbours = the adjacent lanes' R[k]; z neighbours = own R[k - p], R[k + 1 - p]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double sxm = shfl_f64(R[k], lxm), sxp = shfl_f64(R[k], lxp), sym = shfl_f64(R[k], lym), syp = shfl_f64(R[k], lyp);
      const double xm = x == 0 ? gxB[k] : sxm, xp = x == 7 ? gxB[k] : sxp, ym = y == 0 ? gyB[k] : sym, yp = y == 7 ? gyB[k] : syp;
      const double below = k > 0 ? R[k > 0 ? k - 1 : 0] : gzm, above = k < 3 ? R[k < 3 ? k + 1 : 3] : gzp;
      const double zm = p ? below : R[k], zp = p ? R[k] : above;
      B[k] = (1.0 / 6.0) * ((xm + xp) + (ym + yp) + (zm + zp) - rB[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    xout[bo + (2 * k) * 64 + l] = p ? B[k] : R[k];
    xout[bo + (2 * k + 1) * 64 + l] = p ? R[k] : B[k];
  }
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

// ---- multi-level meshes
// b of the level's leaf nodes <- the solver's vector; the solver's vector <- x of the leaf nodes
__global__ void __launch_bounds__(256) k_mg_gather(const int32_t *__restrict__ leaf, const double *__restrict__ in, double *__restrict__ b) {
  const int ls = leaf[blockIdx.x];
  if (ls < 0) return;
  for (int i = threadIdx.x; i < 512; i += 256) b[(size_t)blockIdx.x * 512 + i] = in[(size_t)ls * 512 + i];
}
__global__ void __launch_bounds__(256) k_mg_scatter(const int32_t *__restrict__ leaf, const double *__restrict__ x, double *__restrict__ out) {
  const int ls = leaf[blockIdx.x];
  if (ls < 0) return;
  for (int i = threadIdx.x; i < 512; i += 256) out[(size_t)ls * 512 + i] = x[(size_t)blockIdx.x * 512 + i];
}
// ghost slab behind a face whose neighbour exists one level coarser only: every fine ghost cell takes the value of the coarse cell
// that contains it (piecewise-constant, like the prolongation).  Slab element order = face1() of tile7.hpp.
__global__ void __launch_bounds__(64) k_mg_cf_ghosts(const int32_t *__restrict__ cf, const double *__restrict__ xc, double *__restrict__ slabs) {
  const int e = blockIdx.x, lane = threadIdx.x, a1 = lane & 7, a2 = lane >> 3;
  const int cs = cf[4 * e], d = cf[4 * e + 1], side = cf[4 * e + 2], par = cf[4 * e + 3];
  const int t1 = 4 * (par & 1) + (a1 >> 1), t2 = 4 * (par >> 1) + (a2 >> 1), q = side ? 0 : 7;  // behind the minus face lies the coarse block's last layer
  const int cell = d == 2 ? q * 64 + t2 * 8 + t1 : (d == 1 ? t2 * 64 + q * 8 + t1 : t2 * 64 + t1 * 8 + q);
  slabs[(size_t)e * 64 + lane] = xc[(size_t)cs * 512 + cell];
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

// Multi-level meshes (one rank).  The hierarchy is the octree itself: level l holds the leaves of level l AND the ancestors (at level l)
// of every finer leaf, so a coarse level is a complete mesh of the region its blocks cover and the fine levels sit on top of parts
// of it.  One V-cycle from a zero guess:
//   down, l = lmax .. 1:  b_l = r on the leaves of level l (on ancestors: the summed residual of their children, from the step before);
//                         smooth A_l x_l = b_l on all nodes of the level, ghosts behind faces whose neighbour exists only one level
//                         coarser = 0 (the coarse iterate is still zero); restrict the residual into the parents;
//   level 0:              mean removed, many sweeps;
//   up, l = 1 .. lmax:    x_l += the parent's value (piecewise constant); ghosts behind coarse/fine faces = the coarse neighbour's
//                         cell values (k_mg_cf_ghosts); smooth;
//   z = x_l on the leaves.
// A fixed linear operator (the smoother is block-Jacobi with frozen ghosts, as on uniform grids).  It ignores the reference's
// coarse/fine flux matching and its quadratic ghost interpolation (those are in A, which BiCGSTAB applies exactly): as a
// preconditioner it only has to be close.
static int mg_setup_amr(Sim *s) {
  const Grid *g = s->grid;
  std::unique_ptr<Multigrid> mg(new Multigrid());
  mg->amr = true;
  // the tables of this rank's share of every level: Grid::mg_hierarchy (grid.cpp) -- built with the rank view where the mesh is spread over
  // ranks (the view does not keep the global mesh), here for a mesh on one rank
  try {
    if (g->n_local >= 0) {
      mg->plan = g->mg_plan_get();  // built now, on first use (std::invalid_argument: the mesh is not 2:1 balanced)
      if (!mg->plan) { set_error("multigrid: this rank view carries no level hierarchy (a tensorial view)"); return CUP3D_EINVAL; }
    } else {
      if (g->nranks > 1) { set_error("multigrid on a multi-level mesh over ranks needs rank views (cup3d_grid_rank_view)"); return CUP3D_EINVAL; }
      mg->plan = g->mg_hierarchy(nullptr, 0, 1, nullptr);
    }
  } catch (const std::invalid_argument &e) {  // the caller's mesh; a std::logic_error would be a bug in the plan and is not turned into "bad argument"
    set_error("multigrid: %s", e.what());
    return CUP3D_EINVAL;
  } catch (const std::exception &e) {
    set_error("multigrid: building the level hierarchy failed: %s", e.what());
    return CUP3D_ESTATE;
  }
  const MGHierarchy &H = *mg->plan;
  mg->local = H.nranks > 1;
  const int nlev = (int)H.lev.size();
  mg->lev.resize(nlev);
  auto up = [&](int32_t **d, const std::vector<int32_t> &v) -> int {
    CUP3D_HIP(hipMalloc((void **)d, std::max<size_t>(v.size(), 1) * sizeof(int32_t)));
    if (!v.empty()) CUP3D_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return CUP3D_OK;
  };
  int rc;
  for (int l = 0; l < nlev; ++l) {
    MGLevel &M = mg->lev[l];
    const MGLevelPlan &P = H.lev[l];
    M.index = l;
    M.plan = &P;
    M.nb = P.n_owned;
    M.nghost = P.n_ghost;
    M.h = P.h;
    M.own_nbr = true;
    M.ncf = (int64_t)P.cf.size() / 4;
    if ((rc = up(&M.d_nbr, P.nbr)) || (rc = up(&M.d_parent, P.parent)) || (rc = up(&M.d_leaf, P.leaf)) || (rc = up(&M.d_cf, P.cf))) return rc;
    const size_t nvis = (size_t)std::max<int64_t>(P.n_owned + P.n_ghost, 1);
    const size_t bytes = nvis * 512 * sizeof(double), sb = (size_t)std::max<int64_t>(M.ncf, 1) * 64 * sizeof(double);
    CUP3D_HIP(hipMalloc((void **)&M.x, bytes));
    CUP3D_HIP(hipMalloc((void **)&M.x2, bytes));
    CUP3D_HIP(hipMalloc((void **)&M.b, bytes));
    CUP3D_HIP(hipMalloc((void **)&M.slabs, sb));
    CUP3D_HIP(hipMemset(M.slabs, 0, sb));
    CUP3D_HIP(hipMemset(M.x, 0, bytes));
    CUP3D_HIP(hipMemset(M.x2, 0, bytes));
    CUP3D_HIP(hipMemset(M.b, 0, bytes));
    s->bytes += 3 * bytes + sb;
    if (H.nranks > 1) {
      if ((rc = up(&M.d_send_slots, P.send_slots)) || (rc = up(&M.d_rsend, P.rsend)) || (rc = up(&M.d_rrecv, P.rrecv))) return rc;
      const size_t ps = std::max<size_t>(P.send_slots.size(), 1) * 512 * sizeof(double);
      const size_t rs = std::max<size_t>(P.rsend.size() / 2, 1) * 64 * sizeof(double), rr = std::max<size_t>(P.rrecv.size() / 2, 1) * 64 * sizeof(double);
      CUP3D_HIP(hipMalloc((void **)&M.pack, ps));
      CUP3D_HIP(hipMalloc((void **)&M.rpack, rs));
      CUP3D_HIP(hipMalloc((void **)&M.runpack, rr));
      s->bytes += ps + rs + rr;
    }
  }
  s->mg = mg.release();
  return CUP3D_OK;
}

static int mg_setup(Sim *s) {
  if (s->mg) return CUP3D_OK;
  const Grid *g = s->grid;
  if (g->multilevel) return mg_setup_amr(s);
  std::unique_ptr<Multigrid> mg(new Multigrid());
  const int L = g->level, N = g->nranks;
  // Several ranks: ONE V-cycle over all ranks.  Every level is partitioned by the same rule as the solver's grid (Hilbert ranges),
  // and its iterate crosses ranks as face slabs before every launch that reads ghosts (halo_exchange: pack kernel + one grouped
  // send/recv per peer on the communication stream) -- so on the levels it has, the cycle computes what the one-rank cycle computes
  // (the smoother is block-Jacobi with frozen ghosts: a ghost is the neighbour's previous iterate wherever the neighbour lives).
  // The hierarchy goes down as far as the partition of the coarser grid still nests in this one (every rank's share a whole number of
  // parent blocks): level 1 for 2, 4 or 8 ranks of a cubic base grid of one block; that level is then solved by sweeps (mean removed
  // over all ranks).  Round 2 ran rank-local cycles with zero ghosts instead (additive Schwarz: 27-29 iterations where this needs 5-8).
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
      M.index = i;
      const Grid *gl = g;
      if (i < nlev - 1) {
        M.grid.reset(new Grid(g->bpd, g->level_max, lmin + i, g->maxextent, g->bc, g->rank, N));
        gl = M.grid.get();
        if (N > 1) {
          M.xch = sim_comm_only(gl, s->comm_stream);
          M.own_xch = true;
          if (!M.xch) { set_error("multigrid setup: exchange buffers of level %d", lmin + i); return CUP3D_ENOMEM; }
        }
        if ((rc = up(&M.d_nbr, gl->nbr))) return rc;
        const size_t bytes = (size_t)gl->nblocks() * 512 * sizeof(double);
        CUP3D_HIP(hipMalloc((void **)&M.x, bytes));
        CUP3D_HIP(hipMalloc((void **)&M.x2, bytes));
        CUP3D_HIP(hipMalloc((void **)&M.b, bytes));
        s->bytes += 3 * bytes;
      } else {
        M.d_nbr = s->d_nbr;
        if (N > 1) M.xch = s;
        CUP3D_HIP(hipMalloc((void **)&M.x2, (size_t)s->nb * 512 * sizeof(double)));
        s->bytes += (size_t)s->nb * 512 * sizeof(double);
      }
      M.nb = gl->nblocks();
      M.h = gl->h;
      max_halo_faces = std::max<int64_t>(max_halo_faces, gl->n_recv_faces);
    }
    (void)max_halo_faces;
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

// `launches` smoothing launches of `sweeps` sweeps each on one level; the iterate alternates between *xa and *xb and ends in *xa.
// Several ranks: the face slabs of the iterate travel before every launch that reads them (not before the first one of a cycle, whose
// iterate is zero everywhere).
static int mg_ghosts(Sim *s, const MGLevel &M, double *x);
// amr_sim != nullptr: a level of a multi-level mesh spread over ranks -- its ghost NODES are refreshed before every launch that reads them
static int mg_smooth(const MGLevel &M, const double *zeros, double **xa, double **xb, const double *rhs, int launches, int sweeps, bool from_zero, Sim *amr_sim = nullptr) {
  const GridDev g = level_gdev(M);
  const dim3 G(launch_groups(g)), B(256);
  const bool wave = !debug_option("mg_smooth_workgroup");  // production: one wavefront per block, shuffles (k_mg_smooth_wave); A/B: the LDS-tile form
  for (int i = 0; i < launches; ++i) {
    const bool zero = from_zero && i == 0;
    if (M.xch && !zero) { int rc = halo_exchange(M.xch, *xa, 1, 1); if (rc) return rc; }
    if (amr_sim && !zero) { int rc = mg_ghosts(amr_sim, M, *xa); if (rc) return rc; }
    const double *halo = M.xch ? (const double *)M.xch->halo_recv : (zeros ? zeros : (const double *)*xa);
    if (!halo) halo = *xa;  // a rank without remote faces
    ProfileScope ps(level_name(zero ? 1 : 0, M.index));
    if (M.nb == 0) { /* a rank that owns no node of this level still took part in the exchange above */ }
    else if (wave && zero) hipLaunchKernelGGL(k_mg_smooth_wave<true>, G, dim3(64), 0, stream(), g, (const double *)nullptr, halo, rhs, *xb, sweeps);
    else if (wave) hipLaunchKernelGGL(k_mg_smooth_wave<false>, G, dim3(64), 0, stream(), g, (const double *)*xa, halo, rhs, *xb, sweeps);
    else if (zero) hipLaunchKernelGGL(k_mg_smooth<true>, G, B, 0, stream(), g, (const double *)nullptr, halo, rhs, *xb, sweeps);
    else hipLaunchKernelGGL(k_mg_smooth<false>, G, B, 0, stream(), g, (const double *)*xa, halo, rhs, *xb, sweeps);
    double *t = *xa;
    *xa = *xb;
    *xb = t;
  }
  return CUP3D_OK;
}

// several ranks, coarsest level: b -= mean(b) over ALL ranks
__global__ void __launch_bounds__(256) k_mg_local_sum(const double *__restrict__ b, long n, double *__restrict__ out) {
  __shared__ double red[4];
  double s = 0;
  for (long i = threadIdx.x; i < n; i += 256) s += b[i];
  s = group_sum<4>(s, red);
  if (threadIdx.x == 0) out[0] = s;
}
__global__ void __launch_bounds__(256) k_mg_sub(double *__restrict__ b, long n, const double *__restrict__ total, double inv_n_global) {
  const double m = total[0] * inv_n_global;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] -= m;
}

// ---- multi-level meshes over ranks: the two exchanges of a level (plans: MGLevelPlan, grid.hpp)
__global__ void __launch_bounds__(256) k_mg_pack_nodes(const double *__restrict__ x, const int32_t *__restrict__ slots, double *__restrict__ out) {
  const double *src = x + (size_t)slots[blockIdx.x] * 512;
  for (int i = threadIdx.x; i < 512; i += 256) out[(size_t)blockIdx.x * 512 + i] = src[i];
}
// octant `oct` of block `slot` of b <-> 64 contiguous doubles
__global__ void __launch_bounds__(64) k_mg_pack_octants(const double *__restrict__ b, const int32_t *__restrict__ items, double *__restrict__ out) {
  const int slot = items[2 * blockIdx.x], oct = items[2 * blockIdx.x + 1], t = threadIdx.x;
  const int cx = 4 * (oct & 1) + (t & 3), cy = 4 * ((oct >> 1) & 1) + ((t >> 2) & 3), cz = 4 * (oct >> 2) + (t >> 4);
  out[(size_t)blockIdx.x * 64 + t] = b[(size_t)slot * 512 + cz * 64 + cy * 8 + cx];
}
__global__ void __launch_bounds__(64) k_mg_unpack_octants(double *__restrict__ b, const int32_t *__restrict__ items, const double *__restrict__ in) {
  const int slot = items[2 * blockIdx.x], oct = items[2 * blockIdx.x + 1], t = threadIdx.x;
  const int cx = 4 * (oct & 1) + (t & 3), cy = 4 * ((oct >> 1) & 1) + ((t >> 2) & 3), cz = 4 * (oct >> 2) + (t >> 4);
  b[(size_t)slot * 512 + cz * 64 + cy * 8 + cx] = in[(size_t)blockIdx.x * 64 + t];
}
// the ghost nodes of `x` (an array of level M) <- their owners' current values.  Collective; no-op on one rank.
static int mg_ghosts(Sim *s, const MGLevel &M, double *x) {
  const MGLevelPlan &P = *M.plan;
  if (P.send_count.size() <= 1) return CUP3D_OK;
  ProfileScope ps("mg_exchange");
  const unsigned ns = (unsigned)P.send_slots.size();
  if (ns) hipLaunchKernelGGL(k_mg_pack_nodes, dim3(ns), dim3(256), 0, stream(), (const double *)x, (const int32_t *)M.d_send_slots, M.pack);
  CUP3D_HIP(hipGetLastError());
  return exchange_items(s, M.pack, P.send_count, x + (size_t)M.nb * 512, P.recv_count, 512);
}
// after the residual of level M was restricted into bc (level below): the octants that landed in GHOST parents travel to the parents' owners
static int mg_restrict_exchange(Sim *s, const MGLevel &M, double *bc) {
  const MGLevelPlan &P = *M.plan;
  if (P.rsend_count.size() <= 1) return CUP3D_OK;
  ProfileScope ps("mg_exchange");
  const unsigned ns = (unsigned)(P.rsend.size() / 2), nr = (unsigned)(P.rrecv.size() / 2);
  if (ns) hipLaunchKernelGGL(k_mg_pack_octants, dim3(ns), dim3(64), 0, stream(), (const double *)bc, (const int32_t *)M.d_rsend, M.rpack);
  CUP3D_HIP(hipGetLastError());
  int rc = exchange_items(s, M.rpack, P.rsend_count, M.runpack, P.rrecv_count, 64);
  if (rc) return rc;
  if (nr) hipLaunchKernelGGL(k_mg_unpack_octants, dim3(nr), dim3(64), 0, stream(), bc, (const int32_t *)M.d_rrecv, (const double *)M.runpack);
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

static int mg_vcycle_amr(Sim *s, Multigrid &mg, const double *in, double *out, int nu, int sw) {
  const int L = (int)mg.lev.size() - 1;
  Sim *xs = mg.local ? s : nullptr;  // several ranks: ghost nodes are exchanged (mg_ghosts), restricted octants travel to remote parents
  int rc;
  std::vector<double *> xa(L + 1), xb(L + 1);
  for (int l = 0; l <= L; ++l) {
    MGLevel &M = mg.lev[l];
    xa[l] = M.x;
    xb[l] = M.x2;
    ProfileScope ps("mg_gather");
    if (M.nb) hipLaunchKernelGGL(k_mg_gather, dim3((unsigned)M.nb), dim3(256), 0, stream(), (const int32_t *)M.d_leaf, in, M.b);
    if (M.ncf) CUP3D_HIP(hipMemsetAsync(M.slabs, 0, (size_t)M.ncf * 64 * sizeof(double), stream()));  // the coarse iterate is zero on the way down
  }
  for (int l = L; l >= 1; --l) {
    MGLevel &M = mg.lev[l];
    if ((rc = mg_smooth(M, M.slabs, &xa[l], &xb[l], M.b, nu, sw, true, xs))) return rc;
    if (xs && (rc = mg_ghosts(xs, M, xa[l]))) return rc;  // the residual reads the neighbours' final iterate
    const GridDev g = level_gdev(M);
    {
      ProfileScope ps(level_name(2, M.index));
      if (M.nb) hipLaunchKernelGGL(k_mg_residual_restrict, dim3(launch_groups(g)), dim3(256), 0, stream(), g, (const double *)xa[l], (const double *)M.slabs, (const double *)M.b,
                                   (const int32_t *)M.d_parent, mg.lev[l - 1].b);
    }
    if (xs && (rc = mg_restrict_exchange(xs, M, mg.lev[l - 1].b))) return rc;
  }
  MGLevel &C = mg.lev[0];
  if (L > 0 && !xs) hipLaunchKernelGGL(k_mg_remove_mean, dim3(1), dim3(256), 0, stream(), C.b, (long)C.nb * 512);
  if (L > 0 && xs) {  // the same over all ranks: local sums, one all-reduce on the communication stream, subtraction
    double *tot = s->d_red + kRedMg;
    const long n = (long)C.nb * 512;
    hipLaunchKernelGGL(k_mg_local_sum, dim3(1), dim3(256), 0, stream(), (const double *)C.b, n, tot);
    hipStream_t cs = scalar_stream(s);
    if (cs != stream()) { CUP3D_HIP(hipEventRecord(s->ev_b, stream())); CUP3D_HIP(hipStreamWaitEvent(cs, s->ev_b, 0)); }
    if ((rc = allreduce(s, tot, 1, false, cs))) return rc;
    if (cs != stream()) { CUP3D_HIP(hipEventRecord(s->ev_a, cs)); CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_a, 0)); }
    if (n) hipLaunchKernelGGL(k_mg_sub, dim3(64), dim3(256), 0, stream(), C.b, n, (const double *)tot, 1.0 / (512.0 * (double)C.plan->n_global));
  }
  if (C.plan->n_global == 1) rc = mg_smooth(C, C.slabs, &xa[0], &xb[0], C.b, 1, 64, true, xs);
  else rc = mg_smooth(C, C.slabs, &xa[0], &xb[0], C.b, 16, 4, true, xs);
  if (rc) return rc;
  for (int l = 1; l <= L; ++l) {
    MGLevel &M = mg.lev[l];
    if (xs && (rc = mg_ghosts(xs, mg.lev[l - 1], xa[l - 1]))) return rc;  // remote parents and coarse neighbours: their final iterate
    {
      ProfileScope ps(level_name(3, M.index));
      if (M.nb) hipLaunchKernelGGL(k_mg_prolong_add, dim3((unsigned)M.nb), dim3(256), 0, stream(), (int)M.nb, xa[l], (const int32_t *)M.d_parent, (const double *)xa[l - 1]);
      if (M.ncf) hipLaunchKernelGGL(k_mg_cf_ghosts, dim3((unsigned)M.ncf), dim3(64), 0, stream(), (const int32_t *)M.d_cf, (const double *)xa[l - 1], M.slabs);
    }
    if ((rc = mg_smooth(M, M.slabs, &xa[l], &xb[l], M.b, nu, sw, false, xs))) return rc;
  }
  for (int l = 0; l <= L; ++l) {
    ProfileScope ps("mg_gather");
    if (mg.lev[l].nb) hipLaunchKernelGGL(k_mg_scatter, dim3((unsigned)mg.lev[l].nb), dim3(256), 0, stream(), (const int32_t *)mg.lev[l].d_leaf, (const double *)xa[l], out);
  }
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

// out = V-cycle(in) from a zero guess; `in` is not modified
int mg_vcycle(Sim *s, const double *in, double *out) {
  int rc = mg_setup(s);
  if (rc) return rc;
  Multigrid &mg = *reinterpret_cast<Multigrid *>(s->mg);
  if (mg.amr) return mg_vcycle_amr(s, mg, in, out, debug_option("mg_launches") > 0 ? debug_option("mg_launches") : 2, debug_option("mg_sweeps") > 0 ? debug_option("mg_sweeps") : 2);
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
    if ((rc = mg_smooth(mg.lev[l], mg.zeros, &xa[l], &xb[l], rhs[l], nu, sw, true))) return rc;
    const GridDev g = level_gdev(mg.lev[l]);
    Sim *xs = mg.lev[l].xch;
    if (xs && (rc = halo_exchange(xs, xa[l], 1, 1))) return rc;
    const double *halo = xs && xs->halo_recv ? (const double *)xs->halo_recv : (const double *)xa[l];
    ProfileScope ps(level_name(2, l));
    hipLaunchKernelGGL(k_mg_residual_restrict, dim3(launch_groups(g)), dim3(256), 0, stream(), g, (const double *)xa[l], halo, rhs[l], (const int32_t *)mg.lev[l].d_parent,
                       mg.lev[l - 1].b);
  }
  // coarsest level.  Its right-hand side loses its mean (the all-Neumann operator is singular) -- except in a ONE-level hierarchy, where
  // that would make the whole M^-1 singular (it would annihilate the constant component of every input and BiCGSTAB could never
  // reduce the residual along it); there the sweeps just carry a multiple of mean(b) along, a fixed linear map like the rest.
  if (L > 0 && !mg.local) hipLaunchKernelGGL(k_mg_remove_mean, dim3(1), dim3(256), 0, stream(), mg.lev[0].b, (long)mg.lev[0].nb * 512);
  if (L > 0 && mg.local) {  // the same over all ranks: local sums, one all-reduce on the communication stream, subtraction
    double *tot = s->d_red + kRedMg;
    const long n = (long)mg.lev[0].nb * 512;
    hipLaunchKernelGGL(k_mg_local_sum, dim3(1), dim3(256), 0, stream(), (const double *)mg.lev[0].b, n, tot);
    hipStream_t cs = scalar_stream(s);
    if (cs != stream()) { CUP3D_HIP(hipEventRecord(s->ev_b, stream())); CUP3D_HIP(hipStreamWaitEvent(cs, s->ev_b, 0)); }
    if ((rc = allreduce(s, tot, 1, false, cs))) return rc;
    if (cs != stream()) { CUP3D_HIP(hipEventRecord(s->ev_a, cs)); CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_a, 0)); }
    const double ncells = 512.0 * (double)mg.lev[0].grid->total_blocks;
    hipLaunchKernelGGL(k_mg_sub, dim3(64), dim3(256), 0, stream(), mg.lev[0].b, n, (const double *)tot, 1.0 / ncells);
  }
  if (mg.lev[0].nb == 1 && !mg.local) rc = mg_smooth(mg.lev[0], mg.zeros, &xa[0], &xb[0], rhs[0], 1, 64, true);
  else rc = mg_smooth(mg.lev[0], mg.zeros, &xa[0], &xb[0], rhs[0], 16, 4, true);
  if (rc) return rc;
  for (int l = 1; l <= L; ++l) {  // upward leg
    {
      ProfileScope ps(level_name(3, l));
      hipLaunchKernelGGL(k_mg_prolong_add, dim3((unsigned)mg.lev[l].nb), dim3(256), 0, stream(), (int)mg.lev[l].nb, xa[l], (const int32_t *)mg.lev[l].d_parent,
                         (const double *)xa[l - 1]);
    }
    if ((rc = mg_smooth(mg.lev[l], mg.zeros, &xa[l], &xb[l], rhs[l], nu, sw, false))) return rc;
  }
  if (xa[L] != out)  // an odd number of buffer swaps on the finest level cannot happen with nu + nu launches, but stay safe
    CUP3D_HIP(hipMemcpyAsync(out, xa[L], (size_t)s->nb * 512 * sizeof(double), hipMemcpyDeviceToDevice, stream()));
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

}  // namespace cup3d
