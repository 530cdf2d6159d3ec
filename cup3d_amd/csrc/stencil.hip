// 7-point (stencil [-1,2)) kernels on the ghosted 10^3 tile: Poisson LHS, pressure RHS,
// div(grad p) increment, pressure gradient.  One 256-thread workgroup per 8^3 block,
// tile staged in LDS from the face neighbours / halo slabs / boundary conditions
// (replaces ScalarLab / VectorLab loads, main.cpp:3623-3743, 5929-6004, 6107-6503).
// All arithmetic keeps the reference's association (-ffp-contract=off).
#include "sim.hpp"
#include "tile.hpp"
#include "tile7.hpp"

namespace cup3d {

// Multi-level meshes: face-flux arrays of the interface faces of this block (KernelLHSPoisson 9216-9267 and friends).
// fn(c, in, ghost, side, d) -> flux of component c given the LDS indices of the face cell and of the ghost behind it.
template <int NFC, class Fn>
__device__ __forceinline__ void write_face_fluxes(const GridDev &g, int slot, Fn fn) {
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int face = wave + 4 * i;
    if (face >= 6) continue;
    const int n = g.nbr[slot * 6 + face];
    if (n < kNbrHalo) continue;
    int nb_cell, own_cell, ghost;
    face1(face, lane, nb_cell, own_cell, ghost);
    const int in = tix(own_cell & 7, (own_cell >> 3) & 7, own_cell >> 6);
#pragma unroll
    for (int c = 0; c < NFC; ++c) g.flux[((size_t)(n - kNbrHalo) * NFC + c) * 64 + lane] = fn(c, in, ghost, face & 1, face >> 1);
  }
}

// component c of a vector field, ghosts only on the two faces normal to axis c (all the
// divergence reads); domain faces: wall negates every component, freespace negates the
// normal one -> the normal component is negated under both (main.cpp:6137-6153, 6384-6394)
__device__ __forceinline__ void load_normal_tile(const GridDev &g, int slot, const double *__restrict__ f, const double *__restrict__ halo,
                                                 int c, double *tile) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double *own = f + (size_t)slot * 1536 + c * 512;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  tile[tix(x, y, z0)] = own[cell0];
  tile[tix(x, y, z0 + 4)] = own[256 + cell0];
  if (wave < 2) {
    const int face = 2 * c + wave;
    const int n = g.nbr[slot * 6 + face];
    int nb_cell, own_cell, lds;
    face1(face, lane, nb_cell, own_cell, lds);
    double v;
    if (n >= kNbrHalo) v = halo[((size_t)(n - kNbrHalo) * 3 + c) * 64 + lane];
    else if (n >= 0) v = f[(size_t)n * 1536 + c * 512 + nb_cell];
    else v = -own[own_cell];
    tile[lds] = v;
  }
}

// all three components with ghosts behind all six faces (VectorLab with BlockLabBC: wall negates every component,
// freespace the one normal to the face, main.cpp:6137-6153, 6384-6394)
__device__ __forceinline__ void load_vector_tile(const GridDev &g, int slot, const double *__restrict__ f, const double *__restrict__ halo, double *tile) {
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const double *own = f + (size_t)slot * 1536;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    tile[c * kT + tix(x, y, z0)] = own[c * 512 + cell0];
    tile[c * kT + tix(x, y, z0 + 4)] = own[c * 512 + 256 + cell0];
  }
  for (int u = wave; u < 18; u += 4) {  // (face, component) units of 64 ghosts
    const int face = u / 3, c = u - 3 * face;
    const int n = g.nbr[slot * 6 + face];
    int nb_cell, own_cell, lds;
    face1(face, lane, nb_cell, own_cell, lds);
    double v;
    if (n >= kNbrHalo) v = halo[((size_t)(n - kNbrHalo) * 3 + c) * 64 + lane];
    else if (n >= 0) v = f[(size_t)n * 1536 + c * 512 + nb_cell];
    else { v = own[c * 512 + own_cell]; if (n == -3 || c == (face >> 1)) v = -v; }
    tile[c * kT + lds] = v;
  }
}

// ---- ComputeVorticity (main.cpp:8624-8746): tmpV = curl(vel) * (h^2/2) / h^3.  The reference's face-flux branch is dead code
// (it reads the BlockCase from the velocity grid's Info, which never has one), so there is nothing to flux-correct.
__global__ void __launch_bounds__(256) k_vorticity(GridDev g, const double *__restrict__ vel, const double *__restrict__ halo, double *__restrict__ tmpV) {
  __shared__ double tv[3 * kT];
  const int slot = block_slot(g);
  if (slot < 0) return;
  load_vector_tile(g, slot, vel, halo, tv);
  __syncthreads();
  const int t = threadIdx.x;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  const double h = block_h(g, slot), inv2h = .5 * h * h, fac = 1.0 / (h * h * h);
  const double *U = tv, *V = tv + kT, *W = tv + 2 * kT;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int b = tix(x, y, z0 + 4 * k);
    const size_t o = (size_t)slot * 1536 + k * 256 + cell0;
    tmpV[o] = (inv2h * ((W[b + 10] - W[b - 10]) - (V[b + kTP] - V[b - kTP]))) * fac;
    tmpV[o + 512] = (inv2h * ((U[b + kTP] - U[b - kTP]) - (W[b + 1] - W[b - 1]))) * fac;
    tmpV[o + 1024] = (inv2h * ((V[b + 1] - V[b - 1]) - (U[b + 10] - U[b - 10]))) * fac;
  }
}

// ---- KernelLHSPoisson (main.cpp:9205-9215) + the per-block partial of sum(p*h^3) that
// ComputeLHS needs for the mean constraint (9283-9294)
template <bool SKIPX = false>
__global__ void __launch_bounds__(256) k_lhs(GridDev g, const double *__restrict__ p, const double *__restrict__ halo, double *__restrict__ out,
                                             double *__restrict__ block_sums, const double *__restrict__ avg, int corner_slot) {
  __shared__ double tile[kT];
  __shared__ double red[4];
  const int slot = block_slot(g);
  if (slot < 0) return;
  double c[2];
  load_scalar_tile<SKIPX>(g, slot, p, halo, tile, c);
  __syncthreads();
  const int t = threadIdx.x;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  const double h = block_h(g, slot);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int z = z0 + 4 * k, b = tix(x, y, z);
    out[(size_t)slot * 512 + k * 256 + cell0] =
        h * (tile[b - 1] + tile[b + 1] + tile[b - 10] + tile[b + 10] + tile[b - kTP] + tile[b + kTP] - 6.0 * c[k]);
  }
  if (avg && slot == corner_slot && cell0 == 0) out[(size_t)slot * 512] = avg[0];  // LHS(0,0,0) = avgP, 9299-9304 (the total is known already)
  if (g.flux) write_face_fluxes<1>(g, slot, [&](int, int in, int gh, int, int) { return h * (tile[in] - tile[gh]); });
  if (block_sums) {
    const double h3 = h * h * h;
    const double s = group_sum<4>(c[0] * h3 + c[1] * h3, red);
    if (t == 0) block_sums[slot] = s;
  }
}

// deterministic sum of n per-block values in ONE launch: 64 workgroups -> 64 partials, the last workgroup to arrive totals them in
// index order (tile.hpp, grid_sum_finish) and, on one rank with bMeanConstraint == 1, writes the row of the corner cell:
// LHS(0,0,0) = avgP (main.cpp:9299-9304).  (Round 1 spent three launches on this, four times per BiCGSTAB iteration.)
__global__ void __launch_bounds__(256) k_mean_finish(const double *__restrict__ v, int n, double *__restrict__ part, unsigned *counter,
                                                     double *__restrict__ total, double *__restrict__ corner_out) {
  __shared__ double red[4];
  __shared__ int is_last;
  double s = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += v[i];
  s = group_sum<4>(s, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s;
    __threadfence();
    is_last = atomicAdd(counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last || threadIdx.x >= 64) return;
  __threadfence();
  double t = (int)threadIdx.x < (int)gridDim.x ? part[threadIdx.x] : 0.0;
  t = wave_sum(t);
  if (threadIdx.x == 0) {
    total[0] = t;
    if (corner_out) corner_out[0] = t;
    *counter = 0;
  }
}
// mean-constraint fix-ups of ComputeLHS::operator(), main.cpp:9299-9326
__global__ void k_lhs_corner(double *__restrict__ out, const double *__restrict__ in, const double *__restrict__ avg, int corner_slot, int mode) {
  if (mode == 1) out[(size_t)corner_slot * 512] = avg[0];          // LHS(0,0,0) = avgP
  else out[(size_t)corner_slot * 512] = in[(size_t)corner_slot * 512];  // bMeanConstraint > 2
}
__global__ void __launch_bounds__(256) k_lhs_add_mean(double *__restrict__ out, long n, const double *__restrict__ avg, double h3,
                                                      const double *__restrict__ hb) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double v = h3;
    if (hb) { const double h = hb[i >> 9]; v = h * h * h; }
    out[i] += avg[0] * v;  // 9314
  }
}

// ---- KernelPressureRHS (main.cpp:14849-14875)
__global__ void __launch_bounds__(256) k_pressure_rhs(GridDev g, const double *__restrict__ vel, const double *__restrict__ udef,
                                                      const double *__restrict__ chi, const double *__restrict__ halo_v,
                                                      const double *__restrict__ halo_u, double dt, double *__restrict__ out) {
  __shared__ double tv[3 * kT];
  __shared__ double tu[3 * kT];
  const int slot = block_slot(g);
  if (slot < 0) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) load_normal_tile(g, slot, vel, halo_v, c, tv + c * kT);
  if (chi)
#pragma unroll
    for (int c = 0; c < 3; ++c) load_normal_tile(g, slot, udef, halo_u, c, tu + c * kT);
  __syncthreads();
  const int t = threadIdx.x;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  const double h = block_h(g, slot), fac = 0.5 * h * h / dt;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int b = tix(x, y, z0 + 4 * k);
    double p = fac * (tv[b + 1] - tv[b - 1] + tv[kT + b + 10] - tv[kT + b - 10] + tv[2 * kT + b + kTP] - tv[2 * kT + b - kTP]);
    if (chi) {
      const double divUs = tu[b + 1] - tu[b - 1] + tu[kT + b + 10] - tu[kT + b - 10] + tu[2 * kT + b + kTP] - tu[2 * kT + b - kTP];
      p += -chi[(size_t)slot * 512 + k * 256 + cell0] * fac * divUs;
    }
    out[(size_t)slot * 512 + k * 256 + cell0] = p;
  }
  if (g.flux)  // main.cpp:14892-14945: the normal component of (u - chi*udef) summed across the face
    write_face_fluxes<1>(g, slot, [&](int, int in, int gh, int side, int d) {
      const double su = tv[d * kT + gh] + tv[d * kT + in];
      double v = side ? -fac * su : fac * su;
      if (chi) {
        const int x_ = (in % kTP) % 10 - 1, y_ = (in % kTP) / 10 - 1, z_ = in / kTP - 1;
        const double cu = chi[(size_t)slot * 512 + z_ * 64 + y_ * 8 + x_] * fac * (tu[d * kT + gh] + tu[d * kT + in]);
        v = side ? v + cu : v - cu;
      }
      return v;
    });
}

// ---- KernelDivPressure (main.cpp:14769-14778): tmpV.u[0] = h*lap(p)
__global__ void __launch_bounds__(256) k_div_pressure(GridDev g, const double *__restrict__ p, const double *__restrict__ halo, double *__restrict__ tmpV) {
  __shared__ double tile[kT];
  const int slot = block_slot(g);
  if (slot < 0) return;
  double c[2];
  load_scalar_tile(g, slot, p, halo, tile, c);
  __syncthreads();
  const int t = threadIdx.x;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  const double fac = block_h(g, slot);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int b = tix(x, y, z0 + 4 * k);
    tmpV[(size_t)slot * 1536 + k * 256 + cell0] =
        fac * (tile[b + 1] + tile[b - 1] + tile[b + 10] + tile[b - 10] + tile[b + kTP] + tile[b - kTP] - 6.0 * c[k]);
  }
  if (g.flux)  // main.cpp:14795-14833
    write_face_fluxes<1>(g, slot, [&](int, int in, int gh, int side, int) { return side ? -fac * (tile[gh] - tile[in]) : fac * (tile[in] - tile[gh]); });
}

// ---- KernelGradP (main.cpp:14990-14999), optionally fused with vel += tmpV/h^3 (15147-15159)
__global__ void __launch_bounds__(256) k_grad_p(GridDev g, const double *__restrict__ p, const double *__restrict__ halo, double dt,
                                                double *__restrict__ tmpV, double *__restrict__ vel) {
  __shared__ double tile[kT];
  const int slot = block_slot(g);
  if (slot < 0) return;
  double c[2];
  load_scalar_tile(g, slot, p, halo, tile, c);
  __syncthreads();
  const int t = threadIdx.x;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  const double h = block_h(g, slot), fac = -0.5 * dt * h * h, ih3 = 1.0 / (h * h * h);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int b = tix(x, y, z0 + 4 * k);
    const double gx = fac * (tile[b + 1] - tile[b - 1]);
    const double gy = fac * (tile[b + 10] - tile[b - 10]);
    const double gz = fac * (tile[b + kTP] - tile[b - kTP]);
    const size_t o = (size_t)slot * 1536 + k * 256 + cell0;
    tmpV[o] = gx; tmpV[o + 512] = gy; tmpV[o + 1024] = gz;
    if (vel) {
      vel[o] += ih3 * gx; vel[o + 512] += ih3 * gy; vel[o + 1024] += ih3 * gz;
    }
  }
  if (g.flux)  // main.cpp:15016-15054: only the component normal to the face is non-zero
    write_face_fluxes<3>(g, slot, [&](int c_, int in, int gh, int side, int d) {
      const double v = side ? -fac * (tile[gh] + tile[in]) : fac * (tile[gh] + tile[in]);
      return c_ == d ? v : 0.0;
    });
}

// vel += tmpV / h^3 (main.cpp:15147-15159) as its own pass: on multi-level meshes tmpV is flux-corrected first
__global__ void __launch_bounds__(256) k_add_scaled(const double *__restrict__ hb, const double *__restrict__ tmpV, double *__restrict__ vel, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double h = hb[i / 1536];
    vel[i] += (1.0 / (h * h * h)) * tmpV[i];
  }
}

// ---- KernelLHSDiffusion (main.cpp:6726-6803): out = h*(sum6 - 6p) + coef*p, coef = -h^3/(dt nu), on the BlockLabBC tile of
// velocity component `dir`
__global__ void __launch_bounds__(256) k_lhs_diffusion(GridDev g, const double *__restrict__ p, const double *__restrict__ halo, double *__restrict__ out,
                                                       int dir, double dt, double nu) {
  __shared__ double tile[kT];
  const int slot = block_slot(g);
  if (slot < 0) return;
  double c[2];
  load_scalar_tile(g, slot, p, halo, tile, c, dir);
  __syncthreads();
  const int t = threadIdx.x;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  const double h = block_h(g, slot), coef = -1.0 / (dt * nu) * h * h * h;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int z = z0 + 4 * k, b = tix(x, y, z);
    out[(size_t)slot * 512 + k * 256 + cell0] =
        h * (tile[b - 1] + tile[b + 1] + tile[b - 10] + tile[b + 10] + tile[b - kTP] + tile[b + kTP] - 6.0 * c[k]) + coef * c[k];
  }
  if (g.flux) write_face_fluxes<1>(g, slot, [&](int, int in, int gh, int, int) { return h * (tile[in] - tile[gh]); });
}

// ---- KernelDiffusionRHS (main.cpp:9729-9848): tmpV = h*lap(vel), each component with its own association
__global__ void __launch_bounds__(256) k_diffusion_rhs(GridDev g, const double *__restrict__ vel, const double *__restrict__ halo, double *__restrict__ tmpV) {
  __shared__ double tv[3 * kT];
  const int slot = block_slot(g);
  if (slot < 0) return;
  load_vector_tile(g, slot, vel, halo, tv);
  __syncthreads();
  const int t = threadIdx.x;
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  const double facD = block_h(g, slot);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int b = tix(x, y, z0 + 4 * k);
    const size_t o = (size_t)slot * 1536 + k * 256 + cell0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double *L = tv + c * kT;
      const double sx = L[b + 1] + L[b - 1], sy = L[b + 10] + L[b - 10], sz = L[b + kTP] + L[b - kTP];
      const double lap = (c == 0 ? (sx + (sy + sz)) : (c == 1 ? (sy + (sz + sx)) : (sz + (sx + sy)))) - 6 * L[b];
      tmpV[o + c * 512] = facD * lap;
    }
  }
  if (g.flux) write_face_fluxes<3>(g, slot, [&](int c, int in, int gh, int, int) { return facD * (tv[c * kT + in] - tv[c * kT + gh]); });
}

int launch_lhs_diffusion(Sim *s, const double *p, double *out, const HelmholtzOp &op) {
  s->scalar_bc_dir = op.direction;  // the coarse shadow tiles behind coarse/fine faces follow the same domain-face rule
  int rc = halo_begin(s, p, 1, 1);
  s->scalar_bc_dir = -1;
  if (rc) return rc;
  const bool split = s->grid->nranks > 1;
  for (int pass = 0; pass < (split ? 2 : 1); ++pass) {
    GridDev g = split ? s->gdev(pass == 1, pass == 0) : s->gdev();
    if (pass == 1 && (rc = halo_finish(s))) return rc;
    if (g.nblocks == 0) continue;
    ProfileScope ps("diffusion_lhs");
    hipLaunchKernelGGL(k_lhs_diffusion, dim3(launch_groups(g)), dim3(256), 0, stream(), g, p, s->halo_recv, out, op.direction, op.dt, op.nu);
  }
  CUP3D_HIP(hipGetLastError());
  if (s->grid->multilevel && (rc = amr_flux_fix(s, 1, out, 1))) return rc;  // compute<Lab>(.., pres, lhs) corrector, 6857-6862
  return CUP3D_OK;
}

int launch_diffusion_rhs(Sim *s) {
  int rc = halo_begin(s, s->vel, 3, 1);
  if (rc) return rc;
  const bool split = s->grid->nranks > 1;
  for (int pass = 0; pass < (split ? 2 : 1); ++pass) {
    GridDev g = split ? s->gdev(pass == 1, pass == 0) : s->gdev();
    if (pass == 1 && (rc = halo_finish(s))) return rc;
    if (g.nblocks == 0) continue;
    ProfileScope ps("diffusion_rhs");
    hipLaunchKernelGGL(k_diffusion_rhs, dim3(launch_groups(g)), dim3(256), 0, stream(), g, s->vel, s->halo_recv, s->tmpV);
  }
  CUP3D_HIP(hipGetLastError());
  if (s->grid->multilevel && (rc = amr_flux_fix(s, 3, s->tmpV, 3))) return rc;  // compute<VectorLab>(.., vel, tmpV), 10057
  return CUP3D_OK;
}

__global__ void __launch_bounds__(256) k_lhs_add_mean_list(double *__restrict__ out, const int32_t *__restrict__ list, const double *__restrict__ avg,
                                                           const double *__restrict__ hb) {
  const int slot = list[blockIdx.x];
  const double h = hb[slot], v = h * h * h;
  for (int i = threadIdx.x; i < 512; i += 256) out[(size_t)slot * 512 + i] += avg[0] * v;  // 9314
}

int launch_lhs(Sim *s, const double *p, double *out, int mc, const int32_t *list, unsigned nlist) {
  if (list) {
    // the interface blocks of a multi-level mesh on one rank (the loop kernels form the LHS of all other blocks): ghost slabs of every
    // interface face, k_lhs on the list (it fills the face fluxes of the list's faces = all of them), flux correction, then the
    // mean-constraint fix-ups of 9299-9326 for the list's blocks from the total the fused iteration has already formed
    const bool need_sum = mc > 0 && mc <= 2;
    if (!s->grid->multilevel || s->grid->nranks != 1 || (need_sum && s->mean_total_of != p)) { set_error("launch_lhs: block list without a known total"); return CUP3D_ESTATE; }
    const double *total = s->mean_total;
    s->mean_total_of = nullptr;
    s->sums_of = nullptr;
    int rc = amr_fill_ghosts(s, p, 1, 1, s->halo_recv);
    if (rc) return rc;
    if (nlist) {
      ProfileScope ps("poisson_lhs");
      const GridDev g = s->gdev_list(list, nlist);
      hipLaunchKernelGGL(k_lhs<false>, dim3(launch_groups(g)), dim3(256), 0, stream(), g, p, s->halo_recv, out, (double *)nullptr, (const double *)nullptr, -1);
    }
    CUP3D_HIP(hipGetLastError());
    if ((rc = amr_flux_fix(s, 1, out, 1))) return rc;
    const int corner = s->corner_is_plain ? -1 : s->grid->corner_slot;
    if (mc == 1 && corner >= 0) hipLaunchKernelGGL(k_lhs_corner, dim3(1), dim3(1), 0, stream(), out, p, total, corner, 1);
    if (mc == 2 && nlist) hipLaunchKernelGGL(k_lhs_add_mean_list, dim3(nlist), dim3(256), 0, stream(), out, list, total, s->d_hb);
    if (mc > 2 && corner >= 0) hipLaunchKernelGGL(k_lhs_corner, dim3(1), dim3(1), 0, stream(), out, p, total, corner, 3);
    CUP3D_HIP(hipGetLastError());
    return CUP3D_OK;
  }
  int rc = halo_begin(s, p, 1, 1);  // overlapped with the inner blocks, as compute<>() does (main.cpp:5598-5618)
  if (rc) return rc;
  const bool need_sum = mc > 0 && mc <= 2;
  const bool split = s->grid->nranks > 1;
  double *block_sums = s->d_partials + (size_t)s->max_groups * 8;
  const bool have_sums = need_sum && s->sums_of == p;  // the block solve that produced p already summed it
  s->sums_of = nullptr;
  // ... and the fused solver loop has even totalled those sums (k_sums_finish<K, true>) and, over ranks, all-reduced the total
  // together with its dot products (Reducer::begin; ev_a marks the arrival)
  const bool have_total = need_sum && s->mean_total_of == p;
  const double *total = have_total ? s->mean_total : s->d_red + kRedMeanLhs;
  s->mean_total_of = nullptr;
  const bool across = scalars_cross_ranks(s);
  const int corner = s->grid->corner_slot;
  // one rank, uniform grid: the kernel writes the constraint row itself (on a multi-level mesh the flux correction runs in between)
  const bool row_in_kernel = have_total && !across && mc == 1 && corner >= 0 && !s->grid->multilevel;
  for (int pass = 0; pass < (split ? 2 : 1); ++pass) {
    GridDev g = split ? s->gdev(pass == 1, pass == 0) : s->gdev();
    if (pass == 1 && (rc = halo_finish(s))) return rc;
    if (g.nblocks == 0) continue;
    ProfileScope ps("poisson_lhs");
#ifdef CUP3D_TUNING_ABLATIONS  // timing ablation with deliberately WRONG results (no x-face ghosts): never in a release build
    if (debug_option("lhs_variant") == 1) {
      hipLaunchKernelGGL(k_lhs<true>, dim3(launch_groups(g)), dim3(256), 0, stream(), g, p, s->halo_recv, out, (double *)nullptr, (const double *)nullptr, corner);
      continue;
    }
#endif
    hipLaunchKernelGGL(k_lhs<false>, dim3(launch_groups(g)), dim3(256), 0, stream(), g, p, s->halo_recv, out, (need_sum && !have_sums && !have_total) ? block_sums : nullptr,
                       row_in_kernel ? total : nullptr, corner);
  }
  CUP3D_HIP(hipGetLastError());
  if (s->grid->multilevel && (rc = amr_flux_fix(s, 1, out, 1))) return rc;  // compute(..., lhs) corrector, main.cpp:9298
  if (mc == 0) return CUP3D_OK;
  if (need_sum && have_total) {
    if (across) CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_a, 0));  // the all-reduced total (communication stream)
    if (mc == 1 && corner >= 0 && !row_in_kernel) hipLaunchKernelGGL(k_lhs_corner, dim3(1), dim3(1), 0, stream(), out, p, total, corner, 1);
    if (mc == 2) {
      const double h = s->grid->h;
      hipLaunchKernelGGL(k_lhs_add_mean, dim3(2048), dim3(256), 0, stream(), out, s->nb * 512L, total, h * h * h, s->d_hb);
    }
  } else if (need_sum) {
    ProfileScope ps("poisson_mean_sum");
    double *corner_out = (!across && mc == 1 && corner >= 0) ? out + (size_t)corner * 512 : nullptr;
    hipLaunchKernelGGL(k_mean_finish, dim3(64), dim3(256), 0, stream(), block_sums, (int)s->nb, s->d_partials, s->d_counters + 1, s->d_red + kRedMeanLhs, corner_out);
    if (across) {  // MPI_Iallreduce, main.cpp:9295 -- on the communication stream like every RCCL call; the fix-up below waits for it
      hipStream_t cs = scalar_stream(s);
      if (cs != stream()) {
        CUP3D_HIP(hipEventRecord(s->ev_b, stream()));
        CUP3D_HIP(hipStreamWaitEvent(cs, s->ev_b, 0));
      }
      if ((rc = allreduce(s, s->d_red + kRedMeanLhs, 1, false, cs))) return rc;
      if (cs != stream()) {
        CUP3D_HIP(hipEventRecord(s->ev_h1, cs));
        CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_h1, 0));
      }
      if (mc == 1 && corner >= 0) hipLaunchKernelGGL(k_lhs_corner, dim3(1), dim3(1), 0, stream(), out, p, s->d_red + kRedMeanLhs, corner, 1);
    }
    if (mc == 2) {
      const double h = s->grid->h;
      hipLaunchKernelGGL(k_lhs_add_mean, dim3(2048), dim3(256), 0, stream(), out, s->nb * 512L, s->d_red + kRedMeanLhs, h * h * h, s->d_hb);
    }
  } else if (corner >= 0) {
    hipLaunchKernelGGL(k_lhs_corner, dim3(1), dim3(1), 0, stream(), out, p, s->d_red + kRedMeanLhs, corner, 3);
  }
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

// the total of the per-block sums currently in the tail of d_partials (a block solve with want_sums, k_lhs, k_refresh_pointwise left them
// there) -> d_red[kRedMeanLhs]; one rank (the fused refresh of the solver, poisson.hip)
int launch_mean_total(Sim *s) {
  ProfileScope ps("poisson_mean_sum");
  double *block_sums = s->d_partials + (size_t)s->max_groups * 8;
  hipLaunchKernelGGL(k_mean_finish, dim3(64), dim3(256), 0, stream(), (const double *)block_sums, (int)s->nb, s->d_partials, s->d_counters + 1, s->d_red + kRedMeanLhs, (double *)nullptr);
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

}  // namespace cup3d

using namespace cup3d;

extern "C" {

int cup3d_compute_lhs(cup3d_sim_t *h, int mean_constraint) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  return launch_lhs(s, s->pres, s->lhs, mean_constraint);
}

int cup3d_pressure_rhs(cup3d_sim_t *h, double dt) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int rc;
  // several ranks: every rank takes the chi / udef path (adding -0*fac*0 where there is no obstacle is the identity), because the
  // udef exchange below is a collective and chi_nonzero is per-rank state (an obstacle covers blocks of some ranks only)
  // (cup3d_sim_set_obstacles(.., 0) says no rank has one: the obstacle-free run then skips the exchange of zeros)
  const bool obst = s->chi_path();
  double *halo_u = nullptr;
  if (obst && s->grid->multilevel) {
    halo_u = s->halo_recv + (size_t)s->grid->n_amr_faces() * 3 * 64;
    if ((rc = view_exchange_blocks(s, s->tmpV, 3, 1))) return rc;  // rank views: udef of the ghost blocks
    if ((rc = amr_fill_ghosts(s, s->tmpV, 3, 1, halo_u))) return rc;
  } else if (obst && s->grid->nranks > 1) {
    // udef slabs go to the second half of the receive buffer (each exchange uses <= 3*64 per face of 9*64)
    if ((rc = halo_exchange(s, s->tmpV, 3, 1))) return rc;
    halo_u = s->halo_recv + (size_t)s->grid->n_recv_faces * 3 * 64;
    CUP3D_HIP(hipMemcpyAsync(halo_u, s->halo_recv, (size_t)s->grid->n_recv_faces * 3 * 64 * sizeof(double), hipMemcpyDeviceToDevice, stream()));
  }
  if ((rc = halo_exchange(s, s->vel, 3, 1))) return rc;
  GridDev g = s->gdev();
  ProfileScope ps("pressure_rhs");
  hipLaunchKernelGGL(k_pressure_rhs, dim3(launch_groups(g)), dim3(256), 0, stream(), g, s->vel, s->tmpV, obst ? s->chi : nullptr, s->halo_recv, halo_u, dt, s->lhs);
  CUP3D_HIP(hipGetLastError());
  if (s->grid->multilevel) return amr_flux_fix(s, 1, s->lhs, 1);
  return CUP3D_OK;
}

int cup3d_div_pressure(cup3d_sim_t *h) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int rc = halo_exchange(s, s->pres, 1, 1);
  if (rc) return rc;
  GridDev g = s->gdev();
  ProfileScope ps("div_pressure");
  hipLaunchKernelGGL(k_div_pressure, dim3(launch_groups(g)), dim3(256), 0, stream(), g, s->pres, s->halo_recv, s->tmpV);
  CUP3D_HIP(hipGetLastError());
  if (s->grid->multilevel) return amr_flux_fix(s, 1, s->tmpV, 3);  // only tmpV.u[0] carries the result
  return CUP3D_OK;
}

int cup3d_compute_vorticity(cup3d_sim_t *h) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int rc = halo_exchange(s, s->vel, 3, 1);
  if (rc) return rc;
  GridDev g = s->gdev();
  ProfileScope ps("vorticity");
  hipLaunchKernelGGL(k_vorticity, dim3(launch_groups(g)), dim3(256), 0, stream(), g, s->vel, s->halo_recv, s->tmpV);
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

static int grad_p(Sim *s, double dt, bool update_vel) {
  int rc = halo_exchange(s, s->pres, 1, 1);
  if (rc) return rc;
  GridDev g = s->gdev();
  ProfileScope ps("grad_p");
  const bool ml = s->grid->multilevel;
  hipLaunchKernelGGL(k_grad_p, dim3(launch_groups(g)), dim3(256), 0, stream(), g, s->pres, s->halo_recv, dt, s->tmpV, (update_vel && !ml) ? s->vel : nullptr);
  CUP3D_HIP(hipGetLastError());
  if (ml) {
    if ((rc = amr_flux_fix(s, 3, s->tmpV, 3))) return rc;
    if (update_vel) hipLaunchKernelGGL(k_add_scaled, dim3(2048), dim3(256), 0, stream(), s->d_hb, s->tmpV, s->vel, s->nb * 1536L);
    CUP3D_HIP(hipGetLastError());
  }
  return CUP3D_OK;
}
int cup3d_grad_p(cup3d_sim_t *h, double dt) {
  if (!h) return CUP3D_EINVAL;
  return grad_p(reinterpret_cast<Sim *>(h), dt, false);
}
__attribute__((visibility("hidden"))) int cup3d_grad_p_update(cup3d_sim_t *h, double dt) {  // internal to cup3d_pressure_project, not exported
  return grad_p(reinterpret_cast<Sim *>(h), dt, true);
}

}  // extern "C"
