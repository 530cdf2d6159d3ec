// Implicit diffusion: AdvectionDiffusionImplicit::euler (main.cpp:10030-10118) on the device.
//
//   cup3d_advect_implicit         <- compute<VectorLab>(KernelAdvect, vel, tmpV)         10038, 9849-10029
//   cup3d_diffusion_rhs           <- compute<VectorLab>(KernelDiffusionRHS, vel, tmpV)   10057, 9729-9848
//   cup3d_diffusion_lhs           <- DiffusionSolver::_lhs / KernelLHSDiffusion          6836-6875, 6726-6803
//   cup3d_diffusion_preconditioner<- diffusion_kernels::getZImplParallel                 10534-10579
//   cup3d_diffusion_solve         <- DiffusionSolver::solve                              6896-7146
//   cup3d_advect_diffuse_implicit <- AdvectionDiffusionImplicit::operator()              10119
//
// One step = one 7-wide upwind kernel, one 7-point kernel, five pointwise passes and three Helmholtz solves (one per velocity
// component, each with the domain-face rule of that component) that reuse the BiCGSTAB driver, the fused vector kernels and
// the block-CG kernel of the pressure solver (poisson.hip) with a different centre coefficient.  The saved copies the
// reference keeps in std::vectors (`velocity`, `pressure`, 10036-10037) live in buffers that are idle during this operator:
// the second velocity buffer and pOld.
#include "sim.hpp"
#include "tile.hpp"

namespace cup3d {

#define GRID_STRIDE(j, n) for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < (n); j += (long)gridDim.x * 256)

__device__ __forceinline__ double cell_h(const double *__restrict__ hb, double h, long block) { return hb ? hb[block] : h; }

// velocity = V ; V = TMPV*ih3 + V   (10047-10053)
__global__ void __launch_bounds__(256) k_imp_guess(const double *__restrict__ hb, double h0, const double *__restrict__ tmpV, double *__restrict__ vel,
                                                   double *__restrict__ saved, long n) {
  GRID_STRIDE(i, n) {
    const double h = cell_h(hb, h0, i / 1536), ih3 = 1.0 / (h * h * h), v = vel[i];
    saved[i] = v;
    vel[i] = tmpV[i] * ih3 + v;
  }
}
// TMPV = -TMPV*ih3 + (V - velocity)/(dt*nu)   (10066-10074)
__global__ void __launch_bounds__(256) k_imp_rhs(const double *__restrict__ hb, double h0, double *__restrict__ tmpV, const double *__restrict__ vel,
                                                 const double *__restrict__ saved, double dtnu, long n) {
  GRID_STRIDE(i, n) {
    const double h = cell_h(hb, h0, i / 1536), ih3 = 1.0 / (h * h * h);
    tmpV[i] = -tmpV[i] * ih3 + (vel[i] - saved[i]) / dtnu;
  }
}
// P = 0 ; RHS = h3*TMPV.u[index]   (10083-10095); n = nb*512 scalar cells
__global__ void __launch_bounds__(256) k_imp_load(const double *__restrict__ hb, double h0, const double *__restrict__ tmpV, int index,
                                                  double *__restrict__ pres, double *__restrict__ rhs, long n) {
  GRID_STRIDE(i, n) {
    const long b = i >> 9;
    const double h = cell_h(hb, h0, b), h3 = h * h * h;
    pres[i] = 0;
    rhs[i] = h3 * tmpV[b * 1536 + index * 512 + (i & 511)];
  }
}
// V.u[index] += P   (10097-10106)
__global__ void __launch_bounds__(256) k_imp_add(const double *__restrict__ pres, int index, double *__restrict__ vel, long n) {
  GRID_STRIDE(i, n) vel[(i >> 9) * 1536 + index * 512 + (i & 511)] += pres[i];
}
__global__ void __launch_bounds__(256) k_imp_copy(const double *__restrict__ src, double *__restrict__ dst, long n) {
  GRID_STRIDE(i, n) dst[i] = src[i];
}

static cup3d_poisson_params diffusion_params(const cup3d_poisson_params *pp) {
  cup3d_poisson_params P;
  cup3d_poisson_default_params(&P);  // 1e-6 / 1e-4 are also sim.DiffusionErrorTol / DiffusionErrorTolRel (15369-15370)
  if (pp) P = *pp;
  if (P.block_solver != 2) P.block_solver = 0;  // block CG only (2 = reference association): the direct block solve is written for the Poisson coefficient
  return P;
}

}  // namespace cup3d

using namespace cup3d;

extern "C" {

int cup3d_advect_implicit(cup3d_sim_t *h, double dt, double nu, const double uinf[3]) {
  if (!h || !uinf) return CUP3D_EINVAL;
  return launch_advect_implicit(reinterpret_cast<Sim *>(h), dt, nu, uinf);
}

int cup3d_diffusion_rhs(cup3d_sim_t *h) {
  if (!h) return CUP3D_EINVAL;
  return launch_diffusion_rhs(reinterpret_cast<Sim *>(h));
}

int cup3d_diffusion_lhs(cup3d_sim_t *h, int direction, double dt, double nu) {
  if (!h || direction < 0 || direction > 2) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  return launch_lhs_diffusion(s, s->pres, s->lhs, HelmholtzOp{direction, dt, nu});
}

int cup3d_diffusion_preconditioner(cup3d_sim_t *h, double dt, double nu) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  return launch_precond_diffusion(s, s->pres, s->pres, HelmholtzOp{0, dt, nu});  // in place, as cup3d_preconditioner
}

int cup3d_diffusion_solve(cup3d_sim_t *h, int direction, double dt, double nu, const cup3d_poisson_params *pp, cup3d_poisson_result *r) {
  if (!h || direction < 0 || direction > 2) return CUP3D_EINVAL;
  return solve_helmholtz(reinterpret_cast<Sim *>(h), diffusion_params(pp), r, HelmholtzOp{direction, dt, nu});
}

int cup3d_advect_diffuse_implicit(cup3d_sim_t *hs, double dt, double nu, const double uinf[3], const cup3d_poisson_params *pp,
                                  cup3d_poisson_result results[3]) {
  if (!hs || !uinf) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(hs);
  const cup3d_poisson_params P = diffusion_params(pp);
  const long N = s->nb * 512L, N3 = 3 * N;
  const double h0 = s->grid->h;
  const dim3 G(2048), B(256);
  int rc;
  if ((rc = launch_advect_implicit(s, dt, nu, uinf))) return rc;  // tmpV = facD*lap(u), vel advected; s->vel2 is free again
  {
    ProfileScope ps("implicit_pointwise");
    hipLaunchKernelGGL(k_imp_copy, G, B, 0, stream(), s->pres, s->pold, N);  // `pressure`, 10046
    hipLaunchKernelGGL(k_imp_guess, G, B, 0, stream(), s->d_hb, h0, s->tmpV, s->vel, s->vel2, N3);
  }
  if ((rc = launch_diffusion_rhs(s))) return rc;
  {
    ProfileScope ps("implicit_pointwise");
    hipLaunchKernelGGL(k_imp_rhs, G, B, 0, stream(), s->d_hb, h0, s->tmpV, s->vel, s->vel2, dt * nu, N3);
  }
  for (int index = 0; index < 3; ++index) {
    {
      ProfileScope ps("implicit_pointwise");
      hipLaunchKernelGGL(k_imp_load, G, B, 0, stream(), s->d_hb, h0, s->tmpV, index, s->pres, s->lhs, N);
    }
    if ((rc = solve_helmholtz(s, P, results ? &results[index] : nullptr, HelmholtzOp{index, dt, nu}))) return rc;
    ProfileScope ps("implicit_pointwise");
    hipLaunchKernelGGL(k_imp_add, G, B, 0, stream(), s->pres, index, s->vel, N);
  }
  {
    ProfileScope ps("implicit_pointwise");
    hipLaunchKernelGGL(k_imp_copy, G, B, 0, stream(), s->pold, s->pres, N);  // 10108-10117
  }
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

}  // extern "C"
