// Obstacle operators that sit between / inside the two hot-path operators: KernelPenalization
// (main.cpp:13841-13912) + kernelFinalizePenalizationForce (13913-13938), and kernelUpdateTmpV (14948-14979).
// The obstacles themselves (geometry, chi/udef rasterisation, rigid-body integration) stay on the host; these kernels take the
// ObstacleBlocks of one obstacle at a time -- chi[8][8][8] and udef[8][8][8][3] in the reference's own (AoS) layout -- so the
// velocity does not have to leave HBM between AdvectionDiffusion and PressureProjection when obstacles are present.
// Velocities and tmpV are bit-exact with the reference; the force / torque sums are reductions (block totals summed in block order
// on the host, cells within a block in tree order on the device).
#include <algorithm>
#include <vector>

#include "sim.hpp"
#include "tile.hpp"

namespace cup3d {

struct ObstItems {
  const int32_t *slots;  // [n] block slot of each ObstacleBlock
  const double *geom;    // [n][4]: h, origin[3] of that block (Info::h, Info::origin)
  const double *chi;     // [n][512]
  const double *udef;    // [n][512][3]
};

__global__ void __launch_bounds__(256) k_penalize(ObstItems it, double *__restrict__ vel, const double *__restrict__ chi_field, double dt,
                                                   double lambdaFac, int implicit, double cm0, double cm1, double cm2, double v0, double v1, double v2,
                                                   double o0, double o1, double o2, double *__restrict__ forces /* [n][6] */) {
  __shared__ double red[4];
  const int i = blockIdx.x, t = threadIdx.x;
  const int slot = it.slots[i];
  const double h = it.geom[4 * i], dv = pow(h, 3.0);
  double F[6] = {0, 0, 0, 0, 0, 0};
  for (int k = 0; k < 2; ++k) {
    const int c = k * 256 + t, ix = c & 7, iy = (c >> 3) & 7, iz = c >> 6;
    const double CHI = it.chi[(size_t)i * 512 + c];
    if (chi_field[(size_t)slot * 512 + c] > CHI) continue;
    if (CHI <= 0) continue;
    double p[3] = {it.geom[4 * i + 1] + h * (ix + 0.5), it.geom[4 * i + 2] + h * (iy + 0.5), it.geom[4 * i + 3] + h * (iz + 0.5)};
    p[0] -= cm0; p[1] -= cm1; p[2] -= cm2;
    const double *U = it.udef + ((size_t)i * 512 + c) * 3;
    const double UT0 = v0 + o1 * p[2] - o2 * p[1] + U[0];
    const double UT1 = v1 + o2 * p[0] - o0 * p[2] + U[1];
    const double UT2 = v2 + o0 * p[1] - o1 * p[0] + U[2];
    const double X = implicit ? (CHI > 0.5 ? 1.0 : 0.0) : CHI;
    const double penalFac = implicit ? X * lambdaFac / (1 + X * lambdaFac * dt) : X * lambdaFac;
    double *b = vel + (size_t)slot * 1536 + c;
    const double FPX = penalFac * (UT0 - b[0]), FPY = penalFac * (UT1 - b[512]), FPZ = penalFac * (UT2 - b[1024]);
    b[0] = b[0] + dt * FPX;
    b[512] = b[512] + dt * FPY;
    b[1024] = b[1024] + dt * FPZ;
    F[0] += dv * FPX; F[1] += dv * FPY; F[2] += dv * FPZ;
    F[3] += dv * (p[1] * FPZ - p[2] * FPY);
    F[4] += dv * (p[2] * FPX - p[0] * FPZ);
    F[5] += dv * (p[0] * FPY - p[1] * FPX);
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const double s = group_sum<4>(F[q], red);
    if (t == 0) forces[(size_t)i * 6 + q] = s;
  }
}

__global__ void __launch_bounds__(256) k_update_tmpv(ObstItems it, double *__restrict__ tmpV, const double *__restrict__ chi_field) {
  const int i = blockIdx.x, t = threadIdx.x;
  const int slot = it.slots[i];
  for (int k = 0; k < 2; ++k) {
    const int c = k * 256 + t;
    if (chi_field[(size_t)slot * 512 + c] > it.chi[(size_t)i * 512 + c]) continue;
    const double *U = it.udef + ((size_t)i * 512 + c) * 3;
    double *b = tmpV + (size_t)slot * 1536 + c;
    b[0] += U[0]; b[512] += U[1]; b[1024] += U[2];
  }
}

namespace {
struct DevBuf {
  void *p = nullptr;
  int alloc(size_t bytes) { CUP3D_HIP(hipMalloc(&p, bytes ? bytes : 8)); return CUP3D_OK; }
  int upload(const void *src, size_t bytes) {
    int rc = alloc(bytes);
    if (rc) return rc;
    if (bytes) CUP3D_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, stream()));
    return CUP3D_OK;
  }
  ~DevBuf() { if (p) hipFree(p); }
};

int stage(const Sim *s, const cup3d_obstacle &o, DevBuf &slots, DevBuf &geom, DevBuf &chi, DevBuf &udef, ObstItems *it) {
  const Grid *g = s->grid;
  std::vector<double> gm(4 * (size_t)o.nblocks);
  for (long i = 0; i < o.nblocks; ++i) {
    const int32_t b = o.slots[i];
    if (b < 0 || b >= s->nb) { set_error("obstacle block slot %d out of range", (int)b); return CUP3D_EINVAL; }
    const double h = g->multilevel ? g->hb[b] : g->h;
    gm[4 * i] = h;
    for (int d = 0; d < 3; ++d) gm[4 * i + 1 + d] = g->index[3 * (size_t)b + d] * kBS * h;  // Info::origin, main.cpp:1066-1068
  }
  int rc;
  if ((rc = slots.upload(o.slots, o.nblocks * sizeof(int32_t))) || (rc = geom.upload(gm.data(), gm.size() * sizeof(double))) ||
      (rc = chi.upload(o.chi, (size_t)o.nblocks * 512 * sizeof(double))) || (rc = udef.upload(o.udef, (size_t)o.nblocks * 1536 * sizeof(double))))
    return rc;
  CUP3D_HIP(hipStreamSynchronize(stream()));  // gm is a local
  it->slots = (const int32_t *)slots.p;
  it->geom = (const double *)geom.p;
  it->chi = (const double *)chi.p;
  it->udef = (const double *)udef.p;
  return CUP3D_OK;
}
}  // namespace

}  // namespace cup3d

using namespace cup3d;

extern "C" int cup3d_penalization(cup3d_sim_t *h, double dt, double lambda, int implicit, int nobst, cup3d_obstacle *obst) {
  if (!h || (nobst > 0 && !obst) || dt <= 0) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  const double lambdaFac = implicit ? lambda : 1.0 / dt;  // 13867
  for (int k = 0; k < nobst; ++k) {  // obstacles one after the other, as KernelPenalization::operator() visits them (13849-13852)
    cup3d_obstacle &o = obst[k];
    for (int d = 0; d < 3; ++d) o.force[d] = o.torque[d] = 0.0;
    // A rank whose share of the grid this obstacle does not touch still takes part in the all-reduce below with M = 0: the
    // reference issues MPI_Allreduce(M, 6) for every obstacle on every rank (13931), and skipping it here would pair this rank's
    // NEXT collective with the other ranks' 6-double sum.
    // M[6] rides along: 0, or 1 from a rank whose local part failed -- the ranks agree on the outcome in the collective they hold
    // anyway, so that one rank's bad obstacle returns an error everywhere instead of leaving the others inside the all-reduce
    double M[7] = {0, 0, 0, 0, 0, 0, 0};
    auto local_part = [&]() -> int {
      if (o.nblocks <= 0) return CUP3D_OK;
      int rc;
      if (!o.slots || !o.chi || !o.udef) { set_error("cup3d_penalization: obstacle %d has blocks but no slots / chi / udef", k); return CUP3D_EINVAL; }
      DevBuf slots, geom, chi, udef, forces;
      ObstItems it;
      if ((rc = stage(s, o, slots, geom, chi, udef, &it))) return rc;
      if ((rc = forces.alloc((size_t)o.nblocks * 6 * sizeof(double)))) return rc;
      {
        ProfileScope ps("penalization");
        hipLaunchKernelGGL(k_penalize, dim3((unsigned)o.nblocks), dim3(256), 0, stream(), it, s->vel, s->chi, dt, lambdaFac, implicit ? 1 : 0, o.cm[0], o.cm[1],
                           o.cm[2], o.vel[0], o.vel[1], o.vel[2], o.omega[0], o.omega[1], o.omega[2], (double *)forces.p);
      }
      CUP3D_HIP(hipGetLastError());
      std::vector<double> F((size_t)o.nblocks * 6);
      CUP3D_HIP(hipMemcpyAsync(F.data(), forces.p, F.size() * sizeof(double), hipMemcpyDeviceToHost, stream()));
      CUP3D_HIP(hipStreamSynchronize(stream()));
      // kernelFinalizePenalizationForce (13913-13938): block totals in block (slot) order
      std::vector<long> order(o.nblocks);
      for (long i = 0; i < o.nblocks; ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](long a, long b) { return o.slots[a] < o.slots[b]; });
      for (long i : order)
        for (int q = 0; q < 6; ++q) M[q] += F[(size_t)i * 6 + q];
      return CUP3D_OK;
    };
    int rc = local_part();
    if (scalars_cross_ranks(s)) {  // MPI_Allreduce(M, 6), 13931; on the stream every RCCL call of the library uses
      M[6] = rc ? 1.0 : 0.0;
      double *d = s->d_red;
      hipStream_t cs = scalar_stream(s);
      int rc2;
      CUP3D_HIP(hipStreamSynchronize(stream()));
      CUP3D_HIP(hipMemcpyAsync(d, M, 7 * sizeof(double), hipMemcpyHostToDevice, cs));
      if ((rc2 = allreduce(s, d, 7, false, cs))) return rc ? rc : rc2;
      CUP3D_HIP(hipMemcpyAsync(M, d, 7 * sizeof(double), hipMemcpyDeviceToHost, cs));
      CUP3D_HIP(hipStreamSynchronize(cs));
      if (!rc && M[6] != 0.0) {
        set_error("cup3d_penalization: obstacle %d failed on %d other rank(s)", k, (int)M[6]);
        rc = CUP3D_ECOMM;
      }
    }
    if (rc) return rc;
    for (int d = 0; d < 3; ++d) { o.force[d] = M[d]; o.torque[d] = M[3 + d]; }
  }
  return CUP3D_OK;
}

extern "C" int cup3d_update_tmpv(cup3d_sim_t *h, int nobst, const cup3d_obstacle *obst) {
  if (!h || (nobst > 0 && !obst)) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  for (int k = 0; k < nobst; ++k) {
    const cup3d_obstacle &o = obst[k];
    if (o.nblocks <= 0) continue;
    if (!o.slots || !o.chi || !o.udef) return CUP3D_EINVAL;
    DevBuf slots, geom, chi, udef;
    ObstItems it;
    int rc = stage(s, o, slots, geom, chi, udef, &it);
    if (rc) return rc;
    ProfileScope ps("update_tmpv");
    hipLaunchKernelGGL(k_update_tmpv, dim3((unsigned)o.nblocks), dim3(256), 0, stream(), it, s->tmpV, s->chi);
    CUP3D_HIP(hipGetLastError());
    CUP3D_HIP(hipStreamSynchronize(stream()));  // the staged arrays are freed on scope exit
  }
  if (nobst > 0) s->udef_nonzero = true;
  return CUP3D_OK;
}
