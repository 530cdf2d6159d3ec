// Mesh-adaptation block operators on the device, for whole-mesh transitions between two
// uniform levels l and l+1 (the data-movement part of MeshAdaptation, main.cpp:5023-5583):
//   cup3d_restrict    <- compress          5272-5329  (8-cell mean with the reference's association)
//   cup3d_prolong     <- refine_1 + RefineBlocks 5227-5249, 5493-5565 (2nd-order Taylor expansion from
//                        the tensorial [-1,2) tile: 26 neighbours + ordered domain-face passes)
//   cup3d_tag_blocks  <- TagLoadedBlock    5566-5582 + the level clamps of TagBlocksVector 5207-5211
// All three are bit-exact with the reference (tests/test_gpu_adapt.py).
#include "sim.hpp"
#include "tile.hpp"

namespace cup3d {

// coarse cell (cx,cy,cz) <- mean of the 2x2x2 fine cells of child (cx>>2, cy>>2, cz>>2)
__global__ void __launch_bounds__(256) k_restrict(const double *__restrict__ fine, double *__restrict__ coarse, const int32_t *__restrict__ child,
                                                  int nc) {
  const int pb = blockIdx.x, t = threadIdx.x;
  for (int k = 0; k < 2; ++k) {
    const int cell = k * 256 + t, cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
    const int fs = child[pb * 8 + (cz >> 2) * 4 + (cy >> 2) * 2 + (cx >> 2)];
    const int i = 2 * (cx & 3), j = 2 * (cy & 3), kk = 2 * (cz & 3);
    for (int c = 0; c < nc; ++c) {
      const double *b = fine + ((size_t)fs * nc + c) * 512;
#define B(a, bb, cc) b[((kk + (cc)) * 8 + (j + (bb))) * 8 + (i + (a))]
      coarse[((size_t)pb * nc + c) * 512 + cell] =
          0.125 * ((B(0, 0, 0) + B(1, 1, 1)) + (B(1, 0, 0) + B(0, 1, 1)) + (B(0, 1, 0) + B(1, 0, 1)) + (B(1, 1, 0) + B(0, 0, 1)));  // 5298-5302
#undef B
    }
  }
}

__device__ __forceinline__ int lix(int x, int y, int z) { return ((z + 1) * 10 + (y + 1)) * 10 + (x + 1); }

// one workgroup per coarse block; NC components staged as [NC][10][10][10] in LDS
template <int NC>
__global__ void __launch_bounds__(256) k_prolong(const double *__restrict__ coarse, double *__restrict__ fine, const int32_t *__restrict__ nbr27,
                                                 const int32_t *__restrict__ nbr6, const int32_t *__restrict__ child) {
  __shared__ double lab[NC * 1000];
  const int pb = blockIdx.x, t = threadIdx.x;
  // 1. centre + the 26 neighbours that BlockLab::load copies (3690-3725); skipped regions start at 0
  for (int e = t; e < 1000; e += 256) {
    const int lx = e % 10 - 1, ly = (e / 10) % 10 - 1, lz = e / 100 - 1;
    const int cx = lx < 0 ? -1 : (lx > 7 ? 1 : 0), cy = ly < 0 ? -1 : (ly > 7 ? 1 : 0), cz = lz < 0 ? -1 : (lz > 7 ? 1 : 0);
    const int n = nbr27[pb * 27 + (cx + 1) + 3 * (cy + 1) + 9 * (cz + 1)];
    const int cell = ((lz - 8 * cz) * 8 + (ly - 8 * cy)) * 8 + (lx - 8 * cx);
#pragma unroll
    for (int c = 0; c < NC; ++c) lab[c * 1000 + e] = n >= 0 ? coarse[((size_t)n * NC + c) * 512 + cell] : 0.0;
  }
  __syncthreads();
  // 2. domain faces in the reference's order x-,x+,y-,y+,z-,z+ (6513-6551 / 6561-6581): the whole ghost
  //    slab behind the face, edge strips included, from the face cell with the same transverse coordinates
  for (int f = 0; f < 6; ++f) {
    const int n = nbr6[pb * 6 + f];  // < 0: domain face, boundary condition -1-n
    if (n < 0) {
      const int d = f >> 1, side = f & 1, ghost = side ? 8 : -1, face = side ? 7 : 0, d1 = (d + 1) % 3, d2 = (d + 2) % 3;
      if (t < 100) {
        int p[3], q[3];
        p[d] = ghost; q[d] = face;
        p[d1] = q[d1] = t % 10 - 1;
        p[d2] = q[d2] = t / 10 - 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          double v = lab[c * 1000 + lix(q[0], q[1], q[2])];
          if (NC == 3 && (n == -3 || c == d)) v = -v;  // wall: all components; freespace: the normal one
          lab[c * 1000 + lix(p[0], p[1], p[2])] = v;
        }
      }
      __syncthreads();
    }
  }
  // 3. RefineBlocks, 5493-5565
  for (int k = 0; k < 2; ++k) {
    const int cell = k * 256 + t, x = cell & 7, y = (cell >> 3) & 7, z = cell >> 6;
    const int fs = child[pb * 8 + (z >> 2) * 4 + (y >> 2) * 2 + (x >> 2)];
    const int i = 2 * (x & 3), j = 2 * (y & 3), kk = 2 * (z & 3);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double *L = lab + c * 1000;
#define Lb(a, b, cc) L[lix(x + (a), y + (b), z + (cc))]
      const double dudx = 0.5 * (Lb(1, 0, 0) - Lb(-1, 0, 0));
      const double dudy = 0.5 * (Lb(0, 1, 0) - Lb(0, -1, 0));
      const double dudz = 0.5 * (Lb(0, 0, 1) - Lb(0, 0, -1));
      const double dudx2 = (Lb(1, 0, 0) + Lb(-1, 0, 0)) - 2.0 * Lb(0, 0, 0);
      const double dudy2 = (Lb(0, 1, 0) + Lb(0, -1, 0)) - 2.0 * Lb(0, 0, 0);
      const double dudz2 = (Lb(0, 0, 1) + Lb(0, 0, -1)) - 2.0 * Lb(0, 0, 0);
      const double dudxdy = 0.25 * ((Lb(1, 1, 0) + Lb(-1, -1, 0)) - (Lb(1, -1, 0) + Lb(-1, 1, 0)));
      const double dudxdz = 0.25 * ((Lb(1, 0, 1) + Lb(-1, 0, -1)) - (Lb(1, 0, -1) + Lb(-1, 0, 1)));
      const double dudydz = 0.25 * ((Lb(0, 1, 1) + Lb(0, -1, -1)) - (Lb(0, 1, -1) + Lb(0, -1, 1)));
      const double u = Lb(0, 0, 0), q2 = 0.03125 * (dudx2 + dudy2 + dudz2);
#undef Lb
      double *b = fine + ((size_t)fs * NC + c) * 512;
#define B(a, bb, cc) b[((kk + (cc)) * 8 + (j + (bb))) * 8 + (i + (a))]
      B(0, 0, 0) = u + 0.25 * (-(1.0) * dudx - dudy - dudz) + q2 + 0.0625 * (dudxdy + dudxdz + dudydz);
      B(1, 0, 0) = u + 0.25 * (dudx - dudy - dudz) + q2 + 0.0625 * (-(1.0) * dudxdy - dudxdz + dudydz);
      B(0, 1, 0) = u + 0.25 * (-(1.0) * dudx + dudy - dudz) + q2 + 0.0625 * (-(1.0) * dudxdy + dudxdz - dudydz);
      B(1, 1, 0) = u + 0.25 * (dudx + dudy - dudz) + q2 + 0.0625 * (dudxdy - dudxdz - dudydz);
      B(0, 0, 1) = u + 0.25 * (-(1.0) * dudx - dudy + dudz) + q2 + 0.0625 * (dudxdy - dudxdz - dudydz);
      B(1, 0, 1) = u + 0.25 * (dudx - dudy + dudz) + q2 + 0.0625 * (-(1.0) * dudxdy + dudxdz - dudydz);
      B(0, 1, 1) = u + 0.25 * (-(1.0) * dudx + dudy + dudz) + q2 + 0.0625 * (-(1.0) * dudxdy - dudxdz + dudydz);
      B(1, 1, 1) = u + 0.25 * (dudx + dudy + dudz) + q2 + 0.0625 * (dudxdy + dudxdz + dudydz);
#undef B
    }
  }
}

// TagLoadedBlock: Linf of |magnitude| per block -> Refine (1) / Compress (-1) / Leave (0)
__global__ void __launch_bounds__(256) k_tag(const double *__restrict__ f, int nc, double rtol, double ctol, int clamp_refine, int clamp_compress,
                                             signed char *__restrict__ states) {
  __shared__ double red[4];
  const int b = blockIdx.x, t = threadIdx.x;
  double linf = 0.0;
  for (int k = 0; k < 2; ++k) {
    const int cell = k * 256 + t;
    double m;
    if (nc == 1) m = f[(size_t)b * 512 + cell];  // ScalarElement::magnitude, 5783
    else {
      double s1 = 0.0;
      for (int c = 0; c < nc; ++c) { const double u = f[((size_t)b * nc + c) * 512 + cell]; s1 += u * u; }
      m = sqrt(s1);  // VectorElement::magnitude, 5873-5879
    }
    linf = fmax(linf, fabs(m));
  }
  linf = wave_max(linf);
  if ((t & 63) == 0) red[t >> 6] = linf;
  __syncthreads();
  if (t == 0) {
    linf = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    signed char st = linf > rtol ? 1 : (linf < ctol ? -1 : 0);
    if (st == 1 && clamp_refine) st = 0;    // already at levelMax-1 (5207-5211)
    if (st == -1 && clamp_compress) st = 0; // already at level 0
    states[b] = st;
  }
}

static int child_table(const Sim *coarse, const Sim *fine, std::vector<int32_t> &tab) {
  const Grid *gc = coarse->grid, *gf = fine->grid;
  if (gc->multilevel || gf->multilevel) { set_error("restrict/prolong/tag operate on uniform levels (whole-mesh transitions)"); return CUP3D_EINVAL; }
  bool same = gc->level + 1 == gf->level && gc->level_max == gf->level_max;
  for (int d = 0; d < 3; ++d) same = same && gc->bpd[d] == gf->bpd[d] && gc->bc[d] == gf->bc[d];
  if (!same) { set_error("restrict/prolong need two sims of the same box at levels l and l+1"); return CUP3D_EINVAL; }
  tab.resize(8 * (size_t)coarse->nb);
  for (int64_t pb = 0; pb < coarse->nb; ++pb)
    for (int K = 0; K < 2; ++K)
      for (int J = 0; J < 2; ++J)
        for (int I = 0; I < 2; ++I) {
          const int32_t s = gf->slot_of_index(2 * gc->index[3 * pb] + I, 2 * gc->index[3 * pb + 1] + J, 2 * gc->index[3 * pb + 2] + K);
          if (s < 0) { set_error("a child block lives on another rank: partitions of the two levels are not aligned"); return CUP3D_ESTATE; }
          tab[8 * pb + K * 4 + J * 2 + I] = s;
        }
  return CUP3D_OK;
}

template <typename T>
static int upload(std::vector<T> &h, T **d) {
  CUP3D_HIP(hipMalloc((void **)d, h.size() * sizeof(T)));
  CUP3D_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return CUP3D_OK;
}

}  // namespace cup3d

using namespace cup3d;

extern "C" {

int cup3d_restrict(cup3d_sim_t *fine_h, cup3d_sim_t *coarse_h, int field) {
  if (!fine_h || !coarse_h) return CUP3D_EINVAL;
  Sim *fine = reinterpret_cast<Sim *>(fine_h), *coarse = reinterpret_cast<Sim *>(coarse_h);
  std::vector<int32_t> tab;
  int rc = child_table(coarse, fine, tab), nc, nc2;
  if (rc) return rc;
  const double *src = fine->field(field, &nc);
  double *dst = coarse->field(field, &nc2);
  if (!src || !dst) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  int32_t *d_tab;
  if ((rc = upload(tab, &d_tab))) return rc;
  {
    ProfileScope ps("restrict_blocks");
    hipLaunchKernelGGL(k_restrict, dim3((unsigned)coarse->nb), dim3(256), 0, stream(), src, dst, d_tab, nc);
  }
  CUP3D_HIP(hipGetLastError());
  CUP3D_HIP(hipStreamSynchronize(stream()));
  CUP3D_HIP(hipFree(d_tab));
  return CUP3D_OK;
}

int cup3d_prolong(cup3d_sim_t *coarse_h, cup3d_sim_t *fine_h, int field) {
  if (!fine_h || !coarse_h) return CUP3D_EINVAL;
  Sim *fine = reinterpret_cast<Sim *>(fine_h), *coarse = reinterpret_cast<Sim *>(coarse_h);
  std::vector<int32_t> tab;
  int rc = child_table(coarse, fine, tab), nc, nc2;
  if (rc) return rc;
  const double *src = coarse->field(field, &nc);
  double *dst = fine->field(field, &nc2);
  if (!src || !dst) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  std::vector<int32_t> n27 = coarse->grid->neighbours27();
  for (int32_t v : n27)
    if (v == -2) { set_error("cup3d_prolong: a neighbour block lives on another rank (multi-rank mesh adaptation is not wired yet)"); return CUP3D_ESTATE; }
  int32_t *d_tab, *d_n27;
  if ((rc = upload(tab, &d_tab)) || (rc = upload(n27, &d_n27))) return rc;
  {
    ProfileScope ps("prolong_blocks");
    if (nc == 3) hipLaunchKernelGGL(k_prolong<3>, dim3((unsigned)coarse->nb), dim3(256), 0, stream(), src, dst, d_n27, coarse->d_nbr, d_tab);
    else hipLaunchKernelGGL(k_prolong<1>, dim3((unsigned)coarse->nb), dim3(256), 0, stream(), src, dst, d_n27, coarse->d_nbr, d_tab);
  }
  CUP3D_HIP(hipGetLastError());
  CUP3D_HIP(hipStreamSynchronize(stream()));
  CUP3D_HIP(hipFree(d_tab));
  CUP3D_HIP(hipFree(d_n27));
  return CUP3D_OK;
}

int cup3d_tag_blocks(cup3d_sim_t *h, int field, double rtol, double ctol, signed char *states) {
  if (!h || !states) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  int nc;
  const double *f = s->field(field, &nc);
  if (!f) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  const Grid *gr = s->grid;
  signed char *d_states;
  CUP3D_HIP(hipMalloc((void **)&d_states, (size_t)s->nb));
  {
    ProfileScope ps("tag_blocks");
    // multi-level meshes: the level clamps differ per block and are applied on the host below
    hipLaunchKernelGGL(k_tag, dim3((unsigned)s->nb), dim3(256), 0, stream(), f, nc, rtol, ctol,
                       (!gr->multilevel && gr->level == gr->level_max - 1) ? 1 : 0, (!gr->multilevel && gr->level == 0) ? 1 : 0, d_states);
  }
  CUP3D_HIP(hipGetLastError());
  CUP3D_HIP(hipMemcpyAsync(states, d_states, (size_t)s->nb, hipMemcpyDeviceToHost, stream()));
  CUP3D_HIP(hipStreamSynchronize(stream()));
  CUP3D_HIP(hipFree(d_states));
  if (gr->multilevel)
    for (int64_t b = 0; b < s->nb; ++b) {  // TagBlocksVector, main.cpp:5207-5211
      if (states[b] == 1 && gr->blevel[b] == gr->level_max - 1) states[b] = 0;
      if (states[b] == -1 && gr->blevel[b] == 0) states[b] = 0;
    }
  return CUP3D_OK;
}

}  // extern "C"
