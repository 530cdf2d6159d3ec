// Multi-level (AMR) meshes on the device: ghost reconstruction at coarse/fine faces and flux correction.
//
// The stencil kernels never see the mesh hierarchy.  Before a kernel runs, two small kernels write, for every
// interface face (a face whose same-level neighbour does not exist), the ghost slab the kernel will read -- in
// the same slab format and through the same `nbr >= kNbrHalo` indirection as the RCCL halo slabs of a
// multi-rank run:
//   * neighbour finer  -> "restrict": 8-cell AverageDown of the fine leaves (BlockLab::FineToCoarseExchange,
//     main.cpp:3907-4065, AverageDown 3877-3882);
//   * neighbour coarser -> "prolong": BlockLab::CoarseFineInterpolation (4236-4614).  Only what a star-shaped
//     stencil reads is built: layers 1-2 behind the face use the finite-difference mode (1-D quadratic
//     interpolation along both tangential axes + mixed term on the coarse layer next to the face, blended with
//     two own cells, 4374-4612); layer 3 (stencil [-3,4) only) uses TestInterp (3883-3906) on the 3x3x3 coarse
//     neighbourhood, which needs the coarse shadow tile (m_CoarsenedBlock) behind the face including its
//     tangential ghosts: coarser leaves copied (CoarseFineExchange 4066-4170), same-level neighbours averaged
//     down (FillCoarseVersion 4171-4235), domain faces mirrored (_apply_bc on the coarse tile, 3781).
// After the kernel, k_flux_fix applies FluxCorrectionMPI::FillBlockCases (2825-2935) to the coarse side.
// All arithmetic keeps the reference's association (-ffp-contract=off): results are bit-identical to it.
#include <memory>
#include <stdexcept>

#include "sim.hpp"
#include "tile.hpp"

namespace cup3d {

struct AmrDev {
  const int32_t *faces;  // [ne][2]: 6*slot + face, kind
  const int32_t *fine;   // [ne][4]
  const int32_t *nbr27;  // [nb][27]
  const int32_t *nbr;    // [nb][6]
  const int32_t *index;  // [nb][3]
  int bc_comp;           // scalar fields: -1 = zero-gradient domain faces (BlockLabNeumann3D); k = element of BlockLabBC<.., direction k>
};

__device__ __forceinline__ double avg_down8(const double v[8]) {  // AverageDown, main.cpp:3877-3882
  return 0.125 * (v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7]);
}
// the 2x2x2 cells with low corner (x,y,z) of a block component, in AverageDown's argument order (x slowest)
__device__ __forceinline__ double avg_block(const double *__restrict__ blk, int x, int y, int z) {
  double v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = blk[(z + (q & 1)) * 64 + (y + ((q >> 1) & 1)) * 8 + (x + (q >> 2))];
  return avg_down8(v);
}

// ---- neighbour finer: slab[(e*nc + c)*W + gl][a2*8 + a1]
template <int W>
__device__ __forceinline__ void ghost_restrict_face(const AmrDev &a, int e, const double *__restrict__ field, int nc,
                                                       double *__restrict__ slabs) {
  const int lane = threadIdx.x;
  const int sf = a.faces[2 * e], f = sf % 6, d = f >> 1, side = f & 1;
  const int a1 = lane & 7, a2 = lane >> 3;
  const int fe = a.fine[4 * e + (a1 >> 2) + 2 * (a2 >> 2)];
  const int fslot = a.faces[2 * fe] / 6;
  const int t1 = 2 * (a1 & 3), t2 = 2 * (a2 & 3);
  for (int c = 0; c < nc; ++c) {
    const double *__restrict__ blk = field + ((size_t)fslot * nc + c) * 512;
#pragma unroll
    for (int gl = 0; gl < W; ++gl) {
      const int n0 = side ? 2 * gl : 6 - 2 * gl;
      const int x = d == 0 ? n0 : t1, y = d == 1 ? n0 : (d == 0 ? t1 : t2), z = d == 2 ? n0 : t2;
      slabs[(((size_t)e * nc + c) * W + gl) * 64 + lane] = avg_block(blk, x, y, z);
    }
  }
}

// ---- neighbour coarser
__constant__ double kCoefPlus[9] = {-0.09375, 0.4375, 0.15625, 0.15625, -0.5625, 0.90625, -0.09375, 0.4375, 0.15625};   // d_coef_plus, 3485-3486
__constant__ double kCoefMinus[9] = {0.15625, -0.5625, 0.90625, -0.09375, 0.4375, 0.15625, 0.15625, 0.4375, -0.09375};  // d_coef_minus, 3487-3488

// 1-D quadratic interpolation at coarse position p (0..3) along one tangential axis of the 6x6 patch layer
// (stride `st` between neighbours along that axis), main.cpp:4419-4441; pm = positions used by the mixed term
__device__ __forceinline__ double interp1d(const double *__restrict__ P0, int p, int st, const double *__restrict__ coef, int &pp, int &pm) {
  if (p != 0 && p != 3) {
    pp = p + 1; pm = p - 1;
    return (coef[6] * P0[-st] + coef[8] * P0[st]) + coef[7] * P0[0];
  } else if (p == 0) {
    pp = p + 1; pm = p;
    return (coef[0] * P0[2 * st] + coef[1] * P0[st]) + coef[2] * P0[0];
  }
  pp = p; pm = p - 1;
  return (coef[3] * P0[-2 * st] + coef[4] * P0[-st]) + coef[5] * P0[0];
}

// CoarseFineInterpolation, finite-difference mode (main.cpp:4419-4600): the coarse-side value `av` of a fine ghost cell -- the two 1-D
// quadratic interpolations along the face's tangential axes plus the mixed term -- from the coarse layer behind the face.  P0 = the coarse
// cell that holds the ghost cell, (p1, p2) its tangential position (0..3), st1 / st2 the strides of the layer along the two axes, bit1 /
// bit2 which child of the coarse cell the ghost cell is.  ONE definition for the three callers (the ghost slabs of the star stencils,
// k_ghost_prolong; the tensorial tiles of the mesh adaptation, k_refine and k_grad_chi): a bit-exactness fix lands once.
__device__ __forceinline__ double fd_mode_av(const double *__restrict__ P0, int p1, int p2, int st1, int st2, int bit1, int bit2) {
  const double dd1 = 0.25 * (2 * bit1 - 1), dd2 = 0.25 * (2 * bit2 - 1);
  const double *coef1 = dd1 > 0 ? kCoefPlus : kCoefMinus, *coef2 = dd2 > 0 ? kCoefPlus : kCoefMinus;
  int pp1, pm1, pp2, pm2;
  const double x1D = interp1d(P0, p1, st1, coef1, pp1, pm1);
  const double x2D = interp1d(P0, p2, st2, coef2, pp2, pm2);
  double mixed_coef = 1.0;
  if (p1 != 0 && p1 != 3) mixed_coef *= 0.5;
  if (p2 != 0 && p2 != 3) mixed_coef *= 0.5;
#define PC(i, j) P0[((i) - p1) * st1 + ((j) - p2) * st2]
  const double mixed = mixed_coef * dd1 * dd2 * ((PC(pm1, pm2) + PC(pp1, pp2)) - (PC(pp1, pm2) + PC(pm1, pp2)));
#undef PC
  return (x1D + x2D) + mixed;
}
// ... blended with the block's own two cells behind the face (bv: the face cell, cv: the one behind it), main.cpp:4601-4608
__device__ __forceinline__ double fd_mode_blend(double av, double bv, double cv, int layer) {
  return layer == 0 ? (1.0 / 15.0) * (8.0 * av + (10.0 * bv - 3.0 * cv)) : (1.0 / 15.0) * (24.0 * av + (-15.0 * bv + 6 * cv));
}
// TestInterp (main.cpp:3883-3906): second-order Taylor expansion around a coarse cell; Cc(i, j, k) = the 3x3x3 coarse cells around it
// (1, 1, 1 = the cell itself), bit[d] = which child along d
template <class F>
__device__ __forceinline__ double test_interp(F Cc, const int (&bit)[3]) {
  const double dudx = 0.125 * (Cc(2, 1, 1) - Cc(0, 1, 1));
  const double dudy = 0.125 * (Cc(1, 2, 1) - Cc(1, 0, 1));
  const double dudz = 0.125 * (Cc(1, 1, 2) - Cc(1, 1, 0));
  const double dudxdy = 0.015625 * (Cc(0, 0, 1) + Cc(2, 2, 1) - Cc(2, 0, 1) - Cc(0, 2, 1));
  const double dudxdz = 0.015625 * (Cc(0, 1, 0) + Cc(2, 1, 2) - Cc(2, 1, 0) - Cc(0, 1, 2));
  const double dudydz = 0.015625 * (Cc(1, 0, 0) + Cc(1, 2, 2) - Cc(1, 2, 0) - Cc(1, 0, 2));
  const double lap = Cc(1, 1, 1) + 0.03125 * (Cc(0, 1, 1) + Cc(2, 1, 1) + Cc(1, 0, 1) + Cc(1, 2, 1) + Cc(1, 1, 0) + Cc(1, 1, 2) + (-6.0) * Cc(1, 1, 1));
  const double sx = bit[0] ? 1.0 : -1.0, sy = bit[1] ? 1.0 : -1.0, sz = bit[2] ? 1.0 : -1.0;
  return lap + sx * dudx + sy * dudy + sz * dudz + (sx * sy) * dudxdy + (sx * sz) * dudxdz + (sy * sz) * dudydz;
}

template <int W>
__device__ __forceinline__ void ghost_prolong_face(const AmrDev &a, int e, const double *__restrict__ field, int nc,
                                                      double *__restrict__ slabs) {
  __shared__ double patch[W][36];  // coarse shadow tile behind the face: [layer][(Z+1)*6 + (Y+1)], Y/Z in [-1,4] along (a1,a2)
  const int lane = threadIdx.x;
  const int sf = a.faces[2 * e], slot = sf / 6, f = sf % 6, ax = f >> 1, side = f & 1;
  const int ax1 = ax == 0 ? 1 : 0, ax2 = ax == 2 ? 1 : 2;  // tangential axes (fast, slow) = the reference's (y,z)/(x,z)/(x,y)
  const int par[3] = {a.index[3 * slot] & 1, a.index[3 * slot + 1] & 1, a.index[3 * slot + 2] & 1};
  const bool is_vector = nc == 3;
  const int a1i = lane & 7, a2i = lane >> 3;
  for (int c = 0; c < nc; ++c) {
    __syncthreads();
    // ---- coarse shadow values
    for (int i = lane; i < W * 36; i += 64) {
      const int L = i / 36, r = i - 36 * L;
      int P[3], code[3];
      P[ax] = side ? 4 + L : -1 - L;
      P[ax1] = r % 6 - 1;
      P[ax2] = r / 6 - 1;
      code[ax] = side ? 1 : -1;
      bool flip = false;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = k ? ax2 : ax1;
        int rg = P[t] < 0 ? -1 : (P[t] > 3 ? 1 : 0);
        if (rg != 0) {
          const int n = a.nbr[slot * 6 + 2 * t + (rg > 0)];
          if (n < 0) {  // domain face: _apply_bc on the coarse tile copies the face cell with the same transverse coordinates
            P[t] = rg < 0 ? 0 : 3;
            rg = 0;
            // wall: all components; freespace: the normal one.  A scalar of BlockLabBC<ScalarGrid, .., direction> (the Helmholtz
            // solves of the implicit diffusion, main.cpp:6853-6862) behaves as component `direction` of a vector
            const int cc = is_vector ? c : a.bc_comp;
            if ((is_vector || a.bc_comp >= 0) && (n == -3 || cc == t)) flip = !flip;
          }
        }
        code[t] = rg;
      }
      const int v = a.nbr27[slot * 27 + (code[0] + 1) + 3 * (code[1] + 1) + 9 * (code[2] + 1)];
      double val = 0.0;
      if (v >= kNbrCoarser) {  // CoarseFineExchange: cell of the coarser leaf
        const double *__restrict__ blk = field + ((size_t)(v - kNbrCoarser) * nc + c) * 512;
        const int q0 = (par[0] * 4 + P[0] + 8) & 7, q1 = (par[1] * 4 + P[1] + 8) & 7, q2 = (par[2] * 4 + P[2] + 8) & 7;
        val = blk[q2 * 64 + q1 * 8 + q0];
      } else if (v >= 0) {  // FillCoarseVersion: same-level neighbour averaged down
        const double *__restrict__ blk = field + ((size_t)v * nc + c) * 512;
        val = avg_block(blk, 2 * P[0] - 8 * code[0], 2 * P[1] - 8 * code[1], 2 * P[2] - 8 * code[2]);
      }
      patch[L][r] = flip ? -val : val;
    }
    __syncthreads();
    // ---- layers 1 (and 2): finite-difference mode
    const double *__restrict__ own = field + ((size_t)slot * nc + c) * 512;
    int cb[3], cc[3];
    cb[ax] = side ? 7 : 0; cc[ax] = side ? 6 : 1;
    cb[ax1] = cc[ax1] = a1i;
    cb[ax2] = cc[ax2] = a2i;
    const double bv = own[cb[2] * 64 + cb[1] * 8 + cb[0]], cv = own[cc[2] * 64 + cc[1] * 8 + cc[0]];
    const int p1 = a1i >> 1, p2 = a2i >> 1;
    const double av = fd_mode_av(&patch[0][(p2 + 1) * 6 + (p1 + 1)], p1, p2, 1, 6, a1i & 1, a2i & 1);
    double *__restrict__ out = slabs + ((size_t)e * nc + c) * W * 64 + lane;
    out[0] = fd_mode_blend(av, bv, cv, 0);
    if (W == 3) {
      out[64] = fd_mode_blend(av, bv, cv, 1);
      // ---- layer 3: TestInterp around the coarse cell two layers behind the face
      auto Cc = [&](int i, int j, int k) -> double {
        const int off[3] = {i, j, k};
        const int L = side ? off[ax] : 2 - off[ax];
        return patch[L][(p2 + off[ax2]) * 6 + (p1 + off[ax1])];
      };
      int bit[3];
      bit[ax] = side ? 0 : 1;  // the fine layer is the upper child of the coarse cell below the block, the lower one above it
      bit[ax1] = a1i & 1;
      bit[ax2] = a2i & 1;
      out[128] = test_interp(Cc, bit);
    }
  }
}

// ---- flux correction of the coarse side, one launch per normal direction (x, then y, then z: FillCase_2 order): rounds 1-4's form, kept in the
// TESTING build as the A/B and bit-for-bit cross-check of k_flux_fix_blocks below ("flux_fix_by_direction").
// out[cell] += own face flux + ((f00 + f10) + (f01 + f11)) of the fine faces; flux arrays [(e*nfc + c)][a2*8+a1]
#ifdef CUP3D_TESTING
__global__ void __launch_bounds__(64) k_flux_fix(AmrDev a, const int32_t *__restrict__ list, const double *__restrict__ flux, int nfc,
                                                 double *__restrict__ out, int out_nc) {
  const int e = list[blockIdx.x], lane = threadIdx.x;
  const int sf = a.faces[2 * e], slot = sf / 6, f = sf % 6, d = f >> 1, side = f & 1;
  const int a1 = lane & 7, a2 = lane >> 3, j = side ? 7 : 0;
  const int cell = d == 0 ? a2 * 64 + a1 * 8 + j : (d == 1 ? a2 * 64 + j * 8 + a1 : j * 64 + a2 * 8 + a1);
  const int fe = a.fine[4 * e + (a1 >> 2) + 2 * (a2 >> 2)];
  const int i2 = 2 * (a1 & 3), i1 = 2 * (a2 & 3);
  for (int c = 0; c < nfc; ++c) {
    const double *__restrict__ F = flux + ((size_t)fe * nfc + c) * 64;
    const double avg = (F[i2 + i1 * 8] + F[i2 + 1 + i1 * 8]) + (F[i2 + (i1 + 1) * 8] + F[i2 + 1 + (i1 + 1) * 8]);
    const double coarse = flux[((size_t)e * nfc + c) * 64 + lane] + avg;
    double *__restrict__ o = out + ((size_t)slot * out_nc + c) * 512 + cell;
    *o = (*o + coarse) + 0.0;  // + 0.0: the three further FillCase_2 calls on the cleared face
  }
}
#endif

// ONE launch for the ghost slabs of a stencil: workgroups [0, nr) restrict (neighbour finer), [nr, nr + np) interpolate (neighbour
// coarser).  These launches are a few microseconds of work each and sit between the loop kernels of every BiCGSTAB iteration on a
// multi-level mesh: what they cost is their number (rounds 1-4: two launches here, three in the flux correction below).
template <int W>
__global__ void __launch_bounds__(64) k_ghosts(AmrDev a, const int32_t *__restrict__ rlist, unsigned nr, const int32_t *__restrict__ plist, const double *__restrict__ field, int nc,
                                               double *__restrict__ slabs) {
  if (blockIdx.x < nr) ghost_restrict_face<W>(a, rlist[blockIdx.x], field, nc, slabs);
  else ghost_prolong_face<W>(a, plist[blockIdx.x - nr], field, nc, slabs);
}

// The flux correction in ONE launch: one wavefront per coarse-side BLOCK walks the block's corrected faces in the order x-, x+, y-, y+,
// z-, z+.  A cell lies on at most one face per direction, so every cell still receives its corrections direction by direction, x first
// (FillCase_2 order, 2918-2926) -- bit-identical to the three launches of k_flux_fix; the barrier between two faces orders the second
// face's read of a cell behind the first face's write of it (edge and corner cells of the block).
__global__ void __launch_bounds__(64) k_flux_fix_blocks(AmrDev a, const int32_t *__restrict__ tab /* [n][6]: interface face behind face f, -1 none */, const double *__restrict__ flux, int nfc,
                                                        double *out, int out_nc) {
  const int lane = threadIdx.x;
  const int a1 = lane & 7, a2 = lane >> 3;
  for (int f = 0; f < 6; ++f) {
    const int e = tab[6 * blockIdx.x + f];
    if (e < 0) continue;
    const int slot = a.faces[2 * e] / 6, d = f >> 1, j = (f & 1) ? 7 : 0;
    const int cell = d == 0 ? a2 * 64 + a1 * 8 + j : (d == 1 ? a2 * 64 + j * 8 + a1 : j * 64 + a2 * 8 + a1);
    const int fe = a.fine[4 * e + (a1 >> 2) + 2 * (a2 >> 2)];
    const int i2 = 2 * (a1 & 3), i1 = 2 * (a2 & 3);
    for (int c = 0; c < nfc; ++c) {
      const double *__restrict__ F = flux + ((size_t)fe * nfc + c) * 64;
      const double avg = (F[i2 + i1 * 8] + F[i2 + 1 + i1 * 8]) + (F[i2 + (i1 + 1) * 8] + F[i2 + 1 + (i1 + 1) * 8]);
      const double coarse = flux[((size_t)e * nfc + c) * 64 + lane] + avg;
      double *o = out + ((size_t)slot * out_nc + c) * 512 + cell;
      *o = (*o + coarse) + 0.0;  // + 0.0: the three further FillCase_2 calls on the cleared face
    }
    __syncthreads();
  }
}

// ---- mesh adaptation: the eight children of a refined block (refine_1 + RefineBlocks, main.cpp:5227-5249, 5493-5565).
// One workgroup per refined parent builds the parent's tensorial [-1,2) tile on the OLD mesh in LDS exactly as BlockLab::load
// does -- same-level neighbours copied, finer ones averaged down, coarser ones interpolated from the coarse shadow tile
// (faces: finite-difference mode, edges and corners: TestInterp), domain faces last -- and expands it.
struct RefineTab {
  const int32_t *items;  // [n][9]: parent slot (old mesh), eight child slots (new mesh, child = I + 2J + 4K)
  const int32_t *finer;  // [n][27][8]: old-mesh slot of the finer leaf behind code for the octant (bits of x,y,z >= 4), -1 unused
};
__device__ __forceinline__ int lix10(int x, int y, int z) { return ((z + 1) * 10 + (y + 1)) * 10 + (x + 1); }
__device__ __forceinline__ int cix8(int X, int Y, int Z) { return ((Z + 2) * 8 + (Y + 2)) * 8 + (X + 2); }

template <int NC>
__global__ void __launch_bounds__(256) k_refine(AmrDev a, RefineTab tab, const double *__restrict__ src, double *__restrict__ dst) {
  __shared__ double lab[NC * 1000];
  __shared__ double Ct[NC * 512];  // coarse shadow tile, coarse cells [-2,6)^3
  const int it = blockIdx.x, t = threadIdx.x;
  const int pb = tab.items[9 * it];
  const int32_t *n27 = a.nbr27 + 27 * pb;
  const int32_t *fin = tab.finer + (size_t)it * 27 * 8;
  const int par[3] = {a.index[3 * pb] & 1, a.index[3 * pb + 1] & 1, a.index[3 * pb + 2] & 1};
  bool has_coarse = false;
  for (int i = 0; i < 27; ++i) has_coarse = has_coarse || n27[i] >= kNbrCoarser;
  // A. centre, same-level neighbours (SameLevelExchange), finer neighbours (FineToCoarseExchange)
  for (int e = t; e < 1000; e += 256) {
    const int l[3] = {e % 10 - 1, (e / 10) % 10 - 1, e / 100 - 1};
    int code[3], loc[3], fl[3], q = 0;
    for (int d = 0; d < 3; ++d) {
      code[d] = l[d] < 0 ? -1 : (l[d] > 7 ? 1 : 0);
      loc[d] = l[d] - 8 * code[d];
      fl[d] = code[d] < 0 ? 6 : (code[d] > 0 ? 0 : (2 * l[d]) & 7);
      if (code[d] == 0 && l[d] >= 4) q |= 1 << d;
    }
    const int icode = (code[0] + 1) + 3 * (code[1] + 1) + 9 * (code[2] + 1);
    const int n = n27[icode];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double v = 0.0;
      if (n >= 0 && n < kNbrCoarser) v = src[((size_t)n * NC + c) * 512 + (loc[2] * 8 + loc[1]) * 8 + loc[0]];
      else if (n == kNbrFiner) v = avg_block(src + ((size_t)fin[icode * 8 + q] * NC + c) * 512, fl[0], fl[1], fl[2]);
      lab[c * 1000 + e] = v;
    }
  }
  __syncthreads();
  if (has_coarse) {
    // B. coarse shadow tile: own block averaged down (post_load 3750-3778), coarser leaves copied (CoarseFineExchange),
    //    same-level neighbours averaged down (FillCoarseVersion)
    for (int e = t; e < 512; e += 256) {
      const int P[3] = {e % 8 - 2, (e / 8) % 8 - 2, e / 64 - 2};
      int code[3];
      for (int d = 0; d < 3; ++d) code[d] = P[d] < 0 ? -1 : (P[d] > 3 ? 1 : 0);
      const int icode = (code[0] + 1) + 3 * (code[1] + 1) + 9 * (code[2] + 1);
      const int n = n27[icode];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        double v = 0.0;
        if (icode == 13) {
          const double *L = lab + c * 1000;
          double w[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) w[q] = L[lix10(2 * P[0] + (q & 1), 2 * P[1] + ((q >> 1) & 1), 2 * P[2] + (q >> 2))];  // x fastest here
          v = avg_down8(w);
        } else if (n >= kNbrCoarser) {
          v = src[((size_t)(n - kNbrCoarser) * NC + c) * 512 + ((par[2] * 4 + P[2] + 8) & 7) * 64 + ((par[1] * 4 + P[1] + 8) & 7) * 8 + ((par[0] * 4 + P[0] + 8) & 7)];
        } else if (n >= 0) {
          v = avg_block(src + ((size_t)n * NC + c) * 512, 2 * P[0] - 8 * code[0], 2 * P[1] - 8 * code[1], 2 * P[2] - 8 * code[2]);
        }
        Ct[c * 512 + e] = v;
      }
    }
    __syncthreads();
    // C. domain faces on the coarse tile (_apply_bc(info, t, true), 3781): both ghost layers behind the face, every
    //    transverse position, from the face cell; order x-,x+,y-,y+,z-,z+
    for (int f = 0; f < 6; ++f) {
      const int n = a.nbr[pb * 6 + f];
      if (n >= 0) continue;
      const int d = f >> 1, side = f & 1, d1 = (d + 1) % 3, d2 = (d + 2) % 3;
      if (t < 128) {
        int p[3], q[3];
        p[d] = side ? 4 + (t >> 6) : -1 - (t >> 6);
        q[d] = side ? 3 : 0;
        p[d1] = q[d1] = (t & 7) - 2;
        p[d2] = q[d2] = ((t >> 3) & 7) - 2;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          double v = Ct[c * 512 + cix8(q[0], q[1], q[2])];
          if (NC == 3 && (n == -3 || c == d)) v = -v;
          Ct[c * 512 + cix8(p[0], p[1], p[2])] = v;
        }
      }
      __syncthreads();
    }
    // D. CoarseFineInterpolation (4236-4614) of the ghosts behind coarser neighbours
    for (int e = t; e < 1000; e += 256) {
      const int l[3] = {e % 10 - 1, (e / 10) % 10 - 1, e / 100 - 1};
      int code[3], X[3], bit[3], ncode = 0;
      for (int d = 0; d < 3; ++d) {
        code[d] = l[d] < 0 ? -1 : (l[d] > 7 ? 1 : 0);
        ncode += code[d] != 0;
        X[d] = code[d] < 0 ? -1 : (code[d] > 0 ? 4 : l[d] >> 1);
        bit[d] = code[d] < 0 ? 1 : (code[d] > 0 ? 0 : l[d] & 1);
      }
      if (ncode == 0 || n27[(code[0] + 1) + 3 * (code[1] + 1) + 9 * (code[2] + 1)] < kNbrCoarser) continue;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const double *C0 = Ct + c * 512;
        double v;
        if (ncode == 1) {  // face: finite-difference mode
          const int ax = code[0] ? 0 : (code[1] ? 1 : 2), ax1 = ax == 0 ? 1 : 0, ax2 = ax == 2 ? 1 : 2;
          const int st1 = ax1 == 0 ? 1 : 8, st2 = ax2 == 1 ? 8 : 64;
          const int p1 = X[ax1], p2 = X[ax2];
          const double av = fd_mode_av(C0 + cix8(X[0], X[1], X[2]), p1, p2, st1, st2, bit[ax1], bit[ax2]);
          int cb[3] = {l[0], l[1], l[2]}, cc[3] = {l[0], l[1], l[2]};
          cb[ax] = code[ax] > 0 ? 7 : 0;
          cc[ax] = code[ax] > 0 ? 6 : 1;
          const double bv = lab[c * 1000 + lix10(cb[0], cb[1], cb[2])], cv = lab[c * 1000 + lix10(cc[0], cc[1], cc[2])];
          v = fd_mode_blend(av, bv, cv, 0);
        } else {  // edge / corner: TestInterp
          v = test_interp([&](int i, int j, int k) -> double { return C0[cix8(X[0] - 1 + i, X[1] - 1 + j, X[2] - 1 + k)]; }, bit);
        }
        lab[c * 1000 + e] = v;
      }
    }
    __syncthreads();
  }
  // E. domain faces on the fine tile, order x-,x+,y-,y+,z-,z+ (as k_prolong)
  for (int f = 0; f < 6; ++f) {
    const int n = a.nbr[pb * 6 + f];
    if (n >= 0) continue;
    const int d = f >> 1, side = f & 1, ghost = side ? 8 : -1, face = side ? 7 : 0, d1 = (d + 1) % 3, d2 = (d + 2) % 3;
    if (t < 100) {
      int p[3], q[3];
      p[d] = ghost; q[d] = face;
      p[d1] = q[d1] = t % 10 - 1;
      p[d2] = q[d2] = t / 10 - 1;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        double v = lab[c * 1000 + lix10(q[0], q[1], q[2])];
        if (NC == 3 && (n == -3 || c == d)) v = -v;
        lab[c * 1000 + lix10(p[0], p[1], p[2])] = v;
      }
    }
    __syncthreads();
  }
  // F. RefineBlocks, 5493-5565
  for (int k = 0; k < 2; ++k) {
    const int cell = k * 256 + t, x = cell & 7, y = (cell >> 3) & 7, z = cell >> 6;
    const int fs = tab.items[9 * it + 1 + (z >> 2) * 4 + (y >> 2) * 2 + (x >> 2)];
    const int i = 2 * (x & 3), j = 2 * (y & 3), kk = 2 * (z & 3);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double *L = lab + c * 1000;
#define Lb(a_, b_, c_) L[lix10(x + (a_), y + (b_), z + (c_))]
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
      double *b = dst + ((size_t)fs * NC + c) * 512;
#define B(a_, b_, c_) b[((kk + (c_)) * 8 + (j + (b_))) * 8 + (i + (a_))]
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

// ---- compute<ScalarLab>(GradChiOnTmp(sim), sim.chi), main.cpp:8540-8600: the chi-driven half of adaptMesh's tagging input.
// One workgroup per block builds the TENSORIAL [-2,3) tile of chi exactly as BlockLab::load does for that stencil -- same-level
// neighbours copied (all 26 positions, two layers), finer ones averaged down, coarser ones interpolated from the 8^3 coarse shadow
// tile (faces: the finite-difference mode for both layers, 4374-4612; edges and corners: TestInterp, 3883-3906), zero-gradient
// domain faces last (Neumann3D 5929-6004, x-,x+,y-,y+,z-,z+) -- and then applies the operator: the cap of the vorticity on level
// levelMaxVorticity - 1, the ordered scan (z, y, x over the block grown by `offset`) for the first cell with 1e-5 < chi < 0.9, which
// flags the block with 1e10 and ends the scan, and the clearing of interior cells with chi > 0.9 met before it.  The scan is
// evaluated in parallel through its one order dependence: the position of that first cell (an LDS atomicMin of the scan index).
__device__ __forceinline__ int lix12(int x, int y, int z) { return ((z + 2) * 12 + (y + 2)) * 12 + (x + 2); }

__global__ void __launch_bounds__(256) k_grad_chi(AmrDev a, const int32_t *__restrict__ finer_row, const int32_t *__restrict__ finer,
                                                  const int32_t *__restrict__ blevel, int level_max, int lmv, double Rtol, double Ctol,
                                                  const double *__restrict__ src, double *__restrict__ tmpV) {
  __shared__ double lab[1728];
  __shared__ double Ct[512];  // coarse shadow tile, coarse cells [-2,6)^3
  __shared__ int first;
  const int pb = blockIdx.x, t = threadIdx.x;
  const int32_t *n27 = a.nbr27 + 27 * pb;
  const int32_t *fin = finer_row[pb] >= 0 ? finer + (size_t)finer_row[pb] * 216 : nullptr;
  const int par[3] = {a.index[3 * pb] & 1, a.index[3 * pb + 1] & 1, a.index[3 * pb + 2] & 1};
  bool has_coarse = false;
  for (int i = 0; i < 27; ++i) has_coarse = has_coarse || n27[i] >= kNbrCoarser;
  if (t == 0) first = 0x7fffffff;
  // A. centre, same-level neighbours, finer neighbours (averaged down)
  for (int e = t; e < 1728; e += 256) {
    const int l[3] = {e % 12 - 2, (e / 12) % 12 - 2, e / 144 - 2};
    int code[3], loc[3], fl[3], q = 0;
    for (int d = 0; d < 3; ++d) {
      code[d] = l[d] < 0 ? -1 : (l[d] > 7 ? 1 : 0);
      loc[d] = l[d] - 8 * code[d];
      fl[d] = code[d] < 0 ? 8 + 2 * l[d] : (code[d] > 0 ? 2 * (l[d] - 8) : (2 * l[d]) & 7);
      if (code[d] == 0 && l[d] >= 4) q |= 1 << d;
    }
    const int icode = (code[0] + 1) + 3 * (code[1] + 1) + 9 * (code[2] + 1);
    const int n = n27[icode];
    double v = 0.0;
    if (n >= 0 && n < kNbrCoarser) v = src[(size_t)n * 512 + (loc[2] * 8 + loc[1]) * 8 + loc[0]];
    else if (n == kNbrFiner && fin) v = avg_block(src + (size_t)fin[icode * 8 + q] * 512, fl[0], fl[1], fl[2]);
    lab[e] = v;
  }
  __syncthreads();
  if (has_coarse) {
    // B. coarse shadow tile: own block averaged down, coarser leaves copied, same-level neighbours averaged down
    for (int e = t; e < 512; e += 256) {
      const int P[3] = {e % 8 - 2, (e / 8) % 8 - 2, e / 64 - 2};
      int code[3];
      for (int d = 0; d < 3; ++d) code[d] = P[d] < 0 ? -1 : (P[d] > 3 ? 1 : 0);
      const int icode = (code[0] + 1) + 3 * (code[1] + 1) + 9 * (code[2] + 1);
      const int n = n27[icode];
      double v = 0.0;
      if (icode == 13) {
        double w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = lab[lix12(2 * P[0] + (q & 1), 2 * P[1] + ((q >> 1) & 1), 2 * P[2] + (q >> 2))];
        v = avg_down8(w);
      } else if (n >= kNbrCoarser) {
        v = src[(size_t)(n - kNbrCoarser) * 512 + ((par[2] * 4 + P[2] + 8) & 7) * 64 + ((par[1] * 4 + P[1] + 8) & 7) * 8 + ((par[0] * 4 + P[0] + 8) & 7)];
      } else if (n >= 0) {
        v = avg_block(src + (size_t)n * 512, 2 * P[0] - 8 * code[0], 2 * P[1] - 8 * code[1], 2 * P[2] - 8 * code[2]);
      }
      Ct[e] = v;
    }
    __syncthreads();
    // C. zero-gradient domain faces on the coarse tile
    for (int f = 0; f < 6; ++f) {
      const int n = a.nbr[pb * 6 + f];
      if (n >= 0) continue;
      const int d = f >> 1, side = f & 1, d1 = (d + 1) % 3, d2 = (d + 2) % 3;
      if (t < 128) {
        int p[3], q[3];
        p[d] = side ? 4 + (t >> 6) : -1 - (t >> 6);
        q[d] = side ? 3 : 0;
        p[d1] = q[d1] = (t & 7) - 2;
        p[d2] = q[d2] = ((t >> 3) & 7) - 2;
        Ct[cix8(p[0], p[1], p[2])] = Ct[cix8(q[0], q[1], q[2])];
      }
      __syncthreads();
    }
    // D. ghosts behind coarser neighbours
    for (int e = t; e < 1728; e += 256) {
      const int l[3] = {e % 12 - 2, (e / 12) % 12 - 2, e / 144 - 2};
      int code[3], X[3], bit[3], ncode = 0;
      for (int d = 0; d < 3; ++d) {
        code[d] = l[d] < 0 ? -1 : (l[d] > 7 ? 1 : 0);
        ncode += code[d] != 0;
        X[d] = l[d] >> 1;   // the coarse cell that holds this fine cell (-1 behind the low face, 4 behind the high one)
        bit[d] = l[d] & 1;  // which of its two children along d
      }
      if (ncode == 0 || n27[(code[0] + 1) + 3 * (code[1] + 1) + 9 * (code[2] + 1)] < kNbrCoarser) continue;
      double v;
      if (ncode == 1) {  // face: finite-difference mode, both layers
        const int ax = code[0] ? 0 : (code[1] ? 1 : 2), ax1 = ax == 0 ? 1 : 0, ax2 = ax == 2 ? 1 : 2;
        const int st1 = ax1 == 0 ? 1 : 8, st2 = ax2 == 1 ? 8 : 64;
        const int p1 = X[ax1], p2 = X[ax2];
        const double av = fd_mode_av(Ct + cix8(X[0], X[1], X[2]), p1, p2, st1, st2, bit[ax1], bit[ax2]);
        int cb[3] = {l[0], l[1], l[2]}, cc[3] = {l[0], l[1], l[2]};
        cb[ax] = code[ax] > 0 ? 7 : 0;
        cc[ax] = code[ax] > 0 ? 6 : 1;
        const double bv = lab[lix12(cb[0], cb[1], cb[2])], cv = lab[lix12(cc[0], cc[1], cc[2])];
        const int layer = code[ax] < 0 ? -1 - l[ax] : l[ax] - 8;
        v = fd_mode_blend(av, bv, cv, layer);
      } else {  // edge / corner: TestInterp
        v = test_interp([&](int i, int j, int k) -> double { return Ct[cix8(X[0] - 1 + i, X[1] - 1 + j, X[2] - 1 + k)]; }, bit);
      }
      lab[e] = v;
    }
    __syncthreads();
  }
  // E. zero-gradient domain faces on the fine tile: both ghost layers, every transverse position, from the face cell
  for (int f = 0; f < 6; ++f) {
    const int n = a.nbr[pb * 6 + f];
    if (n >= 0) continue;
    const int d = f >> 1, side = f & 1, d1 = (d + 1) % 3, d2 = (d + 2) % 3;
    for (int i = t; i < 288; i += 256) {
      const int layer = i / 144, r = i - 144 * layer;
      int p[3], q[3];
      p[d] = side ? 8 + layer : -1 - layer;
      q[d] = side ? 7 : 0;
      p[d1] = q[d1] = r % 12 - 2;
      p[d2] = q[d2] = r / 12 - 2;
      lab[lix12(p[0], p[1], p[2])] = lab[lix12(q[0], q[1], q[2])];
    }
    __syncthreads();
  }
  // F. the operator
  double *T = tmpV + (size_t)pb * 1536;
  const int level = blevel[pb];
  if (level == lmv - 1 && lmv < level_max) {  // 8546-8557
    for (int c = t; c < 512; c += 256) {
      const double u0 = T[c], u1 = T[512 + c], u2 = T[1024 + c];
      if (sqrt(u0 * u0 + u1 * u1 + u2 * u2) >= Rtol) { T[c] = 0.5 * (Rtol + Ctol); T[512 + c] = 0.0; T[1024 + c] = 0.0; }
    }
  }
  const int off = level == level_max - 1 ? 2 : 1;
  for (int e = t; e < 1728; e += 256) {
    const int x = e % 12 - 2, y = (e / 12) % 12 - 2, z = e / 144 - 2;
    if (x < -off || x >= 8 + off || y < -off || y >= 8 + off || z < -off || z >= 8 + off) continue;
    double v = lab[e];
    v = v < 1.0 ? v : 1.0;
    v = v > 0.0 ? v : 0.0;
    if (v > 0.00001 && v < 0.9) atomicMin(&first, e);  // e grows in the reference's scan order (z, y, x)
  }
  __syncthreads();
  const int stop = first;
  for (int c = t; c < 512; c += 256) {
    const int x = c & 7, y = (c >> 3) & 7, z = c >> 6, e = lix12(x, y, z);
    double v = lab[e];
    v = v < 1.0 ? v : 1.0;
    if (v > 0.9 && e < stop) { T[c] = 0.0; T[512 + c] = 0.0; T[1024 + c] = 0.0; }  // 8592-8597
  }
  __syncthreads();
  if (stop != 0x7fffffff && t < 6) {  // 8567-8590 (six distinct cells among the eight assignments)
    const int cx[6] = {3, 4, 3, 3, 4, 4}, cy[6] = {3, 3, 4, 3, 4, 3}, cz[6] = {3, 3, 3, 4, 4, 4};
    T[(cz[t] * 8 + cy[t]) * 8 + cx[t]] = 1e10;
  }
}

// unchanged blocks: copy; parents of compressed octets: compress (5272-5329) -- pairs[n][2] = dst slot, src slot;
// octets[n][9] = dst slot, eight src slots (child = I + 2J + 4K)
__global__ void __launch_bounds__(256) k_copy_blocks(const int32_t *__restrict__ pairs, const double *__restrict__ src, double *__restrict__ dst, int nc) {
  const int d = pairs[2 * blockIdx.x], s = pairs[2 * blockIdx.x + 1];
  for (int i = threadIdx.x; i < 512 * nc; i += 256) dst[(size_t)d * 512 * nc + i] = src[(size_t)s * 512 * nc + i];
}
__global__ void __launch_bounds__(256) k_compress_blocks(const int32_t *__restrict__ octets, const double *__restrict__ src, double *__restrict__ dst, int nc) {
  const int pb = octets[9 * blockIdx.x], t = threadIdx.x;
  for (int k = 0; k < 2; ++k) {
    const int cell = k * 256 + t, cx = cell & 7, cy = (cell >> 3) & 7, cz = cell >> 6;
    const int fs = octets[9 * blockIdx.x + 1 + (cz >> 2) * 4 + (cy >> 2) * 2 + (cx >> 2)];
    const int i = 2 * (cx & 3), j = 2 * (cy & 3), kk = 2 * (cz & 3);
    for (int c = 0; c < nc; ++c) {
      const double *b = src + ((size_t)fs * nc + c) * 512;
#define B(a_, b_, c_) b[((kk + (c_)) * 8 + (j + (b_))) * 8 + (i + (a_))]
      dst[((size_t)pb * nc + c) * 512 + cell] =
          0.125 * ((B(0, 0, 0) + B(1, 1, 1)) + (B(1, 0, 0) + B(0, 1, 1)) + (B(0, 1, 0) + B(1, 0, 1)) + (B(1, 1, 0) + B(0, 0, 1)));  // 5298-5302
#undef B
    }
  }
}

// ---- host side
int amr_fill_ghosts(Sim *s, const double *field, int nc, int w, double *slabs, int part) {
  AmrDev a{s->d_amr_faces, s->d_amr_fine, s->d_nbr27, s->d_nbr, s->d_index, nc == 1 ? s->scalar_bc_dir : -1};
  ProfileScope ps("amr_ghosts");
  // both lists hold the faces of inner blocks first (sim_build): a rank view produces those while its ghost blocks travel
  const unsigned r0 = part == 2 ? s->n_restrict_inner : 0, r1 = part == 1 ? s->n_restrict_inner : s->n_restrict;
  const unsigned p0 = part == 2 ? s->n_prolong_inner : 0, p1 = part == 1 ? s->n_prolong_inner : s->n_prolong;
  if (r1 + p1 > r0 + p0) {  // one launch: the restricting faces first, then the interpolating ones
    const unsigned nr = r1 - r0, np = p1 - p0;
    if (w == 3) hipLaunchKernelGGL(k_ghosts<3>, dim3(nr + np), dim3(64), 0, stream(), a, (const int32_t *)s->d_restrict_list + r0, nr, (const int32_t *)s->d_prolong_list + p0, field, nc, slabs);
    else hipLaunchKernelGGL(k_ghosts<1>, dim3(nr + np), dim3(64), 0, stream(), a, (const int32_t *)s->d_restrict_list + r0, nr, (const int32_t *)s->d_prolong_list + p0, field, nc, slabs);
  }
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

int amr_flux_fix(Sim *s, int nfc, double *out, int out_nc) {
  AmrDev a{s->d_amr_faces, s->d_amr_fine, s->d_nbr27, s->d_nbr, s->d_index, -1};
  {  // rank views: the fluxes of fine faces owned by other ranks arrive first (FluxCorrectionMPI, main.cpp:2848-2945)
    int rc = view_exchange_flux(s, nfc);
    if (rc) return rc;
  }
  ProfileScope ps("amr_flux_fix");
#ifdef CUP3D_TESTING
  if (debug_option("flux_fix_by_direction")) {  // A/B and the bit-for-bit cross-check: rounds 1-4's three launches, x faces, then y, then z
    for (int d = 0; d < 3; ++d) {
      const unsigned n = (unsigned)s->grid->fix_faces[d].size();
      if (n) hipLaunchKernelGGL(k_flux_fix, dim3(n), dim3(64), 0, stream(), a, s->d_fix_list[d], s->d_flux, nfc, out, out_nc);
    }
    CUP3D_HIP(hipGetLastError());
    return CUP3D_OK;
  }
#endif
  if (s->n_fix_blocks) hipLaunchKernelGGL(k_flux_fix_blocks, dim3(s->n_fix_blocks), dim3(64), 0, stream(), a, (const int32_t *)s->d_fix_blocks, (const double *)s->d_flux, nfc, out, out_nc);
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

}  // namespace cup3d

using namespace cup3d;

namespace {
struct DevInts {
  int32_t *p = nullptr;
  int upload(const std::vector<int32_t> &v) {
    if (v.empty()) return CUP3D_OK;
    CUP3D_HIP(hipMalloc((void **)&p, v.size() * sizeof(int32_t)));
    CUP3D_HIP(hipMemcpyAsync(p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream()));
    return CUP3D_OK;
  }
  ~DevInts() { if (p) hipFree(p); }
};
struct DevBuf {
  void *p = nullptr;
  int alloc(size_t bytes) {
    CUP3D_HIP(hipMalloc(&p, bytes));
    return CUP3D_OK;
  }
  ~DevBuf() { if (p) hipFree(p); }
};
__global__ void __launch_bounds__(256) k_pack_blocks(const double *__restrict__ field, const int32_t *__restrict__ slots, int nc, double *__restrict__ out) {
  const double *src = field + (size_t)slots[blockIdx.x] * nc * 512;
  double *dst = out + (size_t)blockIdx.x * nc * 512;
  for (int i = threadIdx.x; i < nc * 512; i += 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_unpack_blocks(const double *__restrict__ in, const int32_t *__restrict__ slots, int nc, double *__restrict__ field) {
  const double *src = in + (size_t)blockIdx.x * nc * 512;
  double *dst = field + (size_t)slots[blockIdx.x] * nc * 512;
  for (int i = threadIdx.x; i < nc * 512; i += 256) dst[i] = src[i];
}
}  // namespace


// ---- data movement of MeshAdaptation::Adapt, shared by the one-rank and the multi-rank entry points.
// `mo`: the OLD mesh as this rank sees it (a whole mesh, or a tensorial rank view whose ghost slots of `fs` hold the owners' data);
// `blocks`: the blocks to produce, each with the slot of `fd` it goes to.  A block is a leaf of the old mesh (copied), a child of
// one (refine_1 + RefineBlocks, 5227-5249, 5493-5565) or the parent of an octet of them (compress, 5272-5329).
struct NewBlock { int level, idx[3]; int32_t dst; };
static int adapt_produce(const Grid *mo, const std::vector<NewBlock> &blocks, const double *fs, double *fd, int nc) {
  std::vector<int32_t> pairs, octets, items, finer;
  try {
    std::vector<int32_t> item_of((size_t)mo->Z.size(), -1);
    for (const NewBlock &nbk : blocks) {
      const int l = nbk.level;
      const int *idx = nbk.idx;
      const int32_t same = mo->leaf(l, idx);
      if (same >= 0) { pairs.push_back(nbk.dst); pairs.push_back(same); continue; }
      const int pidx[3] = {idx[0] >> 1, idx[1] >> 1, idx[2] >> 1};
      const int32_t par = l > 0 ? mo->leaf(l - 1, pidx) : -1;
      if (par >= 0) {
        if (par >= mo->nblocks()) throw std::invalid_argument("a refined block is not local to the rank that has to refine it");
        if (item_of[par] < 0) {
          item_of[par] = (int32_t)(items.size() / 9);
          items.push_back(par);
          for (int q = 0; q < 8; ++q) items.push_back(-1);
        }
        items[9 * (size_t)item_of[par] + 1 + (idx[0] & 1) + 2 * (idx[1] & 1) + 4 * (idx[2] & 1)] = nbk.dst;
        continue;
      }
      octets.push_back(nbk.dst);
      for (int q = 0; q < 8; ++q) {
        const int ci[3] = {2 * idx[0] + (q & 1), 2 * idx[1] + ((q >> 1) & 1), 2 * idx[2] + (q >> 2)};
        const int32_t cb = l + 1 < mo->level_max ? mo->leaf(l + 1, ci) : -1;
        if (cb < 0) throw std::invalid_argument("a block of the new mesh is neither a block, a child nor the parent of blocks of the old mesh");
        octets.push_back(cb);
      }
    }
    // children of one parent are produced together or not at all on one rank; slots that stay -1 are children another rank's
    // list holds -- impossible, since all eight go where the parent is refined
    for (size_t i = 0; i < items.size(); ++i)
      if (items[i] < 0) throw std::invalid_argument("a refined block lacks some of its children in the new mesh");
    // finer leaves behind every code of the refined parents, by octant of the parent
    finer.assign(items.size() / 9 * 27 * 8, -1);
    for (size_t it = 0; it < items.size() / 9; ++it) {
      const int32_t pb = items[9 * it];
      const int l = mo->blevel[pb];
      for (int icode = 0; icode < 27; ++icode) {
        if (mo->nbr27[27 * (size_t)pb + icode] != kNbrFiner) continue;
        const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, icode / 9 - 1};
        for (int q = 0; q < 8; ++q) {
          int fi[3];
          bool used = true;
          for (int d = 0; d < 3; ++d) {
            const int bit = (q >> d) & 1;
            if (code[d] != 0 && bit) used = false;
            fi[d] = 2 * mo->index[3 * (size_t)pb + d] + (code[d] < 0 ? -1 : (code[d] > 0 ? 2 : bit));
          }
          if (!used) continue;
          const int32_t fl = mo->leaf(l + 1, fi);
          if (fl < 0) throw std::invalid_argument("a finer neighbour of a refined block is not visible on this rank");
          finer[(it * 27 + icode) * 8 + q] = fl;
        }
      }
    }
  } catch (const std::exception &e) {
    set_error("mesh adaptation: %s", e.what());
    return CUP3D_EINVAL;
  }
  DevInts d_pairs, d_octets, d_items, d_finer, d_n27, d_nbr, d_index;
  int rc;
  if ((rc = d_pairs.upload(pairs)) || (rc = d_octets.upload(octets)) || (rc = d_items.upload(items)) || (rc = d_finer.upload(finer))) return rc;
  ProfileScope ps("adapt_transfer");
  if (!pairs.empty()) hipLaunchKernelGGL(k_copy_blocks, dim3((unsigned)(pairs.size() / 2)), dim3(256), 0, stream(), d_pairs.p, fs, fd, nc);
  if (!octets.empty()) hipLaunchKernelGGL(k_compress_blocks, dim3((unsigned)(octets.size() / 9)), dim3(256), 0, stream(), d_octets.p, fs, fd, nc);
  if (!items.empty()) {
    if ((rc = d_n27.upload(mo->nbr27)) || (rc = d_nbr.upload(mo->nbr)) || (rc = d_index.upload(mo->index))) return rc;
    AmrDev a{nullptr, nullptr, d_n27.p, d_nbr.p, d_index.p, -1};
    RefineTab tab{d_items.p, d_finer.p};
    const unsigned n = (unsigned)(items.size() / 9);
    if (nc == 3) hipLaunchKernelGGL(k_refine<3>, dim3(n), dim3(256), 0, stream(), a, tab, fs, fd);
    else hipLaunchKernelGGL(k_refine<1>, dim3(n), dim3(256), 0, stream(), a, tab, fs, fd);
  }
  CUP3D_HIP(hipGetLastError());
  CUP3D_HIP(hipStreamSynchronize(stream()));  // the index tables above are freed on return
  return CUP3D_OK;
}

extern "C" int cup3d_adapt_transfer(cup3d_sim_t *src_h, cup3d_sim_t *dst_h, int field) {
  if (!src_h || !dst_h) return CUP3D_EINVAL;
  Sim *src = reinterpret_cast<Sim *>(src_h), *dst = reinterpret_cast<Sim *>(dst_h);
  int nc, nc2;
  const double *fs = src->field(field, &nc);
  double *fd = dst->field(field, &nc2);
  if (!fs || !fd) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
  if (src->grid->n_local >= 0 || dst->grid->n_local >= 0) { set_error("cup3d_adapt_transfer: meshes spread over ranks go through cup3d_adapt_migrate"); return CUP3D_EINVAL; }
  std::unique_ptr<Grid> mo_tmp, mn_tmp;
  const Grid *mo = src->grid, *mn = dst->grid;
  std::vector<NewBlock> blocks;
  try {
    if (!mo->multilevel) { mo_tmp = mo->as_mesh(); mo = mo_tmp.get(); }
    if (!mn->multilevel) { mn_tmp = mn->as_mesh(); mn = mn_tmp.get(); }
    for (int d = 0; d < 3; ++d)
      if (mo->bpd[d] != mn->bpd[d] || mo->bc[d] != mn->bc[d] || mo->level_max != mn->level_max) throw std::invalid_argument("the two meshes belong to different boxes");
  } catch (const std::exception &e) {
    set_error("cup3d_adapt_transfer: %s", e.what());
    return CUP3D_EINVAL;
  }
  blocks.resize((size_t)mn->nblocks());
  for (int64_t b = 0; b < mn->nblocks(); ++b)
    blocks[b] = NewBlock{mn->blevel[b], {mn->index[3 * b], mn->index[3 * b + 1], mn->index[3 * b + 2]}, (int32_t)b};
  return adapt_produce(mo, blocks, fs, fd, nc);
}

// ---- the same over ranks: MeshAdaptation::Adapt + the block traffic of the LoadBalancer (PrepareCompression 4729-4804, Balance_Diffusion
// 4805-4905, Balance_Global 4906-5021).  The reference refines where the parent lives, gathers an octet on the rank of its base block,
// compresses there, then ships whole blocks to even out the load; the END STATE -- every block of the adapted mesh, filled from the old
// mesh's data, on the rank cup3d_grid_adapted_owners names -- is produced here in one hop: the rank that owns the ORIGIN of a new block
// (the leaf itself, the refined parent, or the base block of the octet) builds it from its tensorial view of the old mesh and sends it
// straight to the new owner.  Refinement and compression are block-local operations on identical inputs, so the fields equal the
// one-rank adaptation bit for bit, wherever the blocks end up.
extern "C" int cup3d_adapt_migrate(const cup3d_grid_t *old_mesh_h, const int32_t *old_owner, cup3d_sim_t *src_h, const cup3d_grid_t *new_mesh_h,
                                   const int32_t *new_owner, cup3d_sim_t *dst_h, int field) {
  if (!src_h || !dst_h) return CUP3D_EINVAL;  // (without the sims there is no communicator to tell the other ranks through)
  Sim *src = reinterpret_cast<Sim *>(src_h), *dst = reinterpret_cast<Sim *>(dst_h);
  const Grid *om = reinterpret_cast<const Grid *>(old_mesh_h), *nm = reinterpret_cast<const Grid *>(new_mesh_h);
  const int me = src->grid->rank, nranks = src->grid->nranks;
  int nc = 0, nc2 = 0;
  const double *fs = nullptr;
  double *fd = nullptr;
  std::unique_ptr<Grid> tv;
  std::vector<NewBlock> mine;               // what this rank produces, ordered by (consumer, new global slot)
  std::vector<int64_t> send_count(nranks, 0), recv_count(nranks, 0);
  std::vector<int32_t> recv_slots;          // local slots of dst in arrival order: by (producer, new global slot)
  size_t per = 0;
  DevBuf F, prod, recv, pack;
  DevInts d_send, d_recv_slots;
  // everything this rank can get wrong on its own -- arguments, the plan, allocations -- happens in here; the ranks then agree on the
  // outcome BEFORE the first exchange, so that a bad call on one rank returns an error on every rank instead of blocking the others
  auto local_part = [&]() -> int {
    if (!old_mesh_h || !old_owner || !new_mesh_h || !new_owner) { set_error("cup3d_adapt_migrate: null argument"); return CUP3D_EINVAL; }
    fs = src->field(field, &nc);
    fd = dst->field(field, &nc2);
    if (!fs || !fd) { set_error("unknown field id %d", field); return CUP3D_EINVAL; }
    if (!om->multilevel || !nm->multilevel || om->n_local >= 0 || nm->n_local >= 0) { set_error("cup3d_adapt_migrate needs the two GLOBAL mesh objects"); return CUP3D_EINVAL; }
    if (dst->grid->rank != me || dst->grid->nranks != nranks) { set_error("cup3d_adapt_migrate: the two sims belong to different ranks"); return CUP3D_EINVAL; }
    try {
      tv = om->rank_view(old_owner, me, nranks, /*tensorial=*/true);
      if (tv->n_local != src->nb) throw std::invalid_argument("the source sim does not hold this rank's blocks of the old mesh");
      std::vector<std::vector<NewBlock>> by_consumer(nranks);
      std::vector<std::vector<int32_t>> by_producer(nranks);
      int32_t my_slot = 0;
      for (int64_t b = 0; b < nm->nblocks(); ++b) {
        const int l = nm->blevel[b];
        const int idx[3] = {nm->index[3 * b], nm->index[3 * b + 1], nm->index[3 * b + 2]};
        int32_t origin = om->leaf(l, idx);
        if (origin < 0 && l > 0) { const int pi[3] = {idx[0] >> 1, idx[1] >> 1, idx[2] >> 1}; origin = om->leaf(l - 1, pi); }
        if (origin < 0 && l + 1 < om->level_max) { const int ci[3] = {2 * idx[0], 2 * idx[1], 2 * idx[2]}; origin = om->leaf(l + 1, ci); }
        if (origin < 0) throw std::invalid_argument("a block of the new mesh is neither a block, a child nor the parent of blocks of the old mesh");
        const int producer = old_owner[origin], consumer = new_owner[b];
        if (producer < 0 || producer >= nranks || consumer < 0 || consumer >= nranks) throw std::invalid_argument("owner out of range");
        if (producer == me) by_consumer[consumer].push_back(NewBlock{l, {idx[0], idx[1], idx[2]}, 0});
        if (consumer == me) by_producer[producer].push_back(my_slot++);
      }
      if (my_slot != dst->nb) throw std::invalid_argument("the destination sim does not hold this rank's blocks of the new mesh");
      for (int p = 0; p < nranks; ++p) {
        send_count[p] = (int64_t)by_consumer[p].size();
        recv_count[p] = (int64_t)by_producer[p].size();
        for (NewBlock &nbk : by_consumer[p]) { nbk.dst = (int32_t)mine.size(); mine.push_back(nbk); }
        recv_slots.insert(recv_slots.end(), by_producer[p].begin(), by_producer[p].end());
      }
    } catch (const std::exception &e) {
      set_error("cup3d_adapt_migrate: %s", e.what());
      return CUP3D_EINVAL;
    }
    per = (size_t)nc * 512;
    const size_t nvis = tv->Z.size();
    int rc;
    if ((rc = F.alloc(nvis * per * sizeof(double))) || (rc = prod.alloc(std::max<size_t>(mine.size(), 1) * per * sizeof(double))) ||
        (rc = recv.alloc(std::max<size_t>(recv_slots.size(), 1) * per * sizeof(double))) ||
        (rc = pack.alloc(std::max<size_t>(tv->send_blocks.size(), 1) * per * sizeof(double))) || (rc = d_send.upload(tv->send_blocks)))
      return rc;
    // the old field on the tensorial view: local blocks (the ghost blocks arrive from their owners below)
    CUP3D_HIP(hipMemcpyAsync(F.p, fs, (size_t)src->nb * per * sizeof(double), hipMemcpyDeviceToDevice, stream()));
    if (!tv->send_blocks.empty())
      hipLaunchKernelGGL(k_pack_blocks, dim3((unsigned)tv->send_blocks.size()), dim3(256), 0, stream(), (const double *)F.p, d_send.p, nc, (double *)pack.p);
    CUP3D_HIP(hipGetLastError());
    return CUP3D_OK;
  };
  int rc = agree(src, local_part(), "cup3d_adapt_migrate");
  if (rc) return rc;
  if ((rc = exchange_items(src, (const double *)pack.p, tv->send_block_count, (double *)F.p + (size_t)tv->n_local * per, tv->recv_block_count, per))) return rc;
  CUP3D_HIP(hipStreamSynchronize(stream()));
  // producing the new blocks is rank-local again (kernels, table uploads): agree once more before they travel
  if ((rc = agree(src, adapt_produce(tv.get(), mine, (const double *)F.p, (double *)prod.p, nc), "cup3d_adapt_migrate (produce)"))) return rc;
  if ((rc = exchange_items(src, (const double *)prod.p, send_count, (double *)recv.p, recv_count, per))) return rc;
  if (!recv_slots.empty()) {
    if ((rc = d_recv_slots.upload(recv_slots))) return rc;
    hipLaunchKernelGGL(k_unpack_blocks, dim3((unsigned)recv_slots.size()), dim3(256), 0, stream(), (const double *)recv.p, d_recv_slots.p, nc, fd);
    CUP3D_HIP(hipGetLastError());
  }
  CUP3D_HIP(hipStreamSynchronize(stream()));
  return CUP3D_OK;
}

// compute<ScalarLab>(GradChiOnTmp(sim), sim.chi) (main.cpp:15182, 8540-8600): edits tmpV (= the vorticity of ComputeVorticity) from
// the resident chi; with cup3d_compute_vorticity before and cup3d_tag_blocks after it, adaptMesh's decision input is complete for
// runs with obstacles.  `mo`: a multi-level mesh object whose first `nloc` slots are the blocks of `tmpV` (the mesh itself on one
// rank, the rank's TENSORIAL view over ranks -- the tensorial chi tile reaches edge / corner neighbours); `chi` lives on mo's slots.
static int grad_chi_run(const Grid *mo, int64_t nloc, const double *chi, double *tmpV, double Rtol, double Ctol, int level_max_vorticity) {
  std::vector<int32_t> finer_row, finer;
  try {
    finer_row.assign((size_t)nloc, -1);
    for (int64_t b = 0; b < nloc; ++b) {
      bool any = false;
      for (int c = 0; c < 27; ++c) any = any || mo->nbr27[27 * (size_t)b + c] == kNbrFiner;
      if (!any) continue;
      finer_row[b] = (int32_t)(finer.size() / 216);
      finer.resize(finer.size() + 216, -1);
      int32_t *row = finer.data() + finer.size() - 216;
      const int l = mo->blevel[b];
      for (int icode = 0; icode < 27; ++icode) {
        if (mo->nbr27[27 * (size_t)b + icode] != kNbrFiner) continue;
        const int code[3] = {icode % 3 - 1, (icode / 3) % 3 - 1, icode / 9 - 1};
        for (int q = 0; q < 8; ++q) {
          int fi[3];
          bool used = true;
          for (int d = 0; d < 3; ++d) {
            const int bit = (q >> d) & 1;
            if (code[d] != 0 && bit) used = false;
            fi[d] = 2 * mo->index[3 * (size_t)b + d] + (code[d] < 0 ? -1 : (code[d] > 0 ? 2 : bit));
          }
          if (used) row[icode * 8 + q] = mo->leaf(l + 1, fi);
        }
      }
    }
  } catch (const std::exception &e) {
    set_error("cup3d_grad_chi_on_tmp: %s", e.what());
    return CUP3D_EINVAL;
  }
  if (finer.empty()) finer.assign(216, -1);
  DevInts d_row, d_finer, d_n27, d_nbr, d_index, d_level;
  int rc;
  if ((rc = d_row.upload(finer_row)) || (rc = d_finer.upload(finer)) || (rc = d_n27.upload(mo->nbr27)) || (rc = d_nbr.upload(mo->nbr)) ||
      (rc = d_index.upload(mo->index)) || (rc = d_level.upload(mo->blevel)))
    return rc;
  AmrDev a{nullptr, nullptr, d_n27.p, d_nbr.p, d_index.p, -1};
  {
    ProfileScope ps("grad_chi_on_tmp");
    hipLaunchKernelGGL(k_grad_chi, dim3((unsigned)nloc), dim3(256), 0, stream(), a, d_row.p, d_finer.p, d_level.p, mo->level_max, level_max_vorticity, Rtol, Ctol, chi,
                       tmpV);
  }
  CUP3D_HIP(hipGetLastError());
  CUP3D_HIP(hipStreamSynchronize(stream()));  // the tables above are freed on return
  return CUP3D_OK;
}

extern "C" int cup3d_grad_chi_on_tmp(cup3d_sim_t *h, double Rtol, double Ctol, int level_max_vorticity) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  if (s->grid->nranks > 1) { set_error("cup3d_grad_chi_on_tmp: a mesh spread over ranks goes through cup3d_grad_chi_on_tmp_over_ranks (its chi tile needs edge / corner ghost blocks)"); return CUP3D_ESTATE; }
  std::unique_ptr<Grid> tmp;
  const Grid *mo = s->grid;
  try {
    if (!mo->multilevel) { tmp = mo->as_mesh(); mo = tmp.get(); }
  } catch (const std::exception &e) {
    set_error("cup3d_grad_chi_on_tmp: %s", e.what());
    return CUP3D_EINVAL;
  }
  return grad_chi_run(mo, mo->nblocks(), s->chi, s->tmpV, Rtol, Ctol, level_max_vorticity);
}

// The same on a mesh spread over ranks: `mesh` / `owner` are the global mesh object and the rank of every leaf (as for
// cup3d_adapt_migrate); chi of the edge / corner / finer neighbours other ranks own arrives first (the plan of the rank's tensorial view),
// then the one-rank kernel runs on the rank's blocks.  Collective: every rank calls it.
extern "C" int cup3d_grad_chi_on_tmp_over_ranks(cup3d_sim_t *h, const cup3d_grid_t *mesh_h, const int32_t *owner, double Rtol, double Ctol, int level_max_vorticity) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  const Grid *gm = reinterpret_cast<const Grid *>(mesh_h);
  const int me = s->grid->rank, nranks = s->grid->nranks;
  std::unique_ptr<Grid> tv;
  DevBuf F, pack;
  DevInts d_send;
  auto local_part = [&]() -> int {  // what one rank can get wrong on its own; the ranks agree on the outcome before anything is exchanged
    if (!mesh_h || !owner) { set_error("cup3d_grad_chi_on_tmp_over_ranks: null argument"); return CUP3D_EINVAL; }
    if (!gm->multilevel || gm->n_local >= 0) { set_error("cup3d_grad_chi_on_tmp_over_ranks needs the GLOBAL mesh object"); return CUP3D_EINVAL; }
    try {
      tv = gm->rank_view(owner, me, nranks, /*tensorial=*/true);
      if (tv->n_local != s->nb) throw std::invalid_argument("the sim does not hold this rank's blocks of the mesh");
    } catch (const std::exception &e) {
      set_error("cup3d_grad_chi_on_tmp_over_ranks: %s", e.what());
      return CUP3D_EINVAL;
    }
    const size_t nvis = tv->Z.size();
    int rc;
    if ((rc = F.alloc(nvis * 512 * sizeof(double))) || (rc = pack.alloc(std::max<size_t>(tv->send_blocks.size(), 1) * 512 * sizeof(double))) ||
        (rc = d_send.upload(tv->send_blocks)))
      return rc;
    CUP3D_HIP(hipMemcpyAsync(F.p, s->chi, (size_t)s->nb * 512 * sizeof(double), hipMemcpyDeviceToDevice, stream()));
    if (!tv->send_blocks.empty())
      hipLaunchKernelGGL(k_pack_blocks, dim3((unsigned)tv->send_blocks.size()), dim3(256), 0, stream(), (const double *)F.p, d_send.p, 1, (double *)pack.p);
    CUP3D_HIP(hipGetLastError());
    return CUP3D_OK;
  };
  int rc = agree(s, local_part(), "cup3d_grad_chi_on_tmp_over_ranks");
  if (rc) return rc;
  if ((rc = exchange_items(s, (const double *)pack.p, tv->send_block_count, (double *)F.p + (size_t)tv->n_local * 512, tv->recv_block_count, 512))) return rc;
  CUP3D_HIP(hipStreamSynchronize(stream()));
  return grad_chi_run(tv.get(), tv->n_local, (const double *)F.p, s->tmpV, Rtol, Ctol, level_max_vorticity);
}

// TEST SUPPORT: the ghost slabs of every interface face for `field` and a w-deep stencil, [(e*nc + c)*w + gl][64]
#ifdef CUP3D_TESTING  // test / tuning support: not in the release library at all
extern "C" int cup3d_debug_amr_slabs(cup3d_sim_t *h, int field, int w, double *out) {
  if (!h || !out || (w != 1 && w != 3)) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  if (!s->grid->multilevel) { set_error("cup3d_debug_amr_slabs: not a multi-level mesh"); return CUP3D_EINVAL; }
  int nc;
  const double *f = s->field(field, &nc);
  if (!f) return CUP3D_EINVAL;
  int rc = amr_fill_ghosts(s, f, nc, w, s->halo_recv);
  if (rc) return rc;
  CUP3D_HIP(hipMemcpyAsync(out, s->halo_recv, (size_t)s->grid->n_amr_faces() * nc * w * 64 * sizeof(double), hipMemcpyDeviceToHost, stream()));
  CUP3D_HIP(hipStreamSynchronize(stream()));
  return CUP3D_OK;
}
#endif
