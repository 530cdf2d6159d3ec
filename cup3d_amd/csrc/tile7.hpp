// The ghosted 10^3 tile of a scalar block field in LDS (stencil [-1,2)), shared by the 7-point kernels (stencil.hip) and the
// multigrid preconditioner (multigrid.hip).
#pragma once
#include "sim.hpp"
#include "tile.hpp"

namespace cup3d {

// LDS bank discipline (as in advdiff.hip): a half-wave computes 8 x by 4 z cells at one y, and the z-plane stride is
// 104 = 8 (mod 32), so the four rows of a half-wave tile the 32 bank pairs of a 64-bit LDS access for every stencil shift.
constexpr int kTP = 104;       // z-plane stride: [10][10] + 4 pad
constexpr int kT = 10 * kTP;   // one component: [10 planes][10][10]
__device__ __forceinline__ int tix(int x, int y, int z) { return (z + 1) * kTP + (y + 1) * 10 + (x + 1); }
// the two cells of thread t: (x, y, z0) and (x, y, z0 + 4); cell0 = z0*64 + y*8 + x is the first one's index in the block
__device__ __forceinline__ void thread_cells(int t, int &x, int &y, int &z0, int &cell0) {
  const int lane = t & 63;
  x = lane & 7;
  z0 = (lane >> 3) & 3;
  y = 2 * (t >> 6) + (lane >> 5);
  cell0 = z0 * 64 + y * 8 + x;
}

// 1-deep face slab element `lane` of face f: neighbour cell, own face cell, LDS slot
__device__ __forceinline__ void face1(int f, int lane, int &nb_cell, int &own_cell, int &lds) {
  const int d = f >> 1, side = f & 1, a1 = lane & 7, a2 = lane >> 3;
  const int qn = side ? 0 : 7, qo = side ? 7 : 0, g = side ? 8 : -1;
  if (d == 2) { nb_cell = qn * 64 + a2 * 8 + a1; own_cell = qo * 64 + a2 * 8 + a1; lds = tix(a1, a2, g); }
  else if (d == 1) { nb_cell = a2 * 64 + qn * 8 + a1; own_cell = a2 * 64 + qo * 8 + a1; lds = tix(a1, g, a2); }
  else { nb_cell = a2 * 64 + a1 * 8 + qn; own_cell = a2 * 64 + a1 * 8 + qo; lds = tix(g, a1, a2); }
}

// scalar tile with zero-gradient domain faces (BlockLabNeumann3D, main.cpp:6561-6581); bc_comp >= 0: the scalar element of
// BlockLabBC<ScalarGrid, .., direction = bc_comp> instead (wall: negated; freespace: negated behind the faces normal to
// bc_comp, main.cpp:6120, 6384-6394) -- the tiles of DiffusionSolver::_lhs (6853-6862)
// SKIPX (timing ablation only, wrong results): the two x faces are not fetched from the neighbours
template <bool SKIPX = false>
__device__ __forceinline__ void load_scalar_tile(const GridDev &g, int slot, const double *__restrict__ f, const double *__restrict__ halo,
                                                 double *tile, double c[2], int bc_comp = -1) {
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const double *own = f + (size_t)slot * 512;
  // all global loads (2 centre cells + up to 2 face elements per thread) are issued before the first LDS write
  int x, y, z0, cell0;
  thread_cells(t, x, y, z0, cell0);
  c[0] = own[cell0];
  c[1] = own[256 + cell0];
  double gv[2];
  int gl[2];
  bool gon[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int face = wave + 4 * i;
    gon[i] = face < 6;
    if (gon[i]) {
      const int n = (SKIPX && face < 2) ? -1 : g.nbr[slot * 6 + face];
      int nb_cell, own_cell, lds;
      face1(face, lane, nb_cell, own_cell, lds);
      const double *__restrict__ base = n >= kNbrHalo ? halo + (size_t)(n - kNbrHalo) * 64 : (n >= 0 ? f + (size_t)n * 512 : own);
      gv[i] = base[n >= kNbrHalo ? lane : (n >= 0 ? nb_cell : own_cell)];
      if (n < 0 && bc_comp >= 0 && (n == -3 || bc_comp == (face >> 1))) gv[i] = -gv[i];
      gl[i] = lds;
    }
  }
  tile[tix(x, y, z0)] = c[0];
  tile[tix(x, y, z0 + 4)] = c[1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (gon[i]) tile[gl[i]] = gv[i];
}

}  // namespace cup3d
