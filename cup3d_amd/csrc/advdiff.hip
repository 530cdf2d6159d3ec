// Fused advection-diffusion Runge-Kutta stage.
//
// Replaces, per RK stage, compute<VectorLab>(KernelAdvectDiffuse, vel, tmpV)
// (main.cpp:9708 -> 5584-5644 -> BlockLab::load 3623-3743 -> kernel 9484-9549) AND the
// pointwise update loop that follows it (9709-9725) with ONE launch:
//   tmpV' = tmpV + facA*(u.grad)u + facD*lap(u) ;  vel' = vel + tmpV'*alpha/h^3 ;  tmpV = tmpV'*beta
// One 256-thread workgroup per 8^3 block (2 cells per thread).  The ghosted tile is
// assembled in LDS straight from the six face neighbours' slots (or the RCCL halo
// slabs, or the boundary condition) -- only the star-shaped part the stencil reads:
// centre 8^3 + six 3-deep face slabs = 1664 cells x 3 components (the reference copies
// the full 14^3 cube).  Arithmetic keeps the reference's association and is compiled
// with -ffp-contract=off, so the result is bit-identical to the CPU reference.
//
// Algorithmic HBM traffic: 96 B/cell/stage (vel 24 in, tmpV 24 in, vel' 24 out,
// tmpV 24 out); stage 1 skips the tmpV read (tmpV is 0 there).  FP64, no MFMA: pure
// stencil, bandwidth-bound once the ~300 FP64 VALU ops per cell are overlapped.
#include <type_traits>

#include "sim.hpp"
#include "tile.hpp"

namespace cup3d {

constexpr int kXYPitch = 14;                    // x and y extended by the 3-deep ghosts
// LDS bank discipline: a 64-bit LDS access is served 32 lanes at a time over 32 bank pairs.  The compute phase maps a
// half-wave to 8 x by 4 z cells at one y (see k_advdiff), so every stencil read of a half-wave is "base + x + z*kPlane"; with
// kPlane = 8 (mod 32) the four z rows tile the 32 bank pairs exactly, for every shift along x, y or z.  (The first version mapped
// a half-wave to 8 x by 4 y at pitch 14: every read was a 2-way conflict.)  The z ghost planes get pitch 72 for the same reason.
constexpr int kPlane = 14 * 14 + 4;             // 200: one z plane [y 14][x 14] + 4 pad
constexpr int kXYSize = 8 * kPlane;             // centre + x/y ghosts: [z 8][y 14][x 14]
constexpr int kZPitch = 72;                     // pitch of the z ghost planes
constexpr int kZGSize = 6 * kZPitch;            // z ghost planes: [-1,-2,-3, 8,9,10][y][x]
constexpr int kCompStride = kXYSize + kZGSize;  // 2032 doubles per component

constexpr int kAdvProduction = 0;  // 0: k_advdiff (one workgroup per block), 6: k_advdiff_pc (one per block and component); measured, profiles/r03

struct AdvArgs {
  const double *vel;   // [nb][3][512] in
  double *vel_out;     // [nb][3][512] out (double buffer: neighbours still read `vel`)
  double *tmp;         // [nb][3][512] in/out
  const double *halo;  // remote face slabs [(e*3+c)*3+gl][64]
  double dt, nu, u0, u1, u2;
  double alpha, beta;  // Williamson RK3 coefficients, main.cpp:9700-9701
};

// KernelAdvectDiffuse::derivative (main.cpp:9474-9483).  The U<=0 branch is the exact
// negation of the U>0 polynomial on mirrored inputs (round-to-nearest is symmetric
// under negation at every step), so one polynomial and one division serve both.
// n / 60 correctly rounded, in 3 FP64 operations instead of the ~11 of the IEEE division
// expansion.  q0 = RN(n*C) with C = RN(1/60) is within 2 ulp of n/60; r = n - 60*q0 is then
// exactly representable and the FMA delivers it exactly; q0 + r/60 equals n/60, and using
// C for 1/60 perturbs it by < 2^-53 ulp, while n/60 (n a 53-bit float) is never closer than
// ulp/30 to a rounding boundary and never on one.  Hence RN(q0 + r*C) = RN(n/60).  Signed
// zero is restored with copysign; numerators whose residual could be subnormal, and
// non-finite ones, take the IEEE division.  (Checked against n/60. on 2e9 random and 1.5e7
// near-midpoint numerators on the host, and by the bit-exact GPU parity tests.)
template <bool EXACT_TRICK>
__device__ __forceinline__ double div60(double n) {
  if constexpr (!EXACT_TRICK) return n / 60.;
  const double C = 1.0 / 60.0;
  const double q0 = n * C;
  const double r = __builtin_fma(-60.0, q0, n);
  double q = __builtin_copysign(__builtin_fma(r, C, q0), n);
  const double an = __builtin_fabs(n);
  if (__builtin_expect(!(an > 1e-280 && an < 1e300) && n != 0.0, 0)) q = n / 60.;
  return q;
}
template <bool EXACT_TRICK>
__device__ __forceinline__ double upwind5(bool pos, double m3, double m2, double m1, double c, double p1, double p2, double p3) {
  const double a = pos ? m3 : p3, b = pos ? m2 : p2, d = pos ? m1 : p1, e = pos ? p1 : m1, f = pos ? p2 : m2;
  const double q = div60<EXACT_TRICK>(-2 * a + 15 * b - 60 * d + 20 * c + 30 * e - 3 * f);
  return pos ? q : -q;
}

// Decode element e (0..191) of the 3-deep slab behind face f into: the cell of the
// same-level neighbour that holds it, the own face cell used by boundary conditions,
// its LDS position and its position inside a packed halo slab (3 layers x 64).
__device__ __forceinline__ void face_element(int f, int e, int &nb_cell, int &own_cell, int &lds, int &halo) {
  const int d = f >> 1, side = f & 1;
  if (d == 2) {
    const int gl = e >> 6, a = e & 63;
    nb_cell = (side ? gl : 7 - gl) * 64 + a;
    own_cell = (side ? 7 : 0) * 64 + a;
    lds = kXYSize + (side * 3 + gl) * kZPitch + a;
    halo = gl * 64 + a;
  } else if (d == 1) {
    const int z = e / 24, r = e - 24 * z, gr = r >> 3, x = r & 7;
    const int ys = side ? gr : 5 + gr, yg = side ? 8 + gr : gr - 3, gl = side ? gr : 2 - gr;
    nb_cell = z * 64 + ys * 8 + x;
    own_cell = z * 64 + (side ? 7 : 0) * 8 + x;
    lds = z * kPlane + (yg + 3) * kXYPitch + (x + 3);
    halo = gl * 64 + z * 8 + x;
  } else {
    const int row = e / 3, xx = e - 3 * row, z = row >> 3, y = row & 7;
    const int xs = side ? xx : 5 + xx, xg = side ? 8 + xx : xx - 3, gl = side ? xx : 2 - xx;
    nb_cell = row * 8 + xs;
    own_cell = row * 8 + (side ? 7 : 0);
    lds = z * kPlane + (y + 3) * kXYPitch + (xg + 3);
    halo = gl * 64 + z * 8 + y;
  }
}

// CPT = cells per thread (2 -> 256 threads, 1 -> 512 threads).  VAR: 0 production,
// 1 IEEE division instead of div60 (A/B), 4 plain instead of nontemporal stores (A/B), 2/3 timing ablations with WRONG results
// (2: no stencil arithmetic, 3: no ghost staging) -- cup3d_debug_set_option only.
// AMR (multi-level meshes): spacing per block, face fluxes facD*(u_in - u_ghost) of the interface faces into g.flux
// (main.cpp:9550-9637).  Only the blocks with a coarse-side interface face (g.raw) are edited by the flux correction afterwards:
// they leave the raw increment in tmpV, which k_flux_fix corrects before k_rk_update_list applies it; every other block fuses the
// Runge-Kutta update as on uniform grids (a.alpha is the bare Williamson coefficient here, divided by the block's h^3 in the kernel).
// IMPLICIT: KernelAdvect of the implicit-diffusion integrator (main.cpp:9849-10029) on the same tile: tmpV = facD*lap(u) (+ the
// same face fluxes), vel' = vel + facA*(u.grad)u/h^3.  The reference updates vel in place while other blocks still load their
// tiles from it (its result depends on block order / thread timing); here every tile comes from the field on entry and the
// result goes to the second buffer.
template <bool FIRST_STAGE, int CPT, int VAR, bool AMR = false, bool IMPLICIT = false>
__global__ void __launch_bounds__(512 / CPT) k_advdiff(GridDev g, AdvArgs a) {
  constexpr int NT = 512 / CPT, NW = NT / 64;
  __shared__ double tile[3 * kCompStride];  // 48,768 B -> 3 workgroups per CU
  const int slot = block_slot(g);
  if (slot < 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double *__restrict__ own = a.vel + (size_t)slot * 1536;

  // ---- stage the tile: centre (each thread's own cells stay in registers too)
  // cells of this thread: (x, y, z0 + 4k).  A half-wave holds 8 x by 4 z at one y (bank discipline above); the two halves of a
  // wave hold y and y+1, i.e. the two 64 B halves of the same 128 B lines in global memory.
  static_assert(CPT == 2, "the lane -> cell map below is written for 256 threads x 2 cells");
  const int x = lane & 7, z0 = (lane >> 3) & 3, y = 2 * wave + (lane >> 5);
  const int cell0 = z0 * 64 + y * 8 + x;  // second cell: + 256
  double uc[CPT][3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < CPT; ++k) uc[k][c] = own[c * 512 + k * 256 + cell0];
  double told[CPT][3];
  if (!FIRST_STAGE) {
    const double *__restrict__ tp = a.tmp + (size_t)slot * 1536;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < CPT; ++k) told[k][c] = VAR != 4 ? __builtin_nontemporal_load(&tp[c * 512 + k * 256 + cell0]) : tp[c * 512 + k * 256 + cell0];
  }
  // ---- ghosts: 18 (face, component) units of 192 values.  Every global load of the tile
  // (centre, tmpV, ghosts: ~27 per thread) is issued before the first LDS write, so ONE memory
  // latency is exposed per block.  Units are dealt so that the face direction of a unit is a
  // compile-time constant of the unrolled loop (x, y, z faces of components 0/1 in rounds 0-2,
  // component 2 in the remaining rounds): the per-element index arithmetic then reduces to a few
  // adds on per-thread constants computed once.
  constexpr int UPW = (18 + NW - 1) / NW;  // rounds (units per wave)
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  double gv[UPW][3];
  int gl[UPW][3];
  bool gflip[UPW], gon[UPW];
  // per-thread decode of element e = j*64 + lane of a y-face (24 = 3 rows x 8) and of an x-face (3 per row)
  int yb[3], yl[3], yg[3], xb[3], xl[3], xg[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int e = j * 64 + lane;
    const int zy = e / 24, r = e - 24 * zy, gr = r >> 3, xx8 = r & 7;
    yb[j] = zy * 64 + xx8;                    // neighbour / own cell without the y term
    yl[j] = zy * kPlane + gr * kXYPitch + xx8 + 3;  // LDS slot for side 0 (ghost row gr-3)
    yg[j] = gr;
    const int row = e / 3, x3 = e - 3 * row;
    xb[j] = row * 8;
    xl[j] = (row >> 3) * kPlane + ((row & 7) + 3) * kXYPitch + x3;  // LDS slot for side 0 (ghost column x3-3)
    xg[j] = x3;
  }
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    // unit of (wave, round): rounds 0..2 -> direction d = i, side = wave & 1, component wave >> 1 (NW = 4)
    int d, side, c;
    bool on = VAR != 3;
    if (NW == 4) {
      if (i < 3) { d = i; side = wave_s & 1; c = wave_s >> 1; }
      else if (i == 3) { d = wave_s >> 1; side = wave_s & 1; c = 2; }
      else { d = 2; side = wave_s & 1; c = 2; on = on && wave_s < 2; }
    } else {  // NW = 8: rounds 0,1 -> faces 0..5 of components (0,1) spread over 8 waves + remainder
      const int u = wave_s + i * NW;
      on = on && u < 18;
      const int f = u / 3;
      d = f >> 1; side = f & 1; c = u - 3 * f;
    }
    gon[i] = on;
    if (on) {
      const int f = 2 * d + side;
      const int n = g.nbr[slot * 6 + f];
      // domain face: BlockLabBC, main.cpp:6513-6551.  wall (n == -3): every component negated
      // (6384-6394); freespace (n == -1): copy, normal component negated (6137-6153)
      gflip[i] = n < 0 && (n == -3 || c == d);
      const int src = n >= kNbrHalo ? 2 : (n >= 0 ? 1 : 0);  // halo slab / neighbour block / own face cell
      const double *__restrict__ base = src == 2 ? a.halo + ((size_t)(n - kNbrHalo) * 3 + c) * 192
                                        : (src == 1 ? a.vel + (size_t)n * 1536 + c * 512 : own + c * 512);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        int nb_cell, own_cell, lds, hal;
        if (d == 2) {
          nb_cell = (side ? j : 7 - j) * 64 + lane;
          own_cell = (side ? 7 : 0) * 64 + lane;
          lds = kXYSize + (side * 3 + j) * kZPitch + lane;
          hal = j * 64 + lane;
        } else if (d == 1) {
          nb_cell = yb[j] + (side ? yg[j] : 5 + yg[j]) * 8;
          own_cell = yb[j] + (side ? 56 : 0);
          lds = yl[j] + (side ? 11 * kXYPitch : 0);
          hal = (side ? yg[j] : 2 - yg[j]) * 64 + (yb[j] >> 6) * 8 + (yb[j] & 7);
        } else {
          nb_cell = xb[j] + (side ? xg[j] : 5 + xg[j]);
          own_cell = xb[j] + (side ? 7 : 0);
          lds = xl[j] + (side ? 11 : 0);
          hal = (side ? xg[j] : 2 - xg[j]) * 64 + (xb[j] >> 3);
        }
        gv[i][j] = base[src == 2 ? hal : (src == 1 ? nb_cell : own_cell)];  // consumed only after all loads are out
        gl[i][j] = c * kCompStride + lds;
      }
    }
  }
  const int xy = (y + 3) * kXYPitch + (x + 3);
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < CPT; ++k) tile[c * kCompStride + (z0 + 4 * k) * kPlane + xy] = uc[k][c];
#pragma unroll
  for (int i = 0; i < UPW; ++i)
    if (gon[i]) {
#pragma unroll
      for (int j = 0; j < 3; ++j) tile[gl[i][j]] = gflip[i] ? -gv[i][j] : gv[i][j];
    }
  __syncthreads();

  // ---- compute
  const double h = (AMR || IMPLICIT) ? block_h(g, slot) : g.h, h3 = h * h * h;
  const double facA = -a.dt / h * h3 * 1.0;                 // main.cpp:9487 (coef = 1)
  const double facD = (a.nu / h) * (a.dt / h) * h3 * 1.0;   // main.cpp:9488
  if ((AMR || IMPLICIT) && g.flux && t < 192) {
    const int c = t >> 6, a1 = lane & 7, a2 = lane >> 3;
    const double *L = tile + c * kCompStride;
    for (int f = 0; f < 6; ++f) {
      const int n = g.nbr[slot * 6 + f];
      if (n < kNbrHalo) continue;
      const int d = f >> 1, side = f & 1;
      int in, gh;
      if (d == 0) { in = a2 * kPlane + (a1 + 3) * kXYPitch + (side ? 10 : 3); gh = in + (side ? 1 : -1); }
      else if (d == 1) { in = a2 * kPlane + (side ? 10 : 3) * kXYPitch + a1 + 3; gh = in + (side ? kXYPitch : -kXYPitch); }
      else { in = (side ? 7 : 0) * kPlane + (a2 + 3) * kXYPitch + a1 + 3; gh = kXYSize + (side * 3) * kZPitch + lane; }
      g.flux[((size_t)(n - kNbrHalo) * 3 + c) * 64 + lane] = facD * (L[in] - L[gh]);
    }
  }
  double *__restrict__ vout = a.vel_out + (size_t)slot * 1536;
  double *__restrict__ tout = a.tmp + (size_t)slot * 1536;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int z = z0 + 4 * k;
    if (VAR == 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        vout[c * 512 + k * 256 + cell0] = uc[k][c] + tile[c * kCompStride + ((t * 7 + c) % kCompStride)];
        tout[c * 512 + k * 256 + cell0] = FIRST_STAGE ? 0.0 : told[k][c];
      }
      continue;
    }
    int zo[7];
#pragma unroll
    for (int dz = -3; dz <= 3; ++dz) {
      const int zz = z + dz;
      zo[dz + 3] = (zz >= 0 && zz < 8) ? zz * kPlane + xy : kXYSize + (zz < 0 ? -1 - zz : zz - 5) * kZPitch + y * 8 + x;
    }
    const int b = z * kPlane + xy;
    const double ua0 = uc[k][0] + a.u0, ua1 = uc[k][1] + a.u1, ua2 = uc[k][2] + a.u2;  // uAbs, 9492-9494
    const bool p0 = ua0 > 0, p1 = ua1 > 0, p2 = ua2 > 0;
    double res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // VAR 7: every stencil operand one ds_read_b64 with an immediate offset (volatile LDS pointer, address space kept) instead of the
      // compiler's ds_read2_b64 pairs, which the LDS serves at half the rate (MI355X_MICROARCH.md, LDS table: 8 cycles for two values
      // against 2 per ds_read_b64) -- the same change that bought the block CG 2.5 % (poisson.hip, EV bit 2)
      typedef const volatile __attribute__((address_space(3))) double lds_cvd;
      typedef typename std::conditional<VAR == 7, lds_cvd, const double>::type lds_t;
      lds_t *L = (lds_t *)(tile + c * kCompStride);
      const double cc = uc[k][c];
      const double xm1 = L[b - 1], xp1 = L[b + 1], ym1 = L[b - kXYPitch], yp1 = L[b + kXYPitch], zm1 = L[zo[2]], zp1 = L[zo[4]];
      const double dx = upwind5<VAR != 1>(p0, L[b - 3], L[b - 2], xm1, cc, xp1, L[b + 2], L[b + 3]);
      const double dy = upwind5<VAR != 1>(p1, L[b - 3 * kXYPitch], L[b - 2 * kXYPitch], ym1, cc, yp1, L[b + 2 * kXYPitch], L[b + 3 * kXYPitch]);
      const double dz = upwind5<VAR != 1>(p2, L[zo[0]], L[zo[1]], zm1, cc, zp1, L[zo[5]], L[zo[6]]);
      const double sx = xp1 + xm1, sy = yp1 + ym1, sz = zp1 + zm1;
      double lap, adv;  // the three components use three association orders, main.cpp:9531-9545
      if (c == 0) {
        lap = (sx + (sy + sz)) - 6 * cc;
        adv = ua0 * dx + (ua1 * dy + ua2 * dz);
      } else if (c == 1) {
        lap = (sy + (sz + sx)) - 6 * cc;
        adv = ua1 * dy + (ua2 * dz + ua0 * dx);
      } else {
        lap = (sz + (sx + sy)) - 6 * cc;
        adv = ua2 * dz + (ua0 * dx + ua1 * dy);
      }
      if (IMPLICIT) {  // 9936-9941
        tout[c * 512 + k * 256 + cell0] = facD * lap;
        vout[c * 512 + k * 256 + cell0] = cc + facA * adv / h3;
      }
      res[c] = facA * adv + facD * lap;
    }
    if (IMPLICIT) continue;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double tn = (FIRST_STAGE ? 0.0 : told[k][c]) + res[c];  // o += ..., main.cpp:9546-9548
      if (AMR) {
        if (g.raw[slot]) { tout[c * 512 + k * 256 + cell0] = tn; continue; }
        const double ih3 = a.alpha / h3;                      // 9711-9712
        vout[c * 512 + k * 256 + cell0] = uc[k][c] + tn * ih3;  // 9718-9720
        tout[c * 512 + k * 256 + cell0] = tn * a.beta;          // 9721-9723
        continue;
      }
      if (VAR != 4) {  // streaming stores (and tmpV load above): both arrays are next touched a full sweep later (2.73 vs 2.87 ms)
        __builtin_nontemporal_store(uc[k][c] + tn * a.alpha, &vout[c * 512 + k * 256 + cell0]);  // V += tmpV*ih3, 9718-9720
        __builtin_nontemporal_store(tn * a.beta, &tout[c * 512 + k * 256 + cell0]);              // tmpV *= beta, 9721-9723
        continue;
      }
      vout[c * 512 + k * 256 + cell0] = uc[k][c] + tn * a.alpha;  // VAR 4: plain stores (A/B)
      tout[c * 512 + k * 256 + cell0] = tn * a.beta;
    }
  }
}

// ---- the same stage, one velocity component at a time (uniform grids).
// k_advdiff stages the star tile of all three components at once: 48.8 KB of LDS and 110-132 VGPRs (every load of the block in
// flight before the first LDS write) allow three workgroups per CU, and stage 1 -- 25 % less HBM traffic -- is no faster than stages
// 2 and 3: the kernel is bound by how little of a block's load -> LDS -> compute -> store sequence three workgroups can overlap, not
// by HBM.  The three components are independent once a cell's own velocity is known (the derivatives and the Laplacian of u_c read
// the tile of u_c only; u, v, w at the cell come from registers), so this kernel keeps ONE component tile in LDS at a time, in two
// alternating buffers (32.5 KB), with the ghost values of the next component requested while the current one is computed: fewer
// registers, more workgroups per CU.  Arithmetic and association are those of k_advdiff: the results are bit-identical.
template <bool FIRST_STAGE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) k_advdiff_c(GridDev g, AdvArgs a) {
  __shared__ double tile[2][kCompStride];
  const int slot = block_slot(g);
  if (slot < 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double *__restrict__ own = a.vel + (size_t)slot * 1536;
  const int x = lane & 7, z0 = (lane >> 3) & 3, y = 2 * wave + (lane >> 5);
  const int cell0 = z0 * 64 + y * 8 + x;  // second cell: + 256
  double uc[2][3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < 2; ++k) uc[k][c] = own[c * 512 + k * 256 + cell0];
  const double *__restrict__ tp = a.tmp + (size_t)slot * 1536;
  double *__restrict__ vout = a.vel_out + (size_t)slot * 1536;
  double *__restrict__ tout = a.tmp + (size_t)slot * 1536;
  // the ghost elements this thread stages, the same for every component: element e = t + 256 j of the 6 x 192 behind the faces
  const double *gsrc[5];
  int gstride[5], glds[5], gdir[5];
  bool gon[5], gwall[5], gbc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int e = t + 256 * j;
    gon[j] = e < 1152;
    const int f = gon[j] ? e / 192 : 0, idx = gon[j] ? e - 192 * f : 0;
    int nb_cell, own_cell, lds, hal;
    face_element(f, idx, nb_cell, own_cell, lds, hal);
    const int n = g.nbr[slot * 6 + f];
    glds[j] = lds;
    gdir[j] = f >> 1;
    gbc[j] = n < 0;
    gwall[j] = n == -3;  // wall: every component negated; freespace: the normal one (main.cpp:6137-6153, 6384-6394)
    if (n >= kNbrHalo) { gsrc[j] = a.halo + (size_t)(n - kNbrHalo) * 3 * 192 + hal; gstride[j] = 192; }
    else if (n >= 0) { gsrc[j] = a.vel + (size_t)n * 1536 + nb_cell; gstride[j] = 512; }
    else { gsrc[j] = own + own_cell; gstride[j] = 512; }
  }
  double gnext[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) gnext[j] = gon[j] ? gsrc[j][0] : 0.0;
  const int xy = (y + 3) * kXYPitch + (x + 3);
  const double h = g.h, h3 = h * h * h;
  const double facA = -a.dt / h * h3 * 1.0;                 // main.cpp:9487 (coef = 1)
  const double facD = (a.nu / h) * (a.dt / h) * h3 * 1.0;   // main.cpp:9488
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double told[2] = {0.0, 0.0};
    if (!FIRST_STAGE) {  // this component's tmpV: requested here, consumed after the stencil below
#pragma unroll
      for (int k = 0; k < 2; ++k) told[k] = __builtin_nontemporal_load(&tp[c * 512 + k * 256 + cell0]);
    }
    double gv[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) gv[j] = gnext[j];
    if (c < 2) {
#pragma unroll
      for (int j = 0; j < 5; ++j) gnext[j] = gon[j] ? gsrc[j][(size_t)(c + 1) * gstride[j]] : 0.0;
    }
    double *L = tile[c & 1];
#pragma unroll
    for (int k = 0; k < 2; ++k) L[(z0 + 4 * k) * kPlane + xy] = uc[k][c];
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (gon[j]) L[glds[j]] = (gbc[j] && (gwall[j] || gdir[j] == c)) ? -gv[j] : gv[j];
    __syncthreads();  // also: every thread is done computing component c - 1, whose buffer component c + 1 will overwrite
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int z = z0 + 4 * k;
      int zo[7];
#pragma unroll
      for (int dz = -3; dz <= 3; ++dz) {
        const int zz = z + dz;
        zo[dz + 3] = (zz >= 0 && zz < 8) ? zz * kPlane + xy : kXYSize + (zz < 0 ? -1 - zz : zz - 5) * kZPitch + y * 8 + x;
      }
      const int b = z * kPlane + xy;
      const double ua0 = uc[k][0] + a.u0, ua1 = uc[k][1] + a.u1, ua2 = uc[k][2] + a.u2;  // uAbs, 9492-9494
      const bool p0 = ua0 > 0, p1 = ua1 > 0, p2 = ua2 > 0;
      const double cc = uc[k][c];
      const double xm1 = L[b - 1], xp1 = L[b + 1], ym1 = L[b - kXYPitch], yp1 = L[b + kXYPitch], zm1 = L[zo[2]], zp1 = L[zo[4]];
      const double dx = upwind5<true>(p0, L[b - 3], L[b - 2], xm1, cc, xp1, L[b + 2], L[b + 3]);
      const double dy = upwind5<true>(p1, L[b - 3 * kXYPitch], L[b - 2 * kXYPitch], ym1, cc, yp1, L[b + 2 * kXYPitch], L[b + 3 * kXYPitch]);
      const double dz = upwind5<true>(p2, L[zo[0]], L[zo[1]], zm1, cc, zp1, L[zo[5]], L[zo[6]]);
      const double sx = xp1 + xm1, sy = yp1 + ym1, sz = zp1 + zm1;
      double lap, adv;  // the three components use three association orders, main.cpp:9531-9545
      if (c == 0) {
        lap = (sx + (sy + sz)) - 6 * cc;
        adv = ua0 * dx + (ua1 * dy + ua2 * dz);
      } else if (c == 1) {
        lap = (sy + (sz + sx)) - 6 * cc;
        adv = ua1 * dy + (ua2 * dz + ua0 * dx);
      } else {
        lap = (sz + (sx + sy)) - 6 * cc;
        adv = ua2 * dz + (ua0 * dx + ua1 * dy);
      }
      const double tn = told[k] + (facA * adv + facD * lap);                                     // o += ..., main.cpp:9546-9548
      __builtin_nontemporal_store(cc + tn * a.alpha, &vout[c * 512 + k * 256 + cell0]);          // V += tmpV*ih3, 9718-9720
      __builtin_nontemporal_store(tn * a.beta, &tout[c * 512 + k * 256 + cell0]);                // tmpV *= beta, 9721-9723
    }
  }
}

// ---- the same stage with one WORKGROUP per (block, component) (uniform grids).
// The components are independent once a cell's own velocity is known (see k_advdiff_c), so three workgroups can share a block: each
// stages the star tile of ITS component only (16.3 KB of LDS instead of 48.8), reads the block's three centre values per cell (the
// advecting velocity; the other two workgroups' reads of the same lines hit the XCD's L2: the three workgroups of a block are
// consecutive workgroups of one XCD) and writes its component of vel' and tmpV.  No serialisation of the components as in
// k_advdiff_c, three times the workgroups, a third of the LDS and fewer registers each: more wavefronts per SIMD to overlap one
// workgroup's load -> LDS -> compute -> store sequence with the others'.  Arithmetic and association are those of k_advdiff:
// bit-identical results.
template <bool FIRST_STAGE>
__global__ void __launch_bounds__(256) k_advdiff_pc(GridDev g, AdvArgs a) {
  __shared__ double tile[kCompStride];
  const int bid = blockIdx.x, jj = bid >> 3, c = jj % 3, bi = (bid & 7) * g.chunk + jj / 3;
  if (bi >= g.nblocks) return;
  const int slot = g.list ? g.list[bi] : bi;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double *__restrict__ own = a.vel + (size_t)slot * 1536;
  const int x = lane & 7, z0 = (lane >> 3) & 3, y = 2 * wave + (lane >> 5);
  const int cell0 = z0 * 64 + y * 8 + x;  // second cell: + 256
  double uc[2][3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int k = 0; k < 2; ++k) uc[k][q] = own[q * 512 + k * 256 + cell0];
  double told[2] = {0.0, 0.0};
  if (!FIRST_STAGE) {
    const double *__restrict__ tp = a.tmp + (size_t)slot * 1536 + c * 512;
#pragma unroll
    for (int k = 0; k < 2; ++k) told[k] = __builtin_nontemporal_load(&tp[k * 256 + cell0]);
  }
  // ghosts of component c: element e = t + 256 j of the 6 x 192 behind the faces; every load is out before the first LDS write
  double gv[5];
  int glds[5];
  bool gon[5], gneg[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int e = t + 256 * j;
    gon[j] = e < 1152;
    const int f = gon[j] ? e / 192 : 0, idx = gon[j] ? e - 192 * f : 0;
    int nb_cell, own_cell, lds, hal;
    face_element(f, idx, nb_cell, own_cell, lds, hal);
    const int n = g.nbr[slot * 6 + f];
    glds[j] = lds;
    gneg[j] = n < 0 && (n == -3 || (f >> 1) == c);  // wall: every component negated; freespace: the normal one (main.cpp:6137-6153, 6384-6394)
    const double *src = n >= kNbrHalo ? a.halo + ((size_t)(n - kNbrHalo) * 3 + c) * 192 + hal
                                      : (n >= 0 ? a.vel + (size_t)n * 1536 + c * 512 + nb_cell : own + c * 512 + own_cell);
    gv[j] = gon[j] ? src[0] : 0.0;
  }
  const int xy = (y + 3) * kXYPitch + (x + 3);
  double *L = tile;
#pragma unroll
  for (int k = 0; k < 2; ++k) L[(z0 + 4 * k) * kPlane + xy] = c == 0 ? uc[k][0] : (c == 1 ? uc[k][1] : uc[k][2]);
#pragma unroll
  for (int j = 0; j < 5; ++j)
    if (gon[j]) L[glds[j]] = gneg[j] ? -gv[j] : gv[j];
  __syncthreads();
  const double h = g.h, h3 = h * h * h;
  const double facA = -a.dt / h * h3 * 1.0;                 // main.cpp:9487 (coef = 1)
  const double facD = (a.nu / h) * (a.dt / h) * h3 * 1.0;   // main.cpp:9488
  double *__restrict__ vout = a.vel_out + (size_t)slot * 1536 + c * 512;
  double *__restrict__ tout = a.tmp + (size_t)slot * 1536 + c * 512;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int z = z0 + 4 * k;
    int zo[7];
#pragma unroll
    for (int dz = -3; dz <= 3; ++dz) {
      const int zz = z + dz;
      zo[dz + 3] = (zz >= 0 && zz < 8) ? zz * kPlane + xy : kXYSize + (zz < 0 ? -1 - zz : zz - 5) * kZPitch + y * 8 + x;
    }
    const int b = z * kPlane + xy;
    const double ua0 = uc[k][0] + a.u0, ua1 = uc[k][1] + a.u1, ua2 = uc[k][2] + a.u2;  // uAbs, 9492-9494
    const bool p0 = ua0 > 0, p1 = ua1 > 0, p2 = ua2 > 0;
    const double cc = c == 0 ? uc[k][0] : (c == 1 ? uc[k][1] : uc[k][2]);
    const double xm1 = L[b - 1], xp1 = L[b + 1], ym1 = L[b - kXYPitch], yp1 = L[b + kXYPitch], zm1 = L[zo[2]], zp1 = L[zo[4]];
    const double dx = upwind5<true>(p0, L[b - 3], L[b - 2], xm1, cc, xp1, L[b + 2], L[b + 3]);
    const double dy = upwind5<true>(p1, L[b - 3 * kXYPitch], L[b - 2 * kXYPitch], ym1, cc, yp1, L[b + 2 * kXYPitch], L[b + 3 * kXYPitch]);
    const double dz = upwind5<true>(p2, L[zo[0]], L[zo[1]], zm1, cc, zp1, L[zo[5]], L[zo[6]]);
    const double sx = xp1 + xm1, sy = yp1 + ym1, sz = zp1 + zm1;
    double lap, adv;  // the three components use three association orders, main.cpp:9531-9545
    if (c == 0) {
      lap = (sx + (sy + sz)) - 6 * cc;
      adv = ua0 * dx + (ua1 * dy + ua2 * dz);
    } else if (c == 1) {
      lap = (sy + (sz + sx)) - 6 * cc;
      adv = ua1 * dy + (ua2 * dz + ua0 * dx);
    } else {
      lap = (sz + (sx + sy)) - 6 * cc;
      adv = ua2 * dz + (ua0 * dx + ua1 * dy);
    }
    const double tn = told[k] + (facA * adv + facD * lap);                               // o += ..., main.cpp:9546-9548
    __builtin_nontemporal_store(cc + tn * a.alpha, &vout[k * 256 + cell0]);              // V += tmpV*ih3, 9718-9720
    __builtin_nontemporal_store(tn * a.beta, &tout[k * 256 + cell0]);                    // tmpV *= beta, 9721-9723
  }
}

// face slabs of `field` behind the faces listed in send_faces -> packed send buffer
// [(s*nc + c)*w + gl][64]  (the device-side `pack`, main.cpp:1128-1157)
__global__ void __launch_bounds__(64) k_pack_faces(const double *__restrict__ field, const int32_t *__restrict__ send_faces, int nc, int w,
                                                   double *__restrict__ out) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int sf = send_faces[s], slot = sf / 6, f = sf - 6 * slot, d = f >> 1, side = f & 1;
  // the receiver sees this slab through ITS face f^1: ghost layer gl is my layer `gl` counted from face f
  const int a1 = lane & 7, a2 = lane >> 3;
  for (int c = 0; c < nc; ++c)
    for (int gl = 0; gl < w; ++gl) {
      const int q = side ? 7 - gl : gl;
      int cell;
      if (d == 2) cell = q * 64 + a2 * 8 + a1;        // (a1,a2) = (x,y)
      else if (d == 1) cell = a2 * 64 + q * 8 + a1;   // (x,z)
      else cell = a2 * 64 + a1 * 8 + q;               // (y,z)
      out[(((size_t)s * nc + c) * w + gl) * 64 + lane] = field[((size_t)slot * nc + c) * 512 + cell];
    }
}

// multi-level meshes: V += tmpV*alpha/h^3 ; tmpV *= beta (main.cpp:9709-9725) after the flux correction of tmpV, for the blocks the
// stage kernel left raw (one workgroup per listed block)
__global__ void __launch_bounds__(256) k_rk_update_list(const int32_t *__restrict__ list, const double *__restrict__ hb, const double *__restrict__ vel,
                                                        double *__restrict__ tmp, double *__restrict__ vel_out, double alpha, double beta) {
  const int slot = list[blockIdx.x];
  const double h = hb[slot], ih3 = alpha / (h * h * h);
  for (int i = threadIdx.x; i < 1536; i += 256) {
    const size_t o = (size_t)slot * 1536 + i;
    const double tn = tmp[o];
    vel_out[o] = vel[o] + tn * ih3;
    tmp[o] = tn * beta;
  }
}

int launch_pack(Sim *src, const double *field, int nc, int w, hipStream_t st) {
  const unsigned nsend = (unsigned)src->grid->send_faces.size();
  if (!nsend) return CUP3D_OK;
  hipLaunchKernelGGL(k_pack_faces, dim3(nsend), dim3(64), 0, st, field, src->d_send_faces, nc, w, src->halo_send);
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

int launch_advect_implicit(Sim *s, double dt, double nu, const double uinf[3]) {
  int rc = halo_begin(s, s->vel, 3, 3);
  if (rc) return rc;
  AdvArgs a;
  a.vel = s->vel; a.vel_out = s->vel2; a.tmp = s->tmpV; a.halo = s->halo_recv;
  a.dt = dt; a.nu = nu; a.u0 = uinf[0]; a.u1 = uinf[1]; a.u2 = uinf[2];
  a.alpha = a.beta = 0;
  const bool split = s->grid->nranks > 1;
  for (int pass = 0; pass < (split ? 2 : 1); ++pass) {
    GridDev g = split ? s->gdev(pass == 1, pass == 0) : s->gdev();
    if (pass == 1 && (rc = halo_finish(s))) return rc;
    if (g.nblocks == 0) continue;
    ProfileScope ps("advect_implicit");
    hipLaunchKernelGGL((k_advdiff<true, 2, 0, false, true>), dim3(launch_groups(g)), dim3(256), 0, stream(), g, a);
  }
  CUP3D_HIP(hipGetLastError());
  if (s->grid->multilevel && (rc = amr_flux_fix(s, 3, s->tmpV, 3))) return rc;  // compute<VectorLab>(.., vel, tmpV), 10038
  std::swap(s->vel, s->vel2);
  return CUP3D_OK;
}

static int advdiff_stage(Sim *s, int rk, double dt, double nu, const double uinf[3]) {
  const double alpha[3] = {1.0 / 3.0, 15.0 / 16.0, 8.0 / 15.0};   // main.cpp:9700
  const double beta[3] = {-5.0 / 9.0, -153.0 / 128.0, 0.0};       // main.cpp:9701
  const double h = s->grid->h;
  {
    // grid->sync(stencil{-3..4}) (main.cpp:5589-5590) overlapped with the inner blocks (5598-5602)
    const bool split = s->grid->nranks > 1;
    int rc = halo_begin(s, s->vel, 3, 3);
    if (rc) return rc;
    AdvArgs a;
    a.vel = s->vel;
    a.vel_out = s->vel2;
    a.tmp = s->tmpV;
    a.halo = s->halo_recv;
    a.dt = dt; a.nu = nu; a.u0 = uinf[0]; a.u1 = uinf[1]; a.u2 = uinf[2];
    a.alpha = alpha[rk] / (h * h * h);  // ih3, main.cpp:9711-9712
    a.beta = beta[rk];
    if (s->grid->multilevel) {
      a.alpha = alpha[rk];  // per-block h: divided in the kernel
      for (int pass = 0; pass < (split ? 2 : 1); ++pass) {  // rank views: inner blocks while the ghost blocks travel, then the rest
        GridDev g = split ? s->gdev(pass == 1, pass == 0) : s->gdev();
        if (pass == 1 && (rc = halo_finish(s))) return rc;
        if (g.nblocks == 0) continue;
        const dim3 G(launch_groups(g));
        ProfileScope ps("advdiff_stage");
        if (rk == 0) hipLaunchKernelGGL((k_advdiff<true, 2, 0, true>), G, dim3(256), 0, stream(), g, a);
        else hipLaunchKernelGGL((k_advdiff<false, 2, 0, true>), G, dim3(256), 0, stream(), g, a);
      }
      CUP3D_HIP(hipGetLastError());
      if ((rc = amr_flux_fix(s, 3, s->tmpV, 3))) return rc;  // compute(..., vel, tmpV) corrector, main.cpp:9708
      ProfileScope ps("advdiff_update");
      if (s->n_raw) hipLaunchKernelGGL(k_rk_update_list, dim3(s->n_raw), dim3(256), 0, stream(), s->d_raw_list, s->d_hb, s->vel, s->tmpV, s->vel2, alpha[rk], beta[rk]);
      CUP3D_HIP(hipGetLastError());
      std::swap(s->vel, s->vel2);
      return CUP3D_OK;
    }
    for (int pass = 0; pass < (split ? 2 : 1); ++pass) {
      GridDev g = split ? s->gdev(pass == 1, pass == 0) : s->gdev();
      if (pass == 1 && (rc = halo_finish(s))) return rc;
      if (g.nblocks == 0) continue;
      ProfileScope ps(rk == 0 ? "advdiff_stage1" : "advdiff_stage");  // stage 1 reads no tmpV: 72 instead of 96 B/cell
      const dim3 G(launch_groups(g));
#define ADV(FIRST, CPT, VAR) hipLaunchKernelGGL((k_advdiff<FIRST, CPT, VAR>), G, dim3(512 / CPT), 0, stream(), g, a)
#define ADV2(CPT, VAR) do { if (rk == 0) ADV(true, CPT, VAR); else ADV(false, CPT, VAR); } while (0)
#ifndef CUP3D_TESTING
      if constexpr (kAdvProduction == 6) {
        if (rk == 0) hipLaunchKernelGGL((k_advdiff_pc<true>), dim3(3 * launch_groups(g)), dim3(256), 0, stream(), g, a);
        else hipLaunchKernelGGL((k_advdiff_pc<false>), dim3(3 * launch_groups(g)), dim3(256), 0, stream(), g, a);
      } else ADV2(2, 0);  // release build: the production kernel only
#else
      switch (debug_option("advdiff_variant") ? debug_option("advdiff_variant") % 16 : kAdvProduction) {  // 0 / 16 = k_advdiff; 1 and 4 are A/B variants with the SAME results
        case 0: ADV2(2, 0); break;
        case 1: ADV2(2, 1); break;
        case 4: ADV2(2, 4); break;
        case 7: ADV2(2, 7); break;   // single-width LDS reads (A/B; the same results)
        case 6:  // one workgroup per (block, component) (k_advdiff_pc)
          if (rk == 0) hipLaunchKernelGGL((k_advdiff_pc<true>), dim3(3 * launch_groups(g)), dim3(256), 0, stream(), g, a);
          else hipLaunchKernelGGL((k_advdiff_pc<false>), dim3(3 * launch_groups(g)), dim3(256), 0, stream(), g, a);
          break;
        case 5:  // one component tile at a time (k_advdiff_c)
          if (rk == 0) hipLaunchKernelGGL((k_advdiff_c<true>), G, dim3(256), 0, stream(), g, a);
          else hipLaunchKernelGGL((k_advdiff_c<false>), G, dim3(256), 0, stream(), g, a);
          break;
#ifdef CUP3D_TUNING_ABLATIONS  // timing ablations with deliberately WRONG results: never in a release build (make TUNING=1)
        case 2: ADV2(2, 2); break;
        case 3: ADV2(2, 3); break;
#endif
        default: set_error("unknown advdiff_variant (the ablation variants 2 and 3 need a build with -DCUP3D_TUNING_ABLATIONS)"); return CUP3D_EINVAL;
      }
#endif
#undef ADV2
#undef ADV
    }
    CUP3D_HIP(hipGetLastError());
    std::swap(s->vel, s->vel2);
  }
  return CUP3D_OK;
}

}  // namespace cup3d

using namespace cup3d;

extern "C" int cup3d_advect_diffuse(cup3d_sim_t *hs, double dt, double nu, const double uinf[3]) {
  if (!hs || !uinf) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(hs);
  for (int rk = 0; rk < 3; ++rk) {
    int rc = advdiff_stage(s, rk, dt, nu, uinf);
    if (rc) return rc;
  }
  return CUP3D_OK;
}
// TEST SUPPORT: a single RK stage (tmpV must be 0 before stage 0), for the virtual-rank tests
#ifdef CUP3D_TESTING  // test / tuning support: not in the release library at all
extern "C" int cup3d_debug_advdiff_stage(cup3d_sim_t *hs, int rk, double dt, double nu, const double uinf[3]) {
  if (!hs || !uinf || rk < 0 || rk > 2) return CUP3D_EINVAL;
  return advdiff_stage(reinterpret_cast<Sim *>(hs), rk, dt, nu, uinf);
}
#endif
