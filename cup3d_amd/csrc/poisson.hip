// Pressure Poisson solve and projection on the device.
//
//   cup3d_preconditioner   <- poisson_kernels::getZImplParallel      main.cpp:14704-14745
//   cup3d_poisson_solve    <- PoissonSolverAMR::solve                main.cpp:14363-14616
//   cup3d_pressure_project <- PressureProjection::operator()         main.cpp:15061-15160
//
// The solver is the reference's pipelined, preconditioned BiCGSTAB, restated step for
// step (same recurrences, same 50-iteration true-residual refresh, same breakdown
// restart, same x_opt bookkeeping, same stopping rule) with the vectors resident in
// HBM as flat [block][512] slabs -- which IS the block layout, so the reference's six
// scatter/gather copies per iteration (9372-9392, 9342-9363) do not exist here.  An
// iteration is two launches (LHS + vector loop + block CG each: k_loop1_cg, k_loop2_cg);
// the scalar recurrences live on the device (SolverCtl) and the host only watches; every
// 50th iteration is four launches of k_refresh; over ranks the all-reduces can start early.
// Elementwise updates keep the reference's association (no FMA contraction); only the
// summation ORDER of the dot products and of the block-CG inner products differs
// from the CPU, so pressure agrees to solver tolerance, not bitwise.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>

#include "sim.hpp"
#include "tile.hpp"
#include "tile7.hpp"

namespace cup3d {

enum { PHAT, RHAT, SHAT, WHAT, ZHAT, QHAT, S_, W_, Z_, T_, V_, Q_, R_, Y_, X_, R0, B_, XOPT, NVEC };

// Everything one wavefront hands to another inside a launch -- per-block values, group sums, counters, totals, flags -- travels by
// AGENT-scope atomic stores / loads / read-modify-writes (sc1 on gfx950: written through and read past the per-XCD L2s, which are not
// coherent with one another), ordered by s_waitcnt alone (a workgroup-scope fence).  NOT by __threadfence(): an agent-scope release fence is
// a write-back of the XCD's whole L2 (buffer_wbl2), there to publish ORDINARY stores that may sit dirty in it -- one per wavefront, 262 144
// per launch, made the loop kernels seven times slower (3.4 instead of 0.5 ms at 256^3, gpurun_out/r05b).  No ordinary store is published here.
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// this lane's agent-scope stores have completed (they are write-through: complete = visible to the agent) before anything that follows
__device__ __forceinline__ void stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }

// ------------------------------------------------------------------ block-local CG
// One wavefront per 8^3 block: lane = (x,y) column, the 8 z-values of r, p, x, Ap in
// registers; z-neighbours come from registers, x/y-neighbours from an LDS copy of p with
// zero rows above and below (the zero Dirichlet halo of the reference's PaddedBlock).
// Two wave reductions per CG iteration (p.Ap and r.r).
// FMA = contract a*b+c where the reference has a separate multiply and add (tuning variant
// only: the production launch keeps the reference's association).
template <bool FMA>
__device__ __forceinline__ double mad(double a, double b, double c) {
  if constexpr (FMA) return __builtin_fma(a, b, c);
  else return a * b + c;
}
// p <- beta p + r written so that the result lands in p's own registers: the compiler turns __builtin_fma(beta, p, r) into the
// two-operand v_fmac_f64 (destination tied to the addend r), which costs a copy of r before and a register rotation after --
// 2 extra moves per cell and iteration in an issue-bound loop.  The three-operand v_fma_f64 has no such tie.
template <bool FMA>
__device__ __forceinline__ double p_update(double beta, double p, double r) {
  if constexpr (!FMA) return beta * p + r;
  double o;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(beta), "v"(p), "v"(r));
  return o;
}
// LDS layout: [z][row = y + 1 (rows 0 and 9 stay zero)][x], pitch 8 doubles and NO x halo.  With the 10x10-pitched tile of the
// first version half of all LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50, and the waves
// spent 35 % of their cycles in SQ_WAIT_INST_LDS: profiles/r01/pmc_block_preconditioner_sq.txt): a 64-bit access is served 32
// lanes at a time, and four 8-wide rows at pitch 10 overlap in banks, while at pitch 8 the four rows tile the 32 bank pairs
// exactly.  The x-1 / x+1 reads of the lanes at x = 0 / 7, which would fetch a cell of the neighbouring row, go to the zero row.
// HELM: diffusion_kernels::getZImplParallel (main.cpp:10534-10579) -- the same block CG with centre coefficient
// -6 - h^2/nu/dt (10570) instead of -6, for the Helmholtz solves of the implicit diffusion.
// V2 (production): the same iteration with three changes that only touch HOW it is evaluated.  The r01 kernel (V2 = false, kept
// for A/B timing and as the reference-association variant) issued 181 VALU instructions per CG iteration per wavefront, 86 of them
// the stencil and the updates; the SIMD's FP64 issue slots (4 cycles per wave64 instruction) AND the CU's LDS pipe (ds_read2_b64 is
// serviced at half the ds_read_b64 rate, MI355X_MICROARCH.md LDS table) were both ~90 % busy, so only fewer instructions help:
//  * the two wave-wide sums go to the otherwise idle FP64 MATRIX pipe: v_mfma_f64_16x16x4_f64 with B = ones sums the lanes
//    {i, i+16, i+32, i+48}; every lane then holds four of the sixteen partial sums, adds them (3 v_add_f64) and a second MFMA
//    leaves the wavefront total in every lane: 3 VALU instructions instead of 12 DPP moves + 6 adds + 2 readlanes + hazard nops
//    per reduction (this is a cross-lane reduction on an idle pipe, not a reformulation of the stencil as a GEMM);
//  * the x/y-neighbour reads are volatile so that the compiler keeps them as 32 ds_read_b64 (2 LDS cycles each) with the z-plane
//    offset in the instruction instead of 16 ds_read2_b64 (8 cycles each) + per-plane address arithmetic;
//  * rr / (a2 + 1e-55) and ss / (rr + 1e-55) use v_rcp_f64 + two Newton steps + one residual correction (8 instructions, result
//    within 1 ulp of the IEEE quotient) instead of the 12-instruction IEEE expansion -- FMA variant only.
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double wave_sum_mfma(double v) {
  const double4_t zero = {0.0, 0.0, 0.0, 0.0};
  double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, zero, 0, 0, 0);  // D[i][j] = sum_k A[i][k]: lanes i, i+16, i+32, i+48
  const double t = (d[0] + d[1]) + (d[2] + d[3]);                              // the four rows of D this lane holds
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(t, 1.0, zero, 0, 0, 0);            // the four 16-lane rows hold disjoint quarters of the rows of D
  return d[0];
}
__device__ __forceinline__ double fast_div(double n, double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  const double q = n * y;
  return __builtin_fma(__builtin_fma(-d, q, n), y, q);
}
// LDS of one block CG: the padded copy of p ([z][10 rows][8], see below) + 4 doubles for the row totals of wave_sum_rows
constexpr int kCgLds = 8 * 80 + 8;
// The wave sum with its last two steps through LDS: after the four in-row DPP steps every lane of a 16-lane row holds the row total;
// the rows park their totals in LDS and every lane adds the four of them -- (R3 + R2) + (R1 + R0), the association of the two
// row_bcast steps of wave_sum, so the result is BIT-IDENTICAL -- then the value goes through the scalar unit like there (the loop
// stays uniform).  15 + 2 vector instructions instead of 24 + 2 hazard nops.  MEASURED SLOWER (the LDS round trip sits on the
// iteration's dependency chain: reduction -> division -> update): the kernel is bound by that chain more than by instruction issue,
// whatever SQ_ACTIVE_INST_VALU suggests (profiles/r02/pmc_fused_kernels_sq_256cubed.txt).  A/B variant, not production.
__device__ __forceinline__ double wave_sum_rows(double v, double *P) {
  typedef volatile __attribute__((address_space(3))) double lds_vd;
  lds_vd *R = (lds_vd *)(P + 8 * 80);
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  R[threadIdx.x >> 4] = v;
  const double r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3];
  const double t = (r3 + r2) + (r1 + r0);
  const long long b = __builtin_bit_cast(long long, t);
  const int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <bool MFMA, bool ROWS = false, bool ZERO_OLD = false>
__device__ __forceinline__ double cg_sum(double v, double *P = nullptr) {
  if constexpr (MFMA) return wave_sum_mfma(v);
  else if constexpr (ROWS) return wave_sum_rows(v, P);
  else if constexpr (ZERO_OLD) return wave_sum_zero_old(v);
  else return wave_sum(v);
}
template <bool FAST>
__device__ __forceinline__ double cg_div(double n, double d) {
  if constexpr (FAST) return fast_div(n, d);
  else return n / d;
}

// EV = how the iteration is evaluated, a bit set: 1 = wave sums on the matrix pipe, 2 = single-width volatile LDS reads,
// 4 = reciprocal divisions (FMA variants only), 8 = three-operand FMA for the p update (p_update above), 16 = the last two steps of
// the wave sums through LDS (wave_sum_rows: bit-identical sums, 168 instead of 182 vector instructions per iteration -- and SLOWER:
// block CG 3.18 instead of 2.88 ms, fused kernels 4.1 instead of 3.8 ms at 512^3; kept for A/B only)
// cg_block: the iteration itself, entered with r = the block's right-hand side / h already in registers (lane = (x, y), 8 z per lane) --
// shared by the stand-alone preconditioner kernel and the kernels that produce that right-hand side on the fly (k_loop1_cg / k_loop2_cg)
// AG: the block sum is handed to another wavefront of the SAME launch (Arrive, below): agent-scope store instead of an ordinary one
template <bool FMA, bool HELM, int EV, bool AG = false>
__device__ __forceinline__ void cg_block(const GridDev &g, int slot, double (&r)[8], double *out, double *__restrict__ block_sums, double nu, double dt,
                                         int *__restrict__ iters_out, double *P) {
  // (r01 kernel: 86 VGPRs -> 5 waves/SIMD.  Forcing 6 with amdgpu_waves_per_eu spills five values that are reloaded every iteration:
  //  0.476 vs 0.431 ms at 256^3, so the natural allocation stays.)
  constexpr bool V2 = (EV & 1) != 0, LDSV = (EV & 2) != 0, FDIV = (EV & 4) != 0 && FMA, ROWS = (EV & 16) != 0, ZOLD = (EV & 64) != 0;
  const int l = threadIdx.x;
  const int base = ((l >> 3) + 1) * 8 + (l & 7);
  for (int i = l; i < 640; i += 64) P[i] = 0.0;
  // x-1 / x+1 reads of the edge lanes are redirected to the zero row of the same plane, at the one bank the other lanes of the
  // half-wave leave free (address 7 for x = 0, address 0 for x = 7): still conflict-free, and no masking arithmetic
  const int am = (l & 7) == 0 ? 7 : base - 1, ap = (l & 7) == 7 ? 0 : base + 1;
  // V2: volatile LDS pointers (address space kept, or the loads become flat): one ds_read_b64 per access, plane offset immediate
  typedef const volatile __attribute__((address_space(3))) double lds_cvd;
  lds_cvd *Pam = (lds_cvd *)(P + am), *Pap = (lds_cvd *)(P + ap), *Pym = (lds_cvd *)(P + base - 8), *Pyp = (lds_cvd *)(P + base + 8);
  double centre = -6.0;
  if constexpr (HELM) { const double hq = block_h(g, slot); centre = -6.0 - hq * hq / nu / dt; }
  double p[8], x[8], Ax[8];
  double rr = 0;
#pragma unroll
  for (int z = 0; z < 8; ++z) {
    rr = mad<FMA>(r[z], r[z], rr);
    p[z] = r[z];
    x[z] = 0;
  }
  rr = cg_sum<V2, ROWS, ZOLD>(rr, P);
  const double kRel = 1e-7 * 1e-7, kAbs = 1e-16 * 1e-16;  // kSqrNorm{Rel,Abs}Criterion, 14619-14624
  const double sqrNorm0 = (double)1 / (512 * 512) * rr;    // 14734
  int kdone = 0;
  if constexpr ((EV & 32) != 0) {
    // EXPERIMENT (testing build, cg_variant 8 + 32): the same CG in the single-reduction form of Chronopoulos & Gear -- the
    // stencil is applied to r instead of p, s = A p follows by recurrence (s <- A r + beta s), and p.Ap by  den <- r.Ar - beta^2 den,
    // so the two inner products of an iteration (r.r and r.Ar) leave ONE dependent reduction point instead of two.  Same Krylov iterates
    // in exact arithmetic, same stopping rule on the same quantity; one stencil application more per block (the last one is wasted),
    // one vector more in registers, 8 more FMAs per iteration.
    if (sqrNorm0 >= 1e-32) {
      double s[8], w[8];
      auto apply = [&](const double (&v)[8], double (&o)[8]) {
        __syncthreads();
#pragma unroll
        for (int z = 0; z < 8; ++z) P[z * 80 + base] = v[z];
        __syncthreads();
#pragma unroll
        for (int z = 0; z < 8; ++z) {
          double t = mad<FMA>(centre, v[z], Pam[z * 80] + Pap[z * 80]);
          t += Pym[z * 80];
          t += Pyp[z * 80];
          if (z > 0) t += v[z - 1];
          if (z < 7) t += v[z + 1];
          o[z] = t;
        }
      };
      apply(r, w);
      double den = 0;
#pragma unroll
      for (int z = 0; z < 8; ++z) { s[z] = w[z]; den = mad<FMA>(r[z], w[z], den); }
      den = cg_sum<false>(den);
      for (int k = 0; k < 100; ++k) {
        kdone = k + 1;
        const double a = cg_div<FDIV>(rr, den + 1e-55);
#pragma unroll
        for (int z = 0; z < 8; ++z) { x[z] = mad<FMA>(a, p[z], x[z]); r[z] = mad<FMA>(-a, s[z], r[z]); }
        apply(r, w);
        double ss = 0, dl = 0;
#pragma unroll
        for (int z = 0; z < 8; ++z) { ss = mad<FMA>(r[z], r[z], ss); dl = mad<FMA>(r[z], w[z], dl); }
        ss = cg_sum<false>(ss);
        dl = cg_sum<false>(dl);
        const double beta = cg_div<FDIV>(ss, rr + 1e-55);
        const double sqrNorm = (double)1 / (512 * 512) * ss;
        if (sqrNorm < kRel * sqrNorm0 || sqrNorm < kAbs) break;
#pragma unroll
        for (int z = 0; z < 8; ++z) { p[z] = __builtin_fma(beta, p[z], r[z]); s[z] = __builtin_fma(beta, s[z], w[z]); }
        den = __builtin_fma(-beta * beta, den, dl);
        rr = ss;
        if (rr <= 0) break;
      }
    }
  } else
  if (sqrNorm0 >= 1e-32) {                                  // else: block stays 0 (14735-14736)
    __syncthreads();
    auto iteration = [&](int k) -> bool {                     // one trip of the loop at 14739; false = leave it
      kdone = k + 1;
#pragma unroll
      for (int z = 0; z < 8; ++z) P[z * 80 + base] = p[z];
      __syncthreads();
      double a2 = 0;
#pragma unroll
      for (int z = 0; z < 8; ++z) {                         // kernelPoissonGetZInner, 14662-14682
        double t;
        if constexpr (LDSV) {
          t = mad<FMA>(centre, p[z], Pam[z * 80] + Pap[z * 80]);
          t += Pym[z * 80];
          t += Pyp[z * 80];
        } else {
          t = mad<FMA>(centre, p[z], P[z * 80 + am] + P[z * 80 + ap]);
          t += P[z * 80 + base - 8];
          t += P[z * 80 + base + 8];
        }
        t += z > 0 ? p[z - 1] : 0.0;
        t += z < 7 ? p[z + 1] : 0.0;
        Ax[z] = t;
        a2 = mad<FMA>(p[z], t, a2);
      }
      __syncthreads();
      a2 = cg_sum<V2, ROWS, ZOLD>(a2, P);
      const double a = cg_div<FDIV>(rr, a2 + 1e-55);        // 14684
      double ss = 0;
#pragma unroll
      for (int z = 0; z < 8; ++z) {
        x[z] = mad<FMA>(a, p[z], x[z]);                     // 14688
        r[z] = mad<FMA>(-a, Ax[z], r[z]);                   // subAndSumSqr, 14636-14638
        ss = mad<FMA>(r[z], r[z], ss);
      }
      ss = cg_sum<V2, ROWS, ZOLD>(ss, P);
      const double beta = cg_div<FDIV>(ss, rr + 1e-55);       // 14690
      const double sqrNorm = (double)1 / (512 * 512) * ss;  // 14691
      if (sqrNorm < kRel * sqrNorm0 || sqrNorm < kAbs) return false;  // 14692-14694 (returns -1)
#pragma unroll
      for (int z = 0; z < 8; ++z) p[z] = (EV & 8) ? p_update<FMA>(beta, p[z], r[z]) : mad<FMA>(beta, p[z], r[z]);   // 14698-14699
      rr = ss;
      if (rr <= 0) return false;                                   // 14741
      return true;
    };
    // (two iterations per trip, to pay the register rotation of p at the back edge -- 8 v_mov_b64 -- every other iteration, costs
    //  101-119 registers instead of 88-94: below 5 wavefronts per SIMD, not kept)
    for (int k = 0; k < 100; ++k)
      if (!iteration(k)) break;
  }
  double sx = 0;
#pragma unroll
  for (int z = 0; z < 8; ++z) {
    out[(size_t)slot * 512 + z * 64 + l] = x[z];
    sx += x[z];
  }
  if (iters_out && l == 0) iters_out[slot] = kdone;  // measurement only (cup3d_profile_enable): CG iterations this block took
  if (block_sums) {  // sum(z*h^3) of this block for the mean constraint of the LHS that follows (9283-9294)
    const double hq = block_h(g, slot), h3 = hq * hq * hq;
    sx = cg_sum<V2, ROWS, ZOLD>(sx * h3, P);
    if (l == 0) { if constexpr (AG) st_agent(block_sums + slot, sx); else block_sums[slot] = sx; }
  }
}

template <bool FMA, bool HELM = false, int EV = 0>
__global__ void __launch_bounds__(64) k_precond(GridDev g, const double *in, double *out, double *__restrict__ block_sums, double nu, double dt,
                                                int *__restrict__ iters_out) {
  __shared__ double P[kCgLds];
  const int slot = block_slot(g);
  if (slot < 0) return;
  const double invh = 1 / block_h(g, slot);  // main.cpp:14723
  double r[8];
#pragma unroll
  for (int z = 0; z < 8; ++z) r[z] = invh * in[(size_t)slot * 512 + z * 64 + threadIdx.x];
  cg_block<FMA, HELM, EV>(g, slot, r, out, block_sums, nu, dt, iters_out, P);
}

#ifdef CUP3D_TESTING
// ------------------------------------------------------------------ block CG, two blocks per wavefront
// The full-wave kernel above spends more than half of its FP64 issue slots on work that does not scale with the cells: two
// wave-wide sums (12 DPP moves + 6 adds + read-lanes + hazard nops each), two divisions, loop control.  Here a HALF-wave owns a
// block -- lane = (x, y pair), 16 cells per lane -- so one instruction stream serves two blocks and that overhead is shared:
//   * sums over 32 lanes: four DPP steps inside the 16-lane rows, then v_permlane16_swap (gfx950) exchanges the two rows of each
//     half, and every lane of a half holds its block's total (no read-lane, no select);
//   * the y-neighbour of row 2j is row 2j+1 of the same lane and vice versa: 3 LDS reads per cell instead of 4;
//   * LDS rows are stored in the order 0,2,4,6,8 | -1,1,3,5,7 (pitch 8, no x halo): the four rows a half-wave touches in any of its
//     six reads / two writes always fall into four different 8-bank groups, and the x-1 / x+1 reads of the edge lanes go to one
//     zero cell at the bank the others leave free -- every DS access is conflict-free and single-width;
//   * a block that has converged (or is skipped, 14735) just stops updating x and r (its half is masked); the wave leaves the loop
//     when both are done.  Block i of the pair runs exactly the iteration the full-wave kernel runs; only the order of the 512-term
//     sums differs (16 per lane, then the lane tree).
__device__ __forceinline__ double half_sum(double v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row total
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // rows 0<->1 and 2<->3
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double a = __builtin_bit_cast(double, ((long long)rh[0] << 32) | (long long)rl[0]);
  const double c = __builtin_bit_cast(double, ((long long)rh[1] << 32) | (long long)rl[1]);
  return a + c;
}

template <bool FMA, bool HELM = false>
__global__ void __launch_bounds__(64) k_precond_pair(GridDev g, int pchunk, const double *in, double *out, double *__restrict__ block_sums, double nu, double dt,
                                                     int *__restrict__ iters_out) {
  __shared__ double P[2 * 8 * 80];
  typedef const volatile __attribute__((address_space(3))) double lds_cvd;
  const int l = threadIdx.x, half = l >> 5, li = l & 31, x = li & 7, yp = li >> 3;
  const int pi = ((int)blockIdx.x & 7) * pchunk + ((int)blockIdx.x >> 3);  // XCD-aware, as block_slot()
  const int bi = 2 * pi + half;
  const bool have = bi < g.nblocks;
  const int slot = have ? (g.list ? g.list[bi] : bi) : 0;
  for (int i = l; i < 1280; i += 64) P[i] = 0.0;
  double *Pb = P + half * 640;
  // row slots: even rows 0,2,4,6,8 -> 0..4, odd rows -1,1,3,5,7 -> 5..9; slots 4 (row 8) and 5 (row -1) stay zero
  const int s0 = yp, s1 = 6 + yp;                       // own rows y0 = 2 yp, y1 = 2 yp + 1
  double *W0 = Pb + s0 * 8 + x, *W1 = Pb + s1 * 8 + x;  // writes
  lds_cvd *Xm0 = (lds_cvd *)(Pb + (x == 0 ? 47 : s0 * 8 + x - 1)), *Xp0 = (lds_cvd *)(Pb + (x == 7 ? 40 : s0 * 8 + x + 1));
  lds_cvd *Xm1 = (lds_cvd *)(Pb + (x == 0 ? 47 : s1 * 8 + x - 1)), *Xp1 = (lds_cvd *)(Pb + (x == 7 ? 40 : s1 * 8 + x + 1));
  lds_cvd *Ym0 = (lds_cvd *)(Pb + (5 + yp) * 8 + x);    // row y0 - 1
  lds_cvd *Yp1 = (lds_cvd *)(Pb + (yp + 1) * 8 + x);    // row y1 + 1
  const double hq = block_h(g, slot), invh = 1 / hq;     // main.cpp:14723
  double centre = -6.0;
  if constexpr (HELM) centre = -6.0 - hq * hq / nu / dt;
  const size_t o0 = (size_t)slot * 512 + (2 * yp) * 8 + x;  // cell (x, y0, z = 0); y1: + 8; z: + 64
  double r0[8], r1[8], p0[8], p1[8], x0[8], x1[8], A0[8], A1[8];
  double rr = 0;
#pragma unroll
  for (int z = 0; z < 8; ++z) {
    r0[z] = have ? invh * in[o0 + z * 64] : 0.0;
    r1[z] = have ? invh * in[o0 + z * 64 + 8] : 0.0;
    rr = mad<FMA>(r0[z], r0[z], rr);
    rr = mad<FMA>(r1[z], r1[z], rr);
    p0[z] = r0[z]; p1[z] = r1[z];
    x0[z] = 0; x1[z] = 0;
  }
  rr = half_sum(rr);
  const double kRel = 1e-7 * 1e-7, kAbs = 1e-16 * 1e-16;  // kSqrNorm{Rel,Abs}Criterion, 14619-14624
  const double sqrNorm0 = (double)1 / (512 * 512) * rr;    // 14734
  bool active = have && sqrNorm0 >= 1e-32;                  // else: block stays 0 (14735-14736)
  int kdone = 0;
  __syncthreads();
  for (int k = 0; k < 100; ++k) {                           // 14739
    if (!__any(active)) break;
    if (active) kdone = k + 1;
#pragma unroll
    for (int z = 0; z < 8; ++z) { W0[z * 80] = p0[z]; W1[z * 80] = p1[z]; }
    __syncthreads();
    double a2 = 0;
#pragma unroll
    for (int z = 0; z < 8; ++z) {                           // kernelPoissonGetZInner, 14662-14682
      double t = mad<FMA>(centre, p0[z], Xm0[z * 80] + Xp0[z * 80]);
      t += Ym0[z * 80];
      t += p1[z];
      t += z > 0 ? p0[z - 1] : 0.0;
      t += z < 7 ? p0[z + 1] : 0.0;
      A0[z] = t;
      a2 = mad<FMA>(p0[z], t, a2);
      double u = mad<FMA>(centre, p1[z], Xm1[z * 80] + Xp1[z * 80]);
      u += p0[z];
      u += Yp1[z * 80];
      u += z > 0 ? p1[z - 1] : 0.0;
      u += z < 7 ? p1[z + 1] : 0.0;
      A1[z] = u;
      a2 = mad<FMA>(p1[z], u, a2);
    }
    __syncthreads();
    a2 = half_sum(a2);
    const double a = cg_div<FMA>(rr, a2 + 1e-55);           // 14684
    double ss = 0;
    if (active) {
#pragma unroll
      for (int z = 0; z < 8; ++z) {
        x0[z] = mad<FMA>(a, p0[z], x0[z]);                  // 14688
        x1[z] = mad<FMA>(a, p1[z], x1[z]);
        r0[z] = mad<FMA>(-a, A0[z], r0[z]);                 // subAndSumSqr, 14636-14638
        r1[z] = mad<FMA>(-a, A1[z], r1[z]);
      }
    }
#pragma unroll
    for (int z = 0; z < 8; ++z) { ss = mad<FMA>(r0[z], r0[z], ss); ss = mad<FMA>(r1[z], r1[z], ss); }
    ss = half_sum(ss);
    const double beta = cg_div<FMA>(ss, rr + 1e-55);        // 14690
    const double sqrNorm = (double)1 / (512 * 512) * ss;    // 14691
    if (sqrNorm < kRel * sqrNorm0 || sqrNorm < kAbs) active = false;  // 14692-14694: this block is done
#pragma unroll
    for (int z = 0; z < 8; ++z) { p0[z] = p_update<FMA>(beta, p0[z], r0[z]); p1[z] = p_update<FMA>(beta, p1[z], r1[z]); }  // 14698-14699
    rr = ss;
    if (rr <= 0) active = false;                            // 14741
  }
  if (!have) return;
  double sx = 0;
#pragma unroll
  for (int z = 0; z < 8; ++z) {
    out[o0 + z * 64] = x0[z];
    out[o0 + z * 64 + 8] = x1[z];
    sx += x0[z];
    sx += x1[z];
  }
  if (iters_out && li == 0) iters_out[slot] = kdone;
  if (block_sums) {  // sum(z*h^3) of this block for the mean constraint of the LHS that follows (9283-9294)
    sx = half_sum(sx * (hq * hq * hq));
    if (li == 0) block_sums[slot] = sx;
  }
}

#endif  // CUP3D_TESTING

// ------------------------------------------------------------------ direct block solve
// The block preconditioner M^-1 is "solve sum6(z) - 6z = r/h on one 8^3 block with zero
// ghosts".  The reference evaluates it by CG to a 1e-7 relative residual (14704-14745);
// the same operator can be evaluated EXACTLY (to rounding) by fast diagonalisation:
// the 1-D operator tridiag(1,-2,1) with Dirichlet ends has the sine eigenvectors
// Q[k][j] = sqrt(2/9) sin(pi (j+1)(k+1)/9) (Q = Q^T = Q^-1) and eigenvalues
// lam_k = 2 cos(pi (k+1)/9) - 2, so  z = (Q x Q x Q) [ (Q x Q x Q) r / (lam_i+lam_j+lam_k) ].
// Six 8-point transforms per lane (40 FP64 ops each thanks to Q[k][7-j] = (-1)^k Q[k][j]),
// four LDS transposes, no reductions, no iteration, no divergence: ~260 FP64 operations per
// lane against ~124 per CG ITERATION.  Its result differs from the reference's CG result by
// the CG's own truncation error (<= cond * 1e-7), i.e. it is the same preconditioner
// evaluated more accurately; selected with cup3d_poisson_params.block_solver = 1.
__constant__ double cQ[8][4];
static double *g_invD = nullptr;  // [ky][kz][kx] = 1 / (lam_kx + lam_ky + lam_kz)

__device__ __forceinline__ void sine_transform8(const double (&v)[8], double (&o)[8]) {
  const double e0 = v[0] + v[7], e1 = v[1] + v[6], e2 = v[2] + v[5], e3 = v[3] + v[4];
  const double d0 = v[0] - v[7], d1 = v[1] - v[6], d2 = v[2] - v[5], d3 = v[3] - v[4];
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    o[k] = __builtin_fma(cQ[k][3], e3, __builtin_fma(cQ[k][2], e2, __builtin_fma(cQ[k][1], e1, cQ[k][0] * e0)));
    o[k + 1] = __builtin_fma(cQ[k + 1][3], d3, __builtin_fma(cQ[k + 1][2], d2, __builtin_fma(cQ[k + 1][1], d1, cQ[k + 1][0] * d0)));
  }
}

constexpr int kFdmLds = 64 * 9;  // transposes; pitch 9 doubles keeps every ds_read/write_b64 conflict-free
// the direct solve of one block by its wavefront: v[z] = (right-hand side / h) of cell (x = lane & 7, y = lane >> 3, z) on entry;
// out receives M^-1, block_sums[slot] (if any) sum(z h^3).  T: kFdmLds doubles of LDS nobody else is using.
template <bool AG = false>
__device__ __forceinline__ void fdm_block(const GridDev &g, int slot, double (&v)[8], double *__restrict__ out, const double *__restrict__ invD,
                                          double *__restrict__ block_sums, double *T) {
  const int l = threadIdx.x, lo = l & 7, hi = l >> 3;
  double w[8], scale[8];
  double rr = 0;
#pragma unroll
  for (int z = 0; z < 8; ++z) {
    scale[z] = invD[z * 64 + l];
    rr = __builtin_fma(v[z], v[z], rr);
  }
  rr = wave_sum(rr);
  const bool tiny = (double)1 / (512 * 512) * rr < 1e-32;  // the reference leaves such a block at 0 (14735-14736)
  // forward: z (registers), x, y
  sine_transform8(v, w);  // lane (x=lo, y=hi), register kz
#pragma unroll
  for (int k = 0; k < 8; ++k) T[(k * 8 + hi) * 9 + lo] = w[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = T[l * 9 + k];  // lane (y=lo, kz=hi), register x
  __syncthreads();
  sine_transform8(v, w);  // register kx
#pragma unroll
  for (int k = 0; k < 8; ++k) T[(hi * 8 + k) * 9 + lo] = w[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = T[l * 9 + k];  // lane (kx=lo, kz=hi), register y
  __syncthreads();
  sine_transform8(v, w);  // register ky
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] *= scale[k];
  // inverse: y, x, z
  sine_transform8(w, v);  // register y, lane (kx, kz)
#pragma unroll
  for (int k = 0; k < 8; ++k) T[(hi * 8 + k) * 9 + lo] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = T[l * 9 + k];  // lane (y=lo, kz=hi), register kx
  __syncthreads();
  sine_transform8(w, v);  // register x
#pragma unroll
  for (int k = 0; k < 8; ++k) T[l * 9 + k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = T[(k * 8 + hi) * 9 + lo];  // lane (x=lo, y=hi), register kz
  sine_transform8(w, v);  // register z
  double sx = 0;
#pragma unroll
  for (int z = 0; z < 8; ++z) {
    const double r = tiny ? 0.0 : v[z];
    out[(size_t)slot * 512 + z * 64 + l] = r;
    sx += r;
  }
  if (block_sums) {
    const double hq = block_h(g, slot), h3 = hq * hq * hq;
    sx = wave_sum(sx * h3);
    if (l == 0) { if constexpr (AG) st_agent(block_sums + slot, sx); else block_sums[slot] = sx; }
  }
}
__global__ void __launch_bounds__(64) k_precond_fdm(GridDev g, const double *in, double *out, const double *__restrict__ invD,
                                                    double *__restrict__ block_sums) {
  __shared__ double T[kFdmLds];
  const int slot = block_slot(g);
  if (slot < 0) return;
  const int l = threadIdx.x;
  const double invh = 1 / block_h(g, slot);
  double v[8];
#pragma unroll
  for (int z = 0; z < 8; ++z) v[z] = invh * in[(size_t)slot * 512 + z * 64 + l];
  fdm_block(g, slot, v, out, invD, block_sums, T);
}

#ifdef CUP3D_TESTING
// TEST SUPPORT: the two wave-wide sums of one 64-value vector: out[0..63] = MFMA form per lane, out[64..127] = DPP form per lane
__global__ void __launch_bounds__(64) k_debug_wave_sum(const double *__restrict__ in, double *__restrict__ out) {
  const double v = in[threadIdx.x];
  out[threadIdx.x] = wave_sum_mfma(v);
  out[64 + threadIdx.x] = wave_sum(v);
}
#endif

static int fdm_setup() {
  if (g_invD) return CUP3D_OK;
  double Q[8][4], lam[8], invD[512];
  const double pi = 3.14159265358979323846;
  for (int k = 0; k < 8; ++k) {
    lam[k] = 2.0 * std::cos(pi * (k + 1) / 9.0) - 2.0;
    for (int j = 0; j < 4; ++j) Q[k][j] = std::sqrt(2.0 / 9.0) * std::sin(pi * (j + 1) * (k + 1) / 9.0);
  }
  for (int ky = 0; ky < 8; ++ky)
    for (int kz = 0; kz < 8; ++kz)
      for (int kx = 0; kx < 8; ++kx) invD[ky * 64 + kz * 8 + kx] = 1.0 / (lam[kx] + lam[ky] + lam[kz]);
  CUP3D_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cQ), Q, sizeof Q));
  CUP3D_HIP(hipMalloc((void **)&g_invD, sizeof invD));
  CUP3D_HIP(hipMemcpy(g_invD, invD, sizeof invD, hipMemcpyHostToDevice));
  return CUP3D_OK;
}

static int *cg_iters_buffer(Sim *s) {  // per-block CG iteration counts of the last launch (measurement only)
  if (!s->d_cg_iters && hipMalloc((void **)&s->d_cg_iters, (size_t)s->nb * sizeof(int)) != hipSuccess) return nullptr;
  return s->d_cg_iters;
}

constexpr int kLoopPrio = 0;    // LhsIn::prio of the production launch (measured: profiles/r03)
// Evaluation of the production block CG (EV bits of cg_block).  Round 6: 6 = single-width LDS reads (bit 2: 32 ds_read_b64 with immediate
// plane offsets instead of 16 half-rate ds_read2_b64 + 8 address adds per CG iteration) + reciprocal divisions (bit 4: v_rcp_f64, two
// Newton steps and a residual correction, within 1 ulp of the IEEE quotient, 8 instead of 12 instructions, twice per iteration).
// Rounds 2-5 ran EV 0: the variants had only been compared on the stand-alone kernel with an input that converges in three CG
// iterations (profiles/r02/probe_block_cg_variants_*.jsonl: all within 2 %).  Behind the loops on the solver's own inputs (27 CG
// iterations per block) the A/B on one box reads 9.17 -> 8.77 ms per BiCGSTAB iteration at 512^3 and 1.194 -> 1.142 at 256^3 (-4.3 %),
// with identical BiCGSTAB counts (profiles/r06/block_cg_evaluation_behind_the_loops/): bit 2 alone -2.5 % (bit-identical results),
// bit 4 alone -1.6 %, bit 8 (three-operand FMA for the p update) nothing.  Like the FMA contraction, the reciprocal division is a
// rounding-level deviation inside a block solve that is truncated at 1e-7; block_solver 2 stays the reference's association and IEEE division.
constexpr int kCgProduction = 6;

int launch_precond(Sim *s, const double *in, double *out, bool want_sums) {
  GridDev g = s->gdev();
  double *sums = want_sums ? s->d_partials + (size_t)s->max_groups * 8 : nullptr;
  if (s->block_solver == 5) {  // one multigrid V-cycle (multigrid.hip); the LHS that follows sums the blocks itself
    s->sums_of = nullptr;
    return mg_vcycle(s, in, out);
  }
  if (s->block_solver == 1) {
    int rc = fdm_setup();
    if (rc) return rc;
    ProfileScope ps("poisson_block_fdm");
    hipLaunchKernelGGL(k_precond_fdm, dim3(launch_groups(g)), dim3(64), 0, stream(), g, in, out, g_invD, sums);
    CUP3D_HIP(hipGetLastError());
    s->sums_of = want_sums ? out : nullptr;
    return CUP3D_OK;
  }
  ProfileScope ps("poisson_block_cg");
  // Production (block_solver 0, kCgProduction) contracts a*b+c into FMAs here (and only here) and divides by reciprocal + correction
  // (within 1 ulp); wave sums by DPP (the matrix-pipe sums described above k_precond were measured slower and exist in the testing
  // flavour only).  Its result sits behind two wave reductions per iteration whose summation order already differs from the CPU's, and the
  // CG's own truncation is 1e-7, so the contraction is a tolerance-level deviation (tests bound it against the reference's z).
  // block_solver 2 = the reference's association (no contraction).
  const dim3 G(launch_groups(g)), B(64);
  int *it = profile_on() ? cg_iters_buffer(s) : nullptr;  // for the FP64 roofline of bench.py (cup3d_profile_block_cg_iterations)
#define CG(FMA_, EV_) hipLaunchKernelGGL((k_precond<FMA_, false, EV_>), G, B, 0, stream(), g, in, out, sums, 0.0, 0.0, it)
  switch (s->block_solver) {
#ifdef CUP3D_TESTING
    case 0:  // production: kCgProduction, or (tuning) the evaluation selected with cup3d_debug_set_option("cg_variant", 8 + bits)
      switch (debug_option("cg_variant") >= 8 ? debug_option("cg_variant") - 8 : kCgProduction) {
        case 0: CG(true, 0); break;
        case 1: CG(true, 1); break;
        case 2: CG(true, 2); break;
        case 3: CG(true, 3); break;
        case 4: CG(true, 4); break;
        case 5: CG(true, 5); break;
        case 6: CG(true, 6); break;
        case 7: CG(true, 7); break;
        case 8: CG(true, 8); break;
        case 9: CG(true, 9); break;
        case 10: CG(true, 10); break;
        case 11: CG(true, 11); break;
        case 12: CG(true, 12); break;
        case 13: CG(true, 13); break;
        case 14: CG(true, 14); break;
        case 15: CG(true, 15); break;
        case 16: CG(true, 16); break;   // row totals through LDS (wave_sum_rows)
        case 18: CG(true, 18); break;
        case 22: CG(true, 22); break;
        case 30: CG(true, 30); break;
        case 32: CG(true, 32); break;   // Chronopoulos-Gear single-reduction form (EXPERIMENT)
        default: set_error("unknown cg_variant"); return CUP3D_EINVAL;
      }
      break;
    case 3: CG(true, 0); break;  // alias of 0 (the round-1 kernel IS the production evaluation), kept for old scripts
    case 4: {  // two blocks per wavefront (A/B timing)
      const int pchunk = ((g.nblocks + 1) / 2 + 7) / 8;
      hipLaunchKernelGGL((k_precond_pair<true, false>), dim3(8 * pchunk), B, 0, stream(), g, pchunk, in, out, sums, 0.0, 0.0, it);
      break;
    }
#else
    case 0: CG(true, kCgProduction); break;
    case 3: case 4: return not_in_release("block_solver 3 / 4 (A/B variants of the block CG)");
#endif
    case 2: CG(false, 0); break;
    default: set_error("unknown block_solver %d", s->block_solver); return CUP3D_EINVAL;
  }
#undef CG
  CUP3D_HIP(hipGetLastError());
  s->sums_of = want_sums ? out : nullptr;  // block sums of `out` are fresh: the next LHS of `out` reuses them
  return CUP3D_OK;
}

int launch_precond_diffusion(Sim *s, const double *in, double *out, const HelmholtzOp &op) {
  GridDev g = s->gdev();
  ProfileScope ps("diffusion_block_cg");
  if (s->block_solver != 2) hipLaunchKernelGGL((k_precond<true, true, kCgProduction>), dim3(launch_groups(g)), dim3(64), 0, stream(), g, in, out, (double *)nullptr, op.nu, op.dt, (int *)nullptr);
  else hipLaunchKernelGGL((k_precond<false, true, 0>), dim3(launch_groups(g)), dim3(64), 0, stream(), g, in, out, (double *)nullptr, op.nu, op.dt, (int *)nullptr);
  CUP3D_HIP(hipGetLastError());
  s->sums_of = nullptr;
  return CUP3D_OK;
}

// ------------------------------------------------------------------ fused BiCGSTAB vector kernels
struct Vecs {
  double *v[NVEC];
  const double *xin;  // where the second loop reads x from: v[X_], or the x_opt snapshot right after one was taken (see solve())
};

// ------------------------------------------------------------------ the scalar recurrences, resident on the device
// The scalars of PoissonSolverAMR::solve -- alpha, beta, omega, r0r_prev (14443, 14493, 14558-14564), the breakdown test (14566), the
// x_opt bookkeeping (14594-14600) and the stopping rule (14601) -- as ONE struct and ONE pair of functions compiled for host and
// device.  In the fused iterations (k % 50 != 0) the struct lives in device memory: the kernel that totals the dot products (or, over
// ranks, a one-thread kernel behind the all-reduce) steps it, the next loop kernel reads alpha / beta / omega from it, and the host
// only WATCHES: it enqueues iteration k + 1 before it has seen the outcome of iteration k, through a ring of pinned status slots.  A
// launch never waits for the host.  When the outcome is "converged" or "serious breakdown", the kernels of the iteration enqueued
// ahead find state != kRun and return at once; the host then finishes, or runs the restart (14567-14593) and re-enqueues.
// The every-50th iterations (true-residual refresh through _lhs) and the other block solvers step the same struct on the host.
enum { kRun = 0, kDone = 1, kRestart = 2 };
struct SolverCtl {
  double alpha, beta, omega, r0r_prev;
  double norm, init_norm, min_norm;
  double tol, tol_rel;
  int state;
  int restarts, max_restarts;
  int xcur, xopt;  // which of the two x buffers holds x / the best iterate so far (x_opt; -1: none yet)
  int iter;        // iterations completed
  unsigned seq;    // sequence number of the fused iteration in flight (the host's Sim::ctl_seq numbering: k_ctl_set places it, ctl_step2 advances
                   // it): the slot of the status ring and the values of the early all-reduce's flags derive from it, so that the kernels of an
                   // iteration take NO per-iteration argument
};
struct CtlSlot { SolverCtl c; unsigned seq; unsigned pad; };  // pinned status ring, slot = seq & 3
// x is updated in place unless the buffer that holds it is also the x_opt snapshot: then the update goes to the other buffer
// (x_opt = x without a copy: x is read once and written once by the second loop anyway)
__host__ __device__ inline int ctl_xwrite(const SolverCtl &c) { return c.xopt == c.xcur ? 1 - c.xcur : c.xcur; }
// after the first loop's dot products (q.y, y.y): 14493
__host__ __device__ inline void ctl_step1(SolverCtl &c, const double *t) { c.omega = t[0] / (t[1] + 1e-100); }
// after the second loop's seven (14546): 14558-14566, 14594-14601.  The restart itself (kernel launches) is the host's.
__host__ __device__ inline void ctl_step2(SolverCtl &c, const double *t) {
  const double eps = 1e-100;
  const double r0r = t[0], r0w = t[1], r0s = t[2], r0z = t[3], norm_1 = t[4], norm_2 = t[5];
  const double norm = sqrt(t[6]);
  const double omega = c.omega;
  double alpha = c.alpha;
  const double beta = alpha / (omega + eps) * r0r / (c.r0r_prev + eps);  // 14558
  alpha = r0r / (r0w + beta * r0s - beta * omega * r0z);                 // 14559
  double alphat = 1.0 / (omega + eps) + r0w / (r0r + eps) - beta * omega * r0z / (r0r + eps);
  alphat = 1.0 / (alphat + eps);
  if (fabs(alphat) < 10 * fabs(alpha)) alpha = alphat;                   // 14563-14564
  c.alpha = alpha;
  c.beta = beta;
  c.r0r_prev = r0r;
  c.norm = norm;
  c.xcur = ctl_xwrite(c);  // x lives where the second loop wrote it
  c.iter++;
  c.seq++;
  int state = kRun;
  if (r0r * r0r < 1e-16 * norm_1 * norm_2 && c.restarts < c.max_restarts) {  // serious breakdown, 14566-14567
    c.restarts++;
    state = kRestart;
  }
  if (norm < c.min_norm) {  // 14594-14600
    c.min_norm = norm;
    c.xopt = c.xcur;
  }
  if (norm < c.tol || norm / (c.init_norm + eps) < c.tol_rel) state = kDone;  // 14601
  c.state = state;
}
__device__ __forceinline__ void ctl_publish(const SolverCtl *c, CtlSlot *ring, unsigned seq) {
  CtlSlot *sl = ring + (seq & 3);
  sl->c = *c;
  __threadfence_system();
  __hip_atomic_store(&sl->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// several ranks: the totals are all-reduced first (communication stream); this one-thread kernel behind the all-reduce steps the
// struct -- identically on every rank, the all-reduced bits are the same everywhere -- and the compute stream waits for its event
template <int STEP>
__global__ void k_ctl_step(SolverCtl *ctl, const double *__restrict__ tot, CtlSlot *ring) {
  if (ctl->state != kRun) return;  // an iteration enqueued ahead of a stop / restart: nothing happened, nothing to step
  SolverCtl c = *ctl;
  const unsigned it = c.seq;
  if (STEP == 1) ctl_step1(c, tot); else ctl_step2(c, tot);
  *ctl = c;
  if (STEP == 2) ctl_publish(ctl, ring, it);
}
__global__ void k_ctl_set(SolverCtl *ctl, SolverCtl v) { *ctl = v; }

#define GRID_STRIDE(j, n) for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < (n); j += (long)gridDim.x * 256)
// 16 B per lane (double2): n is a multiple of 512
#define GRID_STRIDE2(j, n) for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < (n) / 2; j += (long)gridDim.x * 256)
// NT: nontemporal (streaming) accesses -- every vector is touched once per launch and the 19 GB working set cannot stay in L2
template <bool NT>
__device__ __forceinline__ double2 ld2(const double *v, long j) {
  if constexpr (!NT) return reinterpret_cast<const double2 *>(v)[j];
  const double *p = v + 2 * j;
  double2 r;
  r.x = __builtin_nontemporal_load(p);
  r.y = __builtin_nontemporal_load(p + 1);
  return r;
}
template <bool NT>
__device__ __forceinline__ void st2(double *v, long j, double2 val) {
  if constexpr (!NT) { reinterpret_cast<double2 *>(v)[j] = val; return; }
  double *p = v + 2 * j;
  __builtin_nontemporal_store(val.x, p);
  __builtin_nontemporal_store(val.y, p + 1);
}
#define LD2(v) ld2<NT>(v, j)
#define ST2(v, val) st2<NT>(v, j, val)
__device__ __forceinline__ double2 operator+(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 operator-(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 operator*(double s, double2 a) { return make_double2(s * a.x, s * a.y); }
__device__ __forceinline__ double dot2(double2 a, double2 b, double acc) { acc += a.x * b.x; acc += a.y * b.y; return acc; }

// ------------------------------------------------------------------ vector loop + block preconditioner in ONE launch
// Per BiCGSTAB iteration the reference runs   loop 1 -> z ;  zhat = M^-1 z ;  v = A zhat   and   loop 2 -> w ;  what = M^-1 w ;  t = A what.
// The fused vector loops are HBM-bound (6.2 TB/s, nothing for the FP64 units to do), the block CG that consumes their output is
// bound by FP64 issue and LDS (0.1 of the HBM roof) -- run back to back they each leave the other resource idle, and together they
// are 93 % of an iteration.  Here one wavefront owns one block from start to end: it streams the block's 512 cells of the 11 (12)
// input vectors, writes the 7 (4) updated ones, keeps the block of z (w) in registers and runs the block CG on it straight away;
// while it iterates, the other wavefronts of the SIMD are in their streaming phase, so the two bounds overlap instead of adding.
// The arithmetic per cell is that of k_loop1 / k_loop2 and of cg_block, unchanged; the dot products are summed per block first
// (wave tree) and the per-block values by k_sums_finish, another order than the grid-stride partials of the unfused kernels.
// block_dots layout: [K][nb].
// ------------------------------------------------------------------ totals of per-block values INSIDE the kernel that produces them
// Rounds 2-4 finished the dot products of a fused loop in a launch of their own (k_sums_finish: 64 / 256 workgroups over the [K][nb]
// per-block values, last workgroup totals and steps the scalars).  That launch -- 16-27 us plus the gap around it, twice per iteration --
// is what the per-rank share of the workload on 8 GPUs (256^3: 1.15 ms per iteration) feels most, and it pins the moment the totals exist
// to the END of the loop kernel, one block-CG phase later than they are complete.  Here the kernel finishes them itself: a wavefront
// that has written its block's values takes a ticket in the counter of its GROUP (64 consecutive slots); the last one of a group adds
// the group's 64 values (one per lane, wave tree) and takes a ticket in the counter of the SUPER-GROUP (64 groups); the last one there
// adds the 64 group sums; the last super-group adds the super-group sums, stores the K totals and runs `then` (the recurrence step on
// one rank; the flag the communication stream waits for over ranks).  Who is last varies from run to run, WHAT is added in which
// order does not: sums of fixed sets in a fixed tree -- deterministic.  Counters count over all launches of a loop (inner / boundary
// pass, plain / interface list): membership is by slot.  Release / acquire at agent scope as in grid_sum_finish (tile.hpp); the values
// of other wavefronts are read with agent-scope loads.
struct Arrive {
  const double *vals;      // [K][nb] per-block values
  double *g1, *g2;         // [K][n1], [K][n2]: sums of 64 blocks / of 64 groups
  unsigned *c1, *c2, *c3;  // arrivals per group [n1], per super-group [n2], super-groups done [1]; all zero between two loops
  long nb, n1, n2;
  double *out;             // [K] totals (device memory)
  int light;               // A/B (testing build, "arrive_light_release"): group leaders release like every wavefront (no L2 write-back); production: 0
};
__device__ __forceinline__ unsigned ticket_of_wave(unsigned *counter) {  // lane 0's values are stored: take a ticket; every lane gets it
  unsigned t = 0;
  if (threadIdx.x == 0) {
    stores_done();
    t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)t);
}
// a ticket taken by a wavefront that speaks for a whole GROUP: its sums were written by agent-scope stores as well, but they are read by a
// wavefront on ANOTHER XCD a moment later, so this rare path (1 wavefront in 64) pays for the full agent-scope release (L2 write-back)
__device__ __forceinline__ unsigned ticket_of_group(unsigned *counter, int light) {
  unsigned t = 0;
  if (threadIdx.x == 0) {
    if (light) stores_done(); else __threadfence();
    t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)t);
}
template <int K, class Then>
__device__ __forceinline__ void arrive(const Arrive &A, int slot, Then then) {
  const int l = threadIdx.x;
  const long g = slot >> 6, first = g << 6;
  const unsigned gsize = (unsigned)(A.nb - first < 64 ? A.nb - first : 64);
  if (ticket_of_wave(A.c1 + g) != gsize - 1) return;  // (the loads below are issued after the ticket has come back: control dependence)
  __threadfence();                                    // ... and behind an agent-scope acquire (the last arrivers only: 1 wavefront in 64)
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = wave_sum((unsigned)l < gsize ? ld_agent(A.vals + (size_t)k * A.nb + first + l) : 0.0);
    if (l == 0) st_agent(A.g1 + (size_t)k * A.n1 + g, v);
  }
  const long sg = g >> 6, gfirst = sg << 6;
  const unsigned sgsize = (unsigned)(A.n1 - gfirst < 64 ? A.n1 - gfirst : 64);
  if (l == 0) st_agent(A.c1 + g, 0u);
  if (ticket_of_group(A.c2 + sg, A.light) != sgsize - 1) return;
  __threadfence();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = wave_sum((unsigned)l < sgsize ? ld_agent(A.g1 + (size_t)k * A.n1 + gfirst + l) : 0.0);
    if (l == 0) st_agent(A.g2 + (size_t)k * A.n2 + sg, v);
  }
  if (l == 0) st_agent(A.c2 + sg, 0u);
  if (ticket_of_group(A.c3, A.light) != (unsigned)A.n2 - 1) return;
  __threadfence();
  double tot[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double v = 0;
    for (long j = l; j < A.n2; j += 64) v += ld_agent(A.g2 + (size_t)k * A.n2 + j);
    tot[k] = wave_sum(v);
    if (l == 0) st_agent(A.out + k, tot[k]);
  }
  if (l == 0) {
    st_agent(A.c3, 0u);
    then(tot);
  }
}
// what the wavefront that completes the dot products does with them
struct DotsThen {
  SolverCtl *ctl; CtlSlot *ring;
  int which;         // 1: first loop (q.y, y.y -> omega, 14493), 2: second loop (the seven of 14546 -> 14558-14601)
  int step;          // != 0: one rank -- step the solver's scalars (ctl_step1 / ctl_step2) and, after the second loop, publish them to the host's ring; 0: totals only
  unsigned *flag;    // several ranks, early all-reduce: raised to 2 seq + which - 1 once the totals are in device memory (k_wait_totals on the communication stream)
  __device__ __forceinline__ void operator()(const double *tot) const;
};
struct LoopSums {  // what a fused loop kernel needs to total its per-block values; constant over a solve, in DEVICE memory (Sim::d_loop_sums): the
                   // kernels take a pointer -- as a by-value argument its 25 words stayed live across the plane loop and cost the occupancy
  Arrive dots;     // K = 2 (first loop) / 7 (second loop) dot products, complete when the last block leaves its vector phase
  Arrive mean;     // K = 1: sum(zhat h^3) / sum(what h^3) of the block solves (mean-constraint row, 9283-9326), complete when the kernel ends; vals == nullptr: not wanted
  DotsThen then;
};
__device__ __forceinline__ void DotsThen::operator()(const double *tot) const {
  const unsigned it = ctl->seq;
  if (step != 0 && ctl->state == kRun) {  // (every wavefront that came this far saw kRun; only this one changes it)
    SolverCtl c = *ctl;
    if (which == 1) ctl_step1(c, tot); else ctl_step2(c, tot);
    *ctl = c;
    if (which == 2) ctl_publish(ctl, ring, it);
  }
  if (flag) {  // the totals (this lane's own agent-scope stores) before the flag: a full agent-scope release, once per launch
    __threadfence();
    st_agent(flag, 2 * it + (unsigned)(which - 1));
  }
}
struct NoThen { __device__ __forceinline__ void operator()(const double *) const {} };
__global__ void k_set_loop_sums(LoopSums *dst, LoopSums a, LoopSums b) { dst[0] = a; dst[1] = b; }

struct Loop1Args { double alpha, beta, omega; };
struct Loop2Args { double alpha, omega; };

#define NTL(v, j) __builtin_nontemporal_load(&(v)[j])
#define NTS(v, j, val) __builtin_nontemporal_store((val), &(v)[j])

// ---- the LHS application folded into the loop kernel that needs its result (uniform grids)
// Per iteration the reference applies  v = A zhat  after the first loop and  t = A what  after the second (14489, 14549), and each
// loop then streams t and v like any other vector.  On the device that was two launches of k_lhs (16 B/cell each, 12 % of an
// iteration, with the all-reduce tucked behind them).  With FLHS the wavefront that owns a block builds the ghosted tile of the
// block's what (first loop) / zhat (second loop) in LDS -- its own column of 8 planes plus the six face slabs, fetched from the
// neighbour slots, the domain-face rule (zero-gradient: own face cell) or the halo slabs of other ranks, i.e. what load_scalar_tile
// does for a 256-thread workgroup -- evaluates  h (xm + xp + ym + yp + zm + zp - 6 c)  in k_lhs's association (BIT-IDENTICAL t and
// v), uses the value in place of the streamed one and stores it for the other loop.  One stream fewer to read, no k_lhs launch.
// Tile layout: 10 planes (0 and 9: the z ghosts) of pitch 96 doubles = 10 rows of 8 (rows 0 and 9: the y ghosts) + 8 x-minus ghosts
// + 8 x-plus ghosts.  960 doubles; the block CG's LDS (zeroed again when the CG starts) is inside it.  Every stencil operand is one
// ds_read_b64 with an immediate plane offset: nothing is carried in registers from plane to plane.
constexpr int kTilePitch = 96, kTileLds = 10 * kTilePitch;
static_assert(kTileLds >= kCgLds && kTileLds >= kFdmLds, "the block solve reuses the tile's LDS");
struct LhsIn {
  const double *halo;   // face slabs received from other ranks (Sim::halo_recv)
  const double *total;  // sum(u h^3) over all ranks, for the mean-constraint row (9283-9326); device memory
  int mode;             // bMeanConstraint as ComputeLHS uses it: 0 none, 1 corner row = total, 2 += total h^3 everywhere, 3 corner row = u
  int corner_slot;      // slot of the block with index (0,0,0) on this rank, or -1
  int prio;             // wave priority (s_setprio) while the wavefront streams its block; back to 0 when the block CG starts
  const double *invD;   // DIRECT form of the block solve (block_solver 1): 1 / (lam_kx + lam_ky + lam_kz), [ky][kz][kx]; else unused
  // early all-reduce over ranks (solve(): early): *total is valid once *mean_flag has reached mean_seq -- the wavefronts that USE the total
  // wait for that (mode 1: the corner block's only); nullptr: the total was complete before the launch
  const unsigned *mean_flag;
  int mean_wait;        // which value: 1 = 2 (seq - 1) + 1 (first loop: the total of the previous iteration's second loop), 2 = 2 seq (second loop); seq = SolverCtl::seq
  unsigned *fail;       // pinned: raised when that wait gives up (10 s)
  double *extra;        // EXPERIMENT (testing build, "extra_streams"): scratch of 2 x nb x 512 doubles the XTRA kernels write their dummy streams to; else nullptr
};
struct TileRegs { double c[8], gv[6]; };
// the 14 loads of a tile in two groups: the block's own column (needs nothing but the slot) and the six face slabs (need the
// neighbour table first); the caller issues the first plane of its streams between the two, then commits
__device__ __forceinline__ void tile_issue_own(int slot, const double *__restrict__ f, int l, TileRegs &R) {
  const double *own = f + (size_t)slot * 512;
#pragma unroll
  for (int z = 0; z < 8; ++z) R.c[z] = own[z * 64 + l];
}
__device__ __forceinline__ void tile_issue_faces(const GridDev &g, int slot, const double *__restrict__ f, const double *__restrict__ halo, int l, TileRegs &R) {
  const double *own = f + (size_t)slot * 512;
#pragma unroll
  for (int face = 0; face < 6; ++face) {
    const int n = g.nbr[slot * 6 + face];
    int nb_cell, own_cell, lds;
    face1(face, l, nb_cell, own_cell, lds);
    const double *__restrict__ base = n >= kNbrHalo ? halo + (size_t)(n - kNbrHalo) * 64 : (n >= 0 ? f + (size_t)n * 512 : own);
    R.gv[face] = base[n >= kNbrHalo ? l : (n >= 0 ? nb_cell : own_cell)];
  }
}
__device__ __forceinline__ void tile_commit(const TileRegs &R, double *T, int l) {
  const int base = ((l >> 3) + 1) * 8 + (l & 7), a1 = l & 7, a2 = (l >> 3) + 1;
#pragma unroll
  for (int z = 0; z < 8; ++z) T[(z + 1) * kTilePitch + base] = R.c[z];
  T[a2 * kTilePitch + 80 + a1] = R.gv[0];  // x faces: lane = (a1 = y, z = a2 - 1)
  T[a2 * kTilePitch + 88 + a1] = R.gv[1];
  T[a2 * kTilePitch + a1] = R.gv[2];       // y faces: lane = (a1 = x, z = a2 - 1) -> rows 0 and 9
  T[a2 * kTilePitch + 72 + a1] = R.gv[3];
  T[base] = R.gv[4];                       // z faces: lane = (x, y) -> planes 0 and 9
  T[9 * kTilePitch + base] = R.gv[5];
  __syncthreads();
}
// per-lane tile offsets of the x neighbours (the edge lanes read the ghost slots behind the rows)
struct TileIdx { int base, ixm, ixp; };
__device__ __forceinline__ TileIdx tile_idx(int l) {
  const int x = l & 7, y = l >> 3, base = (y + 1) * 8 + x;
  return TileIdx{base, x > 0 ? base - 1 : 80 + y, x < 7 ? base + 1 : 88 + y};
}
// the mean-constraint fix-ups of ComputeLHS (9299-9326), decided once per wavefront so that the plane loop stays one basic block
// (a branch per plane makes the compiler keep every stream's address in a VGPR pair: +34 registers)
struct LhsFix {
  double total, add;  // sum(u h^3) over all ranks; total * h^3 (mode 2)
  bool add_mean;      // mode 2: t += total h^3 in every cell (9314)
  bool row_total;     // this lane holds the corner cell (plane 0) and mode 1: t = total (9299-9304)
  bool row_self;      // ... and mode > 2: t = u (9316-9325)
};
__device__ __forceinline__ LhsFix lhs_fix(const LhsIn &L, const SolverCtl *ctl, int slot, int l, double h) {
  LhsFix f;
  const bool uses_total = L.mode == 2 || (L.mode == 1 && slot == L.corner_slot);  // wave-uniform
  if (uses_total && L.mean_flag) {
    const unsigned want = L.mean_wait == 1 ? 2 * (ctl->seq - 1) + 1 : 2 * ctl->seq;
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(L.mean_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > 1000000000LL) { if (l == 0) __hip_atomic_store(L.fail, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
  }
  f.total = uses_total ? ld_agent(L.total) : 0.0;
  f.add = f.total * (h * h * h);
  f.add_mean = L.mode == 2;
  const bool corner = slot == L.corner_slot && l == 0;
  f.row_total = corner && L.mode == 1;
  f.row_self = corner && L.mode > 2;
  return f;
}
// KernelLHSPoisson (9211-9214) for the cell of lane l in plane zz, in k_lhs's association; cc = the cell's own value
template <int ZZ>
__device__ __forceinline__ double tile_lhs(const double *T, const TileIdx &ix, double &cc, double h, const LhsFix &f) {
  // volatile LDS pointers (address space kept): one ds_read_b64 per operand, in this order, plane offset in the instruction
  typedef const volatile __attribute__((address_space(3))) double lds_cvd;
  lds_cvd *Q = (lds_cvd *)(T + (ZZ + 1) * kTilePitch);
  cc = Q[ix.base];
  double t = Q[ix.ixm] + Q[ix.ixp];
  t += Q[ix.base - 8];
  t += Q[ix.base + 8];
  t += Q[ix.base - kTilePitch];
  t += Q[ix.base + kTilePitch];
  t = h * (t - 6.0 * cc);
  if (ZZ == 0) {  // the corner cell is cell 0 of its block: selects, no branches
    t = f.row_total ? f.total : t;
    t = f.row_self ? cc : t;
  }
  const double t2 = t + f.add;
  return f.add_mean ? t2 : t;
}

// DIRECT: the block solve behind the loop is the fast diagonalisation (fdm_block: the same M^-1, exact instead of by CG -- block_solver 1,
// bench.py's `alt`), not the reference's CG: no iteration, no reductions, so the kernel is what the streams alone allow
// TOT: the kernel totals its per-block values itself (Arrive; the early all-reduce over ranks) -- else a launch of k_sums_finish does
// XTRA (EXPERIMENT, testing build): what does the iteration pay per byte?  XTRA = 1: the first loop reads one more stream (b, folded into a
// dot product with weight 0: same bits) and the second writes one more (w again, to scratch): +16 B/cell per iteration, the mirror image of
// forming v inside the first loop instead of streaming it.  XTRA = 2: one more read AND one more write in both loops: +32 B/cell.
template <bool FMA, int EV, bool FLHS, bool DIRECT = false, bool TOT = false, int XTRA = 0>
__device__ __forceinline__ void loop1_cg_body(const GridDev &g, const Vecs &V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb,
                                              double *block_sums, int *__restrict__ iters_out, const LhsIn &L, const LoopSums *__restrict__ Z) {
  __shared__ double P[FLHS ? kTileLds : (DIRECT ? kFdmLds : kCgLds)];
  const int slot = block_slot(g);
  if (slot < 0) return;
  if (ctl->state != kRun) return;  // enqueued ahead of a stop or a restart (see SolverCtl)
  const Loop1Args a{ctl->alpha, ctl->beta, ctl->omega};
  const int l = threadIdx.x;
  const double hq = block_h(g, slot), invh = 1 / hq;
  double r[8], d0 = 0, d1 = 0;
  // plane zz + 1 is requested before plane zz is computed and stored (two planes = 22 x 512 B per wavefront in flight): the loads
  // may alias the stores as far as the compiler knows, so the order has to be written out
  // (block base pointers are wave-uniform -> scalar registers; the per-lane part of every address is one 32-bit offset)
  const size_t bo = (size_t)slot * 512;
  // FLHS: what comes from the tile and t is computed from it -- streams 6 and 8 are not loaded
  enum { iRHAT, iW, iSHAT, iZ, iPHAT, iS, iWHAT, iZHAT, iT, iV, iR, NS };
  const double *const src[NS] = {V.v[RHAT] + bo, V.v[W_] + bo, V.v[SHAT] + bo, V.v[Z_] + bo, V.v[PHAT] + bo, V.v[S_] + bo, V.v[WHAT] + bo, V.v[ZHAT] + bo,
                                 V.v[T_] + bo, V.v[V_] + bo, V.v[R_] + bo};
  double *const oP = V.v[PHAT] + bo, *const oS = V.v[S_] + bo, *const oSH = V.v[SHAT] + bo, *const oZ = V.v[Z_] + bo, *const oQ = V.v[Q_] + bo,
               *const oQH = V.v[QHAT] + bo, *const oY = V.v[Y_] + bo, *const oT = V.v[T_] + bo;
  double in[2][NS + 1];
  const double *const xsrc = V.v[B_] + bo;
  double *const xdst = XTRA ? L.extra + bo : nullptr;
#define LOAD_PLANE(buf, off)                                                            \
  _Pragma("unroll") for (int i = 0; i < NS; ++i)                                        \
    if (!(FLHS && (i == iWHAT || i == iT))) in[buf][i] = NTL(src[i], off);              \
  if constexpr (XTRA != 0) in[buf][NS] = NTL(xsrc, off);
  TileIdx ix{0, 0, 0};
  LhsFix fx{};
  TileRegs tr;
  if constexpr (FLHS) {  // the tile's loads first, the first plane of the streams right behind them, then the tile goes to LDS
    fx = lhs_fix(L, ctl, slot, l, hq);
    tile_issue_own(slot, V.v[WHAT], l, tr);
    ix = tile_idx(l);
  }
  // (a streaming wavefront's loads and stores go out ahead of the arithmetic of the wavefronts that sit in their block CG)
  if (L.prio == 1) __builtin_amdgcn_s_setprio(1); else if (L.prio == 2) __builtin_amdgcn_s_setprio(2); else if (L.prio == 3) __builtin_amdgcn_s_setprio(3);
  LOAD_PLANE(0, l)
  if constexpr (FLHS) {
    tile_issue_faces(g, slot, V.v[WHAT], L.halo, l, tr);  // (before or behind the first plane: no measurable difference, profiles/r03)
    tile_commit(tr, P, l);
  }
#pragma unroll
  for (int zz = 0; zz < 8; ++zz) {  // first fused loop, 14454-14464, on plane zz of this block
    const int j = zz * 64 + l;
    if (zz < 7) { LOAD_PLANE((zz + 1) & 1, j + 64) }
    const double *c = in[zz & 1];
    double what = c[iWHAT], t = c[iT];
    if constexpr (FLHS) {
      t = zz == 0 ? tile_lhs<0>(P, ix, what, hq, fx) : tile_lhs<1>(P + (zz - 1) * kTilePitch, ix, what, hq, fx);   // t = A what, 14549
      NTS(oT, j, t);                                                              // the second loop streams it
    }
    const double rhat = c[iRHAT], w = c[iW], shat0 = c[iSHAT], z0 = c[iZ];
    const double phat = rhat + a.beta * (c[iPHAT] - a.omega * shat0);
    const double sv = w + a.beta * (c[iS] - a.omega * z0);
    const double shat = what + a.beta * (shat0 - a.omega * c[iZHAT]);
    const double z = t + a.beta * (z0 - a.omega * c[iV]);
    const double q = c[iR] - a.alpha * sv;
    const double qhat = rhat - a.alpha * shat;
    const double y = w - a.alpha * z;
    NTS(oP, j, phat); NTS(oS, j, sv); NTS(oSH, j, shat); NTS(oZ, j, z); NTS(oQ, j, q); NTS(oQH, j, qhat); NTS(oY, j, y);
    d0 += q * y;
    d1 += y * y;
    if constexpr (XTRA != 0) d0 += 0.0 * c[NS];
    if constexpr (XTRA == 2) NTS(xdst, j, y);
    r[zz] = invh * z;  // the right-hand side of the block solve, main.cpp:14723
  }
#undef LOAD_PLANE
  d0 = wave_sum(d0);
  d1 = wave_sum(d1);
  if constexpr (TOT) {
    if (l == 0) { st_agent(block_dots + slot, d0); st_agent(block_dots + nb + slot, d1); }
    arrive<2>(Z->dots, slot, Z->then);  // q.y, y.y are complete when the last block passes here: the totals exist one block solve before the kernel ends
  } else if (l == 0) { block_dots[slot] = d0; block_dots[nb + slot] = d1; }
  if (L.prio) __builtin_amdgcn_s_setprio(0);
  if constexpr (FLHS) __syncthreads();  // the tile is read no more: the block solve takes over its LDS
  if constexpr (DIRECT) fdm_block<TOT>(g, slot, r, V.v[ZHAT], L.invD, block_sums, P);
  else cg_block<FMA, false, EV, TOT>(g, slot, r, V.v[ZHAT], block_sums, 0.0, 0.0, iters_out, P);  // zhat = M^-1 z, 14488
  if constexpr (TOT) if (Z->mean.vals) arrive<1>(Z->mean, slot, NoThen());  // sum(zhat h^3) for the mean-constraint row of v = A zhat
}
// (with the LHS inside the compiler takes 110 registers -> 4 wavefronts per SIMD; held to 5 wavefronts it fits 94 without a spill and is
//  SLOWER: 0.54 instead of 0.51 ms at 256^3, 3.96 instead of 3.93 at 512^3 -- gpurun_out r03c / r03d, profiles/r03)
template <bool FMA, int EV, bool FLHS>
__global__ void __launch_bounds__(64) k_loop1_cg(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums,
                                                 int *__restrict__ iters_out, LhsIn L, const LoopSums *__restrict__ Z) {
  loop1_cg_body<FMA, EV, FLHS>(g, V, ctl, block_dots, nb, block_sums, iters_out, L, Z);
}
// The same kernel held to 5 wavefronts per SIMD (94 registers, no spill; the 7.5 KB tile allows 21 per CU).  Round 3 measured this
// SLOWER with the ds_read2_b64 form of the block CG; with the single-width reads of round 6, which leave the LDS headroom for a fifth
// wavefront, it is 3 % FASTER at 512^3 (3.78 against 3.90 ms, 262 144 blocks = 51 rounds of wavefronts) and 1.5 % slower at 256^3
// (0.512 against 0.505 ms: 6.4 rounds, the tail of the last round weighs more) -- profiles/r06/loop1_five_waves/.  Production takes it
// from kFiveWavesFrom blocks per launch; same body, same bits.
constexpr int kFiveWavesFrom = 131072;
template <bool FMA, int EV, bool FLHS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5)))
k_loop1_cg_w5(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums, int *__restrict__ iters_out, LhsIn L, const LoopSums *__restrict__ Z) {
  loop1_cg_body<FMA, EV, FLHS>(g, V, ctl, block_dots, nb, block_sums, iters_out, L, Z);
}
// block_solver 1: first loop + the direct block solve (`alt`)
template <bool FLHS>
__global__ void __launch_bounds__(64) k_loop1_fdm(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums,
                                                  LhsIn L, const LoopSums *__restrict__ Z) {
  loop1_cg_body<true, 0, FLHS, true>(g, V, ctl, block_dots, nb, block_sums, nullptr, L, Z);
}

template <bool FMA, int EV, bool FLHS, bool DIRECT = false, bool TOT = false, int XTRA = 0>
__device__ __forceinline__ void loop2_cg_body(const GridDev &g, const Vecs &V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb,
                                              double *block_sums, int *__restrict__ iters_out, const LhsIn &L, const LoopSums *__restrict__ Z) {
  __shared__ double P[FLHS ? kTileLds : (DIRECT ? kFdmLds : kCgLds)];
  const int slot = block_slot(g);
  if (slot < 0) return;
  if (ctl->state != kRun) return;
  const Loop2Args a{ctl->alpha, ctl->omega};
  // the two x buffers are v[X_] and v[XOPT] for the whole solve; which one holds x and which one receives the update is the struct's
  const int xc = ctl->xcur, xw = ctl_xwrite(*ctl);
  const double *const xin = xc ? V.v[XOPT] : V.v[X_];
  const int l = threadIdx.x;
  const double hq = block_h(g, slot), invh = 1 / hq;
  double r[8], acc[6] = {0, 0, 0, 0, 0, 0};
  const size_t bo = (size_t)slot * 512;
  // FLHS: zhat comes from the tile and v is computed from it -- streams 7 and 9 are not loaded
  enum { iQHAT, iY, iR0, iX, iPHAT, iQ, iWHAT, iZHAT, iT, iV, iS, iZ, NS };
  const double *const src[NS] = {V.v[QHAT] + bo, V.v[Y_] + bo, V.v[R0] + bo, xin + bo, V.v[PHAT] + bo, V.v[Q_] + bo, V.v[WHAT] + bo, V.v[ZHAT] + bo,
                                 V.v[T_] + bo, V.v[V_] + bo, V.v[S_] + bo, V.v[Z_] + bo};
  double *const oX = (xw ? V.v[XOPT] : V.v[X_]) + bo, *const oR = V.v[R_] + bo, *const oRH = V.v[RHAT] + bo, *const oW = V.v[W_] + bo, *const oV = V.v[V_] + bo;
  double in[2][NS + 1];
  const double *const xsrc = V.v[B_] + bo;
  double *const xdst = XTRA ? L.extra + (size_t)nb * 512 + bo : nullptr;
#define LOAD_PLANE(buf, off)                                                            \
  _Pragma("unroll") for (int i = 0; i < NS; ++i)                                        \
    if (!(FLHS && (i == iZHAT || i == iV))) in[buf][i] = NTL(src[i], off);              \
  if constexpr (XTRA == 2) in[buf][NS] = NTL(xsrc, off);
  TileIdx ix{0, 0, 0};
  LhsFix fx{};
  TileRegs tr;
  if constexpr (FLHS) {  // the tile's loads first, the first plane of the streams right behind them, then the tile goes to LDS
    fx = lhs_fix(L, ctl, slot, l, hq);
    tile_issue_own(slot, V.v[ZHAT], l, tr);
    ix = tile_idx(l);
  }
  if (L.prio == 1) __builtin_amdgcn_s_setprio(1); else if (L.prio == 2) __builtin_amdgcn_s_setprio(2); else if (L.prio == 3) __builtin_amdgcn_s_setprio(3);
  LOAD_PLANE(0, l)
  if constexpr (FLHS) {
    tile_issue_faces(g, slot, V.v[ZHAT], L.halo, l, tr);  // (before or behind the first plane: no measurable difference, profiles/r03)
    tile_commit(tr, P, l);
  }
#pragma unroll
  for (int zz = 0; zz < 8; ++zz) {  // second fused loop, 14503-14515
    const int j = zz * 64 + l;
    if (zz < 7) { LOAD_PLANE((zz + 1) & 1, j + 64) }
    const double *c = in[zz & 1];
    double zhat = c[iZHAT], v = c[iV];
    if constexpr (FLHS) {
      v = zz == 0 ? tile_lhs<0>(P, ix, zhat, hq, fx) : tile_lhs<1>(P + (zz - 1) * kTilePitch, ix, zhat, hq, fx);   // v = A zhat, 14489
      NTS(oV, j, v);                                                              // the next first loop streams it
    }
    const double qhat = c[iQHAT], y = c[iY], r0 = c[iR0];
    const double x = c[iX] + a.alpha * c[iPHAT] + a.omega * qhat;
    const double rv = c[iQ] - a.omega * y;
    const double rhat = qhat - a.omega * (c[iWHAT] - a.alpha * zhat);
    const double w = y - a.omega * (c[iT] - a.alpha * v);
    NTS(oX, j, x); NTS(oR, j, rv); NTS(oRH, j, rhat); NTS(oW, j, w);
    if constexpr (XTRA != 0) NTS(xdst, j, w);
    if constexpr (XTRA == 2) acc[5] += 0.0 * c[NS];
    acc[0] += r0 * rv;
    acc[1] += r0 * w;
    acc[2] += r0 * c[iS];
    acc[3] += r0 * c[iZ];
    acc[4] += rv * rv;   // norm_1
    acc[5] += r0 * r0;   // norm_2
    r[zz] = invh * w;
  }
#undef LOAD_PLANE
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double t = wave_sum(acc[i]);
    if constexpr (TOT) {
      if (l == 0) st_agent(block_dots + (size_t)i * nb + slot, t);
      if (i == 4 && l == 0) st_agent(block_dots + (size_t)6 * nb + slot, t);  // norm = the same sum as norm_1 (14512-14514)
    } else {
      if (l == 0) block_dots[(size_t)i * nb + slot] = t;
      if (i == 4 && l == 0) block_dots[(size_t)6 * nb + slot] = t;  // norm = the same sum as norm_1 (14512-14514)
    }
  }
  if constexpr (TOT) arrive<7>(Z->dots, slot, Z->then);  // the seven of 14546: complete while the block solves still run
  if (L.prio) __builtin_amdgcn_s_setprio(0);
  if constexpr (FLHS) __syncthreads();
  if constexpr (DIRECT) fdm_block<TOT>(g, slot, r, V.v[WHAT], L.invD, block_sums, P);
  else cg_block<FMA, false, EV, TOT>(g, slot, r, V.v[WHAT], block_sums, 0.0, 0.0, iters_out, P);  // what = M^-1 w, 14548
  if constexpr (TOT) if (Z->mean.vals) arrive<1>(Z->mean, slot, NoThen());  // sum(what h^3) for the mean-constraint row of t = A what
}
// WITHOUT the LHS inside (FLHS = false: multi-level meshes, the no_fuse_lhs A/B): held to 96 registers (2 of the 122 the body asks for
// are spilled, outside the CG loop) -> 5 wavefronts per SIMD: 3.63-3.70 ms instead of 3.75 at 512^3, 0.457-0.461 instead of 0.497 at
// 256^3 (profiles/r02/probe_fused_kernel_occupancy.jsonl).  The same test on the other side -- the first kernel or the stand-alone block
// CG held to 80 registers for 6 wavefronts -- loses (12-14 spills inside the loops: 5.3 ms instead of 3.88; CG 0.43 instead of 0.40).
// (With FLHS held to 96 it spills 30 registers inside the plane loop: the production kernel of uniform grids is k_loop2_cg_w4 below.)
template <bool FMA, int EV, bool FLHS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5)))
k_loop2_cg(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums, int *__restrict__ iters_out, LhsIn L, const LoopSums *__restrict__ Z) {
  loop2_cg_body<FMA, EV, FLHS>(g, V, ctl, block_dots, nb, block_sums, iters_out, L, Z);
}
// PRODUCTION on uniform grids (FLHS = true; what bench.py's `value` runs): the register allocation the compiler picks on its own, 128
// registers -> 4 wavefronts per SIMD, no spills.  (Also the "loop2_four_waves" A/B of the FLHS = false form.)
template <bool FMA, int EV, bool FLHS>
__global__ void __launch_bounds__(64) k_loop2_cg_w4(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums,
                                                    int *__restrict__ iters_out, LhsIn L, const LoopSums *__restrict__ Z) {
  loop2_cg_body<FMA, EV, FLHS>(g, V, ctl, block_dots, nb, block_sums, iters_out, L, Z);
}

#ifdef CUP3D_TESTING
// EXPERIMENT ("extra_streams" = XTRA): the production kernels of uniform grids with dummy streams added (see loop1_cg_body)
template <int XTRA>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))  // (left alone the compiler holds XTRA = 1, 2 to 94 registers, 5 wavefronts: another kernel)
k_loop1_cg_x(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums, int *__restrict__ iters_out, LhsIn L, const LoopSums *__restrict__ Z) {
  loop1_cg_body<true, kCgProduction, true, false, false, XTRA>(g, V, ctl, block_dots, nb, block_sums, iters_out, L, Z);
}
template <int XTRA>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_loop2_cg_x(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums, int *__restrict__ iters_out, LhsIn L, const LoopSums *__restrict__ Z) {
  loop2_cg_body<true, kCgProduction, true, false, false, XTRA>(g, V, ctl, block_dots, nb, block_sums, iters_out, L, Z);
}
// EXPERIMENT (single-reduction block CG, EV 32): the body asks for 130 registers; held to 128 for 4 wavefronts per SIMD
template <bool FMA, int EV, bool FLHS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_loop2_cg_w4f(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums, int *__restrict__ iters_out, LhsIn L, const LoopSums *__restrict__ Z) {
  loop2_cg_body<FMA, EV, FLHS>(g, V, ctl, block_dots, nb, block_sums, iters_out, L, Z);
}
#endif
// The two kernels of an iteration with the totals INSIDE (TOT; uniform grids, block CG): the early all-reduce over ranks (solve(): `early`).
// The second one is held to 128 registers (the compiler would take 136 -> 3 wavefronts per SIMD): one 8-byte value is parked in scratch
// before the plane loop and fetched back when the block CG starts, never inside a loop.
template <bool FMA, int EV>
__global__ void __launch_bounds__(64) k_loop1_cg_tot(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums, int *__restrict__ iters_out, LhsIn L,
                                                     const LoopSums *__restrict__ Z) {
  loop1_cg_body<FMA, EV, true, false, true>(g, V, ctl, block_dots, nb, block_sums, iters_out, L, Z);
}
template <bool FMA, int EV>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_loop2_cg_tot(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums, int *__restrict__ iters_out, LhsIn L, const LoopSums *__restrict__ Z) {
  loop2_cg_body<FMA, EV, true, false, true>(g, V, ctl, block_dots, nb, block_sums, iters_out, L, Z);
}

// block_solver 1: second loop + the direct block solve (`alt`)
template <bool FLHS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) k_loop2_fdm(GridDev g, Vecs V, const SolverCtl *__restrict__ ctl, double *block_dots, long nb, double *block_sums,
                                                  LhsIn L, const LoopSums *__restrict__ Z) {
  loop2_cg_body<true, 0, FLHS, true>(g, V, ctl, block_dots, nb, block_sums, nullptr, L, Z);
}

// DEFAULT totalling of a fused loop's per-block values: K sums of nb values each ([K][nb]) finished in one launch of 64 / 256 workgroups,
// the last one to arrive totals the partials (grid_sum_finish, tile.hpp).  MEAN: one more sum rides along -- the per-block sums of
// zhat h^3 / what h^3 the fused kernel left in mean_src; the total lands in ro.out[K], where the LHS application that follows takes its
// mean-constraint row from (no k_mean_finish launch, and over ranks no second all-reduce: the total travels with the dot products).
// step 1 / 2: one rank -- the last workgroup also steps the solver's scalar struct with the totals (ctl_step1 / ctl_step2) and, after
// the second loop, publishes it to the host's status ring; step 0: totals only (several ranks: the all-reduce comes first, k_ctl_step).
// (Round 5 measured the alternative -- the loop kernels totalling these values themselves, Arrive above -- on one GPU: the launches it saves
//  (16-27 us each) are paid back by the loop kernels (agent-scope stores whose completion a wavefront must wait for before it takes its
//  ticket, +2 % on the second kernel at 512^3, +7-9 % on the smaller kernels of a multi-level mesh): neutral at 256^3, a loss elsewhere.  So
//  this launch stays the default and the in-kernel totals serve what only they can do: the early all-reduce.)
struct CtlThen {
  SolverCtl *ctl; CtlSlot *ring; int step;
  __device__ __forceinline__ void operator()(const double *tot) const {
    if (step == 0 || ctl->state != kRun) return;  // (an iteration enqueued ahead of a stop / restart summed stale partials: dropped)
    SolverCtl c = *ctl;
    const unsigned it = c.seq;
    if (step == 1) ctl_step1(c, tot); else ctl_step2(c, tot);
    *ctl = c;
    if (step == 2) ctl_publish(ctl, ring, it);
  }
};
inline int sums_groups(int64_t nb) { return nb >= (1 << 17) ? 256 : 64; }  // (0.027 instead of 0.051 ms per launch at 512^3, 0.016 instead of 0.014 at 256^3: profiles/r03)
template <int K, bool MEAN>
__global__ void __launch_bounds__(256) k_sums_finish(const double *__restrict__ v, long nb, RedOut ro, const double *__restrict__ mean_src, CtlThen then) {
  static_assert(K + (MEAN ? 1 : 0) <= kRedDotsEnd - kRedDots, "the totals of a loop must fit the kRedDots range of Sim::d_red");
  double acc[K + (MEAN ? 1 : 0)];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double t = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (long)gridDim.x * 256) t += v[(size_t)k * nb + i];
    acc[k] = t;
  }
  if (MEAN) {
    double t = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nb; i += (long)gridDim.x * 256) t += mean_src[i];
    acc[K] = t;
  }
  grid_sum_finish<K + (MEAN ? 1 : 0)>(acc, ro, then);
}

// several ranks, all-reduce started EARLY (solve(): early): the communication stream holds this one-thread kernel in front of the
// all-reduce; it returns when the loop kernel's last block has left its vector phase and the totals are in device memory (DotsThen
// raises *flag to seq) -- one block-solve phase before that kernel ends, so the all-reduce and the recurrence step behind it run while
// the compute stream is still busy.  Bounded: if the flag never comes (a loop kernel that died), *fail is raised and the stream moves on.
__global__ void k_wait_totals(const SolverCtl *ctl, const unsigned *flag, unsigned seq, unsigned *fail, long long limit_ticks) {
  // an iteration enqueued ahead of a stop or a restart: its loop kernels return at once and nobody will raise the flag.  (The struct is
  // stepped on THIS stream only, k_ctl_step: what this kernel reads is what those loop kernels read.)
  if (ctl->state != kRun) return;
  const long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > limit_ticks) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
  }
}
__global__ void k_raise(unsigned *flag, unsigned seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// b = r = rhs, x = pres   (main.cpp:14408-14415)
__global__ void __launch_bounds__(256) k_solver_init(Vecs V, const double *__restrict__ rhs, const double *__restrict__ pres, long n) {
  GRID_STRIDE(j, n) { const double b = rhs[j]; V.v[B_][j] = b; V.v[R_][j] = b; V.v[X_][j] = pres[j]; }
}
// r0 = r - r0 ; r = r0   (14419-14422)
__global__ void __launch_bounds__(256) k_resid0(Vecs V, long n) {
  GRID_STRIDE(j, n) { const double d = V.v[R_][j] - V.v[R0][j]; V.v[R0][j] = d; V.v[R_][j] = d; }
}
// r0.r0, r0.w  (14436-14440 and 14578-14581)
__global__ void __launch_bounds__(256) k_dots_r0(Vecs V, long n, RedOut ro) {
  double acc[2] = {0, 0};
  GRID_STRIDE(j, n) { const double a = V.v[R0][j]; acc[0] += a * a; acc[1] += a * V.v[W_][j]; }
  grid_sum_finish<2>(acc, ro);
}
// first fused loop, k % 50 != 0   (14454-14464)
template <bool NT>
__global__ void __launch_bounds__(256) k_loop1(Vecs V, long n, double alpha, double beta, double omega, RedOut ro) {
  double acc[2] = {0, 0};
  GRID_STRIDE2(j, n) {
    const double2 rhat = LD2(V.v[RHAT]), w = LD2(V.v[W_]), shat0 = LD2(V.v[SHAT]), z0 = LD2(V.v[Z_]);
    const double2 phat = rhat + beta * (LD2(V.v[PHAT]) - omega * shat0);
    const double2 s = w + beta * (LD2(V.v[S_]) - omega * z0);
    const double2 shat = LD2(V.v[WHAT]) + beta * (shat0 - omega * LD2(V.v[ZHAT]));
    const double2 z = LD2(V.v[T_]) + beta * (z0 - omega * LD2(V.v[V_]));
    const double2 q = LD2(V.v[R_]) - alpha * s;
    const double2 qhat = rhat - alpha * shat;
    const double2 y = w - alpha * z;
    ST2(V.v[PHAT], phat); ST2(V.v[S_], s); ST2(V.v[SHAT], shat); ST2(V.v[Z_], z); ST2(V.v[Q_], q); ST2(V.v[QHAT], qhat); ST2(V.v[Y_], y);
    acc[0] = dot2(q, y, acc[0]);
    acc[1] = dot2(y, y, acc[1]);
  }
  grid_sum_finish<2>(acc, ro);
}
// k % 50 == 0 variants   (14467-14480)
__global__ void __launch_bounds__(256) k_loop1_phat(Vecs V, long n, double beta, double omega) {
  GRID_STRIDE(j, n) V.v[PHAT][j] = V.v[RHAT][j] + beta * (V.v[PHAT][j] - omega * V.v[SHAT][j]);
}
__global__ void __launch_bounds__(256) k_loop1_tail(Vecs V, long n, double alpha, RedOut ro) {
  double acc[2] = {0, 0};
  GRID_STRIDE(j, n) {
    const double q = V.v[R_][j] - alpha * V.v[S_][j];
    const double qhat = V.v[RHAT][j] - alpha * V.v[SHAT][j];
    const double y = V.v[W_][j] - alpha * V.v[Z_][j];
    V.v[Q_][j] = q; V.v[QHAT][j] = qhat; V.v[Y_][j] = y;
    acc[0] += q * y;
    acc[1] += y * y;
  }
  grid_sum_finish<2>(acc, ro);
}
// second fused loop, k % 50 != 0   (14503-14515)
template <bool NT>
__global__ void __launch_bounds__(256) k_loop2(Vecs V, long n, double alpha, double omega, RedOut ro) {
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  GRID_STRIDE2(j, n) {
    const double2 qhat = LD2(V.v[QHAT]), y = LD2(V.v[Y_]), r0 = LD2(V.v[R0]);
    const double2 x = LD2(V.xin) + alpha * LD2(V.v[PHAT]) + omega * qhat;
    const double2 r = LD2(V.v[Q_]) - omega * y;
    const double2 rhat = qhat - omega * (LD2(V.v[WHAT]) - alpha * LD2(V.v[ZHAT]));
    const double2 w = y - omega * (LD2(V.v[T_]) - alpha * LD2(V.v[V_]));
    ST2(V.v[X_], x); ST2(V.v[R_], r); ST2(V.v[RHAT], rhat); ST2(V.v[W_], w);
    acc[0] = dot2(r0, r, acc[0]);
    acc[1] = dot2(r0, w, acc[1]);
    acc[2] = dot2(r0, LD2(V.v[S_]), acc[2]);
    acc[3] = dot2(r0, LD2(V.v[Z_]), acc[3]);
    acc[4] = dot2(r, r, acc[4]);    // norm_1
    acc[5] = dot2(r0, r0, acc[5]);  // norm_2
    acc[6] = dot2(r, r, acc[6]);    // norm
  }
  grid_sum_finish<7>(acc, ro);
}
// k % 50 == 0 variants   (14518-14537)
__global__ void __launch_bounds__(256) k_loop2_x(Vecs V, long n, double alpha, double omega) {
  GRID_STRIDE(j, n) V.v[X_][j] = V.xin[j] + alpha * V.v[PHAT][j] + omega * V.v[QHAT][j];
}
__global__ void __launch_bounds__(256) k_true_resid(Vecs V, long n) {
  GRID_STRIDE(j, n) V.v[R_][j] = V.v[B_][j] - V.v[R_][j];
}
// q.y, y.y of the refresh (14478-14480) from the q and y that k_refresh<kRefZ> stored: k_loop1_tail's two sums, thread for thread and term for term
__global__ void __launch_bounds__(256) k_dots2(Vecs V, long n, RedOut ro) {
  double acc[2] = {0, 0};
  GRID_STRIDE(j, n) {
    const double q = V.v[Q_][j], y = V.v[Y_][j];
    acc[0] += q * y;
    acc[1] += y * y;
  }
  grid_sum_finish<2>(acc, ro);
}
__global__ void __launch_bounds__(256) k_dots7(Vecs V, long n, RedOut ro) {
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  GRID_STRIDE(j, n) {
    const double r0 = V.v[R0][j], r = V.v[R_][j];
    acc[0] += r0 * r;
    acc[1] += r0 * V.v[W_][j];
    acc[2] += r0 * V.v[S_][j];
    acc[3] += r0 * V.v[Z_][j];
    acc[4] += r * r;
    acc[5] += r0 * r0;
    acc[6] += r * r;
  }
  grid_sum_finish<7>(acc, ro);
}
// ------------------------------------------------------------------ the every-50th iteration, fused (uniform grids, one rank)
// Every 50th iteration the reference recomputes s, z and the true residual through _lhs instead of the recurrences (14465-14481,
// 14516-14538): four block-CG applications, six LHS applications and six pointwise passes -- rounds 1-4 ran them as sixteen launches
// (24 ms at 512^3, 3.4 times per step).  Here the chain is cut where it MUST be cut -- a block's LHS needs its neighbours' values of the
// vector the previous block solve produced -- and nowhere else: four launches of ONE kernel form (k_refresh), each a tile LHS of its input
// (tile_lhs: bit-identical to k_lhs), the pointwise work that consumes the result, and the block CG on it, by the wavefront that owns the
// block; plus the two pointwise updates that precede an LHS of their own output (k_refresh_pointwise), which also leave the block sums
// the mean-constraint row of that LHS needs -- in k_lhs's cell-to-thread mapping and order, so that the totals, and with them every
// vector of the refresh, are BIT-IDENTICAL to the unfused launches ("no_fuse_refresh", tests).  The per-block dot products it also leaves
// behind are NOT what solve() uses: the refresh's sums come from k_dots2 / k_dots7 over the stored vectors (refresh_iteration says why).
//   kRefS:  s = A phat ; shat = M^-1 s                                   (14468-14469)
//   kRefZ:  z = A shat ; q = r - alpha s, qhat = rhat - alpha shat, y = w - alpha z ; q.y, y.y ; zhat = M^-1 z   (14470-14480, 14488)
//   kRefR:  r = b - A x ; rhat = M^-1 r                                   (14519-14523)
//   kRefW:  w = A rhat ; the seven dot products ; what = M^-1 w           (14524-14537, 14548)
enum { kRefS = 0, kRefZ = 1, kRefR = 2, kRefW = 3 };
template <bool FMA, int EV, int KIND>
__global__ void __launch_bounds__(64) k_refresh(GridDev g, Vecs V, double alpha, const double *__restrict__ xnew, double *__restrict__ block_dots, long nb,
                                                double *__restrict__ block_sums, int *__restrict__ iters_out, LhsIn L) {
  __shared__ double P[kTileLds];
  const int slot = block_slot(g);
  if (slot < 0) return;
  const int l = threadIdx.x;
  const double hq = block_h(g, slot), invh = 1 / hq;
  const size_t bo = (size_t)slot * 512;
  const double *const tin = KIND == kRefS ? V.v[PHAT] : (KIND == kRefZ ? V.v[SHAT] : (KIND == kRefR ? xnew : V.v[RHAT]));
  double *const out = KIND == kRefS ? V.v[SHAT] : (KIND == kRefZ ? V.v[ZHAT] : (KIND == kRefR ? V.v[RHAT] : V.v[WHAT]));
  const LhsFix fx = lhs_fix(L, nullptr, slot, l, hq);  // (no flag to wait for: the total of the input was complete before the launch)
  const TileIdx ix = tile_idx(l);
  TileRegs tr;
  tile_issue_own(slot, tin, l, tr);
  tile_issue_faces(g, slot, tin, L.halo, l, tr);
  tile_commit(tr, P, l);
  double r[8], acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int zz = 0; zz < 8; ++zz) {
    const size_t j = bo + zz * 64 + l;
    double cc;
    const double lhs = zz == 0 ? tile_lhs<0>(P, ix, cc, hq, fx) : tile_lhs<1>(P + (zz - 1) * kTilePitch, ix, cc, hq, fx);
    if constexpr (KIND == kRefS) {
      NTS(V.v[S_], j, lhs);
      r[zz] = invh * lhs;
    } else if constexpr (KIND == kRefZ) {
      const double sv = NTL(V.v[S_], j), w = NTL(V.v[W_], j);
      const double q = NTL(V.v[R_], j) - alpha * sv;
      const double qhat = NTL(V.v[RHAT], j) - alpha * cc;   // cc = shat of this cell (the tile's centre)
      const double y = w - alpha * lhs;
      NTS(V.v[Z_], j, lhs); NTS(V.v[Q_], j, q); NTS(V.v[QHAT], j, qhat); NTS(V.v[Y_], j, y);
      acc[0] += q * y;
      acc[1] += y * y;
      r[zz] = invh * lhs;
    } else if constexpr (KIND == kRefR) {
      const double rv = NTL(V.v[B_], j) - lhs;
      NTS(V.v[R_], j, rv);
      r[zz] = invh * rv;
    } else {
      const double r0 = NTL(V.v[R0], j), rv = NTL(V.v[R_], j);
      NTS(V.v[W_], j, lhs);
      acc[0] += r0 * rv;
      acc[1] += r0 * lhs;
      acc[2] += r0 * NTL(V.v[S_], j);
      acc[3] += r0 * NTL(V.v[Z_], j);
      acc[4] += rv * rv;   // norm_1
      acc[5] += r0 * r0;   // norm_2
      r[zz] = invh * lhs;
    }
  }
  if constexpr (KIND == kRefZ) {
    const double d0 = wave_sum(acc[0]), d1 = wave_sum(acc[1]);
    if (l == 0) { block_dots[slot] = d0; block_dots[nb + slot] = d1; }
  } else if constexpr (KIND == kRefW) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double t = wave_sum(acc[i]);
      if (l == 0) block_dots[(size_t)i * nb + slot] = t;
      if (i == 4 && l == 0) block_dots[(size_t)6 * nb + slot] = t;  // norm = the same sum as norm_1
    }
  }
  __syncthreads();  // the tile is read no more: the block solve takes over its LDS
  cg_block<FMA, false, EV>(g, slot, r, out, block_sums, 0.0, 0.0, iters_out, P);
}
// the two pointwise updates whose OUTPUT the next kernel applies the LHS to -- WHICH 0: phat = rhat + beta (phat - omega shat) (14467),
// WHICH 1: x = x + alpha phat + omega qhat (14518) -- with the block sums of that output for the mean-constraint row: one workgroup per
// block, k_lhs's cell-to-thread mapping and its sum (stencil.hip), so that the total is the one launch_lhs would have formed
template <int WHICH>
__global__ void __launch_bounds__(256) k_refresh_pointwise(GridDev g, Vecs V, double a, double b, double *__restrict__ block_sums) {
  __shared__ double red[4];
  const int slot = block_slot(g);
  if (slot < 0) return;
  int x, y, z0, cell0;
  thread_cells(threadIdx.x, x, y, z0, cell0);
  double c[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const size_t j = (size_t)slot * 512 + k * 256 + cell0;
    if constexpr (WHICH == 0) c[k] = V.v[RHAT][j] + a * (V.v[PHAT][j] - b * V.v[SHAT][j]);
    else c[k] = V.xin[j] + a * V.v[PHAT][j] + b * V.v[QHAT][j];
    (WHICH == 0 ? V.v[PHAT] : V.v[X_])[j] = c[k];
  }
  if (block_sums) {
    const double h = block_h(g, slot), h3 = h * h * h;
    const double sum = group_sum<4>(c[0] * h3 + c[1] * h3, red);
    if (threadIdx.x == 0) block_sums[slot] = sum;
  }
}

__global__ void __launch_bounds__(256) k_copy(const double *__restrict__ src, double *__restrict__ dst, long n) {
  GRID_STRIDE(j, n) dst[j] = src[j];
}
__global__ void k_set_one(double *p, size_t i, double v) { p[i] = v; }
// cup3d_poisson_path_checksum: vector `vec` of block blockIdx.x, a function of (vec, level, global cell index) -- integer hashing and one
// exact scaling, so the bits are the same on every device and under every sharding; values in [-1, 1)
__global__ void __launch_bounds__(256) k_selfcheck_fill(double *__restrict__ v, int vec, const int32_t *__restrict__ index, const int32_t *__restrict__ level, int level0) {
  const int b = blockIdx.x;
  const unsigned lv = (unsigned)(level ? level[b] : level0);
  for (int c = threadIdx.x; c < 512; c += 256) {
    const unsigned gx = (unsigned)index[3 * b] * 8u + (c & 7), gy = (unsigned)index[3 * b + 1] * 8u + ((c >> 3) & 7), gz = (unsigned)index[3 * b + 2] * 8u + (c >> 6);
    unsigned hsh = gx * 73856093u ^ gy * 19349663u ^ gz * 83492791u ^ (unsigned)(vec + 1) * 2654435761u ^ (lv + 1u) * 40503u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13; hsh *= 3266489917u; hsh ^= hsh >> 16;
    // a smooth part (so that the block solve sees a right-hand side like the solver's) + the hashed part
    const double smooth = (double)((int)((gx + 2 * gy + 3 * gz + 5u * (unsigned)vec) & 63u) - 32) * (1.0 / 64.0);
    v[(size_t)b * 512 + c] = 0.5 * smooth + (double)((int)(hsh & 0xfffffu) - 0x80000) * (1.0 / 2097152.0);
  }
}
// lhs -= tmpV.u[0] ; pres = 0   (main.cpp:15090-15099)
__global__ void __launch_bounds__(256) k_sub_divp(double *__restrict__ lhs, const double *__restrict__ tmpV, double *__restrict__ pres, long n) {
  GRID_STRIDE(j, n) { lhs[j] -= tmpV[(j >> 9) * 1536 + (j & 511)]; pres[j] = 0; }
}
// sum(p*vv), sum(vv)   (15111-15121)
__global__ void __launch_bounds__(256) k_mean_dots(const double *__restrict__ p, long n, double vv, const double *__restrict__ hb,
                                                   RedOut ro) {
  double acc[2] = {0, 0};
  GRID_STRIDE(j, n) {
    if (hb) { const double h = hb[j >> 9]; vv = h * h * h; }
    acc[0] += p[j] * vv; acc[1] += vv;
  }
  grid_sum_finish<2>(acc, ro);
}
// p -= avg ; (p += pOld)   (15127-15145)
__global__ void __launch_bounds__(256) k_shift_mean(double *__restrict__ p, const double *__restrict__ pold, long n, double avg) {
  GRID_STRIDE(j, n) { double v = p[j] - avg; if (pold) v += pold[j]; p[j] = v; }
}

static unsigned vec_groups_simple(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g > 2048 ? 2048 : g);
}
static unsigned vec_groups(long n) {
  long g = (n + 255) / 256;
  // One 256-thread workgroup per CU: with 18 concurrent streams per loop, fewer in-flight wavefronts keep the DRAM pages of each
  // stream open longer -- measured at 512^3 (profiles/r01/probe_bicgstab_loops_512.jsonl): 2048 groups 3.50 / 3.42 ms for the two
  // fused loops, 512 groups 3.18 / 2.86, 256 groups 3.08 / 2.82 (6.3 / 6.1 TB/s, the copy ceiling of the chip).
  const int cap = debug_option("vec_groups") > 0 ? debug_option("vec_groups") : 256;  // tuning knob; <= Sim::max_groups
  return (unsigned)(g > cap ? cap : g);
}

// several ranks: the all-reduced totals (device) -> the pinned host mirror, then the sequence word the host spins on
__global__ void k_publish_totals(const double *__restrict__ d, int k, double *__restrict__ host, unsigned *flag, unsigned seq) {
  if ((int)threadIdx.x < k) host[threadIdx.x] = d[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct Reducer {
  Sim *s;
  // where the kernel that ends with grid_sum_finish puts its totals: d_red always; the pinned host mirror directly when no
  // all-reduce has to run in between
  bool direct() const { return !(s->grid->nranks > 1 || (debug_option("force_allreduce") && comm())); }
  // The totals reach the host through pinned memory followed by a sequence word that wait() spins on -- a few microseconds instead
  // of the wake-up latency of hipEventSynchronize, which at <= 256^3 per GPU (the 8-GPU share of the 512^3 workload) is what the LHS
  // enqueued behind the reduction no longer hides.  direct: written by the reducing kernel itself; several ranks: by
  // k_publish_totals behind the all-reduce on the communication stream
  RedOut out() {
    if (!direct()) return RedOut{s->d_partials, s->d_counters, s->d_red, nullptr, nullptr, 0u};
    return RedOut{s->d_partials, s->d_counters, s->d_red, s->h_red_dev, reinterpret_cast<unsigned *>(s->h_red_dev + 16), ++s->red_seq};
  }
  // the k totals are in d_red when the work enqueued so far completes: all-reduce (communication stream), start the read-back
  int begin(int k) {
    if (direct()) {
      CUP3D_HIP(hipEventRecord(s->ev_a, stream()));
      return CUP3D_OK;
    }
    // MPI_Iallreduce (14486, 14546): on the communication stream, so that the preconditioner + LHS enqueued next on the compute
    // stream overlap it; every RCCL call of the library is issued from that one stream, in the same order on all ranks
    hipStream_t cs = scalar_stream(s);
    if (cs != stream()) {
      CUP3D_HIP(hipEventRecord(s->ev_b, stream()));
      CUP3D_HIP(hipStreamWaitEvent(cs, s->ev_b, 0));
    }
    ProfileScope pc("comm_allreduce", cs);
    int rc = allreduce(s, s->d_red, k, false, cs);
    if (rc) return rc;
    hipLaunchKernelGGL(k_publish_totals, dim3(1), dim3(64), 0, cs, (const double *)s->d_red, k, s->h_red_dev, reinterpret_cast<unsigned *>(s->h_red_dev + 16), ++s->red_seq);
    CUP3D_HIP(hipGetLastError());
    CUP3D_HIP(hipEventRecord(s->ev_a, cs));
    return CUP3D_OK;
  }
  int wait() {
    const volatile unsigned *flag = reinterpret_cast<const volatile unsigned *>(s->h_red + 16);
    const unsigned want = s->red_seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 1; *flag != want; ++spin) {
      __builtin_ia32_pause();
      if ((spin & 0x3fff) == 0) {  // every ~16k polls: has the stream finished (or failed) without raising the flag?
        const hipError_t e = hipEventQuery(s->ev_a);
        if (e == hipSuccess) break;  // completed: the totals are in place (an event wait makes them visible as well)
        if (e != hipErrorNotReady) return hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    stats_host_wait(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return CUP3D_OK;
  }
};

static int ensure_vectors(Sim *s) {
  if (!s->d_block_dots) {
    int rc = sim_alloc(&s->d_block_dots, (size_t)7 * s->nb, s);
    if (rc) return rc;
  }
  // Arrive: [dots | mean] x (n1 + n2 + 1) counters + 2 flags; (7 + 1) x (n1 + n2) partial sums.  Each allocation is guarded by its OWN
  // pointer (one guard over four allocations left the later ones null for ever when one of them failed once)
  {
    const size_t n1 = ((size_t)s->nb + 63) / 64, n2 = (n1 + 63) / 64, ncnt = 2 * (n1 + n2 + 1) + 2;
    if (!s->d_arrive_sums) {
      int rc = sim_alloc(&s->d_arrive_sums, 8 * (n1 + n2) + 8, s);
      if (rc) return rc;
    }
    if (!s->d_arrive) {
      CUP3D_HIP(hipMalloc((void **)&s->d_arrive, ncnt * sizeof(unsigned)));
      const hipError_t e = hipMemsetAsync(s->d_arrive, 0, ncnt * sizeof(unsigned), stream());
      if (e != hipSuccess) { (void)hipFree(s->d_arrive); s->d_arrive = nullptr; return hip_fail(e, "hipMemsetAsync(d_arrive)", __FILE__, __LINE__); }
    }
    if (!s->d_loop_sums) CUP3D_HIP(hipMalloc((void **)&s->d_loop_sums, 2 * sizeof(LoopSums)));
    if (!s->h_early_fail_dev) {
      if (!s->h_early_fail) CUP3D_HIP(hipHostMalloc((void **)&s->h_early_fail, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
      *s->h_early_fail = 0;
      CUP3D_HIP(hipHostGetDevicePointer((void **)&s->h_early_fail_dev, s->h_early_fail, 0));
    }
  }
  if (!s->h_ctl_dev) {  // the solver's scalar struct (device) and the pinned ring its outcome reaches the host through: all three or none
    hipError_t e = s->d_ctl ? hipSuccess : hipMalloc(&s->d_ctl, sizeof(SolverCtl));
    if (e == hipSuccess && !s->h_ctl) {
      e = hipHostMalloc(&s->h_ctl, 4 * sizeof(CtlSlot), hipHostMallocMapped | hipHostMallocCoherent);
      if (e == hipSuccess) memset(s->h_ctl, 0, 4 * sizeof(CtlSlot));
    }
    if (e == hipSuccess) e = hipHostGetDevicePointer(&s->h_ctl_dev, s->h_ctl, 0);
    if (e != hipSuccess) {  // leave nothing half-built behind: the next solve starts over
      if (s->d_ctl) (void)hipFree(s->d_ctl);
      if (s->h_ctl) (void)hipHostFree(s->h_ctl);
      s->d_ctl = s->h_ctl = s->h_ctl_dev = nullptr;
      return hip_fail(e, "SolverCtl allocation", __FILE__, __LINE__);
    }
  }
  if (s->sv[0]) return CUP3D_OK;
  for (int i = 0; i < NVEC; ++i) {
    int rc = sim_alloc(&s->sv[i], (size_t)s->nvis * 512, s);  // nvis: the LHS reads ghost blocks of any of them on a rank view
    if (rc) return rc;
  }
  return CUP3D_OK;
}

#define LAUNCH_VEC(kern, ...) hipLaunchKernelGGL(kern, dim3(G), dim3(256), 0, stream(), __VA_ARGS__)
// few-stream pointwise kernels without partial sums: these want the wide grid (0.88 vs 1.26 ms for k_loop1_phat at 512^3)
#define LAUNCH_VEC_S(kern, ...) hipLaunchKernelGGL(kern, dim3(Gs), dim3(256), 0, stream(), __VA_ARGS__)
#define TRY(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

// the host waits for the outcome of a fused iteration: slot seq & 3 of the pinned ring carries `seq` once the struct is in place
static int wait_status(Sim *s, unsigned seq, SolverCtl *out) {
  const CtlSlot *sl = reinterpret_cast<const CtlSlot *>(s->h_ctl) + (seq & 3);
  const volatile unsigned *flag = &sl->seq;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 1; *flag != seq; ++spin) {
    __builtin_ia32_pause();
    if ((spin & 0x3fff) == 0) {  // every ~16k polls: have the streams drained (or failed) without the slot being written?
      if (s->h_early_fail && *(volatile unsigned *)s->h_early_fail) {
        set_error("BiCGSTAB: a device-side wait of the early all-reduce gave up after 10 s (%s): the loop kernel it waits for never delivered",
                  *(volatile unsigned *)s->h_early_fail == 1 ? "the communication stream waiting for the dot products" : "the corner block's wavefront waiting for the mean-constraint total");
        return CUP3D_ESTATE;
      }
      hipError_t e = hipStreamQuery(stream());
      if (e == hipSuccess && s->comm_stream) e = hipStreamQuery(s->comm_stream);
      if (e == hipSuccess) {
        if (*flag == seq) break;
        set_error("BiCGSTAB: the device finished without reporting iteration status %u", seq);
        return CUP3D_ESTATE;
      }
      if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery", __FILE__, __LINE__);
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  memcpy(out, (const void *)&sl->c, sizeof(SolverCtl));
  stats_host_wait(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return CUP3D_OK;
}

// helm != nullptr: DiffusionSolver::solve (main.cpp:6896-7146) -- the same routine on the Helmholtz operator of one velocity
// component, with no mean constraint and no cap on the breakdown restarts
// sc != nullptr: cup3d_poisson_path_checksum -- no solve; the work vectors are filled with a function of the global cell index, the
// scalars are set by hand and ONE iteration's kernels run as they do inside a solve; sc[NVEC] receives the vectors' wrapping sums
static int solve(Sim *s, const cup3d_poisson_params &P, cup3d_poisson_result *res, const HelmholtzOp *helm = nullptr, unsigned long long *sc = nullptr) {
  TRY(ensure_vectors(s));
  s->block_solver = P.block_solver;
  Vecs V;
  for (int i = 0; i < NVEC; ++i) V.v[i] = s->sv[i];
  double *const XB[2] = {s->sv[X_], s->sv[XOPT]};  // the two x buffers; SolverCtl::xcur / xopt say which is which
  const long N = s->nb * 512L;
  const unsigned G = vec_groups(N), Gs = vec_groups_simple(N);
  const int mc = helm ? 0 : P.mean_constraint;
  const double eps = 1e-100;
  Reducer red{s};
  auto LHS = [&](int in, int out) {  // _lhs, 9365-9393 / 6836-6875
    return helm ? launch_lhs_diffusion(s, V.v[in], V.v[out], *helm) : launch_lhs(s, V.v[in], V.v[out], mc);
  };
  auto PRE = [&](int in, int out) {  // _preconditioner, 9334-9364 / 6804-6835
    return helm ? launch_precond_diffusion(s, V.v[in], V.v[out], *helm) : launch_precond(s, V.v[in], V.v[out], mc > 0 && mc <= 2);
  };
  // vector loop + block CG in one launch (k_loop1_cg / k_loop2_cg): the production path of the pressure solver with the block CG
  // block_solver 1 (the direct block solve, `alt`): the same two kernels with fdm_block behind the loops ("no_fuse_fdm": A/B, round 3's launches)
  const bool fuse = !helm && (P.block_solver == 0 || P.block_solver == 2 || (P.block_solver == 1 && !debug_option("no_fuse_fdm"))) && !debug_option("no_fuse");
  const bool direct_solve = P.block_solver == 1;
  if (fuse && direct_solve) TRY(fdm_setup());
  const bool want_sums = mc > 0 && mc <= 2;
  const bool four_waves = debug_option("loop2_four_waves") != 0;  // A/B of the second fused kernel's occupancy
  double *const sums = want_sums ? s->d_partials + (size_t)s->max_groups * 8 : nullptr;
  int *const cg_it = profile_on() ? cg_iters_buffer(s) : nullptr;
  const GridDev gd = s->gdev();
  SolverCtl *const d_ctl = reinterpret_cast<SolverCtl *>(s->d_ctl);
  CtlSlot *const ring = reinterpret_cast<CtlSlot *>(s->h_ctl_dev);

  SolverCtl hs;  // the host's copy of the scalars: current whenever no fused iteration is in flight
  memset(&hs, 0, sizeof hs);
  hs.tol = P.tol; hs.tol_rel = P.tol_rel;
  hs.max_restarts = helm ? 0x7fffffff : P.max_restarts;
  hs.min_norm = 1e50;
  hs.xcur = 0; hs.xopt = -1;
  hs.state = kRun;
  // host-driven launches address x through V: v[X_] = the buffer the next update of x writes, xin = the one that holds x
  auto x_ptrs = [&]() { const int xw = ctl_xwrite(hs); V.v[X_] = XB[xw]; V.v[XOPT] = XB[1 - xw]; V.xin = XB[hs.xcur]; };
  x_ptrs();

  if (!sc) {
    if ((mc == 1 || mc > 2) && s->grid->corner_slot >= 0)  // rhs(0,0,0) = 0, 14404-14407
      hipLaunchKernelGGL(k_set_one, dim3(1), dim3(1), 0, stream(), s->lhs, (size_t)s->grid->corner_slot * 512, 0.0);
    { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC_S(k_solver_init, V, s->lhs, s->pres, N); }
    TRY(LHS(X_, R0));
    { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC_S(k_resid0, V, N); }
    TRY(PRE(R0, RHAT)); TRY(LHS(RHAT, W_)); TRY(PRE(W_, WHAT)); TRY(LHS(WHAT, T_));
    { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC(k_dots_r0, V, N, red.out()); }
    TRY(red.begin(2)); TRY(red.wait());
    hs.alpha = s->h_red[0] / (s->h_red[1] + eps);  // 14443
    hs.r0r_prev = s->h_red[0];
    hs.norm = hs.init_norm = std::sqrt(s->h_red[0]);
  }

  // the mean-constraint total of `what` for the first fused loop (FLHS): d_red[7] after a fused iteration (k_sums_finish<7, true>),
  // d_red[8] after a host-driven LHS(WHAT, T_) (k_mean_finish inside launch_lhs); of `zhat` for the second loop it is d_red[2]
  const double *what_total = s->d_red + kRedMeanLhs;
  bool first_after_host = true;  // the next fused iteration follows a host-driven LHS(WHAT, T_): its total is complete before the launch
  // the restart of 14567-14593 / 7096-7120 (the breakdown was detected, and counted, by ctl_step2)
  auto restart = [&]() -> int {
    { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC_S(k_copy, V.v[R_], V.v[R0], N); }
    TRY(PRE(R0, RHAT)); TRY(LHS(RHAT, W_));
    { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC(k_dots_r0, V, N, red.out()); }
    TRY(red.begin(2));
    TRY(PRE(W_, WHAT)); TRY(LHS(WHAT, T_));
    TRY(red.wait());
    hs.alpha = s->h_red[0] / (s->h_red[1] + eps);
    hs.r0r_prev = s->h_red[0];
    hs.beta = 0.0;
    hs.omega = 0.0;
    hs.state = kRun;
    what_total = s->d_red + kRedMeanLhs;
    first_after_host = true;
    return CUP3D_OK;
  };

  // one iteration driven by the host: every 50th one (s, z and the true residual recomputed through _lhs), and all of them for the
  // block solvers without a fused kernel and for the Helmholtz solves
  auto host_iteration = [&](int k) -> int {
    x_ptrs();
    if (k % 50 != 0) {
      ProfileScope ps("bicgstab_loop1");
      if (!debug_option("loops_no_nt")) LAUNCH_VEC(k_loop1<true>, V, N, hs.alpha, hs.beta, hs.omega, red.out());
      else LAUNCH_VEC(k_loop1<false>, V, N, hs.alpha, hs.beta, hs.omega, red.out());
    } else {
      { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC_S(k_loop1_phat, V, N, hs.beta, hs.omega); }
      TRY(LHS(PHAT, S_)); TRY(PRE(S_, SHAT)); TRY(LHS(SHAT, Z_));
      { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC(k_loop1_tail, V, N, hs.alpha, red.out()); }
    }
    TRY(red.begin(2));         // MPI_Iallreduce(2), 14486
    TRY(PRE(Z_, ZHAT));        // overlapped with the reduction read-back, 14488-14489
    TRY(LHS(ZHAT, V_));
    TRY(red.wait());
    ctl_step1(hs, s->h_red);   // 14493
    if (k % 50 != 0) {
      ProfileScope ps("bicgstab_loop2");
      if (!debug_option("loops_no_nt")) LAUNCH_VEC(k_loop2<true>, V, N, hs.alpha, hs.omega, red.out());
      else LAUNCH_VEC(k_loop2<false>, V, N, hs.alpha, hs.omega, red.out());
    } else {
      { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC_S(k_loop2_x, V, N, hs.alpha, hs.omega); }
      TRY(LHS(X_, R_));        // v[X_] is where x was just written
      { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC_S(k_true_resid, V, N); }
      TRY(PRE(R_, RHAT)); TRY(LHS(RHAT, W_));
      { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC(k_dots7, V, N, red.out()); }
    }
    TRY(red.begin(7));         // MPI_Iallreduce(7), 14546
    TRY(PRE(W_, WHAT));        // 14548-14549
    TRY(LHS(WHAT, T_));
    TRY(red.wait());
    ctl_step2(hs, s->h_red);   // 14558-14566, 14594-14601 (moves xcur to the buffer just written)
    if (hs.state == kRestart) TRY(restart());
    what_total = s->d_red + kRedMeanLhs;  // LHS(WHAT, T_) left sum(what h^3) there (k_mean_finish)
    first_after_host = true;
    return CUP3D_OK;
  };

  // the every-50th iteration as four launches of k_refresh + two of k_refresh_pointwise (see there): uniform grids, one rank, block CG
  const bool fuse_refresh = fuse && !sc && !s->grid->multilevel && s->grid->nranks == 1 && red.direct() && (P.block_solver == 0 || P.block_solver == 2) &&
                            !debug_option("no_fuse_lhs") && !debug_option("no_fuse_refresh");
  auto refresh_iteration = [&]() -> int {
    x_ptrs();
    const GridDev g = s->gdev();
    const dim3 GG(launch_groups(g));
    double *const bsum = want_sums ? sums : nullptr;
    const int lhs_mode = mc > 2 ? 3 : mc;
    const double *const tot = s->d_red + kRedMeanLhs;
    const LhsIn Lin{s->halo_recv, tot, lhs_mode, s->grid->corner_slot, 0, g_invD, nullptr, 0, nullptr};
    auto REF = [&](int kind) -> int {  // one launch of k_refresh; the total of its input's h^3-weighted sum is in *tot
      ProfileScope ps("bicgstab_refresh");
#define REF_ARGS g, V, hs.alpha, (const double *)V.v[X_], s->d_block_dots, (long)s->nb, bsum, cg_it, Lin
#define REF_LAUNCH(FMA_, EV_)                                                                                             \
      switch (kind) {                                                                                                     \
        case kRefS: hipLaunchKernelGGL((k_refresh<FMA_, EV_, kRefS>), GG, dim3(64), 0, stream(), REF_ARGS); break;          \
        case kRefZ: hipLaunchKernelGGL((k_refresh<FMA_, EV_, kRefZ>), GG, dim3(64), 0, stream(), REF_ARGS); break;          \
        case kRefR: hipLaunchKernelGGL((k_refresh<FMA_, EV_, kRefR>), GG, dim3(64), 0, stream(), REF_ARGS); break;          \
        default: hipLaunchKernelGGL((k_refresh<FMA_, EV_, kRefW>), GG, dim3(64), 0, stream(), REF_ARGS); break;             \
      }
      if (P.block_solver == 0) { REF_LAUNCH(true, kCgProduction) } else { REF_LAUNCH(false, 0) }
#undef REF_LAUNCH
#undef REF_ARGS
      CUP3D_HIP(hipGetLastError());
      s->sums_of = nullptr;
      s->mean_total_of = nullptr;
      return CUP3D_OK;
    };
    // The dot products of a refresh are summed by the kernels of the launch-by-launch form, in THEIR order (grid-stride partials: k_dots2 =
    // k_loop1_tail's two sums, k_dots7 itself), from the vectors k_refresh stored -- 0.35 + 1.27 ms per refresh for a solver that is bit for
    // bit the launch-by-launch one: the same omega, alpha, iterates, iteration and restart counts ("no_fuse_refresh", tests).  (k_refresh also
    // leaves per-block dot products behind, which this flow does not use: measured with them -- another order -- the iteration is 0.06 ms
    // cheaper and the driver's window takes 179 iterations per step instead of 171, profiles/r05.)
    auto totals = [&](int K) -> int {
      {
        ProfileScope ps("bicgstab_vector");
        if (K == 2) LAUNCH_VEC(k_dots2, V, N, red.out());
        else LAUNCH_VEC(k_dots7, V, N, red.out());
        CUP3D_HIP(hipGetLastError());
      }
      TRY(red.begin(K));
      if (want_sums) TRY(launch_mean_total(s));
      return CUP3D_OK;
    };
    { ProfileScope ps("bicgstab_vector"); hipLaunchKernelGGL(k_refresh_pointwise<0>, GG, dim3(256), 0, stream(), g, V, hs.beta, hs.omega, bsum); }  // phat, 14467
    if (want_sums) TRY(launch_mean_total(s));
    TRY(REF(kRefS));                       // s = A phat, shat = M^-1 s
    if (want_sums) TRY(launch_mean_total(s));
    TRY(REF(kRefZ));                       // z = A shat; q, qhat, y; q.y, y.y; zhat = M^-1 z
    TRY(totals(2));                        // MPI_Iallreduce(2), 14486
    if (want_sums) { s->mean_total_of = V.v[ZHAT]; s->mean_total = tot; }
    TRY(LHS(ZHAT, V_));                    // v = A zhat, 14489
    TRY(red.wait());
    ctl_step1(hs, s->h_red);               // 14493
    { ProfileScope ps("bicgstab_vector"); hipLaunchKernelGGL(k_refresh_pointwise<1>, GG, dim3(256), 0, stream(), g, V, hs.alpha, hs.omega, bsum); }  // x, 14518
    if (want_sums) TRY(launch_mean_total(s));
    TRY(REF(kRefR));                       // r = b - A x, rhat = M^-1 r
    if (want_sums) TRY(launch_mean_total(s));
    TRY(REF(kRefW));                       // w = A rhat; the seven dot products; what = M^-1 w
    TRY(totals(7));                        // MPI_Iallreduce(7), 14546
    TRY(red.wait());                       // (t = A what, 14549: the next iteration is a fused one and forms it from the tile of what)
    ctl_step2(hs, s->h_red);               // 14558-14566, 14594-14601 (moves xcur to the buffer just written)
    if (want_sums) { s->mean_total_of = V.v[WHAT]; s->mean_total = tot; }
    what_total = tot;                      // sum(what h^3): k_mean_finish left it in d_red[kRedMeanLhs], where the unfused LHS(WHAT, T_) leaves it
    first_after_host = true;
    if (hs.state == kRestart) TRY(restart());
    return CUP3D_OK;
  };

  // one fused iteration, enqueued without waiting for anything: both loop kernels take their scalars from d_ctl
  const bool direct = red.direct();
  // The loop kernels total their dot products themselves (Arrive): buffers of the two in-kernel sums of a loop
  const long n1 = (s->nb + 63) / 64, n2 = (n1 + 63) / 64;
  unsigned *const cnt = s->d_arrive;
  unsigned *const dots_flag = cnt + 2 * (n1 + n2 + 1), *const mean_flag = dots_flag + 1;
  auto arrive_args = [&](int which, const double *vals, double *out) {  // which 0: the dot products, 1: the block sums of the mean constraint
    double *gs = s->d_arrive_sums + (which ? 7 * (n1 + n2) : 0);
    unsigned *c = cnt + which * (n1 + n2 + 1);
    return Arrive{vals, gs, gs + (which ? 1 : 7) * n1, c, c + n1, c + n1 + n2, (long)s->nb, n1, n2, out, debug_option("arrive_light_release")};
  };
  // Several ranks.  DEFAULT: when the loop kernel has ended, one all-reduce of its K totals (+ the mean-constraint total: two RCCL calls
  // per iteration where the reference makes four) on the communication stream, the recurrence step behind it, the next loop kernel
  // waits -- both all-reduces of an iteration are exposed.  EARLY (process-per-rank transports, CUP3D_EARLY_ALLREDUCE / "early_allreduce"):
  // the all-reduce of the dot products starts when the LAST BLOCK LEAVES ITS VECTOR PHASE (k_wait_totals holds the communication
  // stream until DotsThen raises the flag), i.e. it runs under the block solves of the kernel's last round; the mean-constraint total,
  // which exists only when the kernel has ended, follows in an all-reduce of its own that nobody waits for on the host or on the
  // compute stream: the ONE wavefront that needs it -- the corner block's, mode 1 -- waits for its flag inside the next loop kernel.
  // (mode 2 adds the total to every cell: there the next kernel as a whole waits, and early buys only the first half.)
  // What MPI_Iallreduce hides behind the preconditioner in the reference (14486-14490, 14546-14550) is hidden behind it here again.
  // FLHS: v = A zhat and t = A what are formed inside the loop kernels (uniform grids; on multi-level meshes the LHS needs the
  // coarse/fine ghost slabs and the flux correction, so it stays a launch of its own)
  const bool flhs = fuse && !s->grid->multilevel && !debug_option("no_fuse_lhs");
  // read at every solve (a getenv against a solve of milliseconds): bench.py --gpus N times its alt_early_allreduce region in the same
  // process, after the default order has produced `value`
  const bool early_env = [] { const char *e = getenv("CUP3D_EARLY_ALLREDUCE"); return e && atoi(e) != 0; }();
  // (uniform grids only: on multi-level meshes the LHS is a launch of its own between the loops and reads the mean-constraint total itself)
  const bool early = !direct && flhs && (P.block_solver == 0 || P.block_solver == 2) && comm() && !virtual_ranks() && !host_transport() && scalar_stream(s) != stream() && (early_env || debug_option("early_allreduce"));
  // (a solve that ended in an error half way through a loop may have left tickets behind; the two flags behind the counters only ever grow;
  //  only the early kernels take tickets)
  if (early) CUP3D_HIP(hipMemsetAsync(cnt, 0, (size_t)(2 * (n1 + n2 + 1)) * sizeof(unsigned), stream()));
  static const long long tick_rate = [] {  // wall_clock64 ticks per millisecond (100 MHz on CDNA3 / CDNA4)
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) return (long long)khz;
    return 100000LL;
  }();
  auto after_loop = [&](int K, int step, unsigned seq) -> int {  // the totals of the loop just enqueued (-> all ranks) -> the struct
    if (!early) {  // DEFAULT: one launch totals the K (+1) block-wise sums; on one rank its last workgroup steps the struct as well
      ProfileScope ps("bicgstab_dots_finish");
      const RedOut ro{s->d_partials, s->d_counters, s->d_red, nullptr, nullptr, 0u};
      const CtlThen then{d_ctl, ring, direct ? step : 0};
      const dim3 SG(debug_option("sums_groups") > 0 ? debug_option("sums_groups") : sums_groups(s->nb));
      if (K == 2) {
        if (want_sums) hipLaunchKernelGGL((k_sums_finish<2, true>), SG, dim3(256), 0, stream(), s->d_block_dots, (long)s->nb, ro, sums, then);
        else hipLaunchKernelGGL((k_sums_finish<2, false>), SG, dim3(256), 0, stream(), s->d_block_dots, (long)s->nb, ro, (const double *)nullptr, then);
      } else {
        if (want_sums) hipLaunchKernelGGL((k_sums_finish<7, true>), SG, dim3(256), 0, stream(), s->d_block_dots, (long)s->nb, ro, sums, then);
        else hipLaunchKernelGGL((k_sums_finish<7, false>), SG, dim3(256), 0, stream(), s->d_block_dots, (long)s->nb, ro, (const double *)nullptr, then);
      }
      CUP3D_HIP(hipGetLastError());
    }
    if (direct) return CUP3D_OK;
    hipStream_t cs = scalar_stream(s);
    const int nmean = want_sums ? 1 : 0;
    if (early && s->nb == 0) {  // a rank without blocks launched nothing: its contribution is zero, and nobody but the host can raise the flag
      CUP3D_HIP(hipMemsetAsync(s->d_red + kRedDots, 0, (size_t)(K + 1) * sizeof(double), stream()));
      CUP3D_HIP(hipMemsetAsync(s->d_red + kRedEarlyMean, 0, 2 * sizeof(double), stream()));
      hipLaunchKernelGGL(k_raise, dim3(1), dim3(1), 0, stream(), dots_flag, seq * 2 + (unsigned)(step - 1));
    }
    if (!early) {
      if (cs != stream()) {
        CUP3D_HIP(hipEventRecord(s->ev_b, stream()));
        CUP3D_HIP(hipStreamWaitEvent(cs, s->ev_b, 0));
      }
      ProfileScope pc("comm_allreduce", cs);  // the all-reduce and the recurrence step behind it, as the communication stream sees them
      TRY(allreduce(s, s->d_red, K + nmean, false, cs));
      if (step == 1) hipLaunchKernelGGL(k_ctl_step<1>, dim3(1), dim3(1), 0, cs, d_ctl, (const double *)s->d_red, ring);
      else hipLaunchKernelGGL(k_ctl_step<2>, dim3(1), dim3(1), 0, cs, d_ctl, (const double *)s->d_red, ring);
      CUP3D_HIP(hipGetLastError());
      CUP3D_HIP(hipEventRecord(s->ev_a, cs));
      return CUP3D_OK;
    }
    hipLaunchKernelGGL(k_wait_totals, dim3(1), dim3(1), 0, cs, (const SolverCtl *)d_ctl, (const unsigned *)dots_flag, seq * 2 + (unsigned)(step - 1), s->h_early_fail_dev, 10000 * tick_rate);
    {
      ProfileScope pc("comm_allreduce", cs);  // (behind the wait for the totals: the all-reduce and the step, not the time the loop kernel took to deliver)
      TRY(allreduce(s, s->d_red, K, false, cs));
      if (step == 1) hipLaunchKernelGGL(k_ctl_step<1>, dim3(1), dim3(1), 0, cs, d_ctl, (const double *)s->d_red, ring);
      else hipLaunchKernelGGL(k_ctl_step<2>, dim3(1), dim3(1), 0, cs, d_ctl, (const double *)s->d_red, ring);
      CUP3D_HIP(hipGetLastError());
      CUP3D_HIP(hipEventRecord(s->ev_a, cs));
    }
    if (nmean) {  // the mean-constraint total: complete when the kernel ends
      CUP3D_HIP(hipEventRecord(s->ev_b, stream()));
      CUP3D_HIP(hipStreamWaitEvent(cs, s->ev_b, 0));
      ProfileScope pc("comm_allreduce_mean", cs);
      TRY(allreduce(s, s->d_red + kRedEarlyMean + (step - 1), 1, false, cs));
      hipLaunchKernelGGL(k_raise, dim3(1), dim3(1), 0, cs, mean_flag, seq * 2 + (unsigned)(step - 1));
      CUP3D_HIP(hipGetLastError());
      if (mc == 2) CUP3D_HIP(hipEventRecord(s->ev_m, cs));  // every cell takes the total: the next kernel as a whole waits (scalars_ready)
    }
    return CUP3D_OK;
  };
  // early: nothing orders the communication stream behind the loop kernel any more (that is the point), so it must be ordered behind the
  // k_ctl_set that starts a run of fused iterations explicitly: k_wait_totals and k_ctl_step read the struct (an unordered k_wait_totals saw
  // the state of the PREVIOUS run -- kDone -- returned at once, and the all-reduce took stale totals: 1000 iterations without converging)
  auto ctl_set_seen_by_comm = [&]() -> int {
    if (!early) return CUP3D_OK;
    CUP3D_HIP(hipEventRecord(s->ev_b, stream()));
    CUP3D_HIP(hipStreamWaitEvent(scalar_stream(s), s->ev_b, 0));
    return CUP3D_OK;
  };
  auto scalars_ready = [&]() -> int {  // the compute stream waits for the struct stepped on the communication stream
    if (!direct && scalar_stream(s) != stream()) {
      ProfileScope pw("comm_exposed_scalar_wait");  // compute stream idle until the all-reduced scalars are stepped (the exposed part of the all-reduce)
      CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_a, 0));
      if (early && want_sums && mc == 2) CUP3D_HIP(hipStreamWaitEvent(stream(), s->ev_m, 0));
    }
    return CUP3D_OK;
  };
  // A/B, test builds only ("fuse_lhs_ml"): on a multi-level mesh (one rank) the blocks none of whose six faces is a coarse/fine interface
  // take the FLHS kernels too, and only the interface blocks keep k_lhs + ghost slabs + flux correction (launch_lhs on the interface
  // list).  Bit-identical t and v -- and NOT faster: each loop kernel becomes two launches (plain list, interface list), which costs what
  // the smaller k_lhs saves (1.574 -> 1.567 ms per iteration on the 39 369-block mesh of bench.py --amr, profiles/r04).  Measured, kept
  // out of the production path.
  const bool flhs_ml = fuse && s->grid->multilevel && s->grid->nranks == 1 && s->n_plain > 0 && !debug_option("no_fuse_lhs") && debug_option("fuse_lhs_ml");
  const bool split = s->grid->nranks > 1;
  // early: the two loops' LoopSums, constant over this solve, in device memory
  if (early) {
    LoopSums Z[2];
    for (int which = 1; which <= 2; ++which) {
      const int K = which == 1 ? 2 : 7;
      double *const mean_out = early ? s->d_red + kRedEarlyMean + (which - 1) : s->d_red + kRedDots + K;
      Z[which - 1].dots = arrive_args(0, s->d_block_dots, s->d_red + kRedDots);
      Z[which - 1].mean = arrive_args(1, want_sums ? sums : nullptr, mean_out);
      // sc (the path checksum): totals only, nothing stepped
      Z[which - 1].then = DotsThen{d_ctl, ring, which, 0, sc ? nullptr : dots_flag};  // (over ranks the struct is stepped behind the all-reduce, k_ctl_step)
    }
    hipLaunchKernelGGL(k_set_loop_sums, dim3(1), dim3(1), 0, stream(), s->d_loop_sums, Z[0], Z[1]);
    CUP3D_HIP(hipGetLastError());
  }
  const LoopSums *const dZ = s->d_loop_sums;
  auto launch_loop = [&](int which, const double *u, const LhsIn &L) -> int {  // one loop kernel; over ranks: inner blocks while u's face slabs travel, then the rest
    const LoopSums *const Z = dZ + (which - 1);
    if (flhs && split) TRY(halo_begin(s, u, 1, 1));
    for (int pass = 0; pass < ((flhs && split) || flhs_ml ? 2 : 1); ++pass) {
      const GridDev gp = flhs_ml ? (pass == 0 ? s->gdev_list(s->d_plain_list, s->n_plain) : s->gdev_list(s->d_iface_list, s->n_iface))
                                 : (flhs && split ? s->gdev(pass == 1, pass == 0) : gd);
      const bool fl = flhs || (flhs_ml && pass == 0);  // does this launch form the LHS itself?
      if (pass == 1 && !flhs_ml) TRY(halo_finish(s));
      if (gp.nblocks == 0) continue;
      ProfileScope ps(direct_solve ? (which == 1 ? "bicgstab_loop1_fdm" : "bicgstab_loop2_fdm") : (which == 1 ? "bicgstab_loop1_cg" : "bicgstab_loop2_cg"));
      const dim3 GG(launch_groups(gp)), BB(64);
#define LOOP_ARGS gp, V, (const SolverCtl *)d_ctl, s->d_block_dots, (long)s->nb, sums, cg_it, L, Z
      if (direct_solve) {
#define FDM_ARGS gp, V, (const SolverCtl *)d_ctl, s->d_block_dots, (long)s->nb, sums, L, Z
        if (which == 1 && fl) hipLaunchKernelGGL(k_loop1_fdm<true>, GG, BB, 0, stream(), FDM_ARGS);
        else if (which == 1) hipLaunchKernelGGL(k_loop1_fdm<false>, GG, BB, 0, stream(), FDM_ARGS);
        else if (fl) hipLaunchKernelGGL(k_loop2_fdm<true>, GG, BB, 0, stream(), FDM_ARGS);
        else hipLaunchKernelGGL(k_loop2_fdm<false>, GG, BB, 0, stream(), FDM_ARGS);
#undef FDM_ARGS
      } else if (which == 1) {
#ifdef CUP3D_TESTING  // EXPERIMENT: the single-reduction block CG behind the loops (cg_variant 8 + 32; uniform grids, one rank)
        if (P.block_solver == 0 && fl && !early && debug_option("cg_variant") == 40) hipLaunchKernelGGL((k_loop1_cg<true, 32, true>), GG, BB, 0, stream(), LOOP_ARGS);
#define FUSED_EV(E) else if (P.block_solver == 0 && fl && !early && debug_option("fused_cg_ev") == (E) + 1) hipLaunchKernelGGL((k_loop1_cg<true, E, true>), GG, BB, 0, stream(), LOOP_ARGS);
        FUSED_EV(0) FUSED_EV(2) FUSED_EV(4) FUSED_EV(8) FUSED_EV(10) FUSED_EV(12) FUSED_EV(14) FUSED_EV(70)   // (70 = production + rounds 1-5's wave sum) A/B of the block CG's evaluation behind the loops: option value = EV + 1
#undef FUSED_EV
        else if (P.block_solver == 0 && fl && !early && L.extra && debug_option("extra_streams") == 1) hipLaunchKernelGGL(k_loop1_cg_x<1>, GG, BB, 0, stream(), LOOP_ARGS);
        else if (P.block_solver == 0 && fl && !early && L.extra && debug_option("extra_streams") == 2) hipLaunchKernelGGL(k_loop1_cg_x<2>, GG, BB, 0, stream(), LOOP_ARGS);
        else
#endif
        if (early && P.block_solver == 0) hipLaunchKernelGGL((k_loop1_cg_tot<true, kCgProduction>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (early) hipLaunchKernelGGL((k_loop1_cg_tot<false, 0>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (P.block_solver == 0 && fl && (debug_option("loop1_five_waves") ? debug_option("loop1_five_waves") == 1 : gp.nblocks >= kFiveWavesFrom))  // (A/B: 1 = always, 2 = never)
          hipLaunchKernelGGL((k_loop1_cg_w5<true, kCgProduction, true>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (P.block_solver == 0 && fl) hipLaunchKernelGGL((k_loop1_cg<true, kCgProduction, true>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (P.block_solver == 0) hipLaunchKernelGGL((k_loop1_cg<true, kCgProduction, false>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (fl) hipLaunchKernelGGL((k_loop1_cg<false, 0, true>), GG, BB, 0, stream(), LOOP_ARGS);
        else hipLaunchKernelGGL((k_loop1_cg<false, 0, false>), GG, BB, 0, stream(), LOOP_ARGS);
      } else {
        // with the LHS inside, the second loop asks for 128 registers: 4 wavefronts per SIMD without spills (k_loop2_cg_w4; held to 96 it
        // spills 30 registers inside the plane loop), and the 7.5 KB tile needs 16 wavefronts per CU or fewer anyway
#ifdef CUP3D_TESTING  // A/B of the occupancy, test builds only: with the LHS inside at 96 registers (30 spilled); without it at 4 wavefronts
        if (P.block_solver == 0 && fl && !early && debug_option("cg_variant") == 40) hipLaunchKernelGGL((k_loop2_cg_w4f<true, 32, true>), GG, BB, 0, stream(), LOOP_ARGS);
#define FUSED_EV(E) else if (P.block_solver == 0 && fl && !early && debug_option("fused_cg_ev") == (E) + 1) hipLaunchKernelGGL((k_loop2_cg_w4f<true, E, true>), GG, BB, 0, stream(), LOOP_ARGS);
        FUSED_EV(0) FUSED_EV(2) FUSED_EV(4) FUSED_EV(8) FUSED_EV(10) FUSED_EV(12) FUSED_EV(14) FUSED_EV(70)
#undef FUSED_EV
        else if (P.block_solver == 0 && fl && !early && L.extra && debug_option("extra_streams") == 1) hipLaunchKernelGGL(k_loop2_cg_x<1>, GG, BB, 0, stream(), LOOP_ARGS);
        else if (P.block_solver == 0 && fl && !early && L.extra && debug_option("extra_streams") == 2) hipLaunchKernelGGL(k_loop2_cg_x<2>, GG, BB, 0, stream(), LOOP_ARGS);
        else if (P.block_solver == 0 && fl && debug_option("loop2_flhs_five_waves")) hipLaunchKernelGGL((k_loop2_cg<true, kCgProduction, true>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (P.block_solver == 0 && !fl && four_waves) hipLaunchKernelGGL((k_loop2_cg_w4<true, kCgProduction, false>), GG, BB, 0, stream(), LOOP_ARGS);
        else
#endif
        if (early && P.block_solver == 0) hipLaunchKernelGGL((k_loop2_cg_tot<true, kCgProduction>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (early) hipLaunchKernelGGL((k_loop2_cg_tot<false, 0>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (P.block_solver == 0 && fl) hipLaunchKernelGGL((k_loop2_cg_w4<true, kCgProduction, true>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (P.block_solver == 0) hipLaunchKernelGGL((k_loop2_cg<true, kCgProduction, false>), GG, BB, 0, stream(), LOOP_ARGS);
        else if (fl) hipLaunchKernelGGL((k_loop2_cg_w4<false, 0, true>), GG, BB, 0, stream(), LOOP_ARGS);
        else hipLaunchKernelGGL((k_loop2_cg<false, 0, false>), GG, BB, 0, stream(), LOOP_ARGS);
      }
#undef LOOP_ARGS
      CUP3D_HIP(hipGetLastError());
    }
    return CUP3D_OK;
  };
  if (sc) {
    if (!fuse) { set_error("cup3d_poisson_path_checksum: block_solver %d has no fused loop kernels (0, 1 and 2 do)", P.block_solver); return CUP3D_EINVAL; }
    // every vector = a function of (vector, level, global cell index): the same cells hold the same bits however the blocks are
    // spread over ranks.  alpha, beta, omega by hand: no dot product (whose rounding depends on the partition) enters the update.
    int32_t *d_index = nullptr, *d_level = nullptr;
    const size_t nidx = (size_t)s->nb * 3;
    CUP3D_HIP(hipMalloc((void **)&d_index, nidx * sizeof(int32_t)));
    hipError_t e = hipMemcpyAsync(d_index, s->grid->index.data(), nidx * sizeof(int32_t), hipMemcpyHostToDevice, stream());
    if (e == hipSuccess && s->grid->multilevel) {
      e = hipMalloc((void **)&d_level, (size_t)s->nb * sizeof(int32_t));
      if (e == hipSuccess) e = hipMemcpyAsync(d_level, s->grid->blevel.data(), (size_t)s->nb * sizeof(int32_t), hipMemcpyHostToDevice, stream());
    }
    auto body = [&]() -> int {
      if (e != hipSuccess) return hip_fail(e, "cup3d_poisson_path_checksum tables", __FILE__, __LINE__);
      for (int i = 0; i < NVEC; ++i) hipLaunchKernelGGL(k_selfcheck_fill, dim3((unsigned)s->nb), dim3(256), 0, stream(), s->sv[i], i, (const int32_t *)d_index, (const int32_t *)d_level, s->grid->level);
      hs.alpha = 0.75; hs.beta = 0.5; hs.omega = 0.625;  // exact in binary
      hs.xcur = 0; hs.xopt = -1; hs.state = kRun;
      hipLaunchKernelGGL(k_ctl_set, dim3(1), dim3(1), 0, stream(), d_ctl, hs);
      hipLaunchKernelGGL(k_set_one, dim3(1), dim3(1), 0, stream(), s->d_red, (size_t)kRedMeanLhs, 0.375);  // the mean-constraint total, by hand as well
      CUP3D_HIP(hipGetLastError());
      V.v[X_] = XB[0]; V.v[XOPT] = XB[1]; V.xin = XB[0];
      // with the LHS inside the loop kernels the mean-constraint row takes the hand-set total; the stand-alone LHS would reduce the
      // field for it (partition-dependent rounding), so there the row is left out: mode 0 gives the same bits on both routes
      const int lhs_mode = flhs ? (mc > 2 ? 3 : mc) : (flhs_ml ? 0 : -1);
      const double *total = s->d_red + kRedMeanLhs;
      auto LHS0 = [&](int in, int out) {  // (multi-level, one rank: the interface blocks only -- the plain ones form theirs in the loop kernels)
        return flhs_ml ? launch_lhs(s, V.v[in], V.v[out], 0, s->d_iface_list, s->n_iface) : launch_lhs(s, V.v[in], V.v[out], 0);
      };
      if (!flhs) TRY(LHS0(WHAT, T_));
      TRY(launch_loop(1, V.v[WHAT], LhsIn{s->halo_recv, total, lhs_mode, s->grid->corner_slot, kLoopPrio, g_invD, nullptr, 0, nullptr}));
      if (!flhs) TRY(LHS0(ZHAT, V_));
      TRY(launch_loop(2, V.v[ZHAT], LhsIn{s->halo_recv, total, lhs_mode, s->grid->corner_slot, kLoopPrio, g_invD, nullptr, 0, nullptr}));
      for (int i = 0; i < NVEC; ++i) TRY(checksum_array(s, s->sv[i], N, &sc[i]));
      return CUP3D_OK;
    };
    const int rc = body();
    (void)hipStreamSynchronize(stream());
    if (d_index) (void)hipFree(d_index);
    if (d_level) (void)hipFree(d_level);
    return rc;
  }
  auto enqueue_fused = [&](unsigned seq) -> int {
    V.v[X_] = XB[0]; V.v[XOPT] = XB[1];  // fixed roles: the kernels pick by SolverCtl::xcur / xopt
    const int lhs_mode = !(flhs || flhs_ml) ? -1 : (mc > 2 ? 3 : mc);
    const int prio = debug_option("loop_prio") ? debug_option("loop_prio") - 1 : kLoopPrio;  // tuning: option value = priority + 1
    // the mean-constraint row inside the kernels belongs to the corner block only if that block forms its own LHS
    const int corner_in = (flhs || (flhs_ml && s->corner_is_plain)) ? s->grid->corner_slot : -1;
    auto LHS_IFACE = [&](int in, int out) { return launch_lhs(s, V.v[in], V.v[out], mc, s->d_iface_list, s->n_iface); };
    // early all-reduce: the mean-constraint totals arrive in slots of their own behind a flag (after_loop); `wait_seq` = what the flag
    // must have reached before the corner block's wavefront may read the total
    const bool em = early && want_sums;
    double *const xtra = debug_option("extra_streams") && !helm ? s->tmpV : nullptr;  // EXPERIMENT (testing build): tmpV is idle during the pressure solve
    const double *const total1 = em && !first_after_host ? s->d_red + kRedEarlyMean + 1 : what_total;
    const double *const total2 = em ? s->d_red + kRedEarlyMean : s->d_red + kRedDots + 2;
    TRY(launch_loop(1, V.v[WHAT], LhsIn{s->halo_recv, total1, lhs_mode, corner_in, prio, g_invD, em && !first_after_host ? mean_flag : nullptr, 1, s->h_early_fail_dev, xtra}));  // (t = A what,) loop 1, zhat = M^-1 z
    s->sums_of = want_sums ? V.v[ZHAT] : nullptr;
    TRY(after_loop(2, 1, seq));
    if (want_sums) { s->mean_total_of = V.v[ZHAT]; s->mean_total = total2; }
    if (flhs_ml) TRY(LHS_IFACE(ZHAT, V_));
    else if (!flhs) TRY(LHS(ZHAT, V_));
    TRY(scalars_ready());
    TRY(launch_loop(2, V.v[ZHAT], LhsIn{s->halo_recv, total2, lhs_mode, corner_in, prio, g_invD, em ? mean_flag : nullptr, 2, s->h_early_fail_dev, xtra}));  // (v = A zhat,) loop 2, what = M^-1 w
    s->sums_of = want_sums ? V.v[WHAT] : nullptr;
    TRY(after_loop(7, 2, seq));
    what_total = em ? s->d_red + kRedEarlyMean + 1 : s->d_red + kRedDots + 7;
    first_after_host = false;
    if (want_sums) { s->mean_total_of = V.v[WHAT]; s->mean_total = what_total; }
    if (flhs_ml) TRY(LHS_IFACE(WHAT, T_));  // t of the interface blocks for the next first loop; the others form theirs in that kernel
    else if (!flhs) TRY(LHS(WHAT, T_));
    TRY(scalars_ready());
    return CUP3D_OK;
  };

  int k = 0;
  while (k < P.max_iter && hs.state != kDone) {
    if (!fuse || k % 50 == 0) {
      if (fuse_refresh) TRY(refresh_iteration()); else
      TRY(host_iteration(k));
      ++k;
      continue;
    }
    // a run of fused iterations, up to the next multiple of 50: the host stays one iteration ahead of the device
    hs.state = kRun;
    hs.seq = s->ctl_seq + 1;  // the number the next enqueued iteration gets
    hipLaunchKernelGGL(k_ctl_set, dim3(1), dim3(1), 0, stream(), d_ctl, hs);
    TRY(ctl_set_seen_by_comm());
    int enq = k;  // next iteration to enqueue; k = next iteration whose outcome the host has not seen
    unsigned seq_of[2] = {0, 0};
    for (;;) {
      while (enq < P.max_iter && enq % 50 != 0 && enq - k < 2) {
        seq_of[enq & 1] = ++s->ctl_seq;
        TRY(enqueue_fused(seq_of[enq & 1]));
        ++enq;
      }
      if (k == enq) break;  // nothing in flight: a host iteration is next, or the cap is reached
      TRY(wait_status(s, seq_of[k & 1], &hs));
      ++k;
      if (hs.state == kDone) break;
      if (hs.state == kRestart) {  // what was enqueued ahead returned at once (state != kRun on the device)
        enq = k;
        x_ptrs();
        TRY(restart());
        hs.seq = s->ctl_seq + 1;
        hipLaunchKernelGGL(k_ctl_set, dim3(1), dim3(1), 0, stream(), d_ctl, hs);
        TRY(ctl_set_seen_by_comm());
      }
    }
  }
  const bool use_xopt = hs.xopt >= 0;
  { ProfileScope ps("bicgstab_vector"); LAUNCH_VEC_S(k_copy, XB[use_xopt ? hs.xopt : hs.xcur], s->pres, N); }  // 14605-14615
  CUP3D_HIP(hipGetLastError());
  stats_solver_iterations(hs.iter);
  if (res) {
    res->iterations = hs.iter;
    res->restarts = hs.restarts;
    res->norm0 = hs.init_norm;
    res->norm = hs.norm;
    res->used_xopt = use_xopt;
  }
  return CUP3D_OK;
}

int solve_helmholtz(Sim *s, const cup3d_poisson_params &P, cup3d_poisson_result *res, const HelmholtzOp &op) { return solve(s, P, res, &op); }

}  // namespace cup3d

using namespace cup3d;

extern "C" {

__attribute__((visibility("hidden"))) int cup3d_grad_p_update(cup3d_sim_t *h, double dt);  // stencil.hip; not exported

void cup3d_poisson_default_params(cup3d_poisson_params *p) {
  if (!p) return;
  p->tol = 1e-6; p->tol_rel = 1e-4; p->mean_constraint = 1; p->max_iter = 1000; p->max_restarts = 100; p->block_solver = 0;
}

// MEASUREMENT SUPPORT: CG iterations of the last block-CG launch made while cup3d_profile_enable(1) was on, summed over the blocks
int cup3d_profile_block_cg_iterations(cup3d_sim_t *h, long *total, long *nblocks) {
  if (!h || !total || !nblocks) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  *total = 0;
  *nblocks = 0;
  if (!s->d_cg_iters) return CUP3D_OK;
  std::vector<int> v((size_t)s->nb);
  CUP3D_HIP(hipMemcpy(v.data(), s->d_cg_iters, v.size() * sizeof(int), hipMemcpyDeviceToHost));
  long t = 0;
  for (int x : v) t += x;
  *total = t;
  *nblocks = (long)s->nb;
  return CUP3D_OK;
}

// TEST SUPPORT (no GPU needed): the scalar recurrences of the solver, host side of the one pair of functions the device runs too.
// io[16] = alpha, beta, omega, r0r_prev, norm, init_norm, min_norm, tol, tol_rel, state, restarts, max_restarts, xcur, xopt, iter, (unused);
// step 1: totals[2] = q.y, y.y (main.cpp:14493); step 2: totals[7] = r0.r, r0.w, r0.s, r0.z, |r|^2 (norm_1), |r0|^2 (norm_2), |r|^2 (14558-14601)
#ifdef CUP3D_TESTING  // test / tuning support: not in the release library at all
int cup3d_debug_ctl_step(int step, double *io, const double *totals) {
  if (!io || !totals || (step != 1 && step != 2)) return CUP3D_EINVAL;
  SolverCtl c;
  c.alpha = io[0]; c.beta = io[1]; c.omega = io[2]; c.r0r_prev = io[3]; c.norm = io[4]; c.init_norm = io[5]; c.min_norm = io[6];
  c.tol = io[7]; c.tol_rel = io[8]; c.state = (int)io[9]; c.restarts = (int)io[10]; c.max_restarts = (int)io[11];
  c.xcur = (int)io[12]; c.xopt = (int)io[13]; c.iter = (int)io[14]; c.seq = 0;
  if (step == 1) ctl_step1(c, totals); else ctl_step2(c, totals);
  io[0] = c.alpha; io[1] = c.beta; io[2] = c.omega; io[3] = c.r0r_prev; io[4] = c.norm; io[5] = c.init_norm; io[6] = c.min_norm;
  io[9] = c.state; io[10] = c.restarts; io[12] = c.xcur; io[13] = c.xopt; io[14] = c.iter;
  return CUP3D_OK;
}
#endif

// TEST SUPPORT: see k_debug_wave_sum (in64 -> out128, host arrays)
#ifdef CUP3D_TESTING  // test / tuning support: not in the release library at all
int cup3d_debug_wave_sum(const double *in64, double *out128) {
  if (!in64 || !out128) return CUP3D_EINVAL;
  double *d = nullptr;
  CUP3D_HIP(hipMalloc((void **)&d, 192 * sizeof(double)));
  hipError_t e = hipMemcpy(d, in64, 64 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_debug_wave_sum, dim3(1), dim3(64), 0, stream(), d, d + 64);
    e = hipStreamSynchronize(stream());
    if (e == hipSuccess) e = hipMemcpy(out128, d + 64, 128 * sizeof(double), hipMemcpyDeviceToHost);
  }
  hipFree(d);
  if (e != hipSuccess) return hip_fail(e, "cup3d_debug_wave_sum", __FILE__, __LINE__);
  return CUP3D_OK;
}
#endif

int cup3d_preconditioner(cup3d_sim_t *h, int block_solver) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  s->block_solver = block_solver;
  return launch_precond(s, s->pres, s->pres, false);  // in place: each wavefront reads its block before writing it
}

int cup3d_poisson_path_checksum(cup3d_sim_t *h, int block_solver, int mean_constraint, unsigned long long *sums) {
  if (!h || !sums) return CUP3D_EINVAL;
  cup3d_poisson_params d;
  cup3d_poisson_default_params(&d);
  d.block_solver = block_solver;
  d.mean_constraint = mean_constraint;
  return solve(reinterpret_cast<Sim *>(h), d, nullptr, nullptr, sums);
}

int cup3d_poisson_solve(cup3d_sim_t *h, const cup3d_poisson_params *p, cup3d_poisson_result *r) {
  if (!h) return CUP3D_EINVAL;
  cup3d_poisson_params d;
  cup3d_poisson_default_params(&d);
  return solve(reinterpret_cast<Sim *>(h), p ? *p : d, r);
}

int cup3d_pressure_project(cup3d_sim_t *h, double dt, int step, const cup3d_poisson_params *pp, cup3d_poisson_result *r) {
  if (!h) return CUP3D_EINVAL;
  Sim *s = reinterpret_cast<Sim *>(h);
  cup3d_poisson_params P;
  cup3d_poisson_default_params(&P);
  if (pp) P = *pp;
  const long N = s->nb * 512L;
  const unsigned G = vec_groups(N), Gs = vec_groups_simple(N);
  const bool second_order = step > 2;  // sim.step > sim.step_2nd_start (= 2), main.cpp:15087, 15355
  // Several ranks: chi written on THIS rank although cup3d_sim_set_obstacles(0) told every rank that none holds an obstacle.  The flag does
  // not steer the sequence of collectives (chi_path() depends on what the ranks were told, which is the same everywhere), so it needs no
  // agreement BEFORE the operator -- that was one stream drain + one blocking all-reduce + one host wait per time step.  It rides along
  // with the mean-pressure all-reduce the operator performs anyway (15123) and every rank returns the error together at the end.
  const bool over_ranks = s->grid->nranks > 1;
  const double conflict = over_ranks && s->chi_conflict() ? 1.0 : 0.0;
  if (second_order) { ProfileScope ps("project_pointwise"); LAUNCH_VEC_S(k_copy, s->pres, s->pold, N); }  // pOld, 15075
  // tmpV = 0 (15076-15078) matters only as the udef lab of KernelPressureRHS; without obstacles the RHS kernel does not read it
  // (adding -0*fac*0 is the identity).  With a resident chi it does: unless the caller has placed udef there since the last
  // projection (upload / fill / cup3d_update_tmpv), tmpV still holds the previous step's gradP scratch and is cleared here.
  if (s->chi_path() && !s->udef_nonzero) TRY(cup3d_sim_fill(h, CUP3D_FIELD_TMPV, 0.0));
  TRY(cup3d_pressure_rhs(h, dt));
  s->udef_nonzero = false;
  if (second_order) {
    TRY(cup3d_div_pressure(h));
    ProfileScope ps("project_pointwise");
    LAUNCH_VEC_S(k_sub_divp, s->lhs, s->tmpV, s->pres, N);
  } else {
    TRY(cup3d_sim_fill(h, CUP3D_FIELD_PRES, 0.0));  // 15102-15106
  }
  TRY(solve(s, P, r));
  const double hh = s->grid->h, vv = hh * hh * hh;
  Reducer red{s};
  { ProfileScope ps("project_pointwise"); LAUNCH_VEC(k_mean_dots, s->pres, N, vv, s->d_hb, red.out()); }
  if (over_ranks) {  // + the chi-conflict flag of the ranks (above)
    hipLaunchKernelGGL(k_set_one, dim3(1), dim3(1), 0, stream(), s->d_red, (size_t)kRedDots + 2, conflict);
    TRY(red.begin(3));
  } else {
    TRY(red.begin(2));                                       // MPI_Allreduce(2), 15123
  }
  TRY(red.wait());
  if (over_ranks && s->h_red[2] != 0.0) {
    set_error(conflict != 0.0 ? "cup3d_pressure_project: chi was written on this rank but cup3d_sim_set_obstacles(0) says no rank holds an obstacle"
                              : "cup3d_pressure_project: another rank holds a chi although cup3d_sim_set_obstacles(0) told every rank that none does");
    return CUP3D_ESTATE;
  }
  const double avg = s->h_red[0] / s->h_red[1];              // 15126
  { ProfileScope ps("project_pointwise"); LAUNCH_VEC_S(k_shift_mean, s->pres, second_order ? s->pold : (const double *)nullptr, N, avg); }
  TRY(cup3d_grad_p_update(h, dt));                           // KernelGradP + vel += tmpV/h^3, 15146-15159
  CUP3D_HIP(hipGetLastError());
  return CUP3D_OK;
}

}  // extern "C"
