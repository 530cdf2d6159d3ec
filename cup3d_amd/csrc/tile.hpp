// Device helpers shared by the stencil kernels: XCD-aware block scheduling, face
// decoding, wave / workgroup reductions.
#pragma once
#include <hip/hip_runtime.h>

#include "sim.hpp"

namespace cup3d {

// Workgroup -> block slot.  MI355X dispatches workgroup b to XCD b % 8 (each XCD has a
// private 4 MiB L2), so XCD x walks the contiguous slot range [x*chunk, (x+1)*chunk):
// consecutive Hilbert-ordered blocks -- which are face neighbours most of the time --
// meet in the same L2 instead of being fetched by eight different ones.
__device__ __forceinline__ int block_slot(const GridDev &g) {
  const int bid = blockIdx.x;
  const int j = bid >> 3;
  const int i = (bid & 7) * g.chunk + j;
  if (i >= g.nblocks) return -1;
  return g.list ? g.list[i] : i;
}
inline unsigned launch_groups(const GridDev &g) { return (unsigned)(8 * g.chunk); }

// ---- wavefront reductions on the DPP data path (no LDS crossbar round trips).
// Classic GCN/CDNA reduction: xor-1 and xor-2 inside each quad (quad_perm), fold the two
// quads of a half row (row_half_mirror), the two halves of a 16-lane row (row_mirror), then
// carry row totals across rows with row_bcast15 / row_bcast31; lane 63 ends up with the
// wavefront total, which is broadcast through the scalar unit (v_readlane).  ~6 dependent
// VALU steps of a few cycles each instead of 6 ds_bpermute round trips (~100 cycles each):
// the block-CG preconditioner is bound by exactly this latency.
template <int CTRL, int ROW_MASK = 0xf, bool BOUND_CTRL = true>
__device__ __forceinline__ double dpp_move(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  int lo = (int)b, hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, BOUND_CTRL);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, BOUND_CTRL);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double broadcast_lane63(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// The two cross-row steps of wave_sum write only the rows of their row mask; the other rows keep `old`.  Only lane 63 of the result is
// used (and, for the second step, lane 31 of the first step's result: row 1), both in written rows -- so `old` may be anything.  With
// old = 0 the compiler emits two v_mov_b32 0 (+ a hazard nop) per step, with old = the source two copies of it (the DPP move's
// destination is tied to `old`); with old = the PREVIOUS step's moved value, which is dead by then, it emits nothing: 8 vector
// instructions fewer per CG iteration of the block preconditioner, same bits.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_rows(double v, double dead) {
  const long long b = __builtin_bit_cast(long long, v), o = __builtin_bit_cast(long long, dead);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)b, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// all 64 lanes receive the same sum
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0xB1>(v);              // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);              // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);             // row_half_mirror
  const double t4 = dpp_move<0x140>(v);  // row_mirror: every lane of a row holds the row total
  v += t4;
  const double t5 = dpp_move_rows<0x142, 0xa>(v, t4);   // row_bcast15 into rows 1 and 3 (rows 0 and 2 hold leftovers, never read)
  v += t5;
  v += dpp_move_rows<0x143, 0xc>(v, t5);                // row_bcast31 into rows 2 and 3: lane 63 = (R3 + R2) + (R1 + R0)
  return broadcast_lane63(v);
}
// rounds 1-5's form of the two cross-row steps (old = 0: two v_mov_b32 0 and a nop per step) -- kept for the A/B (EV bit 64 of cg_block)
__device__ __forceinline__ double wave_sum_zero_old(double v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  v += dpp_move<0x142, 0xa>(v);
  v += dpp_move<0x143, 0xc>(v);
  return broadcast_lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_move<0xB1>(v));
  v = fmax(v, dpp_move<0x4E>(v));
  v = fmax(v, dpp_move<0x141>(v));
  v = fmax(v, dpp_move<0x140>(v));
#pragma unroll
  for (int m = 32; m >= 16; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
  return v;
}

// workgroup sum for blockDim.x == 64 * NW; result valid in every thread
template <int NW>
__device__ __forceinline__ double group_sum(double v, double *scratch /* >= NW doubles */) {
  v = wave_sum(v);
  if constexpr (NW == 1) return v;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  double t = scratch[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t += scratch[i];
  return t;
}

// Grid-wide deterministic sums finished inside the producing launch (no separate 1-workgroup reduction kernel and, on one rank, no
// device-to-host copy): every workgroup stores its K partial sums, takes a ticket from an agent-scope counter, and the LAST one to
// arrive adds the partials of all workgroups in index order (the order does not depend on which workgroup is last) and stores the
// totals -- to device memory for the all-reduce / later kernels and, when `host` is set, to the pinned host mirror the solver's
// scalar recurrences read.  Release/acquire at agent scope: the partials cross XCDs (per-XCD L2s are not coherent).
struct RedOut {
  double *partials;   // [gridDim.x][8]
  unsigned *counter;  // zero before the launch; reset by the last workgroup
  double *out;        // [K] device
  double *host;       // [K] pinned host memory mapped into the device, or nullptr
  unsigned *flag;     // with `host`: pinned word that receives `seq` once the K totals are visible to the host (it spins on it)
  unsigned seq;
};
struct NoContinuation { __device__ __forceinline__ void operator()(const double *) const {} };
// `then(totals)`: run by thread 0 of the last workgroup once the K totals are stored and before the host flag is raised -- where the
// solver's scalar recurrences live on one rank (poisson.hip), so that the launch that follows finds them in device memory
template <int K, class Then = NoContinuation>
__device__ __forceinline__ void grid_sum_finish(double (&acc)[K], const RedOut &ro, Then then = Then()) {
  __shared__ double red[4];
  __shared__ int is_last;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const double s = group_sum<4>(acc[i], red);
    if (threadIdx.x == 0) ro.partials[(size_t)blockIdx.x * 8 + i] = s;
  }
  if (threadIdx.x == 0) {
    __threadfence();  // release the partials at agent scope
    is_last = atomicAdd(ro.counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();  // acquire the other workgroups' partials
  double tot[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    double s = 0;
    for (int j = threadIdx.x; j < (int)gridDim.x; j += 256) s += ro.partials[(size_t)j * 8 + i];
    s = group_sum<4>(s, red);
    tot[i] = s;
    if (threadIdx.x == 0) {
      ro.out[i] = s;
      if (ro.host) ro.host[i] = s;
    }
  }
  if (threadIdx.x == 0) {
    then(tot);
    *ro.counter = 0;
    if (ro.host) {
      __threadfence_system();  // the totals (this thread's own stores) before the flag
      if (ro.flag) __hip_atomic_store(ro.flag, ro.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace cup3d
