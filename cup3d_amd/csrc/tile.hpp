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

// all 64 lanes receive the sum (commutative butterfly => identical in every lane)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
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

}  // namespace cup3d
