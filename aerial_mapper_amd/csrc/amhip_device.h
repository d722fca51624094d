// amhip_device.h -- device-side helpers shared by the HIP translation units.
#ifndef AMHIP_DEVICE_H_
#define AMHIP_DEVICE_H_

#include <hip/hip_runtime.h>

namespace amhip {

// ---------------------------------------------------------------------------
// wave / workgroup scans
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// Exclusive scan of one value per thread across a block of NT threads.
// Returns the exclusive prefix; *total receives the block sum.
template <int NT>
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* total,
                                                    unsigned* lds /* NT/64+1 */) {
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  constexpr int NW = NT / 64;
  const unsigned incl = wave_incl_scan(v, lane);
  if (lane == 63) lds[wid] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const unsigned t = lds[w];
      lds[w] = run;
      run += t;
    }
    lds[NW] = run;
  }
  __syncthreads();
  const unsigned base = lds[wid];
  *total = lds[NW];
  __syncthreads();
  return base + incl - v;
}

}  // namespace amhip

#endif  // AMHIP_DEVICE_H_
