// amhip_device.h -- device-side helpers shared by the HIP translation units.
#ifndef AMHIP_DEVICE_H_
#define AMHIP_DEVICE_H_

#include <hip/hip_runtime.h>

namespace amhip {

// ---------------------------------------------------------------------------
// wave / workgroup scans
// ---------------------------------------------------------------------------
// Inclusive prefix sum across the 64 lanes of a wave (EVERY lane of the wave must be active:
// callers sit in wave-uniform control flow).  Six DPP adds: a Kogge-Stone scan inside each row
// of 16 lanes (row_shr:1/2/4/8; a lane without a source adds the `old` operand, 0), then the
// row totals travel on (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int /*lane*/) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

// min / max of a float across the wave, delivered in LANE 63 (other lanes hold partial results).
// Same six DPP steps; a lane without a source combines with itself.
__device__ __forceinline__ float wave_min_to_lane63(float v) {
#define AMHIP_DPP_MIN(CTRL, ROWS)                                                              \
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), \
                                                          CTRL, ROWS, 0xf, false)))
  AMHIP_DPP_MIN(0x111, 0xf);
  AMHIP_DPP_MIN(0x112, 0xf);
  AMHIP_DPP_MIN(0x114, 0xf);
  AMHIP_DPP_MIN(0x118, 0xf);
  AMHIP_DPP_MIN(0x142, 0xa);
  AMHIP_DPP_MIN(0x143, 0xc);
#undef AMHIP_DPP_MIN
  return v;
}
__device__ __forceinline__ float wave_max_to_lane63(float v) {
#define AMHIP_DPP_MAX(CTRL, ROWS)                                                              \
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), \
                                                          CTRL, ROWS, 0xf, false)))
  AMHIP_DPP_MAX(0x111, 0xf);
  AMHIP_DPP_MAX(0x112, 0xf);
  AMHIP_DPP_MAX(0x114, 0xf);
  AMHIP_DPP_MAX(0x118, 0xf);
  AMHIP_DPP_MAX(0x142, 0xa);
  AMHIP_DPP_MAX(0x143, 0xc);
#undef AMHIP_DPP_MAX
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

// ---------------------------------------------------------------------------
// running [min, max] of doubles in global memory (atomicMin / atomicMax on an
// order-preserving 64-bit key); NaN never enters
// ---------------------------------------------------------------------------
constexpr unsigned long long kOrderedPlusInf = 0xFFF0000000000000ull;   // key(+inf)
constexpr unsigned long long kOrderedMinusInf = 0x000FFFFFFFFFFFFFull;  // key(-inf)

__device__ __forceinline__ unsigned long long ordered_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ double from_ordered_key(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

// Every lane brings its own [lo, hi] (lo = +inf, hi = -inf when it has
// nothing); each WAVE stores its result to part[2 * wave_index .. +1] with plain
// stores (no barrier; thousands of same-address atomics would serialize) and
// k_range_reduce folds the partials into the running range.
__device__ __forceinline__ void range_commit_wave(double lo, double hi, double* __restrict__ part,
                                                  size_t wave_index) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, d, 64));
    hi = fmax(hi, __shfl_xor(hi, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    part[2 * wave_index] = lo;
    part[2 * wave_index + 1] = hi;
  }
}

// Workgroup `block` of `nblocks` folds its share of the partial pairs into the running range
// (may be null: the mosaic's coarse cull reads it) and the call's own range (the gather's
// rounding guard: max |z|); one pair of atomics per workgroup.  Any block size that is a multiple
// of 64, up to 1024; s_pair: 32 doubles of LDS.
__device__ __forceinline__ void range_reduce_block(const double* __restrict__ part, size_t nparts,
                                                   unsigned long long* __restrict__ range,
                                                   unsigned long long* __restrict__ call_range,
                                                   unsigned block, unsigned nblocks, double* s_pair) {
  double lo = __builtin_huge_val(), hi = -__builtin_huge_val();
  const size_t stride = (size_t)nblocks * blockDim.x;
  for (size_t k = (size_t)block * blockDim.x + threadIdx.x; k < nparts; k += stride) {
    const double2 v = reinterpret_cast<const double2*>(part)[k];
    lo = fmin(lo, v.x);
    hi = fmax(hi, v.y);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, d, 64));
    hi = fmax(hi, __shfl_xor(hi, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    s_pair[2 * (threadIdx.x >> 6)] = lo;
    s_pair[2 * (threadIdx.x >> 6) + 1] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (int)(blockDim.x >> 6);
    for (int w = 1; w < nw; ++w) {
      lo = fmin(lo, s_pair[2 * w]);
      hi = fmax(hi, s_pair[2 * w + 1]);
    }
    if (lo <= hi) {
      if (range) {
        atomicMin(&range[0], ordered_key(lo));
        atomicMax(&range[1], ordered_key(hi));
      }
      atomicMin(&call_range[0], ordered_key(lo));
      atomicMax(&call_range[1], ordered_key(hi));
    }
  }
}

}  // namespace amhip

#endif  // AMHIP_DEVICE_H_
