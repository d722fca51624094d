// ubench3 -- round 3: the issue ceiling of the DSM gather's candidate loop in
// REAL shader cycles, and the point-centric alternative's LDS atomics under the real access pattern.
//
//   part A  cycles per wave-instruction per SIMD of the instructions the loop is made of, from
//           s_memtime deltas inside the kernel (shader clock) at 1 / 2 / 4 / 8 resident waves per
//           SIMD; the shader clock itself from s_memtime / s_memrealtime (100 MHz) in the same
//           waves.  (ubench2 converted wall time with an assumed 2.4 GHz.)
//   part B  the candidate loop's body itself (the asm block of k_dsm_gather_f32, one ds_read_b128 per
//           candidate), full EXEC, 8 waves per SIMD: cycles per candidate and wave.
//   part C  point-centric scatter prototype: lanes = points sorted by cell, a 69-offset disc
//           stencil per point, w = rcp(d2), two ds_add_f32 per hit into an LDS cell image; one
//           64 x 64-cell tile per workgroup at cfg2's density.  ms per 1e8 cells, against the
//           cell-centric loop's 1.41 ms.
// Build: hipcc --offload-arch=gfx950 -O3 ubench3.hip -o ubench3 ; run: ./ubench3 [A|B|C]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);         \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

// ---------------------------------------------------------------------------------------------
// part A
// ---------------------------------------------------------------------------------------------
// %0..%7 float registers (in/out), %8 an integer register, %9 an SGPR pair
#define DEF_KERNEL(NAME, BODY)                                                                \
  __global__ void __launch_bounds__(256) NAME(float* out, unsigned long long* stamps, int iters) { \
    extern __shared__ unsigned char dyn_lds[];                                                \
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,        \
          a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                              \
    unsigned b = threadIdx.x * 2654435761u;                                                   \
    unsigned long long sv;                                                                    \
    if (iters < 0) dyn_lds[threadIdx.x] = 1; /* (keeps the LDS allocation) */                 \
    __syncthreads();                                                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                           \
    for (int it = 0; it < iters; ++it) {                                                      \
      asm volatile(BODY BODY BODY BODY                                                        \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),    \
                     "+v"(a7), "+v"(b), "=&s"(sv)                                             \
                   :                                                                          \
                   : "vcc");                                                                  \
    }                                                                                         \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                           \
    if ((threadIdx.x & 63) == 0) {                                                            \
      unsigned long long* s = stamps + 4 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6));     \
      s[0] = t0; s[1] = t1; s[2] = r0; s[3] = r1;                                             \
    }                                                                                         \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)b;   \
  }

DEF_KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n")
DEF_KERNEL(k_mul, "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %4\n v_mul_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_mul_f32 %7, %7, %0\n")
DEF_KERNEL(k_add, "v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %0\n")
DEF_KERNEL(k_max, "v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %4\n v_max_f32 %4, %4, %5\n v_max_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_max_f32 %7, %7, %0\n")
DEF_KERNEL(k_subu, "v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %4\n v_sub_u32 %4, %4, %5\n v_sub_u32 %5, %5, %6\n v_sub_u32 %6, %6, %7\n v_sub_u32 %7, %7, %0\n")
DEF_KERNEL(k_cvt, "v_cvt_f32_i32 %0, %1\n v_cvt_f32_i32 %1, %2\n v_cvt_f32_i32 %2, %3\n v_cvt_f32_i32 %3, %4\n v_cvt_f32_i32 %4, %5\n v_cvt_f32_i32 %5, %6\n v_cvt_f32_i32 %6, %7\n v_cvt_f32_i32 %7, %0\n")
DEF_KERNEL(k_rcp, "v_rcp_f32 %0, %1\n v_rcp_f32 %1, %2\n v_rcp_f32 %2, %3\n v_rcp_f32 %3, %4\n v_rcp_f32 %4, %5\n v_rcp_f32 %5, %6\n v_rcp_f32 %6, %7\n v_rcp_f32 %7, %0\n")
DEF_KERNEL(k_cmp, "v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %4\n v_cmp_gt_f32 vcc, %4, %5\n v_cmp_gt_f32 vcc, %5, %6\n v_cmp_gt_f32 vcc, %6, %7\n v_cmp_gt_f32 vcc, %7, %0\n")
DEF_KERNEL(k_cnd, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n")
// integer / logical forms of the half-rate float instructions (non-negative floats order like
// their bit patterns): are they full rate?
DEF_KERNEL(k_maxu, "v_max_u32 %0, %0, %1\n v_max_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_max_u32 %3, %3, %4\n v_max_u32 %4, %4, %5\n v_max_u32 %5, %5, %6\n v_max_u32 %6, %6, %7\n v_max_u32 %7, %7, %0\n")
DEF_KERNEL(k_maxi, "v_max_i32 %0, %0, %1\n v_max_i32 %1, %1, %2\n v_max_i32 %2, %2, %3\n v_max_i32 %3, %3, %4\n v_max_i32 %4, %4, %5\n v_max_i32 %5, %5, %6\n v_max_i32 %6, %6, %7\n v_max_i32 %7, %7, %0\n")
DEF_KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n")
DEF_KERNEL(k_max3u, "v_max3_u32 %0, %0, %1, %2\n v_max3_u32 %1, %1, %2, %3\n v_max3_u32 %2, %2, %3, %4\n v_max3_u32 %3, %3, %4, %5\n v_max3_u32 %4, %4, %5, %6\n v_max3_u32 %5, %5, %6, %7\n v_max3_u32 %6, %6, %7, %0\n v_max3_u32 %7, %7, %0, %1\n")
DEF_KERNEL(k_cmpu, "v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %1, %2\n v_cmp_gt_u32 vcc, %2, %3\n v_cmp_gt_u32 vcc, %3, %4\n v_cmp_gt_u32 vcc, %4, %5\n v_cmp_gt_u32 vcc, %5, %6\n v_cmp_gt_u32 vcc, %6, %7\n v_cmp_gt_u32 vcc, %7, %0\n")
DEF_KERNEL(k_and, "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %0\n")
DEF_KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %4\n v_or3_b32 %3, %3, %4, %5\n v_or3_b32 %4, %4, %5, %6\n v_or3_b32 %5, %5, %6, %7\n v_or3_b32 %6, %6, %7, %0\n v_or3_b32 %7, %7, %0, %1\n")
DEF_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %3, %3, %4, %5\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_add3_u32 %6, %6, %7, %0\n v_add3_u32 %7, %7, %0, %1\n")
DEF_KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %1, 3, %2\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %3, 3, %4\n v_lshlrev_b32 %4, 3, %5\n v_lshlrev_b32 %5, 3, %6\n v_lshlrev_b32 %6, 3, %7\n v_lshlrev_b32 %7, 3, %0\n")
DEF_KERNEL(k_cnds, "v_cndmask_b32 %0, %0, %1, %9\n v_cndmask_b32 %1, %1, %2, %9\n v_cndmask_b32 %2, %2, %3, %9\n v_cndmask_b32 %3, %3, %4, %9\n v_cndmask_b32 %4, %4, %5, %9\n v_cndmask_b32 %5, %5, %6, %9\n v_cndmask_b32 %6, %6, %7, %9\n v_cndmask_b32 %7, %7, %0, %9\n")
DEF_KERNEL(k_fmaclamp, "v_fma_f32 %0, %0, %1, %2 clamp\n v_fma_f32 %1, %1, %2, %3 clamp\n v_fma_f32 %2, %2, %3, %4 clamp\n v_fma_f32 %3, %3, %4, %5 clamp\n v_fma_f32 %4, %4, %5, %6 clamp\n v_fma_f32 %5, %5, %6, %7 clamp\n v_fma_f32 %6, %6, %7, %0 clamp\n v_fma_f32 %7, %7, %0, %1 clamp\n")
DEF_KERNEL(k_cmpxu, "s_mov_b64 %9, exec\n v_cmpx_gt_u32 %0, %1\n v_max_u32 %2, %2, %3\n s_mov_b64 exec, %9\n v_cmpx_gt_u32 %1, %2\n v_max_u32 %3, %3, %4\n s_mov_b64 exec, %9\n v_cmpx_gt_u32 %4, %5\n v_max_u32 %6, %6, %7\n s_mov_b64 exec, %9\n v_cmpx_gt_u32 %5, %6\n v_max_u32 %7, %7, %0\n s_mov_b64 exec, %9\n")
// v_cmpx + one masked VALU + s_mov restore, four times (unit = one cmpx/op/restore group)
DEF_KERNEL(k_cmpx, "s_mov_b64 %9, exec\n v_cmpx_gt_f32 %0, %1\n v_max_f32 %2, %2, %3\n s_mov_b64 exec, %9\n v_cmpx_gt_f32 %1, %2\n v_max_f32 %3, %3, %4\n s_mov_b64 exec, %9\n v_cmpx_gt_f32 %4, %5\n v_max_f32 %6, %6, %7\n s_mov_b64 exec, %9\n v_cmpx_gt_f32 %5, %6\n v_max_f32 %7, %7, %0\n s_mov_b64 exec, %9\n")
// the product kernel's hit block, twice (unit = one block: cmpx, rcp, max, add, fmac + exec restore)
DEF_KERNEL(k_hit, "s_mov_b64 %9, exec\n v_cmpx_gt_f32 %7, %0\n v_rcp_f32 %1, %0\n v_max_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n v_fmac_f32 %4, %1, %7\n s_mov_b64 exec, %9\n v_cmpx_gt_f32 %7, %5\n v_rcp_f32 %1, %5\n v_max_f32 %2, %2, %5\n v_add_f32 %6, %6, %1\n v_fmac_f32 %4, %1, %7\n s_mov_b64 exec, %9\n")

// FP64: %0..%3 doubles
#define DEF_KERNEL64(NAME, BODY)                                                              \
  __global__ void __launch_bounds__(256) NAME(float* out, unsigned long long* stamps, int iters) { \
    extern __shared__ unsigned char dyn_lds[];                                                \
    double a0 = threadIdx.x + 1.5, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                      \
    if (iters < 0) dyn_lds[threadIdx.x] = 1;                                                  \
    __syncthreads();                                                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                           \
    for (int it = 0; it < iters; ++it) {                                                      \
      asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY                                    \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));                                 \
    }                                                                                         \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                           \
    if ((threadIdx.x & 63) == 0) {                                                            \
      unsigned long long* s = stamps + 4 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6));     \
      s[0] = t0; s[1] = t1; s[2] = r0; s[3] = r1;                                             \
    }                                                                                         \
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);                         \
  }
DEF_KERNEL64(k_fma64, "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1\n")
DEF_KERNEL64(k_mul64, "v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0\n")
DEF_KERNEL64(k_add64, "v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0\n")

struct Stat {
  double cyc_per_unit_simd, ghz, wall_ms, spread;
};

template <typename K>
static int run_a(K kernel, int waves_per_simd, int units_per_body_x4, int iters, float* out,
                 unsigned long long* stamps, Stat* st) {
  // blocks of 256 threads = one wave per SIMD each; W blocks per CU by the LDS they ask for
  const int blocks = 256 * waves_per_simd;
  const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                         (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, 0, out, stamps, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<unsigned long long> h((size_t)blocks * 16);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  double sum_c = 0, sum_r = 0, mn = 1e30, mx = 0;
  for (int w = 0; w < blocks * 4; ++w) {
    const double c = (double)(h[4 * w + 1] - h[4 * w + 0]), r = (double)(h[4 * w + 3] - h[4 * w + 2]);
    sum_c += c;
    sum_r += r;
    mn = std::min(mn, c);
    mx = std::max(mx, c);
  }
  const double cyc_wave = sum_c / (blocks * 4);
  const double units = (double)iters * units_per_body_x4;  // per wave
  st->cyc_per_unit_simd = cyc_wave / units / waves_per_simd;
  st->ghz = sum_c / sum_r * 0.1;  // s_memrealtime ticks at 100 MHz
  st->wall_ms = ms;
  st->spread = mx / mn;
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// part B: the candidate loop's body
// ---------------------------------------------------------------------------------------------
template <int UNROLL>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_loop(float* out, unsigned long long* stamps, int len, int trips) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* s_rec = reinterpret_cast<uint4*>(smem);
  for (int k = threadIdx.x; k < 1024; k += 512)
    s_rec[k] = make_uint4((k * 2654435761u) >> 4, (k * 40503u) << 8, __float_as_uint(0.25f * (k & 15)), 0u);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned Ui = (unsigned)lane << 28, Vj = (unsigned)(threadIdx.x >> 6) << 28;
  const float one_cell = (float)(1u << 28);
  const float thi = 16.0f * one_cell * one_cell, thiB = thi;
  float NA = 0, DA = 0, NB = 0, DB = 0, mA = 0, mB = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < trips; ++t) {
    float nA = 0, dA = 0, nB = 0, dB = 0;
    const uint4* pr = s_rec + ((lane * 3 + t * 7) & 511);
    const uint4* const pe = pr + len;
    auto cand = [&](const uint4 rec) __attribute__((always_inline)) {
      float t0_, t1_, t3_;
      unsigned long long sv;
      asm volatile(
          "v_sub_u32 %[t0], %[Ui], %[x]\n\t"
          "v_sub_u32 %[t1], %[Vj], %[y]\n\t"
          "v_cvt_f32_i32 %[t0], %[t0]\n\t"
          "v_cvt_f32_i32 %[t1], %[t1]\n\t"
          "v_mul_f32 %[t0], %[t0], %[t0]\n\t"
          "v_add_f32 %[t3], %[one], %[t1]\n\t"
          "v_fma_f32 %[t1], %[t1], %[t1], %[t0]\n\t"
          "v_fma_f32 %[t3], %[t3], %[t3], %[t0]\n\t"
          "s_mov_b64 %[sv], exec\n\t"
          "v_cmpx_gt_f32 %[thi], %[t1]\n\t"
          "v_rcp_f32 %[t0], %[t1]\n\t"
          "v_max_f32 %[mA], %[mA], %[t1]\n\t"
          "v_add_f32 %[dA], %[dA], %[t0]\n\t"
          "v_fmac_f32 %[nA], %[t0], %[z]\n\t"
          "s_mov_b64 exec, %[sv]\n\t"
          "v_cmpx_gt_f32 %[thiB], %[t3]\n\t"
          "v_rcp_f32 %[t0], %[t3]\n\t"
          "v_max_f32 %[mB], %[mB], %[t3]\n\t"
          "v_add_f32 %[dB], %[dB], %[t0]\n\t"
          "v_fmac_f32 %[nB], %[t0], %[z]\n\t"
          "s_mov_b64 exec, %[sv]"
          : [t0] "=&v"(t0_), [t1] "=&v"(t1_), [t3] "=&v"(t3_), [sv] "=&s"(sv), [mA] "+v"(mA),
            [mB] "+v"(mB), [nA] "+v"(nA), [dA] "+v"(dA), [nB] "+v"(nB), [dB] "+v"(dB)
          : [Ui] "v"(Ui), [Vj] "v"(Vj), [x] "v"(rec.x), [y] "v"(rec.y), [z] "v"(rec.z),
            [one] "v"(one_cell), [thi] "v"(thi), [thiB] "v"(thiB)
          : "vcc");
    };
    if (UNROLL == 1) {
      for (; pr < pe; ++pr) cand(*pr);
    } else {
      for (; pr + 1 < pe; pr += 2) {
        const uint4 r0_ = pr[0], r1_ = pr[1];
        cand(r0_);
        cand(r1_);
      }
      if (pr < pe) cand(*pr);
    }
    NA += nA; DA += dA; NB += nB; DB += dB;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    unsigned long long* s = stamps + 4 * ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6));
    s[0] = t0; s[1] = t1; s[2] = r0; s[3] = r1;
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = NA + DA + NB + DB + mA + mB;
}

// ---------------------------------------------------------------------------------------------
// part C: point-centric scatter into an LDS cell image with native ds_add_f32
// ---------------------------------------------------------------------------------------------
constexpr int kTile = 64, kRing = 4, kReg = kTile + 2 * kRing;  // points of a 72 x 72 region
constexpr int kImg = kReg + 2 * kRing;                          // image 80 x 80 (+ pitch)
constexpr int kPitch = kImg + 1;
constexpr int kImgCells = kImg * kPitch;

// offsets (a, b), |a|, |b| <= 4, of the cells a point CAN reach (69) / ALWAYS reaches (37)
__host__ __device__ constexpr bool st_reach(int a, int b) {
  const float ax = (a < 0 ? -a : a) - 0.5f, by = (b < 0 ? -b : b) - 0.5f;
  const float x = ax > 0 ? ax : 0, y = by > 0 ? by : 0;
  return x * x + y * y < 16.0f;
}
__host__ __device__ constexpr bool st_always(int a, int b) {
  const float ax = (a < 0 ? -a : a) + 0.5f, by = (b < 0 ? -b : b) + 0.5f;
  return ax * ax + by * by < 16.0f;
}

struct PtRec {
  float fx, fy, dz;  // offset from the home cell's centre in cells; height offset
  int home;          // hx + hy * kPitch in the image (region cell + kRing)
};

template <bool ATOMIC>
__global__ void __launch_bounds__(512)
k_scatter_tile(const PtRec* __restrict__ pts, const int* __restrict__ tile_first, int ntiles_data,
               float* __restrict__ out) {
  __shared__ float sN[kImgCells], sD[kImgCells];
  for (int k = threadIdx.x; k < kImgCells; k += 512) {
    sN[k] = 0.f;
    sD[k] = 0.f;
  }
  __syncthreads();
  const int td = blockIdx.x % ntiles_data;
  const int p0 = tile_first[td], p1 = tile_first[td + 1];
  for (int k = p0 + threadIdx.x; k < p1; k += 512) {
    const PtRec p = pts[k];
    float dx2[9], dy2[9];
#pragma unroll
    for (int a = -4; a <= 4; ++a) {
      const float dx = (float)a - p.fx, dy = (float)a - p.fy;
      dx2[a + 4] = dx * dx;
      dy2[a + 4] = dy * dy;
    }
    // (bases moved to the stencil's corner: every offset is then a non-negative immediate of the
    // DS instruction, no address arithmetic per pair)
    int corner = p.home - 4 - 4 * kPitch;
    asm volatile("" : "+v"(corner));  // (opaque: keeps the compiler from re-associating the offsets negative)
    float* const nbase = sN + corner;
    float* const dbase = sD + corner;
#pragma unroll
    for (int b = -4; b <= 4; ++b) {
#pragma unroll
      for (int a = -4; a <= 4; ++a) {
        if (!st_reach(a, b)) continue;
        const float d2 = dx2[a + 4] + dy2[b + 4];
        if (st_always(a, b) || d2 < 16.0f) {
          const float w = __builtin_amdgcn_rcpf(d2);
          if (ATOMIC) {
            __hip_atomic_fetch_add(dbase + (a + 4) + (b + 4) * kPitch, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(nbase + (a + 4) + (b + 4) * kPitch, w * p.dz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {  // (same VALU work, plain stores: what the atomics themselves cost)
            dbase[(a + 4) + (b + 4) * kPitch] = w;
            nbase[(a + 4) + (b + 4) * kPitch] = w * p.dz;
          }
        }
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < kTile * kTile; c += 512) {
    const int i = c & 63, j = c >> 6;
    const int q = (i + 2 * kRing) + (j + 2 * kRing) * kPitch;
    const float D = sD[q];
    out[(size_t)blockIdx.x * kTile * kTile + c] = D > 0.f ? sN[q] * __builtin_amdgcn_rcpf(D) : __builtin_nanf("");
  }
}

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "ABC";
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* out;
  unsigned long long* stamps;
  CK(hipMalloc(&out, (size_t)256 * 8 * 512 * 4 * 8));
  CK(hipMalloc(&stamps, (size_t)256 * 8 * 16 * 8 * 8));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.gcnArchName, prop.multiProcessorCount,
         prop.clockRate);

  if (strchr(which, 'A')) {
    struct T {
      const char* name;
      void (*fn)(float*, unsigned long long*, int);
      int units_x4;  // units per (BODY x 4) asm block
    } tests[] = {
        {"v_fma_f32", k_fma, 32},   {"v_mul_f32", k_mul, 32},       {"v_add_f32", k_add, 32},
        {"v_max_f32", k_max, 32},   {"v_sub_u32", k_subu, 32},      {"v_cvt_f32_i32", k_cvt, 32},
        {"v_rcp_f32", k_rcp, 32},   {"v_cmp_gt_f32", k_cmp, 32},    {"v_cndmask_b32", k_cnd, 32},
        {"v_cmpx+v_max+s_mov exec (group)", k_cmpx, 16},
        {"hit block: cmpx rcp max add fmac + exec restore", k_hit, 8},
        {"v_fma_f64", k_fma64, 32}, {"v_mul_f64", k_mul64, 32},     {"v_add_f64", k_add64, 32},
        {"v_max_u32", k_maxu, 32},  {"v_max_i32", k_maxi, 32},      {"v_max3_f32", k_max3, 32},
        {"v_max3_u32", k_max3u, 32}, {"v_cmp_gt_u32", k_cmpu, 32},  {"v_and_b32", k_and, 32},
        {"v_or3_b32", k_or3, 32},   {"v_add3_u32", k_add3, 32},     {"v_lshlrev_b32", k_lshl, 32},
        {"v_cndmask_b32 (sgpr mask)", k_cnds, 32}, {"v_fma_f32 clamp", k_fmaclamp, 32},
        {"v_cmpx_gt_u32+v_max_u32+s_mov exec (group)", k_cmpxu, 16},
    };
    const bool only_new = strchr(which, 'N') != nullptr;   // "AN": the round-3 additions only
    int tindex = 0;
    for (auto& t : tests) {
      if (only_new && tindex++ < 14) continue;
      printf("{\"part\": \"A\", \"instr\": \"%s\"", t.name);
      for (int w : {1, 2, 4, 8}) {
        Stat st;
        if (run_a(t.fn, w, t.units_x4, 2048, out, stamps, &st)) return 1;
        // wall: wall time x measured clock / wave-instructions per SIMD (what a saturated SIMD pays
        // per instruction, incl. the launch); cyc: from the waves' own s_memtime spans
        const double wall_cyc = st.wall_ms * 1e-3 * st.ghz * 1e9 / ((double)w * 2048 * t.units_x4);
        printf(", \"w%d\": {\"wall_cyc\": %.2f, \"cyc\": %.3f, \"GHz\": %.3f, \"ms\": %.3f, \"spread\": %.2f}", w,
               wall_cyc, st.cyc_per_unit_simd, st.ghz, st.wall_ms, st.spread);
      }
      printf("}\n");
    }
  }

  if (strchr(which, 'B')) {
    for (int unroll = 1; unroll <= 2; ++unroll)
      for (int len : {9, 13, 45}) {
        const int blocks = 256 * 4, trips = 45 * 40 / len;
        const size_t lds = 37 * 1024;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          if (unroll == 1) hipLaunchKernelGGL(k_loop<1>, dim3(blocks), dim3(512), lds, 0, out, stamps, len, trips);
          else hipLaunchKernelGGL(k_loop<2>, dim3(blocks), dim3(512), lds, 0, out, stamps, len, trips);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<unsigned long long> h((size_t)blocks * 8 * 4);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double sc = 0, sr = 0;
        for (int w = 0; w < blocks * 8; ++w) {
          sc += (double)(h[4 * w + 1] - h[4 * w]);
          sr += (double)(h[4 * w + 3] - h[4 * w + 2]);
        }
        const double cyc_wave = sc / (blocks * 8);
        const double cands = (double)trips * len;
        printf("{\"part\": \"B\", \"unroll\": %d, \"trip_len\": %d, \"cycles_per_candidate_per_wave\": %.2f, "
               "\"cycles_per_candidate_per_simd\": %.2f, \"GHz\": %.3f, \"ms\": %.3f}\n",
               unroll, len, cyc_wave / cands, cyc_wave / cands / 8.0, sc / sr * 0.1, ms);
      }
  }

  if (strchr(which, 'C')) {
    // 64 tile data sets: points uniform in the 72 x 72 region at 0.5 points per cell, sorted by
    // home cell (x fastest: the order the LDS binning of the product leaves them in)
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const int nd = 64;
    std::vector<PtRec> pts;
    std::vector<int> first(nd + 1, 0);
    for (int t = 0; t < nd; ++t) {
      std::poisson_distribution<int> P(0.5 * kReg * kReg);
      const int n = P(rng);
      std::vector<PtRec> tp(n);
      for (auto& p : tp) {
        const double x = U(rng) * kReg, y = U(rng) * kReg;
        const int hx = (int)x, hy = (int)y;
        p.fx = (float)(x - hx - 0.5);
        p.fy = (float)(y - hy - 0.5);
        p.dz = (float)(U(rng) * 4.0 - 2.0);
        p.home = (hx + kRing) + (hy + kRing) * kPitch;
      }
      std::sort(tp.begin(), tp.end(), [](const PtRec& a, const PtRec& b) { return a.home < b.home; });
      pts.insert(pts.end(), tp.begin(), tp.end());
      first[t + 1] = (int)pts.size();
    }
    PtRec* dp;
    int* df;
    float* dout;
    const int ntiles = 24576;  // x 4096 cells = 1.0066e8 cells
    CK(hipMalloc(&dp, pts.size() * sizeof(PtRec)));
    CK(hipMalloc(&df, first.size() * 4));
    CK(hipMalloc(&dout, (size_t)ntiles * kTile * kTile * 4));
    CK(hipMemcpy(dp, pts.data(), pts.size() * sizeof(PtRec), hipMemcpyHostToDevice));
    CK(hipMemcpy(df, first.data(), first.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int atomic = 1; atomic >= 0; --atomic) {
      float ms = 0, best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        if (atomic) hipLaunchKernelGGL(k_scatter_tile<true>, dim3(ntiles), dim3(512), 0, 0, dp, df, nd, dout);
        else hipLaunchKernelGGL(k_scatter_tile<false>, dim3(ntiles), dim3(512), 0, 0, dp, df, nd, dout);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
      }
      // spot check of tile 0 against a double evaluation (atomic variant only)
      double worst = 0;
      if (atomic) {
        std::vector<float> h(kTile * kTile);
        CK(hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost));
        for (int c = 0; c < kTile * kTile; c += 37) {
          const int i = c & 63, j = c >> 6;
          double N = 0, D = 0;
          for (int k = first[0]; k < first[1]; ++k) {
            const int hx = pts[k].home % kPitch - kRing, hy = pts[k].home / kPitch - kRing;
            const double dx = (i + kRing) - (hx + pts[k].fx), dy = (j + kRing) - (hy + pts[k].fy);
            const double d2 = dx * dx + dy * dy;
            if (d2 < 16.0) {
              N += pts[k].dz / d2;
              D += 1.0 / d2;
            }
          }
          if (D > 0) worst = std::max(worst, std::fabs(N / D - (double)h[c]));
        }
      }
      printf("{\"part\": \"C\", \"variant\": \"%s\", \"tiles\": %d, \"cells\": %.4g, \"points_per_tile_region\": %.0f, "
             "\"ms\": %.3f, \"ms_per_1e8_cells\": %.3f, \"spot_check_max_abs_err\": %.3g}\n",
             atomic ? "ds_add_f32 x2 per hit" : "plain ds_write x2 per hit (no accumulation: VALU + LDS issue only)",
             ntiles, (double)ntiles * kTile * kTile, (double)pts.size() / nd, best,
             best * 1e8 / ((double)ntiles * kTile * kTile), worst);
    }
  }
  return 0;
}
