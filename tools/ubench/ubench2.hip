// Per-instruction issue cost on gfx950 (cycles per wave-instruction per SIMD), for the
// instructions the DSM gather's candidate loop is made of.  Each pattern = 8 independent
// instructions in an asm block, looped; 8 waves per SIMD resident.
// Build: hipcc --offload-arch=gfx950 -O3 ubench2.hip -o ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(S) S S S S S S S S
// operands: %0..%7 float regs (in/out), %8 int-ish reg, %9 sgpr pair scratch
#define DEF_KERNEL(NAME, BODY)                                                              \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                      \
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,      \
          a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                            \
    unsigned b = threadIdx.x * 2654435761u;                                                 \
    unsigned long long sv;                                                                  \
    for (int it = 0; it < iters; ++it) {                                                    \
      asm volatile(BODY                                                                     \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),  \
                     "+v"(a7), "+v"(b), "=&s"(sv)                                           \
                   :                                                                        \
                   : "vcc");                                                                \
    }                                                                                       \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)b; \
  }

DEF_KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n")
DEF_KERNEL(k_mul, "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %4\n v_mul_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_mul_f32 %7, %7, %0\n")
DEF_KERNEL(k_max, "v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %4\n v_max_f32 %4, %4, %5\n v_max_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_max_f32 %7, %7, %0\n")
DEF_KERNEL(k_subu, "v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %4\n v_sub_u32 %4, %4, %5\n v_sub_u32 %5, %5, %6\n v_sub_u32 %6, %6, %7\n v_sub_u32 %7, %7, %0\n")
DEF_KERNEL(k_cvt_i32, "v_cvt_f32_i32 %0, %1\n v_cvt_f32_i32 %1, %2\n v_cvt_f32_i32 %2, %3\n v_cvt_f32_i32 %3, %4\n v_cvt_f32_i32 %4, %5\n v_cvt_f32_i32 %5, %6\n v_cvt_f32_i32 %6, %7\n v_cvt_f32_i32 %7, %0\n")
DEF_KERNEL(k_cvt_u32, "v_cvt_f32_u32 %0, %1\n v_cvt_f32_u32 %1, %2\n v_cvt_f32_u32 %2, %3\n v_cvt_f32_u32 %3, %4\n v_cvt_f32_u32 %4, %5\n v_cvt_f32_u32 %5, %6\n v_cvt_f32_u32 %6, %7\n v_cvt_f32_u32 %7, %0\n")
DEF_KERNEL(k_cvt_ub, "v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte1 %1, %2\n v_cvt_f32_ubyte2 %2, %3\n v_cvt_f32_ubyte3 %3, %4\n v_cvt_f32_ubyte0 %4, %5\n v_cvt_f32_ubyte1 %5, %6\n v_cvt_f32_ubyte2 %6, %7\n v_cvt_f32_ubyte3 %7, %0\n")
DEF_KERNEL(k_cvt_f16, "v_cvt_f32_f16 %0, %1\n v_cvt_f32_f16 %1, %2\n v_cvt_f32_f16 %2, %3\n v_cvt_f32_f16 %3, %4\n v_cvt_f32_f16 %4, %5\n v_cvt_f32_f16 %5, %6\n v_cvt_f32_f16 %6, %7\n v_cvt_f32_f16 %7, %0\n")
DEF_KERNEL(k_cvt_f16_sdwa, "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")
DEF_KERNEL(k_rcp, "v_rcp_f32 %0, %1\n v_rcp_f32 %1, %2\n v_rcp_f32 %2, %3\n v_rcp_f32 %3, %4\n v_rcp_f32 %4, %5\n v_rcp_f32 %5, %6\n v_rcp_f32 %6, %7\n v_rcp_f32 %7, %0\n")
DEF_KERNEL(k_cmp, "v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %4\n v_cmp_gt_f32 vcc, %4, %5\n v_cmp_gt_f32 vcc, %5, %6\n v_cmp_gt_f32 vcc, %6, %7\n v_cmp_gt_f32 vcc, %7, %0\n")
DEF_KERNEL(k_cmp_cnd, "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_gt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc\n")
// v_cmpx + restore, 4 times (8 VALU-ish: 4 cmpx + 4 max under the mask; 4 s_mov)
DEF_KERNEL(k_cmpx, "s_mov_b64 %9, exec\n v_cmpx_gt_f32 %0, %1\n v_max_f32 %2, %2, %3\n s_mov_b64 exec, %9\n v_cmpx_gt_f32 %1, %2\n v_max_f32 %3, %3, %4\n s_mov_b64 exec, %9\n v_cmpx_gt_f32 %4, %5\n v_max_f32 %6, %6, %7\n s_mov_b64 exec, %9\n v_cmpx_gt_f32 %5, %6\n v_max_f32 %7, %7, %0\n s_mov_b64 exec, %9\n")
// cmp -> sgpr, s_and_saveexec-free variant: v_cmp into vcc, s_and exec, op, restore
DEF_KERNEL(k_cmp_sand, "s_mov_b64 %9, exec\n v_cmp_gt_f32 vcc, %0, %1\n s_and_b64 exec, exec, vcc\n v_max_f32 %2, %2, %3\n s_mov_b64 exec, %9\n v_cmp_gt_f32 vcc, %1, %2\n s_and_b64 exec, exec, vcc\n v_max_f32 %3, %3, %4\n s_mov_b64 exec, %9\n v_cmp_gt_f32 vcc, %4, %5\n s_and_b64 exec, exec, vcc\n v_max_f32 %6, %6, %7\n s_mov_b64 exec, %9\n v_cmp_gt_f32 vcc, %5, %6\n s_and_b64 exec, exec, vcc\n v_max_f32 %7, %7, %0\n s_mov_b64 exec, %9\n")
// the current hit block (rcp form) and the division-free form, each twice = 10 VALU
DEF_KERNEL(k_hit_rcp, "s_mov_b64 %9, exec\n v_cmpx_gt_f32 %7, %0\n v_rcp_f32 %1, %0\n v_max_f32 %2, %2, %0\n v_add_f32 %3, %3, %1\n v_fmac_f32 %4, %1, %7\n s_mov_b64 exec, %9\n v_cmpx_gt_f32 %7, %5\n v_rcp_f32 %1, %5\n v_max_f32 %2, %2, %5\n v_add_f32 %6, %6, %1\n v_fmac_f32 %4, %1, %7\n s_mov_b64 exec, %9\n")
DEF_KERNEL(k_hit_prod, "s_mov_b64 %9, exec\n v_cmpx_gt_f32 %7, %0\n v_mul_f32 %1, %2, %7\n v_fma_f32 %3, %3, %0, %1\n v_fma_f32 %4, %4, %0, %2\n v_mul_f32 %2, %2, %0\n s_mov_b64 exec, %9\n v_cmpx_gt_f32 %7, %5\n v_mul_f32 %1, %6, %7\n v_fma_f32 %3, %3, %5, %1\n v_fma_f32 %4, %4, %5, %6\n v_mul_f32 %6, %6, %5\n s_mov_b64 exec, %9\n")

// LDS read patterns next to VALU work: every lane reads consecutive 16/20-byte records
template <int MODE>
__global__ void __launch_bounds__(512) k_lds(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int k = threadIdx.x; k < 8192; k += 512) reinterpret_cast<float*>(smem)[k] = k * 0.001f;
  __syncthreads();
  float acc = 0.f, acc2 = 0.f;
  unsigned a = (threadIdx.x & 63) * 16 * 3;  // neighbouring lanes 3 records apart (overlapping spans)
  for (int it = 0; it < iters; ++it) {
    const unsigned addr = (a + it * 16) & 16383u;
    if (MODE == 0) {
      const float4 v = *reinterpret_cast<const float4*>(smem + addr);
      acc += v.x * v.y + v.z * v.w;
    } else if (MODE == 1) {
      const float4 v = *reinterpret_cast<const float4*>(smem + addr);
      const float w = *reinterpret_cast<const float*>(smem + 16384 + (addr >> 2));
      acc += v.x * v.y + v.z * v.w + w;
    } else {
      const float4 v = *reinterpret_cast<const float4*>(smem + addr);
      float t = v.x;
#pragma unroll
      for (int q = 0; q < 20; ++q) t = fmaf(t, v.y, v.z);
      acc2 += t * v.w;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc + acc2;
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;  // run one test (separate processes: a hang costs one test)
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* rout; CK(hipMalloc(&rout, 256 * 8 * 512 * 4));
  const int blocks = 256 * 8, iters = 4096;
  struct { const char* name; void (*fn)(float*, int); int n; } tests[] = {
    {"v_fma_f32", k_fma, 8}, {"v_mul_f32", k_mul, 8}, {"v_max_f32", k_max, 8}, {"v_sub_u32", k_subu, 8},
    {"v_cvt_f32_i32", k_cvt_i32, 8}, {"v_cvt_f32_u32", k_cvt_u32, 8}, {"v_cvt_f32_ubyteN", k_cvt_ub, 8},
    {"v_cvt_f32_f16", k_cvt_f16, 8}, {"v_cvt_f32_f16 sdwa WORD_1", k_cvt_f16_sdwa, 8},
    {"v_rcp_f32", k_rcp, 8}, {"v_cmp_gt_f32 vcc", k_cmp, 8}, {"v_cmp + v_cndmask", k_cmp_cnd, 8},
    {"v_cmpx + v_max + s_mov exec (per pair)", k_cmpx, 4}, {"v_cmp + s_and exec + v_max + s_mov exec (per group)", k_cmp_sand, 4},
    {"hit block rcp form (per block of 5 VALU)", k_hit_rcp, 2}, {"hit block product form (per block of 5 VALU)", k_hit_prod, 2},
  };
  int ti = -1;
  for (auto& t : tests) {
    ++ti;
    if (only >= 0 && only != ti) continue;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, rout, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double units = (double)blocks * 4 * iters * t.n / 1024.0;  // per SIMD
    printf("%-55s %7.3f ms  %6.2f ns per unit per SIMD  (= %.2f cycles @2.4 GHz)\n", t.name, ms,
           ms * 1e6 / units, ms * 1e-3 * 2.4e9 / units);
  }
  for (int mode = 0; mode < 3; ++mode) {
    if (only >= 0 && only != 100 + mode) continue;
    float ms = 0;
    const int lb = 256 * 4, li = 8192;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(lb), dim3(512), 49152, 0, rout, li);
      else if (mode == 1) hipLaunchKernelGGL(k_lds<1>, dim3(lb), dim3(512), 49152, 0, rout, li);
      else hipLaunchKernelGGL(k_lds<2>, dim3(lb), dim3(512), 49152, 0, rout, li);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double wave_iters_per_cu = (double)lb * 8 * li / 256.0;
    printf("LDS %-50s %7.3f ms  %.2f cycles per wave-iteration per CU @2.4 GHz\n",
           mode == 0 ? "b128 per iteration" : mode == 1 ? "b128 + b32 per iteration" : "b128 + 20 fma per iteration", ms,
           ms * 1e-3 * 2.4e9 / wave_iters_per_cu);
  }
  return 0;
}
