// Micro-benchmarks that ground the DSM gather design on gfx950:
//   - accuracy of v_rcp_f64 (+0/1/2 Newton steps) against IEEE division
//   - issue rate of v_fma_f64, v_rcp_f64, v_rcp_f32, v_cndmask, v_cmp_f64
//   - throughput of LDS f64 atomic adds (random cells, like an IDW scatter)
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_rcp_acc(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double r0 = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r0, 1.0);
  double r1 = fma(r0, e, r0);
  e = fma(-d, r1, 1.0);
  double r2 = fma(r1, e, r1);
  out[3 * i + 0] = r0;
  out[3 * i + 1] = r1;
  out[3 * i + 2] = r2;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(double* out, int iters, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  double a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 8 independent FMA f64
      a0 = fma(a0, 1.0000001, 0.5); a1 = fma(a1, 1.0000001, 0.5); a2 = fma(a2, 1.0000001, 0.5); a3 = fma(a3, 1.0000001, 0.5);
      a4 = fma(a4, 1.0000001, 0.5); a5 = fma(a5, 1.0000001, 0.5); a6 = fma(a6, 1.0000001, 0.5); a7 = fma(a7, 1.0000001, 0.5);
    } else if (MODE == 1) {  // 8 independent rcp f64
      a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
      a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7);
    } else if (MODE == 2) {  // 8 rcp f32 (4 regs x2)
      f0 = __builtin_amdgcn_rcpf(f0); f1 = __builtin_amdgcn_rcpf(f1); f2 = __builtin_amdgcn_rcpf(f2); f3 = __builtin_amdgcn_rcpf(f3);
      f0 = __builtin_amdgcn_rcpf(f0); f1 = __builtin_amdgcn_rcpf(f1); f2 = __builtin_amdgcn_rcpf(f2); f3 = __builtin_amdgcn_rcpf(f3);
    } else if (MODE == 3) {  // 8 compare+select f64
      a0 = a0 < a1 ? a2 : a3; a1 = a1 < a2 ? a3 : a4; a2 = a2 < a3 ? a4 : a5; a3 = a3 < a4 ? a5 : a6;
      a4 = a4 < a5 ? a6 : a7; a5 = a5 < a6 ? a7 : a0; a6 = a6 < a7 ? a0 : a1; a7 = a7 < a0 ? a1 : a2;
    } else if (MODE == 4) {  // 8 add f64
      a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0;
    } else if (MODE == 5) {  // 8 cvt f64->f32->f64
      a0 = (double)(float)a0 + 1; a1 = (double)(float)a1 + 1; a2 = (double)(float)a2 + 1; a3 = (double)(float)a3 + 1;
    } else if (MODE == 6) {  // 8 v_fma_f32 (two dependent rounds over 4 registers)
      f0 = fmaf(f0, 1.0000001f, 0.5f); f1 = fmaf(f1, 1.0000001f, 0.5f); f2 = fmaf(f2, 1.0000001f, 0.5f); f3 = fmaf(f3, 1.0000001f, 0.5f);
      f0 = fmaf(f0, 0.9999999f, 0.25f); f1 = fmaf(f1, 0.9999999f, 0.25f); f2 = fmaf(f2, 0.9999999f, 0.25f); f3 = fmaf(f3, 0.9999999f, 0.25f);
    } else if (MODE == 7) {  // 8 v_pk_fma_f32 (16 f32 fma)
      typedef float fpair __attribute__((ext_vector_type(2)));
      fpair p0 = {f0, f1}, p1 = {f2, f3}, c = {1.0000001f, 0.9999999f}, d = {0.5f, 0.25f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(c), "v"(d));
      }
      f0 = p0.x; f1 = p0.y; f2 = p1.x; f3 = p1.y;
    } else if (MODE == 8) {  // 8 x (v_sub_u32 + v_cvt_f32_i32): counted as 16 instructions
      int i0 = __float_as_int(f0), i1 = __float_as_int(f1), i2 = __float_as_int(f2), i3 = __float_as_int(f3);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f0 = (float)(i0 - i1); f1 = (float)(i1 - i2); f2 = (float)(i2 - i3); f3 = (float)(i3 - i0);
        i0 = __float_as_int(f0) ^ it; i1 = __float_as_int(f1); i2 = __float_as_int(f2); i3 = __float_as_int(f3);
      }
    } else if (MODE == 9) {  // the gather's masked hit block, twice: 2 x (cmpx + rcp + max + add + fmac)
      unsigned long long sv;
      asm volatile(
          "s_mov_b64 %4, exec\n\t"
          "v_cmpx_gt_f32 %5, %0\n\tv_rcp_f32 %1, %0\n\tv_max_f32 %2, %2, %0\n\tv_add_f32 %3, %3, %1\n\tv_fmac_f32 %0, %1, %5\n\t"
          "s_mov_b64 exec, %4\n\t"
          "v_cmpx_gt_f32 %5, %3\n\tv_rcp_f32 %1, %3\n\tv_max_f32 %2, %2, %3\n\tv_add_f32 %0, %0, %1\n\tv_fmac_f32 %3, %1, %5\n\t"
          "s_mov_b64 exec, %4"
          : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "=&s"(sv) : "v"(1.5f) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3;
}

// LDS f64 atomic adds into 2048 accumulators at pseudo-random cells
template <int KIND>
__global__ void __launch_bounds__(512) k_lds_atomic(double* out, int iters) {
  __shared__ double acc[4096];
  for (int k = threadIdx.x; k < 4096; k += blockDim.x) acc[k] = 0.0;
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
  double v = 1.0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    unsigned cell = (KIND == 0) ? ((s >> 8) & 2047u)                       // random
                                : ((threadIdx.x * 1u + (it * 37u)) & 2047u);  // lane-contiguous
    atomicAdd(&acc[cell], v);
    atomicAdd(&acc[2048 + cell], v * 0.5);
  }
  __syncthreads();
  double t = 0;
  for (int k = threadIdx.x; k < 4096; k += blockDim.x) t += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main() {
  // ---- accuracy -------------------------------------------------------------
  const int n = 1 << 20;
  std::vector<double> hx(n), ho(3 * n);
  unsigned long long st = 88172645463325252ULL;
  for (int i = 0; i < n; ++i) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    double u = (st >> 11) * (1.0 / 9007199254740992.0);
    hx[i] = std::pow(10.0, -8.0 + 9.0 * u);  // 1e-8 .. 10
  }
  double *dx, *dout;
  CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dout, 3 * n * 8));
  CK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
  k_rcp_acc<<<n / 256, 256>>>(dx, dout, n);
  CK(hipMemcpy(ho.data(), dout, 3 * n * 8, hipMemcpyDeviceToHost));
  double m[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    long double ex = 1.0L / (long double)hx[i];
    for (int k = 0; k < 3; ++k) {
      double rel = (double)fabsl(((long double)ho[3 * i + k] - ex) / ex);
      if (rel > m[k]) m[k] = rel;
    }
  }
  printf("v_rcp_f64 max rel err: raw %.3e (2^%.1f)  +1 Newton %.3e (2^%.1f)  +2 Newton %.3e (2^%.1f)\n",
         m[0], log2(m[0]), m[1], log2(m[1]), m[2], log2(m[2] > 0 ? m[2] : 1e-30));

  // ---- issue rates ------------------------------------------------------------
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double* rout; CK(hipMalloc(&rout, 256 * 8 * 256 * 8 * 2));
  const int blocks = 256 * 8, iters = 4096;
  const char* names[] = {"v_fma_f64", "v_rcp_f64", "v_rcp_f32", "cmp+select f64", "v_add_f64", "cvt f64<->f32 (+add)",
                         "v_fma_f32", "v_pk_fma_f32", "v_sub_u32+v_cvt_f32_i32", "masked hit block (10 VALU)"};
  for (int mode = 0; mode < 10; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      switch (mode) {
        case 0: k_rate<0><<<blocks, 256>>>(rout, iters, 1.0); break;
        case 1: k_rate<1><<<blocks, 256>>>(rout, iters, 1.0); break;
        case 2: k_rate<2><<<blocks, 256>>>(rout, iters, 1.0); break;
        case 3: k_rate<3><<<blocks, 256>>>(rout, iters, 1.0); break;
        case 4: k_rate<4><<<blocks, 256>>>(rout, iters, 1.0); break;
        case 5: k_rate<5><<<blocks, 256>>>(rout, iters, 1.0); break;
        case 6: k_rate<6><<<blocks, 256>>>(rout, iters, 1.0); break;
        case 7: k_rate<7><<<blocks, 256>>>(rout, iters, 1.0); break;
        case 8: k_rate<8><<<blocks, 256>>>(rout, iters, 1.0); break;
        case 9: k_rate<9><<<blocks, 256>>>(rout, iters, 1.0); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double ops = (double)blocks * 256 * iters * (mode == 5 ? 4 : (mode == 8 ? 16 : (mode == 9 ? 10 : 8)));
    // 8 blocks of 4 waves per CU = 8 waves/SIMD; cycles per wave-instruction per SIMD
    const double wave_instr_per_simd = ops / 64.0 / 1024.0;
    printf("%-22s %8.3f ms  %7.2f Gop/s/lane-total  ~%.2f cycles per wave-instr per SIMD @2.4GHz\n", names[mode], ms,
           ops / ms * 1e-6, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
  }
  for (int kind = 0; kind < 2; ++kind) {
    float ms = 0;
    const int ab = 256 * 2, ai = 2048;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (kind == 0) k_lds_atomic<0><<<ab, 512>>>(rout, ai); else k_lds_atomic<1><<<ab, 512>>>(rout, ai);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double atom = (double)ab * 512 * ai * 2;
    printf("LDS atomicAdd f64 (%s): %8.3f ms, %.1f G atomics/s chip, %.2f cycles per wave-atomic per CU @2.4GHz\n",
           kind == 0 ? "random cells" : "lane-contiguous", ms, atom / ms * 1e-6,
           ms * 1e-3 * 2.4e9 / (atom / 64.0 / 256.0));
  }
  return 0;
}
