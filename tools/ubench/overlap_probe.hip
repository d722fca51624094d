// overlap_probe -- can an HBM-bound pass (the sort's scatter / placement shape: chunk -> LDS -> coalesced
// copy-out) hide under a VALU-issue-bound kernel (the gather's shape: 512 threads, 37 KB of LDS, 4
// workgroups per CU) when both are launched on two HIP streams?  Measures V alone, H alone, V || H.
// Build: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// VALU-bound: the gather's mix (full-rate + half-rate ops), ~1500 instructions per wave, LDS allocated
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_valu(float* out, int iters) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float a = lds[(threadIdx.x * 7) & 511], b = a + 1.f, c = a + 2.f, d = a + 3.f;
  for (int i = 0; i < iters; ++i) {
    a = fmaf(a, b, c); b = fmaf(b, c, d); c = fmaxf(c, a); d = fmaf(d, a, b);
    a = __builtin_amdgcn_rcpf(a + 3.f); b = fmaf(b, c, d); c = fmaf(c, d, a); d = fmaxf(d, b);
  }
  if (a + b + c + d == 1234.5f) out[blockIdx.x] = a;
}

// HBM-bound: THREADS x PER 16-byte records through LDS, coalesced in and out
template <int THREADS, int PER>
__global__ void __launch_bounds__(THREADS) k_stream(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  extern __shared__ float4 stage[];
  const size_t base = (size_t)blockIdx.x * THREADS * PER;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const size_t i = base + threadIdx.x + (size_t)k * THREADS;
    if (i < n) stage[(threadIdx.x + k * THREADS) ^ 1] = src[i];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const size_t i = base + threadIdx.x + (size_t)k * THREADS;
    if (i < n) dst[i] = stage[threadIdx.x + k * THREADS];
  }
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t n = (size_t)75 << 20;  // 75 M x 16 B = 1.2 GB
  float4 *src, *dst;
  float* out;
  CK(hipMalloc(&src, n * 16));
  CK(hipMalloc(&dst, n * 16));
  CK(hipMalloc(&out, 1 << 22));
  CK(hipMemset(src, 1, n * 16));
  hipStream_t s1, s2, s2hi;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s2hi, hipStreamNonBlocking, hi));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  const int vblocks = 98304, viters = 48;
  const size_t vlds = 37 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_valu), hipFuncAttributeMaxDynamicSharedMemorySize, (int)vlds));
  auto launch_v = [&](hipStream_t s) { hipLaunchKernelGGL(k_valu, dim3(vblocks), dim3(512), vlds, s, out, viters); };
  // H variants: (512 thr, 4480 rec = 70 KB: the scatter pass) / (256 thr, 3072 rec = 48 KB: the placement pass)
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<512, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<256, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<256, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024));
  auto launch_h = [&](hipStream_t s, int variant) {
    if (variant == 0) hipLaunchKernelGGL((k_stream<512, 8>), dim3((unsigned)((n + 4095) / 4096)), dim3(512), 70 * 1024, s, src, dst, n);
    else if (variant == 1) hipLaunchKernelGGL((k_stream<256, 12>), dim3((unsigned)((n + 3071) / 3072)), dim3(256), 48 * 1024, s, src, dst, n);
    else hipLaunchKernelGGL((k_stream<256, 4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 16 * 1024, s, src, dst, n);
  };
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, s1)); launch_v(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const float v_alone = ms;
  printf("{\"V_alone_ms\": %.3f}\n", v_alone);
  for (int variant = 0; variant < 3; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s2)); launch_h(s2, variant); launch_h(s2, variant); CK(hipEventRecord(e1, s2));
      CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const float h_alone = ms;
    for (int prio = 0; prio < 2; ++prio) {
      hipStream_t sh = prio ? s2hi : s2;
      float total = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1));
        CK(hipStreamWaitEvent(sh, e0, 0));
        launch_v(s1);
        launch_h(sh, variant); launch_h(sh, variant);
        CK(hipEventRecord(e1, s1));
        CK(hipEventRecord(e2, sh));
        CK(hipStreamWaitEvent(s1, e2, 0));
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&total, e0, e1));
      }
      printf("{\"H_variant\": \"%s\", \"H_alone_ms (2 passes, 4.8 GB)\": %.3f, \"H_stream_priority\": \"%s\", \"V_and_H_ms\": %.3f, "
             "\"serial_ms\": %.3f, \"hidden_frac_of_H\": %.2f}\n",
             variant == 0 ? "512 thr / 70 KB" : variant == 1 ? "256 thr / 48 KB" : "256 thr / 16 KB", h_alone,
             prio ? "high" : "default", total, v_alone + h_alone, (v_alone + h_alone - total) / h_alone);
    }
  }
  return 0;
}
