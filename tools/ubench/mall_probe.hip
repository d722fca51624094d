// mall_probe.hip -- does the 256 MB Infinity Cache serve a streaming copy faster than HBM?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mall_probe.hip -o tools/ubench/mall_probe
// copy (16 B per lane, grid-stride) of working sets from 16 MB to 2 GB, each repeated so that the
// second and later passes find the set where the previous pass left it; and "produce then consume":
// kernel A writes a buffer, kernel B reads it (what two passes of a sort over one partition do).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ src, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 v = src[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_write(uint4* __restrict__ dst, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dst[i] = make_uint4(seed, (unsigned)i, seed ^ (unsigned)i, 7u);
}
int main() {
  const size_t maxb = (size_t)2 << 30;
  uint4 *a, *b;
  unsigned* sink;
  CK(hipMalloc(&a, maxb));
  CK(hipMalloc(&b, maxb));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, maxb));
  CK(hipMemset(b, 2, maxb));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1200, 2048}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    const int reps = (int)(((size_t)8 << 30) / bytes) + 2;
    float ms;
    // copy a -> b over the same `bytes` (working set 2 x bytes)
    hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double copy = 2.0 * bytes * reps / (ms * 1e-3) / 1e12;
    // read only
    hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, sink);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double rd = 1.0 * bytes * reps / (ms * 1e-3) / 1e12;
    // produce (write b) then consume (read b), alternating
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n, (unsigned)r);
      hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, b, n, sink);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double pc = 2.0 * bytes * reps / (ms * 1e-3) / 1e12;
    std::printf("%5zu MB: copy %.2f TB/s (r+w)   read %.2f TB/s   write-then-read %.2f TB/s (r+w)   [%d reps]\n", mb,
                copy, rd, pc, reps);
  }
  return 0;
}
