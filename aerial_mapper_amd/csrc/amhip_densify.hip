// amhip_densify.hip -- disparity map -> world points.
//
// Replaces the per-pixel loop of stereo::Densifier::computePointCloud
// (aerial_mapper_dense_pcl/src/densifier.cpp:48-107): valid pixels
// (disparity > kMaxInvalidDisparity, finite z) become world points in RASTER
// order, with the left image's gray value as intensity.  The output stays in
// HBM, in exactly the layout amhip_dsm_process_dev / amhip_ortho_from_pcl_
// process_dev take -- the incremental pipeline never ships the cloud over PCIe.
//
// Order-preserving compaction: pass 1 counts the valid pixels of every
// 1024-pixel block, an exclusive scan turns the counts into offsets, pass 2
// recomputes the points and writes them at offset + rank-in-block.
#include "amhip_common.h"

namespace amhip {

constexpr int kDensifyThreads = 256;
constexpr int kDensifyPerThread = 4;
constexpr int kDensifyBlock = kDensifyThreads * kDensifyPerThread;

__device__ __forceinline__ bool densify_pixel(const DensifyParams& p, const float* disparity,
                                              long long lin, double* gx, double* gy,
                                              double* gz) {
  const int v = (int)(lin / p.width);
  const int u = (int)(lin - (long long)v * p.width);
  const float d = *reinterpret_cast<const float*>(
      reinterpret_cast<const unsigned char*>(disparity) + (size_t)v * p.disp_step +
      (size_t)u * sizeof(float));
  if (!(d > 1.0f)) return false;  // kMaxInvalidDisparity (densifier.cpp:60)
  const double w = p.Q32 * (double)d;
  const double px = ((double)u + p.Q03) / w;
  const double py = (p.Q11 * (double)v + p.Q13) / w;
  const double pz = p.Q23 / w;
  *gx = ((p.R[0] * px + p.R[1] * py) + p.R[2] * pz) + p.t[0];
  *gy = ((p.R[3] * px + p.R[4] * py) + p.R[5] * pz) + p.t[1];
  *gz = ((p.R[6] * px + p.R[7] * py) + p.R[8] * pz) + p.t[2];
  const float zf = (float)*gz;
  return !isinf(zf);  // densifier.cpp:76
}

__device__ __forceinline__ unsigned block_scan_256(unsigned v, unsigned* total, unsigned* lds) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) lds[wid] = incl;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kDensifyThreads / 64; ++w) {
    const unsigned t = lds[w];
    if (w < wid) base += t;
    tot += t;
  }
  *total = tot;
  __syncthreads();
  return base + incl - v;
}

__global__ void __launch_bounds__(kDensifyThreads)
k_densify_count(DensifyParams p, const float* __restrict__ disparity,
                uint32_t* __restrict__ block_counts) {
  __shared__ unsigned lds[kDensifyThreads / 64];
  const long long npix = (long long)p.width * p.height;
  // thread t owns kDensifyPerThread CONSECUTIVE pixels (keeps raster order)
  const long long base = (long long)blockIdx.x * kDensifyBlock + (long long)threadIdx.x * kDensifyPerThread;
  unsigned c = 0;
#pragma unroll
  for (int k = 0; k < kDensifyPerThread; ++k) {
    double gx, gy, gz;
    if (base + k < npix && densify_pixel(p, disparity, base + k, &gx, &gy, &gz)) ++c;
  }
  unsigned total;
  (void)block_scan_256(c, &total, lds);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// One block: exclusive scan of the per-block counts; grand total to *total.
__global__ void __launch_bounds__(kDensifyThreads)
k_densify_scan(uint32_t* __restrict__ block_counts, int nblocks, long long* __restrict__ total) {
  __shared__ unsigned lds[kDensifyThreads / 64];
  unsigned carry = 0;
  for (int b0 = 0; b0 < nblocks; b0 += kDensifyThreads) {
    const int i = b0 + threadIdx.x;
    const unsigned v = i < nblocks ? block_counts[i] : 0u;
    unsigned tot;
    const unsigned ex = block_scan_256(v, &tot, lds);
    if (i < nblocks) block_counts[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = (long long)carry;
}

__global__ void __launch_bounds__(kDensifyThreads)
k_densify_emit(DensifyParams p, const float* __restrict__ disparity,
               const uint8_t* __restrict__ image_left, const uint32_t* __restrict__ block_offsets,
               double* __restrict__ xyz_out, int32_t* __restrict__ intensity_out,
               unsigned long long capacity) {
  __shared__ unsigned lds[kDensifyThreads / 64];
  const long long npix = (long long)p.width * p.height;
  const long long base = (long long)blockIdx.x * kDensifyBlock + (long long)threadIdx.x * kDensifyPerThread;
  double gx[kDensifyPerThread], gy[kDensifyPerThread], gz[kDensifyPerThread];
  bool ok[kDensifyPerThread];
  unsigned c = 0;
#pragma unroll
  for (int k = 0; k < kDensifyPerThread; ++k) {
    ok[k] = base + k < npix && densify_pixel(p, disparity, base + k, &gx[k], &gy[k], &gz[k]);
    c += ok[k] ? 1u : 0u;
  }
  unsigned total;
  unsigned long long slot = (unsigned long long)block_offsets[blockIdx.x] + block_scan_256(c, &total, lds);
#pragma unroll
  for (int k = 0; k < kDensifyPerThread; ++k) {
    if (!ok[k]) continue;
    if (slot < capacity) {
      const long long lin = base + k;
      const int v = (int)(lin / p.width);
      const int u = (int)(lin - (long long)v * p.width);
      xyz_out[3 * slot + 0] = gx[k];
      xyz_out[3 * slot + 1] = gy[k];
      xyz_out[3 * slot + 2] = gz[k];
      intensity_out[slot] = (int32_t)image_left[(size_t)v * p.img_step + u];
    }
    ++slot;
  }
}

int densify_run(Ctx* c, const DensifyParams& p, const float* dev_disparity,
                const uint8_t* dev_image_left, double* dev_xyz_out, int32_t* dev_intensity_out,
                size_t capacity, long long* dev_count) {
  const long long npix = (long long)p.width * p.height;
  const int nblocks = (int)((npix + kDensifyBlock - 1) / kDensifyBlock);
  int rc;
  if ((rc = ensure_capacity(&c->scan_partials, &c->partial_cap, (size_t)nblocks + 4))) return rc;
  ScopedTimer t(c, AMHIP_K_MISC);
  hipLaunchKernelGGL(k_densify_count, dim3(nblocks), dim3(kDensifyThreads), 0, c->stream, p,
                     dev_disparity, c->scan_partials);
  hipLaunchKernelGGL(k_densify_scan, dim3(1), dim3(kDensifyThreads), 0, c->stream,
                     c->scan_partials, nblocks, dev_count);
  hipLaunchKernelGGL(k_densify_emit, dim3(nblocks), dim3(kDensifyThreads), 0, c->stream, p,
                     dev_disparity, dev_image_left, c->scan_partials, dev_xyz_out,
                     dev_intensity_out, (unsigned long long)capacity);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

}  // namespace amhip
