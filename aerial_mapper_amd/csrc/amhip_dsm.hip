// amhip_dsm.hip -- point cloud -> DSM on MI355X (gfx950).
//
// Replaces the kd-tree + per-cell radius search of the reference
//   Dsm::initializeAndFillKdTree            aerial_mapper_dsm/src/dsm.cc:36-52
//   Dsm::updateElevationLayerMultiThreaded  dsm.cc:113-184
// with a counting sort of the points into a uniform grid of bins aligned to
// the map cells, followed by a per-cell gather over the bins that can hold a
// point inside the search radius.  The SET of neighbours of every cell is the
// reference's (same double-precision, non-fused d2 = dx*dx + dy*dy, same
// strict `d2 < T`, same expanding-radius ladder); only the summation ORDER of
// the inverse-squared-distance weights differs (bin order instead of kd-tree
// visiting order), which moves the double sums by ~1e-16 relative.
//
// Kernels (integer / FP64 streaming work, no MFMA), see DESIGN.md section 4:
//   sort     k_dsm_p3_*      three-pass partition sort (default for >= 1 M
//                            points): count -> two LDS-staged scatter passes ->
//                            in-LDS placement per sub-partition
//            k_dsm_stripe_*  two-level stripe sort (smaller clouds)
//            k_dsm_bin_count / k_scan_* / k_dsm_scatter
//                            one-level counting sort with global atomics
//                            (very wide grids; fallback)
//            All three leave the points of one bin contiguous, bins row-major,
//            and bin_start[] = exclusive offsets.
//   gather   k_dsm_gather_tiled   one workgroup per 64 x 16 (or 64 x 32) cells:
//                            stage the tile's points in LDS once, re-bin them
//                            at cell granularity, per-lane radius search +
//                            division-free IDW; fallback ladder / overflow /
//                            numerically extreme cells go to
//            cell_global / cell_fallback_global / k_dsm_gather
//                            the same search on the global bins (also used for
//                            grids finer than 16 cells per radius and the
//                            adaptive OrthoFromPcl passes)
//   k_halo_select           multi-GPU: compact the points other windows need
#include <cstdlib>

#include "amhip_common.h"

namespace amhip {

// ---------------------------------------------------------------------------
// small helpers
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

__global__ void k_fill_f32(float* __restrict__ dst, size_t n, float value) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    dst[i] = value;
}

int launch_fill(Ctx* c, float* dst, size_t n, float value) {
  if (n == 0) return AMHIP_OK;
  ScopedTimer t(c, AMHIP_K_MISC);
  const int block = 256;
  size_t grid = (n + block - 1) / block;
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)grid), dim3(block), 0,
                     c->stream, dst, n, value);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

// ---------------------------------------------------------------------------
// binning
// ---------------------------------------------------------------------------
constexpr uint32_t kNoRank = 0xFFFFFFFFu;

// Bin of a (centre-shifted) point, or false if it lies more than M cells
// outside the grid (it can then never be inside any cell's last fallback
// radius).  Cell i has its centre at continuous coordinate ci == i.
__device__ __forceinline__ bool point_bin(const DsmParams& p, double px,
                                          double py, uint32_t* bin) {
  const double cx = (p.base_x - px) * p.inv_res - (double)p.i_off;
  const double cy = (p.base_y - py) * p.inv_res - (double)p.j_off;
  const double lo = -(double)p.M - 0.5;
  const double hx = (double)(p.rows + p.M) - 0.5;
  const double hy = (double)(p.cols + p.M) - 0.5;
  if (!(cx >= lo && cx < hx && cy >= lo && cy < hy)) return false;  // NaN too
  int ix = (int)floor(cx + 0.5) + p.M;
  int iy = (int)floor(cy + 0.5) + p.M;
  ix = min(max(ix, 0), p.rows + 2 * p.M - 1);
  iy = min(max(iy, 0), p.cols + 2 * p.M - 1);
  const int bx = ix / p.B;
  const int by = iy / p.B;
  *bin = (uint32_t)by * (uint32_t)p.nbx + (uint32_t)bx;
  return true;
}

__global__ void __launch_bounds__(256)
k_dsm_bin_count(const double* __restrict__ xyz, size_t n, DsmParams p,
                uint32_t* __restrict__ cnt, uint32_t* __restrict__ rank) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += stride) {
    const double x = xyz[3 * idx + 0];
    const double y = xyz[3 * idx + 1];
    const double px = x - p.sub_x;  // dsm.cc:42
    const double py = y - p.sub_y;  // dsm.cc:43
    uint32_t bin;
    uint32_t r = kNoRank;
    if (point_bin(p, px, py, &bin)) r = atomicAdd(&cnt[bin], 1u);
    rank[idx] = r;
  }
}

__global__ void __launch_bounds__(256)
k_dsm_scatter(const double* __restrict__ xyz, const int32_t* __restrict__ values, size_t n,
              DsmParams p, const uint32_t* __restrict__ start,
              const uint32_t* __restrict__ rank, double* __restrict__ sorted) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += stride) {
    const uint32_t r = rank[idx];
    if (r == kNoRank) continue;
    const double x = xyz[3 * idx + 0];
    const double y = xyz[3 * idx + 1];
    const double z = values ? (double)values[idx] : xyz[3 * idx + 2];
    const double px = x - p.sub_x;
    const double py = y - p.sub_y;
    uint32_t bin;
    point_bin(p, px, py, &bin);  // same arithmetic as the count pass
    const size_t slot = (size_t)start[bin] + r;
    sorted[3 * slot + 0] = px;
    sorted[3 * slot + 1] = py;
    sorted[3 * slot + 2] = z;
  }
}

// ---------------------------------------------------------------------------
// two-level stripe sort (clouds below the partition sort's threshold; AMHIP_SORT_TWO_LEVEL=1)
// ---------------------------------------------------------------------------
// The one-level counting sort above pays one device-scope atomic and two
// random 24..64-byte HBM transactions per point.  The stripe sort replaces it:
//   level 1  points -> STRIPES (a few consecutive bin rows, ~1000 stripes).
//            Per-workgroup LDS histograms aggregate the global atomics (one
//            per stripe per 16 K points) and every workgroup appends runs of
//            consecutive points to each stripe -> near-streaming writes.
//   level 2  one workgroup per stripe: LDS histogram over the stripe's bins,
//            LDS scan -> bin_start[] for those bins, then the points are
//            placed; a stripe is ~1 MB, so the second read and the random
//            placement stay inside the XCD's L2.
// Stripes are whole bin rows, so the final order is still row-major by bin.
constexpr int kMaxStripes = 8192;
constexpr int kMaxStripeBins = 8192;
constexpr int kL1Threads = 256;
constexpr int kL1Chunk = 65536;  // points per workgroup in the level-1 scatter (A/B: 16K..128K)
constexpr int kL2Threads = 512;

__device__ __forceinline__ bool point_bin_xy(const DsmParams& p, double px, double py,
                                             int* bx, int* by) {
  const double cx = (p.base_x - px) * p.inv_res - (double)p.i_off;
  const double cy = (p.base_y - py) * p.inv_res - (double)p.j_off;
  const double lo = -(double)p.M - 0.5;
  const double hx = (double)(p.rows + p.M) - 0.5;
  const double hy = (double)(p.cols + p.M) - 0.5;
  if (!(cx >= lo && cx < hx && cy >= lo && cy < hy)) return false;  // NaN too
  int ix = (int)floor(cx + 0.5) + p.M;
  int iy = (int)floor(cy + 0.5) + p.M;
  ix = min(max(ix, 0), p.rows + 2 * p.M - 1);
  iy = min(max(iy, 0), p.cols + 2 * p.M - 1);
  *bx = ix / p.B;
  *by = iy / p.B;
  return true;
}

__global__ void __launch_bounds__(kL1Threads)
k_dsm_stripe_count(const double* __restrict__ xyz, size_t n, DsmParams p,
                   uint32_t* __restrict__ stripe_cnt) {
  extern __shared__ uint32_t s_hist[];
  for (int k = threadIdx.x; k < p.nstripes; k += kL1Threads) s_hist[k] = 0;
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * kL1Threads;
  for (size_t idx = (size_t)blockIdx.x * kL1Threads + threadIdx.x; idx < n; idx += stride) {
    const double px = xyz[3 * idx + 0] - p.sub_x;  // dsm.cc:42
    const double py = xyz[3 * idx + 1] - p.sub_y;  // dsm.cc:43
    int bx, by;
    if (point_bin_xy(p, px, py, &bx, &by)) atomicAdd(&s_hist[by / p.stripe_rows], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < p.nstripes; k += kL1Threads) {
    const uint32_t c = s_hist[k];
    if (c) atomicAdd(&stripe_cnt[k], c);
  }
}

// One block: stripe_start = exclusive scan of stripe_cnt (+ total), and a copy
// that the level-1 scatter uses as its append cursors.
__global__ void __launch_bounds__(1024)
k_dsm_stripe_scan(const uint32_t* __restrict__ stripe_cnt, int nstripes,
                  uint32_t* __restrict__ stripe_start, uint32_t* __restrict__ cursor) {
  __shared__ unsigned lds[1024 / 64 + 1];
  unsigned carry = 0;
  for (int base = 0; base < nstripes; base += 1024) {
    const int i = base + threadIdx.x;
    const unsigned v = (i < nstripes) ? stripe_cnt[i] : 0u;
    unsigned total;
    const unsigned ex = block_excl_scan<1024>(v, &total, lds);
    if (i < nstripes) {
      stripe_start[i] = carry + ex;
      cursor[i] = carry + ex;
    }
    carry += total;
  }
  if (threadIdx.x == 0) stripe_start[nstripes] = carry;
}

__global__ void __launch_bounds__(kL1Threads)
k_dsm_stripe_scatter(const double* __restrict__ xyz, const int32_t* __restrict__ values,
                     size_t n, DsmParams p, uint32_t* __restrict__ cursor,
                     double* __restrict__ tmp) {
  extern __shared__ uint32_t s_mem[];
  uint32_t* s_cnt = s_mem;                // points of this chunk per stripe / local rank
  uint32_t* s_base = s_mem + p.nstripes;  // where this chunk's run of a stripe starts
  const size_t c0 = (size_t)blockIdx.x * kL1Chunk;
  const size_t c1 = min(c0 + (size_t)kL1Chunk, n);
  for (int k = threadIdx.x; k < p.nstripes; k += kL1Threads) s_cnt[k] = 0;
  __syncthreads();
  for (size_t idx = c0 + threadIdx.x; idx < c1; idx += kL1Threads) {
    const double px = xyz[3 * idx + 0] - p.sub_x;
    const double py = xyz[3 * idx + 1] - p.sub_y;
    int bx, by;
    if (point_bin_xy(p, px, py, &bx, &by)) atomicAdd(&s_cnt[by / p.stripe_rows], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < p.nstripes; k += kL1Threads) {
    const uint32_t c = s_cnt[k];
    s_base[k] = c ? atomicAdd(&cursor[k], c) : 0u;
    s_cnt[k] = 0;
  }
  __syncthreads();
  for (size_t idx = c0 + threadIdx.x; idx < c1; idx += kL1Threads) {  // (L2 hits)
    const double px = xyz[3 * idx + 0] - p.sub_x;
    const double py = xyz[3 * idx + 1] - p.sub_y;
    int bx, by;
    if (point_bin_xy(p, px, py, &bx, &by)) {
      const int st = by / p.stripe_rows;
      const size_t slot = (size_t)s_base[st] + atomicAdd(&s_cnt[st], 1u);
      tmp[3 * slot + 0] = px;
      tmp[3 * slot + 1] = py;
      tmp[3 * slot + 2] = values ? (double)values[idx] : xyz[3 * idx + 2];
    }
  }
}

__global__ void __launch_bounds__(kL2Threads)
k_dsm_stripe_sort(const double* __restrict__ tmp, DsmParams p,
                  const uint32_t* __restrict__ stripe_start,
                  uint32_t* __restrict__ bin_start, double* __restrict__ sorted) {
  extern __shared__ uint32_t s_bins[];  // bins of this stripe (+ scan scratch behind)
  const int st = blockIdx.x;
  const int row0 = st * p.stripe_rows;
  const int nrow = min(p.stripe_rows, p.nby - row0);
  const int nb = nrow * p.nbx;
  uint32_t* s_scan = s_bins + nb;
  const uint32_t g0 = stripe_start[st];
  const uint32_t g1 = stripe_start[st + 1];
  for (int k = threadIdx.x; k < nb; k += kL2Threads) s_bins[k] = 0;
  __syncthreads();
  for (uint32_t idx = g0 + threadIdx.x; idx < g1; idx += kL2Threads) {
    const double px = tmp[3 * (size_t)idx + 0];
    const double py = tmp[3 * (size_t)idx + 1];
    int bx, by;
    point_bin_xy(p, px, py, &bx, &by);  // same arithmetic as level 1: always inside
    atomicAdd(&s_bins[(by - row0) * p.nbx + bx], 1u);
  }
  __syncthreads();
  {
    const int per = (nb + kL2Threads - 1) / kL2Threads;
    const int lo = threadIdx.x * per;
    const int hi = min(lo + per, nb);
    unsigned sum = 0;
    for (int k = lo; k < hi; ++k) sum += s_bins[k];
    unsigned total;
    unsigned run = block_excl_scan<kL2Threads>(sum, &total, s_scan);
    for (int k = lo; k < hi; ++k) {
      const unsigned t = s_bins[k];
      s_bins[k] = run;
      run += t;
    }
  }
  __syncthreads();
  uint32_t* out_start = bin_start + (size_t)row0 * p.nbx;
  for (int k = threadIdx.x; k < nb; k += kL2Threads) out_start[k] = g0 + s_bins[k];
  if (st == p.nstripes - 1 && threadIdx.x == 0)
    bin_start[(size_t)p.nbx * p.nby] = stripe_start[p.nstripes];
  __syncthreads();
  for (uint32_t idx = g0 + threadIdx.x; idx < g1; idx += kL2Threads) {  // (L2 hits)
    const double px = tmp[3 * (size_t)idx + 0];
    const double py = tmp[3 * (size_t)idx + 1];
    const double pz = tmp[3 * (size_t)idx + 2];
    int bx, by;
    point_bin_xy(p, px, py, &bx, &by);
    const size_t slot = (size_t)g0 + atomicAdd(&s_bins[(by - row0) * p.nbx + bx], 1u);
    sorted[3 * slot + 0] = px;
    sorted[3 * slot + 1] = py;
    sorted[3 * slot + 2] = pz;
  }
}

// ---------------------------------------------------------------------------
// three-pass partition sort (the default for clouds that are worth it)
// ---------------------------------------------------------------------------
// The stripe sort above appends 24-byte records to ~1000 open runs per
// workgroup straight from registers; the partially written cache lines do not
// survive in the L2 until their neighbours arrive, and the PMC counters show
// 2-3x the algorithmic write traffic.  Here every pass sorts its chunk in LDS
// first and then writes each run with consecutive lanes on consecutive
// addresses, so whole lines leave the CU at once; the price is a third pass
// (partition counts per pass are limited by run length = chunk / partitions):
//   count    one read of the cloud: private LDS histograms over (k1, k2)
//            [k1 = group of p3_r1 bin rows, k2 = (row in group, column block)],
//            one row of counters per workgroup (no global atomics), then a
//            reduction + scan -> the exact final position of every (k1, k2).
//   pass 1   cloud -> k1 partitions      (<= 128, LDS-staged runs)
//   pass 2   k1 partition -> its k2 sub-partitions (<= 256, LDS-staged runs)
//   pass 3   one workgroup per sub-partition (~1.5 K points, fits LDS):
//            counting sort by bin in LDS, bin_start[] for its bins, one
//            contiguous coalesced copy out.
// Sub-partitions are ordered (bin row, column block), so the result is the
// same row-major-by-bin order the gather kernels expect.
constexpr int kP3CountThreads = 1024;
constexpr int kP3Threads = 512;
constexpr int kP3Chunk = 2560;  // points staged per scatter workgroup (70 KB of LDS: 2 per CU)
constexpr int kP3PerThread = kP3Chunk / kP3Threads;
constexpr int kP3MaxKeys = 256;
constexpr int kP3PlaceThreads = 256;
constexpr int kP3PlaceMaxCap = 2048;  // LDS capacity (points) the register-resident path handles
constexpr int kP3PlacePer = kP3PlaceMaxCap / kP3PlaceThreads;

__device__ __forceinline__ bool p3_keys(const DsmParams& p, double px, double py, int* k1,
                                        int* k2) {
  int bx, by;
  if (!point_bin_xy(p, px, py, &bx, &by)) return false;
  const int a = by / p.p3_r1;
  *k1 = a;
  *k2 = (by - a * p.p3_r1) * p.p3_c + bx / p.p3_w;
  return true;
}

__global__ void __launch_bounds__(kP3CountThreads)
k_dsm_p3_count(const double* __restrict__ xyz, size_t n, DsmParams p,
               uint32_t* __restrict__ hist_rows) {
  extern __shared__ uint32_t s_hist[];
  const int nk = p.p3_n1 * p.p3_n2;
  for (int k = threadIdx.x; k < nk; k += kP3CountThreads) s_hist[k] = 0;
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * kP3CountThreads;
  for (size_t idx = (size_t)blockIdx.x * kP3CountThreads + threadIdx.x; idx < n; idx += stride) {
    const double px = xyz[3 * idx + 0] - p.sub_x;  // dsm.cc:42
    const double py = xyz[3 * idx + 1] - p.sub_y;  // dsm.cc:43
    int k1, k2;
    if (p3_keys(p, px, py, &k1, &k2)) atomicAdd(&s_hist[k1 * p.p3_n2 + k2], 1u);
  }
  __syncthreads();
  uint32_t* row = hist_rows + (size_t)blockIdx.x * nk;
  for (int k = threadIdx.x; k < nk; k += kP3CountThreads) row[k] = s_hist[k];
}

__global__ void __launch_bounds__(256)
k_dsm_p3_reduce(const uint32_t* __restrict__ hist_rows, int nrows, int nk,
                uint32_t* __restrict__ cnt) {
  // 64 counters per workgroup, the rows dealt to its four waves
  __shared__ uint32_t s_part[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  uint32_t s = 0;
  if (k < nk)
    for (int r = wid; r < nrows; r += 4) s += hist_rows[(size_t)r * nk + k];
  s_part[wid][lane] = s;
  __syncthreads();
  if (wid == 0 && k < nk) cnt[k] = s_part[0][lane] + s_part[1][lane] + s_part[2][lane] + s_part[3][lane];
}

// One block.  start2 = exclusive scan of the (k1, k2) counts (+ total) and a
// copy as the pass-2 append cursors; start1 / cursor1 for pass 1; blk2 = first
// pass-2 workgroup of every k1 partition (partitions are cut into chunks).
__global__ void __launch_bounds__(1024)
k_dsm_p3_scan(const uint32_t* __restrict__ cnt, int n1, int n2,
              uint32_t* __restrict__ start2, uint32_t* __restrict__ cursor2,
              uint32_t* __restrict__ start1, uint32_t* __restrict__ cursor1,
              uint32_t* __restrict__ blk2) {
  __shared__ unsigned lds[1024 / 64 + 1];
  __shared__ unsigned s_start1[kP3MaxKeys + 1];
  const int nk = n1 * n2;
  unsigned carry = 0;
  for (int base = 0; base < nk; base += 1024) {
    const int i = base + threadIdx.x;
    const unsigned v = (i < nk) ? cnt[i] : 0u;
    unsigned total;
    const unsigned ex = block_excl_scan<1024>(v, &total, lds);
    if (i < nk) {
      start2[i] = carry + ex;
      cursor2[i] = carry + ex;
      if (i % n2 == 0) s_start1[i / n2] = carry + ex;
    }
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    start2[nk] = carry;
    s_start1[n1] = carry;
  }
  __syncthreads();
  const int k = threadIdx.x;
  unsigned nblk = 0;
  if (k < n1) {
    start1[k] = s_start1[k];
    cursor1[k] = s_start1[k];
    nblk = (s_start1[k + 1] - s_start1[k] + kP3Chunk - 1) / kP3Chunk;
  }
  if (k == 0) start1[n1] = carry;
  unsigned total;
  const unsigned ex = block_excl_scan<1024>(nblk, &total, lds);
  if (k < n1) blk2[k] = ex;
  if (k == 0) blk2[n1] = total;
}

// Passes 1 and 2.  kFirst: chunk of the input cloud, key k1, values/centre
// handling of the reference; else: chunk of one k1 partition, key k2.
template <bool kFirst>
__global__ void __launch_bounds__(kP3Threads)
k_dsm_p3_scatter(const double* __restrict__ src, const int32_t* __restrict__ values, size_t n,
                 DsmParams p, const uint32_t* __restrict__ start1,
                 const uint32_t* __restrict__ blk2, uint32_t* __restrict__ cursor,
                 double* __restrict__ dst) {
  extern __shared__ double s_pts[];                                       // 3 * kP3Chunk
  uint32_t* s_dest = reinterpret_cast<uint32_t*>(s_pts + 3 * kP3Chunk);   // kP3Chunk
  uint32_t* s_cnt = s_dest + kP3Chunk;                                    // kP3MaxKeys
  uint32_t* s_off = s_cnt + kP3MaxKeys;
  uint32_t* s_base = s_off + kP3MaxKeys;
  uint32_t* s_scan = s_base + kP3MaxKeys;  // 24
  const int tid = threadIdx.x;
  int nkeys;
  size_t c0, c1;
  if (kFirst) {
    c0 = (size_t)blockIdx.x * kP3Chunk;
    c1 = min(c0 + (size_t)kP3Chunk, n);
    nkeys = p.p3_n1;
  } else {
    const int n1 = p.p3_n1;
    const uint32_t b = blockIdx.x;
    if (b >= blk2[n1]) return;
    int lo = 0, hi = n1;  // blk2[lo] <= b < blk2[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (blk2[mid] <= b) lo = mid; else hi = mid;
    }
    c0 = (size_t)start1[lo] + (size_t)(b - blk2[lo]) * kP3Chunk;
    c1 = min(c0 + (size_t)kP3Chunk, (size_t)start1[lo + 1]);
    nkeys = p.p3_n2;
    cursor += (size_t)lo * p.p3_n2;
  }
  if (tid < kP3MaxKeys) s_cnt[tid] = 0;
  __syncthreads();
  double px[kP3PerThread], py[kP3PerThread], pz[kP3PerThread];
  uint32_t slot[kP3PerThread];  // key << 12 | rank in the chunk's run of that key
#pragma unroll
  for (int k = 0; k < kP3PerThread; ++k) {
    const size_t idx = c0 + tid + (size_t)k * kP3Threads;
    slot[k] = 0xFFFFFFFFu;
    if (idx < c1) {
      double x = src[3 * idx + 0], y = src[3 * idx + 1];
      double z;
      if (kFirst) {
        x -= p.sub_x;
        y -= p.sub_y;
        z = values ? (double)values[idx] : src[3 * idx + 2];
      } else {
        z = src[3 * idx + 2];
      }
      px[k] = x;
      py[k] = y;
      pz[k] = z;
      int k1, k2;
      if (p3_keys(p, x, y, &k1, &k2)) {
        const int key = kFirst ? k1 : k2;
        slot[k] = ((uint32_t)key << 12) | atomicAdd(&s_cnt[key], 1u);
      }
    }
  }
  __syncthreads();
  {
    const unsigned c = (tid < nkeys) ? s_cnt[tid] : 0u;
    unsigned total;
    const unsigned ex = block_excl_scan<kP3Threads>(c, &total, s_scan);
    if (tid < nkeys) {
      s_off[tid] = ex;
      s_base[tid] = c ? atomicAdd(&cursor[tid], c) : 0u;
    }
    if (tid == 0) s_scan[23] = total;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kP3PerThread; ++k) {
    if (slot[k] != 0xFFFFFFFFu) {
      const uint32_t key = slot[k] >> 12, rank = slot[k] & 0xFFFu;
      const uint32_t q = s_off[key] + rank;
      s_pts[3 * q + 0] = px[k];
      s_pts[3 * q + 1] = py[k];
      s_pts[3 * q + 2] = pz[k];
      s_dest[q] = s_base[key] + rank;
    }
  }
  __syncthreads();
  const uint32_t ne = 3u * s_scan[23];
  for (uint32_t e = tid; e < ne; e += kP3Threads) {
    const uint32_t q = e / 3u;
    dst[3 * (size_t)s_dest[q] + (e - 3u * q)] = s_pts[e];
  }
}

// Pass 3: one workgroup per (k1, k2) sub-partition.
__global__ void __launch_bounds__(kP3PlaceThreads)
k_dsm_p3_place(const double* __restrict__ src, DsmParams p, int cap,
               const uint32_t* __restrict__ start2, uint32_t* __restrict__ bin_start,
               double* __restrict__ sorted) {
  extern __shared__ double s_pts[];                                  // 3 * cap
  uint32_t* s_bins = reinterpret_cast<uint32_t*>(s_pts + 3 * cap);   // p3_w
  uint32_t* s_scan = s_bins + p.p3_w;                                // 24
  const int tid = threadIdx.x;
  const int sp = blockIdx.x;
  const int k1 = sp / p.p3_n2, k2 = sp - k1 * p.p3_n2;
  const int rr = k2 / p.p3_c;
  const int row = k1 * p.p3_r1 + rr;
  const int bx0 = (k2 - rr * p.p3_c) * p.p3_w;
  const int nbw = min(p.p3_w, p.nbx - bx0);
  if (sp == 0 && tid == 0)
    bin_start[(size_t)p.nbx * p.nby] = start2[p.p3_n1 * p.p3_n2];
  if (row >= p.nby || nbw <= 0) return;  // no bins (and therefore no points)
  const uint32_t g0 = start2[sp], g1 = start2[sp + 1];
  for (int k = tid; k < nbw; k += kP3PlaceThreads) s_bins[k] = 0;
  __syncthreads();
  const bool in_lds = (int)(g1 - g0) <= cap && cap <= kP3PlaceMaxCap;
  // the sub-partition is read ONCE: a thread keeps its points (<= 8) in
  // registers between the count and the placement
  double px[kP3PlacePer], py[kP3PlacePer], pz[kP3PlacePer];
  int pb[kP3PlacePer];
  if (in_lds) {
#pragma unroll
    for (int k = 0; k < kP3PlacePer; ++k) {
      const uint32_t idx = g0 + tid + (uint32_t)k * kP3PlaceThreads;
      pb[k] = -1;
      if (idx < g1) {
        px[k] = src[3 * (size_t)idx + 0];
        py[k] = src[3 * (size_t)idx + 1];
        pz[k] = src[3 * (size_t)idx + 2];
        int bx, by;
        point_bin_xy(p, px[k], py[k], &bx, &by);
        pb[k] = bx - bx0;
        atomicAdd(&s_bins[pb[k]], 1u);
      }
    }
  } else {
    for (uint32_t idx = g0 + tid; idx < g1; idx += kP3PlaceThreads) {
      int bx, by;
      point_bin_xy(p, src[3 * (size_t)idx + 0], src[3 * (size_t)idx + 1], &bx, &by);
      atomicAdd(&s_bins[bx - bx0], 1u);
    }
  }
  __syncthreads();
  {
    const int per = (nbw + kP3PlaceThreads - 1) / kP3PlaceThreads;
    const int lo = tid * per;
    const int hi = min(lo + per, nbw);
    unsigned sum = 0;
    for (int k = lo; k < hi; ++k) sum += s_bins[k];
    unsigned total;
    unsigned run = block_excl_scan<kP3PlaceThreads>(sum, &total, s_scan);
    for (int k = lo; k < hi; ++k) {
      const unsigned t = s_bins[k];
      s_bins[k] = run;
      run += t;
    }
  }
  __syncthreads();
  uint32_t* out_start = bin_start + (size_t)row * p.nbx + bx0;
  for (int k = tid; k < nbw; k += kP3PlaceThreads) out_start[k] = g0 + s_bins[k];
  __syncthreads();
  if (!in_lds) {
    // over-full sub-partition (clustered cloud): second read, direct placement
    for (uint32_t idx = g0 + tid; idx < g1; idx += kP3PlaceThreads) {
      const double x = src[3 * (size_t)idx + 0];
      const double y = src[3 * (size_t)idx + 1];
      const double z = src[3 * (size_t)idx + 2];
      int bx, by;
      point_bin_xy(p, x, y, &bx, &by);
      const size_t o = (size_t)g0 + atomicAdd(&s_bins[bx - bx0], 1u);
      sorted[3 * o + 0] = x;
      sorted[3 * o + 1] = y;
      sorted[3 * o + 2] = z;
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < kP3PlacePer; ++k) {
    if (pb[k] >= 0) {
      const uint32_t q = atomicAdd(&s_bins[pb[k]], 1u);
      s_pts[3 * q + 0] = px[k];
      s_pts[3 * q + 1] = py[k];
      s_pts[3 * q + 2] = pz[k];
    }
  }
  __syncthreads();
  const uint32_t ne = 3u * (g1 - g0);
  double* out = sorted + 3 * (size_t)g0;
  for (uint32_t e = tid; e < ne; e += kP3PlaceThreads) out[e] = s_pts[e];
}

// ---------------------------------------------------------------------------
// multi-GPU: compact the points other windows need (their halo)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_halo_select(const double* __restrict__ xyz, size_t n, HaloParams hp,
              double* __restrict__ out, unsigned long long* __restrict__ counts) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += stride) {
    const double x = xyz[3 * idx + 0];
    const double y = xyz[3 * idx + 1];
    // continuous cell coordinates in the full map (same frame as point_bin)
    const double cx = (hp.base_x - (x - hp.sub_x)) * hp.inv_res;
    const double cy = (hp.base_y - (y - hp.sub_y)) * hp.inv_res;
#pragma unroll
    for (int d = 0; d < kMaxHaloDests; ++d) {
      if (d < hp.nd && cx >= hp.lo_i[d] && cx <= hp.hi_i[d] && cy >= hp.lo_j[d] &&
          cy <= hp.hi_j[d]) {
        const unsigned long long slot = atomicAdd(&counts[d], 1ull);
        if (slot < hp.cap) {
          double* o = out + ((size_t)d * hp.cap + slot) * 3;
          o[0] = x;
          o[1] = y;
          o[2] = xyz[3 * idx + 2];
        }
      }
    }
  }
}

int halo_select_run(Ctx* c, const double* dev_xyz, size_t n, const HaloParams& hp,
                    double* dev_out, unsigned long long* dev_counts) {
  ScopedTimer t(c, AMHIP_K_HALO_SELECT);
  AMHIP_TRY(hipMemsetAsync(dev_counts, 0, sizeof(unsigned long long) * hp.nd, c->stream));
  if (n == 0) return AMHIP_OK;
  size_t grid = (n + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(k_halo_select, dim3((unsigned)grid), dim3(256), 0, c->stream,
                     dev_xyz, n, hp, dev_out, dev_counts);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

// ---------------------------------------------------------------------------
// exclusive scan of a u32 array, in place (3 launches)
// ---------------------------------------------------------------------------
constexpr int kScanT = 256;
constexpr int kScanI = 16;
constexpr int kScanE = kScanT * kScanI;

__global__ void __launch_bounds__(kScanT)
k_scan_partials(const uint32_t* __restrict__ in, size_t n,
                uint32_t* __restrict__ partials) {
  __shared__ unsigned lds[kScanT / 64 + 1];
  const size_t base = (size_t)blockIdx.x * kScanE + (size_t)threadIdx.x * kScanI;
  unsigned s = 0;
  if (base + kScanI <= n) {
    const uint4* v = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int k = 0; k < kScanI / 4; ++k) {
      const uint4 q = v[k];
      s += q.x + q.y + q.z + q.w;
    }
  } else {
    for (int k = 0; k < kScanI; ++k)
      if (base + k < n) s += in[base + k];
  }
  unsigned total;
  (void)block_excl_scan<kScanT>(s, &total, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// One block; exclusive scan of partials[0..nb) in place, grand total to *total.
__global__ void __launch_bounds__(1024)
k_scan_top(uint32_t* __restrict__ partials, size_t nb,
           uint32_t* __restrict__ total_out) {
  __shared__ unsigned lds[1024 / 64 + 1];
  unsigned carry = 0;
  for (size_t base = 0; base < nb; base += 1024) {
    const size_t i = base + threadIdx.x;
    const unsigned v = (i < nb) ? partials[i] : 0u;
    unsigned total;
    const unsigned ex = block_excl_scan<1024>(v, &total, lds);
    if (i < nb) partials[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(kScanT)
k_scan_final(uint32_t* __restrict__ data, size_t n,
             const uint32_t* __restrict__ partials) {
  __shared__ unsigned lds[kScanT / 64 + 1];
  const size_t base = (size_t)blockIdx.x * kScanE + (size_t)threadIdx.x * kScanI;
  unsigned v[kScanI];
  const bool full = base + kScanI <= n;
  if (full) {
    const uint4* src = reinterpret_cast<const uint4*>(data + base);
#pragma unroll
    for (int k = 0; k < kScanI / 4; ++k) {
      const uint4 q = src[k];
      v[4 * k + 0] = q.x;
      v[4 * k + 1] = q.y;
      v[4 * k + 2] = q.z;
      v[4 * k + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kScanI; ++k) v[k] = (base + k < n) ? data[base + k] : 0u;
  }
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < kScanI; ++k) s += v[k];
  unsigned total;
  unsigned run = block_excl_scan<kScanT>(s, &total, lds) + partials[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanI; ++k) {
    const unsigned t = v[k];
    v[k] = run;
    run += t;
  }
  if (full) {
    uint4* dst = reinterpret_cast<uint4*>(data + base);
#pragma unroll
    for (int k = 0; k < kScanI / 4; ++k)
      dst[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < kScanI; ++k)
      if (base + k < n) data[base + k] = v[k];
  }
}

// ---------------------------------------------------------------------------
// gather
// ---------------------------------------------------------------------------
struct Accum {
  double num, den;
  unsigned cnt;
  bool exact;
  double exact_z;  // value of (one of) the point(s) with d2 == 0
};

// One IDW term.  1/d2 through v_rcp_f64 + two Newton steps (relative error
// ~1e-16; the accumulation order already differs from the kd-tree's, the
// result is rounded to float afterwards and the parity bar is 1e-4 m).
__device__ __forceinline__ void idw_add(double d2, double z, double* num,
                                        double* den) {
  double r = __builtin_amdgcn_rcp(d2);
  double e = fma(-d2, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d2, r, 1.0);
  r = fma(r, e, r);
  *num = fma(z, r, *num);
  *den += r;
}

// Visit every binned point that can lie within the window of half-width w
// cells around cell (i, j).  MODE 0: accumulate IDW over d2 < T.
// MODE 1: track the minimum d2.
template <int MODE>
__device__ __forceinline__ void scan_window(const DsmParams& p,
                                            const uint32_t* __restrict__ start,
                                            const double* __restrict__ sorted,
                                            double qx, double qy, int i, int j,
                                            int w, double T, Accum* acc,
                                            double* dmin) {
  const int bx0 = (i - w + p.M) / p.B;
  const int bx1 = (i + w + p.M) / p.B;
  const int by0 = (j - w + p.M) / p.B;
  const int by1 = (j + w + p.M) / p.B;
  for (int by = by0; by <= by1; ++by) {
    const uint32_t* row = start + (size_t)by * p.nbx;
    const uint32_t s = row[bx0];
    const uint32_t e = row[bx1 + 1];
    for (uint32_t k = s; k < e; ++k) {
      const double px = sorted[3 * (size_t)k + 0];
      const double py = sorted[3 * (size_t)k + 1];
      // L2_Adaptor with size == 2 (nanoflann.hpp:319-322): 0 + dx*dx, + dy*dy
      const double dx = qx - px;
      const double dy = qy - py;
      double d2 = dx * dx;
      d2 = d2 + dy * dy;
      if (MODE == 0) {
        if (d2 < T) {  // RadiusResultSet::addPoint, strict (nanoflann.hpp:157)
          if (d2 > 0.0) {
            idw_add(d2, sorted[3 * (size_t)k + 2], &acc->num, &acc->den);
          } else {
            acc->exact = true;  // dsm.cc:165 CHECK / ortho-from-pcl.cc:91-96
            acc->exact_z = sorted[3 * (size_t)k + 2];
          }
          acc->cnt++;
        }
      } else {
        *dmin = fmin(*dmin, d2);
      }
    }
  }
}

// Where a cell's result goes.
struct CellOut {
  float* __restrict__ layer;          // elevation (DSM) or ortho (OrthoFromPcl)
  unsigned char* __restrict__ mask;   // optional: set where a value was written
  unsigned* __restrict__ unfilled;    // optional: counts cells left without one
  unsigned* __restrict__ dev_err;
  // The layer is logically in its initial state but its memory has not been
  // filled (amhip_layers_reset is lazy): every cell this call leaves without a
  // value gets the initial value written instead of being left untouched.
  int fill_untouched;
  float init_value;
};

__device__ __forceinline__ void leave_untouched(const DsmParams& p, const CellOut& o, int i,
                                                int j) {
  if (o.fill_untouched) o.layer[(size_t)i + (size_t)j * (size_t)p.rows] = o.init_value;
}

__device__ __forceinline__ void emit_value(const DsmParams& p, const CellOut& o, int i, int j,
                                           double v) {
  const size_t at = (size_t)i + (size_t)j * (size_t)p.rows;
  o.layer[at] = (float)v;
  if (o.mask) o.mask[at] = 1;
}

// Turns an accumulated search into the cell's value.  true = the cell is done.
__device__ __forceinline__ bool finish_accum(const DsmParams& p, const CellOut& o, int i, int j,
                                             const Accum& acc) {
  if (acc.exact) {
    if (p.pcl_mode)
      emit_value(p, o, i, j, acc.exact_z);  // ortho-from-pcl.cc:91-96 perfect match
    else
      atomicOr(o.dev_err, kDevErrExactHit);  // dsm.cc:165 CHECK(distances[i] > 0.0)
    return true;
  }
  if (acc.cnt > 0) {
    emit_value(p, o, i, j, acc.num / acc.den);
    return true;
  }
  return false;
}

// Expanding-radius fallback for a cell whose first search (T[0]) was empty
// (dsm.cc:133-144): the reference retries with T[1], T[2], ... until a search
// returns something.  Equivalent: find the nearest point within the LAST
// radius, pick the first level whose threshold exceeds its d2, gather with
// that threshold.  Works on the global bin structure.
__device__ __forceinline__ bool cell_fallback_global(const DsmParams& p,
                                                     const uint32_t* __restrict__ start,
                                                     const double* __restrict__ sorted, int i,
                                                     int j, double qx, double qy,
                                                     const CellOut& o) {
  if (p.nlevels <= 1) return false;
  Accum acc = {0.0, 0.0, 0u, false, 0.0};
  const int last = p.nlevels - 1;
  double dmin = __builtin_huge_val();
  scan_window<1>(p, start, sorted, qx, qy, i, j, p.w[last], 0.0, &acc, &dmin);
  int level = -1;
  for (int k = 1; k <= last; ++k) {
    if (dmin < p.T[k]) {
      level = k;
      break;
    }
  }
  if (level < 0) return false;  // nothing within the last radius: cell untouched
  scan_window<0>(p, start, sorted, qx, qy, i, j, p.w[level], p.T[level], &acc, &dmin);
  return finish_accum(p, o, i, j, acc);
}

// Whole cell through the global bins (first level + fallback).
__device__ __forceinline__ void cell_global(const DsmParams& p,
                                            const uint32_t* __restrict__ start,
                                            const double* __restrict__ sorted, int i, int j,
                                            const CellOut& o) {
  if (p.only_unfilled && o.mask[(size_t)i + (size_t)j * (size_t)p.rows]) return;
  // grid_map_core getPosition (oracle/amo_compat.h cell_position)
  const double qx = p.base_x + p.res * (-(double)(i + p.i_off));
  const double qy = p.base_y + p.res * (-(double)(j + p.j_off));
  Accum acc = {0.0, 0.0, 0u, false, 0.0};
  double dmin = 0.0;
  scan_window<0>(p, start, sorted, qx, qy, i, j, p.w[0], p.T[0], &acc, &dmin);
  bool done = finish_accum(p, o, i, j, acc);
  if (!done) done = cell_fallback_global(p, start, sorted, i, j, qx, qy, o);
  if (!done) {
    leave_untouched(p, o, i, j);
    if (o.unfilled) atomicAdd(o.unfilled, 1u);
  }
}

// Pure global-memory gather: used when the first-level window is too wide for
// the LDS image (very fine grids) and by the adaptive OrthoFromPcl passes.
__global__ void __launch_bounds__(256)
k_dsm_gather(DsmParams p, const uint32_t* __restrict__ start,
             const double* __restrict__ sorted, CellOut o) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= p.rows || j >= p.cols) return;
  cell_global(p, start, sorted, i, j, o);
}

// ---------------------------------------------------------------------------
// LDS-tiled gather
// ---------------------------------------------------------------------------
// One workgroup owns a tile of 64 x 32 cells.  It copies every binned point
// that can lie within the first search radius of any of its cells (the tile's
// bins plus one ring) from HBM into LDS exactly once, re-bins them there at
// CELL granularity (LDS atomics + scan), and then every lane walks, for each
// of its 8 cells, the 2*w0+1 rows of the disc-shaped window: one contiguous
// span of LDS points per row.  Cells whose first search is empty are queued
// in LDS and finished by the fallback path on the global bins.
constexpr int kTileI = 64;
constexpr int kMaxRegionRows = 96;  // bin rows of a region

// |v| outside [2^-960, 2^960] (within ~1e19 of the ends of the double range)
__device__ __forceinline__ bool exponent_extreme(double v) {
  const unsigned e = ((unsigned)__double2hiint(v) >> 20) & 0x7FFu;
  return (e - 63u) > 1920u;
}

// |v| outside [2^-332, 2^332] (about 1e-100 .. 1e100), zero, denormal, inf or NaN
__device__ __forceinline__ bool exponent_far_from_one(double v) {
  const unsigned e = ((unsigned)__double2hiint(v) >> 20) & 0x7FFu;
  return (e - 691u) > 664u;
}

template <int NT, int kTileJ, int kCap>
__global__ void __launch_bounds__(NT)
k_dsm_gather_tiled(DsmParams p, const uint32_t* __restrict__ start,
                   const double* __restrict__ sorted, CellOut o) {
  constexpr int kWaves = NT / 64;
  constexpr int kCellsPerLane = kTileJ / kWaves;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [xy: cap+1 double2][z: cap+1 double (+pad)][cell offsets][rows][ctl][flags]
  double2* s_xy = reinterpret_cast<double2*>(smem);
  double* s_z = reinterpret_cast<double*>(smem + (size_t)(p.lds_cap + 1) * 16);
  uint32_t* s_off = reinterpret_cast<uint32_t*>(smem + (size_t)(p.lds_cap + 2) * 24);
  uint32_t* s_rowg = s_off + p.lds_cells + 1;        // global start of a region bin-row
  uint32_t* s_rowp = s_rowg + kMaxRegionRows;        // prefix of the row lengths (+1)
  uint32_t* s_scan = s_rowp + kMaxRegionRows + 1;    // block-scan scratch
  uint32_t* s_ctl = s_scan + 24;                     // [0] np, [1] nflag, [2] np_ext
  uint16_t* s_flag = reinterpret_cast<uint16_t*>(s_ctl + 4);  // kTileI*kTileJ entries

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;

  // XCD-aware tile order: consecutive tiles (which share halo points) go to
  // the same XCD's L2.  Blocks are dealt round-robin to the 8 XCDs, so give
  // XCD x the x-th contiguous chunk of the tile list (bijective for any count).
  const int ntiles = p.tiles_i * p.tiles_j;
  int tile;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, k = b >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ti = tile % p.tiles_i;
  const int tj = tile / p.tiles_i;
  const int i0 = ti * kTileI;
  const int j0 = tj * kTileJ;
  const int i_hi = min(i0 + kTileI, p.rows) - 1;
  const int j_hi = min(j0 + kTileJ, p.cols) - 1;
  const int w0 = p.w[0];

  // region of bins holding every first-level candidate of the tile
  const int rbx0 = (i0 - w0 + p.M) / p.B;
  const int rbx1 = (i_hi + w0 + p.M) / p.B;
  const int rby0 = (j0 - w0 + p.M) / p.B;
  const int rby1 = (j_hi + w0 + p.M) / p.B;
  const int nrb = rby1 - rby0 + 1;
  const int RW = (rbx1 - rbx0 + 1) * p.B;
  const int RH = nrb * p.B;
  // Cell-offset table layout: window rows are walked two at a time, so the
  // cells of a ROW PAIR are interleaved (x-major, then the row of the pair):
  // the candidates of one trip -- both rows, columns ci-w .. ci+w -- are then
  // ONE contiguous span of the LDS point array.  `sh` aligns the pairs with
  // the tile's first window row (all cell pairs of a tile start on even rows).
  const int sh = (j0 - w0 + p.M - (rby0 * p.B)) & 1;
  const int RW2 = 2 * RW;
  const int ncell = (RH / 2 + 2) * RW2;
  const int ox = rbx0 * p.B;  // region origin in M-shifted cell coordinates
  const int oy = rby0 * p.B;

  const bool geom_ok = nrb <= kMaxRegionRows && ncell <= p.lds_cells;
  if (geom_ok && tid < nrb) {
    const uint32_t* row = start + (size_t)(rby0 + tid) * p.nbx;
    const uint32_t gs = row[rbx0];
    s_rowg[tid] = gs;
    s_rowp[tid + 1] = row[rbx1 + 1] - gs;
  }
  if (tid == NT - 1) {
    // anything at all within the LAST fallback radius of the tile?
    const int wl = p.w[p.nlevels - 1];
    const int ex0 = (i0 - wl + p.M) / p.B, ex1 = (i_hi + wl + p.M) / p.B;
    const int ey0 = (j0 - wl + p.M) / p.B, ey1 = (j_hi + wl + p.M) / p.B;
    uint32_t tot = 0;
    for (int by = ey0; by <= ey1; ++by) {
      const uint32_t* row = start + (size_t)by * p.nbx;
      tot += row[ex1 + 1] - row[ex0];
    }
    s_ctl[2] = tot;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    s_rowp[0] = 0;
    if (geom_ok) {
      for (int r = 0; r < nrb; ++r) {
        run += s_rowp[r + 1];
        s_rowp[r + 1] = run;
      }
    }
    s_ctl[0] = run;
    s_ctl[1] = 0;
  }
  __syncthreads();
  const int np = (int)s_ctl[0];
  if (s_ctl[2] == 0) {  // empty neighbourhood: every cell stays untouched
    if (o.unfilled && tid == 0)
      atomicAdd(o.unfilled, (unsigned)((i_hi - i0 + 1) * (j_hi - j0 + 1)));
    if (o.fill_untouched) {
      for (int c = 0; c < kCellsPerLane; ++c) {
        const int i = i0 + lane, j = j0 + wid * kCellsPerLane + c;
        if (i <= i_hi && j <= j_hi) leave_untouched(p, o, i, j);
      }
    }
    return;
  }

  const bool use_lds = geom_ok && np <= p.lds_cap;
  if (!use_lds) {
    // over-full tile (very dense / clustered cloud): global path for all cells
    for (int c = 0; c < kCellsPerLane; ++c) {
      const int i = i0 + lane, j = j0 + wid * kCellsPerLane + c;
      if (i <= i_hi && j <= j_hi) cell_global(p, start, sorted, i, j, o);
    }
    return;
  }

  // ---- stage + cell-bin the region's points in LDS --------------------------
  // pass 1: count per cell (LDS atomics), remember (cell, rank) per point
  for (int k = tid; k <= ncell; k += NT) s_off[k] = 0;
  __syncthreads();
  static_assert(kCellsPerLane % 2 == 0, "cell pairs must start on even rows of the tile");
  constexpr int kMaxK = (kCap + NT - 1) / NT;  // p.lds_cap == kCap
  uint32_t pslot[kMaxK];                       // cell << 12 | rank  (rank < 4096)
  double ppx[kMaxK], ppy[kMaxK], ppz[kMaxK];   // the thread's points (placed after the scan)
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    const int idx = tid + k * NT;
    pslot[k] = 0xFFFFFFFFu;
    if (idx < np) {
      int r = 0;
      while (idx >= (int)s_rowp[r + 1]) ++r;
      const size_t g = (size_t)s_rowg[r] + (size_t)(idx - (int)s_rowp[r]);
      const double px = sorted[3 * g + 0];
      const double py = sorted[3 * g + 1];
      ppx[k] = px;
      ppy[k] = py;
      ppz[k] = sorted[3 * g + 2];
      // same arithmetic as point_bin(): the point's cell in shifted coordinates
      const double cx = (p.base_x - px) * p.inv_res - (double)p.i_off;
      const double cy = (p.base_y - py) * p.inv_res - (double)p.j_off;
      int ix = (int)floor(cx + 0.5) + p.M;
      int iy = (int)floor(cy + 0.5) + p.M;
      ix = min(max(ix, 0), p.rows + 2 * p.M - 1) - ox;
      iy = min(max(iy, 0), p.cols + 2 * p.M - 1) - oy;
      ix = min(max(ix, 0), RW - 1);  // (always inside: the bins are aligned)
      iy = min(max(iy, 0), RH - 1);
      iy += sh;
      const uint32_t cell = (uint32_t)((iy >> 1) * RW2 + 2 * ix + (iy & 1));
      pslot[k] = (cell << 12) | atomicAdd(&s_off[cell], 1u);
    }
  }
  __syncthreads();
  {
    // exclusive scan of the ncell counters, in place
    const int per = (ncell + NT - 1) / NT;
    const int lo = tid * per;
    const int hi = min(lo + per, ncell);
    unsigned sum = 0;
    for (int k = lo; k < hi; ++k) sum += s_off[k];
    unsigned total;
    unsigned run = block_excl_scan<NT>(sum, &total, s_scan);
    for (int k = lo; k < hi; ++k) {
      const unsigned t = s_off[k];
      s_off[k] = run;
      run += t;
    }
    if (tid == 0) s_off[ncell] = total;
  }
  __syncthreads();
  // pass 2: drop the points into their sorted slot
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    if (pslot[k] != 0xFFFFFFFFu) {
      const uint32_t pos = s_off[pslot[k] >> 12] + (pslot[k] & 0xFFFu);
      s_xy[pos] = make_double2(ppx[k], ppy[k]);
      s_z[pos] = ppz[k];
    }
  }
  __syncthreads();

  // ---- gather: lane = row index i; the lane's cells are taken two at a time
  // (columns j, j+1): every candidate read from LDS is tested against both,
  // which halves the LDS traffic per test (the LDS pipe is shared by the CU's
  // four SIMDs and would otherwise co-limit with the FP64 VALU work).
  const int i = i0 + lane;
  const double T0 = p.T[0];
  if (i <= i_hi) {
    const double qx = p.base_x + p.res * (-(double)(i + p.i_off));
    const int ci = i + p.M - ox;  // this cell's column in the region
    for (int c = 0; c < kCellsPerLane; c += 2) {
      const int jA = j0 + wid * kCellsPerLane + c;
      if (jA > j_hi) break;
      const bool haveB = (jA + 1 <= j_hi) && (c + 1 < kCellsPerLane);
      const double qyA = p.base_y + p.res * (-(double)(jA + p.j_off));
      const double qyB = p.base_y + p.res * (-(double)(jA + 1 + p.j_off));
      const double TB = haveB ? T0 : -1.0;  // d2 < -1 never holds
      const int cj = jA + p.M - oy;
      // Division-free IDW: h = (sum z_i/d_i) / (sum 1/d_i) is kept as N/D with
      //   N = sum_i z_i * prod_{j!=i} d_j,  D = sum_i prod_{j!=i} d_j,  P = prod_j d_j
      // so a hit costs  N = N*d + z*P;  D = D*d + P;  P = P*d  (4 FP64 ops; the
      // alternative v_rcp_f64 is a quarter-rate instruction: 17.7 vs 5.3 cycles,
      // tools/ubench).  D > 0 <=> at least one hit;  P == 0 <=> some hit had
      // d2 == 0 (dsm.cc:165 CHECK(distances[i] > 0.0)).  One division per cell.
      double NA = 0.0, DA = 0.0, PA = 1.0, NB = 0.0, DB = 0.0, PB = 1.0;
      bool suspectA = false, suspectB = false;
      // rows jA-w0 .. jA+1+w0 (the last one only matters for cell B), one row
      // pair = one contiguous span per trip (lanes wait for each other per
      // trip, and the spread of a two-row candidate count is relatively smaller).
      const uint32_t* orow = s_off + ((cj - w0 + sh) >> 1) * RW2 + 2 * ci;
      for (int r = 0; r <= w0; ++r) {
        const int w = p.wrp[r];
        const uint32_t kb = orow[-2 * w];
        const uint32_t ke = orow[2 * w + 2];
        orow += RW2;
        for (uint32_t k = kb; k < ke; ++k) {
          const double2 xy = s_xy[k];
          const double z = s_z[k];
          // L2_Adaptor, size == 2 (nanoflann.hpp:319-322): 0 + dx*dx, + dy*dy
          const double dx = qx - xy.x;
          const double dx2 = dx * dx;
          const double dyA = qyA - xy.y;
          const double dyB = qyB - xy.y;
          const double d2A = dx2 + dyA * dyA;
          const double d2B = dx2 + dyB * dyB;
          // N = N*d + z*P;  D = D*d + P;  P = P*d  as four in-place FP64
          // instructions under the EXEC mask of the hit test.  (Left to the
          // compiler this becomes either 6 v_cndmask selects per test or
          // compute-into-temporaries + 3 masked 64-bit moves.)
          if (d2A < T0) {  // strict (nanoflann.hpp:157)
            double t;
            asm volatile(
                "v_mul_f64 %3, %2, %4\n\t"
                "v_fma_f64 %0, %0, %5, %3\n\t"
                "v_fma_f64 %1, %1, %5, %2\n\t"
                "v_mul_f64 %2, %2, %5"
                : "+v"(NA), "+v"(DA), "+v"(PA), "=&v"(t)
                : "v"(z), "v"(d2A));
          }
          if (d2B < TB) {
            double t;
            asm volatile(
                "v_mul_f64 %3, %2, %4\n\t"
                "v_fma_f64 %0, %0, %5, %3\n\t"
                "v_fma_f64 %1, %1, %5, %2\n\t"
                "v_mul_f64 %2, %2, %5"
                : "+v"(NB), "+v"(DB), "+v"(PB), "=&v"(t)
                : "v"(z), "v"(d2B));
          }
        }
        // keep the running products inside the double range (exact scaling by
        // powers of two; N, D, P share the factor so N/D is unaffected).  The
        // test looks at the exponent field only: outside 2^-332 .. 2^332
        // (three integer instructions per trip instead of a chain of FP64
        // compares; zero falls out of the range too and is left alone inside).
        if (exponent_far_from_one(PA) && PA != 0.0) {
          // within ONE trip the product may even have left the double range
          // (dozens of points within millimetres of the centre): remember it
          if (exponent_extreme(PA)) suspectA = true;
          const double sc = PA < 1.0 ? 0x1p+400 : 0x1p-400;
          NA *= sc;
          DA *= sc;
          PA *= sc;
        }
        if (exponent_far_from_one(PB) && PB != 0.0) {
          if (exponent_extreme(PB)) suspectB = true;
          const double sc = PB < 1.0 ? 0x1p+400 : 0x1p-400;
          NB *= sc;
          DB *= sc;
          PB *= sc;
        }
      }
      // P == 0 <=> a hit with d2 == 0 -- or a product that underflowed inside
      // one trip.  Either way (and whenever the running values came close to
      // the ends of the double range) the cell is re-done by the global
      // routine, which weights with reciprocals and tracks exact hits
      // explicitly: DSM -> CHECK failure (dsm.cc:165), OrthoFromPcl -> that
      // point's value (ortho-from-pcl.cc:91-96).
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && !haveB) break;
        const double Nn = h ? NB : NA, Dd = h ? DB : DA, Pp = h ? PB : PA;
        const bool suspect = h ? suspectB : suspectA;
        const int jj = jA + h;
        int queue = 0;  // 1: no first-level neighbour (ladder), 2: redo the whole cell
        if (Pp == 0.0 || suspect || !(Dd < 0x1p+1000) || !(fabs(Nn) < 0x1p+1000)) {
          queue = 2;
        } else if (Dd > 0.0) {
          emit_value(p, o, i, jj, Nn / Dd);
        } else {
          queue = 1;
        }
        if (queue) {
          const uint32_t slot = atomicAdd(&s_ctl[1], 1u);
          s_flag[slot] = (uint16_t)(((wid * kCellsPerLane + c + h) * kTileI + lane) |
                                    (queue == 2 ? 0x8000 : 0));
        }
      }
    }
  }
  __syncthreads();

  // ---- queued cells (dense over the workgroup): the fallback ladder, or for
  // OrthoFromPcl the full global routine (several coincident exact hits) ------
  const int nflag = (int)s_ctl[1];
  for (int f = tid; f < nflag; f += NT) {
    const int code = s_flag[f] & 0x7FFF;
    const bool redo = (s_flag[f] & 0x8000) != 0;
    const int fi = i0 + (code % kTileI);
    const int fj = j0 + (code / kTileI);
    if (p.pcl_mode || redo) {
      cell_global(p, start, sorted, fi, fj, o);
    } else {
      const double fqx = p.base_x + p.res * (-(double)(fi + p.i_off));
      const double fqy = p.base_y + p.res * (-(double)(fj + p.j_off));
      const bool done = cell_fallback_global(p, start, sorted, fi, fj, fqx, fqy, o);
      if (!done) {
        leave_untouched(p, o, fi, fj);
        if (o.unfilled) atomicAdd(o.unfilled, 1u);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
int dsm_run(Ctx* c, const double* dev_xyz, const int32_t* dev_values, size_t n,
            const DsmParams& p, float* out, unsigned char* mask, unsigned* unfilled,
            bool fill_untouched, float init_value) {
  const CellOut cell_out = {out, mask, unfilled, c->dev_err, fill_untouched ? 1 : 0, init_value};
  const size_t nbins = (size_t)p.nbx * (size_t)p.nby;
  const size_t nblocks_scan = (nbins + kScanE - 1) / kScanE;
  {
    int rc;
    if ((rc = ensure_capacity(&c->sorted, &c->sorted_cap, 3 * n))) return rc;
    if ((rc = ensure_capacity(&c->bin_start, &c->bin_cap, nbins + 4))) return rc;
  }
  c->last_num_bins = (int64_t)nbins;
  c->last_bin_cells = p.B;

  static const bool force_one_level = getenv("AMHIP_SORT_ONE_LEVEL") != nullptr;
  static const bool force_two_level = getenv("AMHIP_SORT_TWO_LEVEL") != nullptr;
  if (p.p3_n1 > 0 && !force_one_level && !force_two_level) {
    // ---- three-pass partition sort ---------------------------------------------
    const int n1 = p.p3_n1, n2 = p.p3_n2, nk = n1 * n2;
    size_t gcount = (n + 8191) / 8192;
    if (gcount > 256) gcount = 256;
    if (gcount < 1) gcount = 1;
    int rc;
    if ((rc = ensure_capacity(&c->tmp_points, &c->tmp_points_cap, 3 * n))) return rc;
    const size_t ws_words = gcount * (size_t)nk + 3 * (size_t)nk + 3 * (size_t)n1 + 16;
    if ((rc = ensure_capacity(&c->stripe_ws, &c->stripe_ws_cap, ws_words))) return rc;
    uint32_t* hist_rows = c->stripe_ws;
    uint32_t* cnt = hist_rows + gcount * (size_t)nk;
    uint32_t* start2 = cnt + nk;       // nk + 1
    uint32_t* cursor2 = start2 + nk + 1;
    uint32_t* start1 = cursor2 + nk;   // n1 + 1
    uint32_t* cursor1 = start1 + n1 + 1;
    uint32_t* blk2 = cursor1 + n1;     // n1 + 1
    {
      ScopedTimer t(c, AMHIP_K_DSM_BIN_COUNT);
      const size_t lds = (size_t)nk * sizeof(uint32_t);
      AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_count),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_dsm_p3_count, dim3((unsigned)gcount), dim3(kP3CountThreads), lds,
                         c->stream, dev_xyz, n, p, hist_rows);
      hipLaunchKernelGGL(k_dsm_p3_reduce, dim3((unsigned)((nk + 63) / 64)), dim3(256), 0,
                         c->stream, hist_rows, (int)gcount, nk, cnt);
      hipLaunchKernelGGL(k_dsm_p3_scan, dim3(1), dim3(1024), 0, c->stream, cnt, n1, n2, start2,
                         cursor2, start1, cursor1, blk2);
      AMHIP_TRY(hipGetLastError());
    }
    {
      ScopedTimer t(c, AMHIP_K_DSM_SCATTER);
      const size_t lds = (size_t)kP3Chunk * 28 + (3 * kP3MaxKeys + 32) * sizeof(uint32_t);
      AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_scatter<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_scatter<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const size_t g1 = (n + kP3Chunk - 1) / kP3Chunk;
      hipLaunchKernelGGL(k_dsm_p3_scatter<true>, dim3((unsigned)g1), dim3(kP3Threads), lds,
                         c->stream, dev_xyz, dev_values, n, p, start1, blk2, cursor1, c->sorted);
      hipLaunchKernelGGL(k_dsm_p3_scatter<false>, dim3((unsigned)(g1 + n1)), dim3(kP3Threads),
                         lds, c->stream, c->sorted, (const int32_t*)nullptr, n, p, start1, blk2,
                         cursor2, c->tmp_points);
      AMHIP_TRY(hipGetLastError());
    }
    {
      ScopedTimer t(c, AMHIP_K_DSM_SCAN);
      const size_t lds = (size_t)p.p3_cap * 24 + ((size_t)p.p3_w + 32) * sizeof(uint32_t);
      AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_place),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_dsm_p3_place, dim3((unsigned)nk), dim3(kP3PlaceThreads), lds,
                         c->stream, c->tmp_points, p, p.p3_cap, start2, c->bin_start, c->sorted);
      AMHIP_TRY(hipGetLastError());
    }
  } else if (p.nstripes > 0 && !force_one_level) {
    // ---- two-level stripe sort ------------------------------------------------
    int rc;
    if ((rc = ensure_capacity(&c->tmp_points, &c->tmp_points_cap, 3 * n))) return rc;
    if ((rc = ensure_capacity(&c->stripe_ws, &c->stripe_ws_cap, 3 * (size_t)p.nstripes + 8)))
      return rc;
    uint32_t* stripe_cnt = c->stripe_ws;
    uint32_t* stripe_start = c->stripe_ws + p.nstripes;          // nstripes + 1
    uint32_t* stripe_cursor = c->stripe_ws + 2 * p.nstripes + 1;  // nstripes
    {
      ScopedTimer t(c, AMHIP_K_DSM_BIN_COUNT);
      AMHIP_TRY(hipMemsetAsync(stripe_cnt, 0, p.nstripes * sizeof(uint32_t), c->stream));
      size_t grid = (n + kL1Threads - 1) / kL1Threads;
      if (grid > 256 * 8) grid = 256 * 8;
      hipLaunchKernelGGL(k_dsm_stripe_count, dim3((unsigned)grid), dim3(kL1Threads),
                         p.nstripes * sizeof(uint32_t), c->stream, dev_xyz, n, p, stripe_cnt);
      hipLaunchKernelGGL(k_dsm_stripe_scan, dim3(1), dim3(1024), 0, c->stream, stripe_cnt,
                         p.nstripes, stripe_start, stripe_cursor);
      AMHIP_TRY(hipGetLastError());
    }
    {
      ScopedTimer t(c, AMHIP_K_DSM_SCATTER);
      const size_t grid = (n + kL1Chunk - 1) / kL1Chunk;
      hipLaunchKernelGGL(k_dsm_stripe_scatter, dim3((unsigned)grid), dim3(kL1Threads),
                         2 * p.nstripes * sizeof(uint32_t), c->stream, dev_xyz, dev_values, n,
                         p, stripe_cursor, c->tmp_points);
      AMHIP_TRY(hipGetLastError());
    }
    {
      ScopedTimer t(c, AMHIP_K_DSM_SCAN);
      const size_t lds = ((size_t)p.stripe_rows * p.nbx + 32) * sizeof(uint32_t);
      hipLaunchKernelGGL(k_dsm_stripe_sort, dim3((unsigned)p.nstripes), dim3(kL2Threads), lds,
                         c->stream, c->tmp_points, p, stripe_start, c->bin_start, c->sorted);
      AMHIP_TRY(hipGetLastError());
    }
  } else {
    // ---- one-level counting sort (fallback: very wide grids, or forced) --------
    {
      int rc;
      if ((rc = ensure_capacity(&c->rank, &c->rank_cap, n))) return rc;
      if ((rc = ensure_capacity(&c->scan_partials, &c->partial_cap, nblocks_scan + 4))) return rc;
    }
    {
      ScopedTimer t(c, AMHIP_K_MISC);
      AMHIP_TRY(hipMemsetAsync(c->bin_start, 0, (nbins + 1) * sizeof(uint32_t), c->stream));
    }
    const int block = 256;
    size_t grid_pts = (n + block - 1) / block;
    if (grid_pts > 256 * 16) grid_pts = 256 * 16;
    {
      ScopedTimer t(c, AMHIP_K_DSM_BIN_COUNT);
      hipLaunchKernelGGL(k_dsm_bin_count, dim3((unsigned)grid_pts), dim3(block), 0, c->stream,
                         dev_xyz, n, p, c->bin_start, c->rank);
      AMHIP_TRY(hipGetLastError());
    }
    {
      ScopedTimer t(c, AMHIP_K_DSM_SCAN);
      hipLaunchKernelGGL(k_scan_partials, dim3((unsigned)nblocks_scan), dim3(kScanT), 0,
                         c->stream, c->bin_start, nbins, c->scan_partials);
      hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, c->scan_partials,
                         nblocks_scan, c->bin_start + nbins);
      hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nblocks_scan), dim3(kScanT), 0, c->stream,
                         c->bin_start, nbins, c->scan_partials);
      AMHIP_TRY(hipGetLastError());
    }
    {
      ScopedTimer t(c, AMHIP_K_DSM_SCATTER);
      hipLaunchKernelGGL(k_dsm_scatter, dim3((unsigned)grid_pts), dim3(block), 0, c->stream,
                         dev_xyz, dev_values, n, p, c->bin_start, c->rank, c->sorted);
      AMHIP_TRY(hipGetLastError());
    }
  }
  {
    ScopedTimer t(c, AMHIP_K_DSM_GATHER);
    if (p.lds_ok) {
      const unsigned ntiles = (unsigned)p.tiles_i * (unsigned)p.tiles_j;
      // AMHIP_GATHER_NT: threads per gather workgroup (tuning knob; 512 measured best)
      static const int nt = getenv("AMHIP_GATHER_NT") ? atoi(getenv("AMHIP_GATHER_NT")) : 512;
#define AMHIP_LAUNCH_TILED(NT_, TJ_, CAP_)                                                    \
  do {                                                                                        \
    AMHIP_TRY(hipFuncSetAttribute(                                                            \
        reinterpret_cast<const void*>(k_dsm_gather_tiled<NT_, TJ_, CAP_>),                    \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes));                       \
    hipLaunchKernelGGL((k_dsm_gather_tiled<NT_, TJ_, CAP_>), dim3(ntiles), dim3(NT_),         \
                       p.lds_bytes, c->stream, p, c->bin_start, c->sorted, cell_out);         \
  } while (0)
      // (tile height, LDS point capacity) picked by make_dsm_params from the
      // cloud's mean density: 64x16 / 1024 points runs 4 workgroups per CU
      if (p.tile_j == 16 && p.lds_cap == 1024) {
        if (nt == 256) AMHIP_LAUNCH_TILED(256, 16, 1024);
        else AMHIP_LAUNCH_TILED(512, 16, 1024);
      } else if (p.tile_j == 16) {
        AMHIP_LAUNCH_TILED(512, 16, 2048);
      } else {
        if (nt == 256) AMHIP_LAUNCH_TILED(256, 32, 2048);
        else if (nt == 1024) AMHIP_LAUNCH_TILED(1024, 32, 2048);
        else AMHIP_LAUNCH_TILED(512, 32, 2048);
      }
#undef AMHIP_LAUNCH_TILED
    } else {
      dim3 grid((unsigned)((p.rows + 63) / 64), (unsigned)((p.cols + 3) / 4));
      hipLaunchKernelGGL(k_dsm_gather, grid, dim3(256), 0, c->stream, p, c->bin_start,
                         c->sorted, cell_out);
    }
    AMHIP_TRY(hipGetLastError());
  }
  return AMHIP_OK;
}

}  // namespace amhip
