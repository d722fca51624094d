// amhip_sort.hip -- binning of the point cloud on MI355X (gfx950).
//
// Replaces Dsm::initializeAndFillKdTree (aerial_mapper_dsm/src/dsm.cc:36-52):
// instead of a kd-tree the points are sorted into a uniform grid of bins of
// B x B cells (B = first search radius in cells), bins row-major, with
// bin_start[] = exclusive offsets; dsm.cc:42-43's centre offsets are applied on
// the way.  Two implementations produce the same order:
//   k_dsm_p3_*      three-pass partition sort (>= 1 M points):
//                   count -> two LDS-staged scatter passes -> in-LDS placement
//   k_dsm_bin_count / k_scan_* / k_dsm_scatter
//                   one-level counting sort with global atomics (smaller clouds: fewer
//                   launches; measured equal or faster there than the two-level stripe
//                   sort that round 1 kept for them -- tools/small_cloud_probe.py)
// plus k_halo_select, the multi-GPU halo compaction (same streaming shape).
#include <algorithm>
#include <cstdlib>

#include "amhip_common.h"
#include "amhip_device.h"

namespace amhip {

// ---------------------------------------------------------------------------
// binning
// ---------------------------------------------------------------------------
constexpr uint32_t kNoRank = 0xFFFFFFFFu;

// n / d through the host's multiplier (DsmParams::mul_*): one v_mul_hi_u32 instead of the ~20
// instructions of a 32-bit division by a run-time divisor; n >= 0.
__device__ __forceinline__ int div_by(int n, int d, unsigned m) {
  if (m == 0u) return n;
  if (m == 0xFFFFFFFFu) return n / d;
  return (int)__umulhi((unsigned)n, m);
}

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
  const int bx = div_by(ix, p.B, p.mul_B);
  const int by = div_by(iy, p.B, p.mul_B);
  *bin = (uint32_t)by * (uint32_t)p.nbx + (uint32_t)bx;
  return true;
}

// zall (may be null): per-wave [min, max] of the binned points' heights (the records' reference height)
__global__ void __launch_bounds__(256)
k_dsm_bin_count(const double* __restrict__ xyz, size_t n, DsmParams p,
                uint32_t* __restrict__ cnt, uint32_t* __restrict__ rank,
                double* __restrict__ zall) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double zlo = __builtin_huge_val(), zhi = -__builtin_huge_val();
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += stride) {
    const double x = xyz[3 * idx + 0];
    const double y = xyz[3 * idx + 1];
    const double z = zall ? xyz[3 * idx + 2] : 0.0;
    const double px = x - p.sub_x;  // dsm.cc:42
    const double py = y - p.sub_y;  // dsm.cc:43
    uint32_t bin;
    uint32_t r = kNoRank;
    if (point_bin(p, px, py, &bin)) {
      r = atomicAdd(&cnt[bin], 1u);
      // (the range of the BINNED points only: a stray height far off the map must not move zref)
      zlo = fmin(zlo, z);
      zhi = fmax(zhi, z);
    }
    rank[idx] = r;
  }
  if (zall) range_commit_wave(zlo, zhi, zall, (size_t)blockIdx.x * 4 + (threadIdx.x >> 6));
}

__device__ __forceinline__ bool make_record(const DsmParams& p, double px, double py, double z,
                                            double zref, uint32_t row, uint32_t* w);

// rec16 / sidx / zref (may be null): the single-precision gather's records next to the doubles
// (small clouds carry both: the exact routines then read the doubles without the detour)
__global__ void __launch_bounds__(256)
k_dsm_scatter(const double* __restrict__ xyz, const int32_t* __restrict__ values, size_t n,
              DsmParams p, const uint32_t* __restrict__ start,
              const uint32_t* __restrict__ rank, double* __restrict__ sorted,
              double* __restrict__ zpart, uint4* __restrict__ rec16, uint32_t* __restrict__ sidx,
              const double* __restrict__ zref) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double zlo = __builtin_huge_val(), zhi = -__builtin_huge_val();
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
    if (rec16) {
      uint32_t w[5];
      make_record(p, px, py, z, zref[0], (uint32_t)idx, w);
      rec16[slot] = make_uint4(w[0], w[1], w[2], w[3]);
      sidx[slot] = w[4];
    }
    zlo = fmin(zlo, z);
    zhi = fmax(zhi, z);
  }
  if (zpart) range_commit_wave(zlo, zhi, zpart, (size_t)blockIdx.x * 4 + (threadIdx.x >> 6));
}

// bin (bx, by) of a (centre-shifted) point: point_bin() without the flattening
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
  *bx = div_by(ix, p.B, p.mul_B);
  *by = div_by(iy, p.B, p.mul_B);
  return true;
}

// ---------------------------------------------------------------------------
// three-pass partition sort (the default for clouds that are worth it)
// ---------------------------------------------------------------------------
// (Round 1's two-level stripe sort appended 24-byte records to ~1000 open runs per
// workgroup straight from registers; the partially written cache lines did not
// survive in the L2 until their neighbours arrived, and the PMC counters showed
// 2-3x the algorithmic write traffic.)  Here every pass sorts its chunk in LDS
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
#ifndef AMHIP_P3_THREADS
#define AMHIP_P3_THREADS 512
#endif
#ifndef AMHIP_P3_PER
#define AMHIP_P3_PER 5
#endif
constexpr int kP3Threads = AMHIP_P3_THREADS;
constexpr int kP3Chunk = AMHIP_P3_THREADS * AMHIP_P3_PER;  // points staged per scatter workgroup (2560: 70 KB of LDS, 2 per CU)
constexpr int kP3PerThread = kP3Chunk / kP3Threads;
constexpr int kP3MaxKeys = 256;
constexpr int kP3PlaceThreads = 256;
constexpr int kP3PlaceMaxCap = 2048;  // LDS capacity (points) the register-resident path handles
constexpr int kP3PlacePer = kP3PlaceMaxCap / kP3PlaceThreads;
constexpr int kP3BigThreads = 1024;
constexpr int kP3BigPer = 6;
constexpr int kP3BigCap = kP3BigThreads * kP3BigPer;  // 6144 points = 147 KB of LDS
constexpr int kRecWords = 5;  // 20-byte sort records of the single-precision mode (below)
// points a thread of the big placement kernel keeps in registers across the rounds of a
// sub-partition beyond one LDS image (place_rounds): 14 x 24 bytes / 16 x 20 bytes
constexpr int kP3RoundsPer = 14;
constexpr int kP3RoundsPerRec = 16;
constexpr size_t kLdsMaxBytes = 160 * 1024 - 512;  // per workgroup on gfx950 (160 KB per CU), a margin kept

__device__ __forceinline__ bool p3_keys(const DsmParams& p, double px, double py, int* k1,
                                        int* k2) {
  int bx, by;
  if (!point_bin_xy(p, px, py, &bx, &by)) return false;
  const int a = div_by(by, p.p3_r1, p.mul_r1);
  *k1 = a;
  *k2 = (by - a * p.p3_r1) * p.p3_c + div_by(bx, p.p3_w, p.mul_w);
  return true;
}

// Halo selection (multi-GPU tiling): the windows a point has to travel to.  Shared by
// k_halo_select and the count pass that selects on the way (k_dsm_p3_count<true>), so both
// pick the same points.  A workgroup stages what it selects in LDS and reserves its rows
// with ONE global atomic per destination at the end: tens of thousands of returning
// atomics on the same counter serialise in L2 (0.45 ms for the 57 K points of a 2.5 km
// edge, more than the pass itself).
constexpr unsigned kHaloStage = 2048;  // staged selections per workgroup; more go direct
struct HaloStage {
  unsigned n;                            // selections so far (those past kHaloStage went direct)
  unsigned cnt[kMaxHaloDests];           // staged per destination
  unsigned long long base[kMaxHaloDests];
  // the destinations' bounds: read here by the few points outside the own window's inside,
  // so that 64 SGPRs of kernel arguments are not kept alive (spilled) across the hot loop
  double lo_i[kMaxHaloDests], hi_i[kMaxHaloDests], lo_j[kMaxHaloDests], hi_j[kMaxHaloDests];
  unsigned long long entry[kHaloStage];  // point index | destination << 32 | local slot << 36
};

__device__ __forceinline__ void halo_stage_init(HaloStage* st, const HaloParams& hp) {
  if (threadIdx.x == 0) {
    st->n = 0;
#pragma unroll
    for (int d = 0; d < kMaxHaloDests; ++d) {
      st->cnt[d] = 0;
      st->lo_i[d] = hp.lo_i[d];
      st->hi_i[d] = hp.hi_i[d];
      st->lo_j[d] = hp.lo_j[d];
      st->hi_j[d] = hp.hi_j[d];
    }
  }
}

__device__ __forceinline__ void halo_write(const HaloParams& hp, const double* __restrict__ xyz,
                                           size_t idx, int d, unsigned long long slot,
                                           double* __restrict__ out) {
  if (slot < hp.cap_d[d]) {
    double* o = out + (size_t)(hp.off[d] + slot) * 3;
    o[0] = xyz[3 * idx + 0];
    o[1] = xyz[3 * idx + 1];
    o[2] = xyz[3 * idx + 2];
  }
}

// true: the point lies outside the inside of the own window, halo_emit() has to look at it
__device__ __forceinline__ bool halo_candidate(const HaloParams& hp, double x, double y) {
  // continuous cell coordinates in the full map (same frame as point_bin)
  const double cx = (hp.base_x - (x - hp.sub_x)) * hp.inv_res;
  const double cy = (hp.base_y - (y - hp.sub_y)) * hp.inv_res;
  return !(cx > hp.in_lo_i && cx < hp.in_hi_i && cy > hp.in_lo_j && cy < hp.in_hi_j);
}

// (the rare path: kept compact -- a rolled loop over the destinations, bounds from LDS)
__device__ __forceinline__ void halo_emit(HaloStage* st, const HaloParams& hp,
                                          const double* __restrict__ xyz, size_t idx,
                                          double* __restrict__ out,
                                          unsigned long long* __restrict__ counts) {
  const double x = xyz[3 * idx + 0], y = xyz[3 * idx + 1];
  const double cx = (hp.base_x - (x - hp.sub_x)) * hp.inv_res;
  const double cy = (hp.base_y - (y - hp.sub_y)) * hp.inv_res;
#pragma nounroll
  for (int d = 0; d < hp.nd; ++d) {
    if (cx >= st->lo_i[d] && cx <= st->hi_i[d] && cy >= st->lo_j[d] && cy <= st->hi_j[d]) {
      const unsigned pos = atomicAdd(&st->n, 1u);
      if (pos < kHaloStage) {
        const unsigned ls = atomicAdd(&st->cnt[d], 1u);
        st->entry[pos] = (unsigned long long)idx | ((unsigned long long)d << 32) |
                         ((unsigned long long)ls << 36);
      } else {
        halo_write(hp, xyz, idx, d, atomicAdd(&counts[d], 1ull), out);
      }
    }
  }
}

// all threads of the workgroup, after their last halo_emit
__device__ __forceinline__ void halo_flush(HaloStage* st, const HaloParams& hp,
                                           const double* __restrict__ xyz,
                                           double* __restrict__ out,
                                           unsigned long long* __restrict__ counts) {
  __syncthreads();
  if ((int)threadIdx.x < hp.nd && st->cnt[threadIdx.x])
    st->base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], (unsigned long long)st->cnt[threadIdx.x]);
  __syncthreads();
  const unsigned m = min(st->n, kHaloStage);
  for (unsigned e = threadIdx.x; e < m; e += blockDim.x) {
    const unsigned long long v = st->entry[e];
    const int d = (int)((v >> 32) & 15u);
    halo_write(hp, xyz, (size_t)(v & 0xFFFFFFFFull), d, st->base[d] + (v >> 36), out);
  }
}

// kHalo: the pass also copies the points other windows need into their send rows -- it
// reads every point anyway (amhip_dsm_tiled_begin_dev).
// zall (may be null; the record pipeline): every wave also leaves the [min, max] of the heights
// of the points it BINS (plain stores, range_commit_wave) -- the records' reference height zref is the
// middle of that range and has to exist before the first scatter pass writes a record.
// (Any zref is correct; the middle of a SAMPLE's range would save this pass 0.015 ms, but then
// the records -- and one height in 500 by a float spacing -- depend on the order of the cloud:
// tests/test_gpu_fullsize.py's permutation property.  The full range is a function of the set.)
template <bool kHalo>
__global__ void __launch_bounds__(kP3CountThreads)
k_dsm_p3_count(const double* __restrict__ xyz, size_t n, DsmParams p,
               uint32_t* __restrict__ hist_rows, HaloParams hp, double* __restrict__ halo_out,
               unsigned long long* __restrict__ halo_counts, double* __restrict__ zall) {
  extern __shared__ uint32_t s_hist[];
  const int nk = p.p3_n1 * p.p3_n2;
  // (kHalo: the staging area follows the histogram, 8-byte aligned)
  HaloStage* const stage = reinterpret_cast<HaloStage*>(s_hist + ((nk + 1) & ~1));
  for (int k = threadIdx.x; k < nk; k += kP3CountThreads) s_hist[k] = 0;
  if (kHalo) halo_stage_init(stage, hp);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * kP3CountThreads;
  // four points' loads in flight per lane before any of the (branchy) bookkeeping
  constexpr int kU = 4;
  double zlo = __builtin_huge_val(), zhi = -__builtin_huge_val();
  for (size_t base = (size_t)blockIdx.x * kP3CountThreads + threadIdx.x; base < n;
       base += kU * stride) {
    double x[kU], y[kU], z[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const size_t idx = base + u * stride;
      z[u] = __builtin_nan("");  // (fmin / fmax pass over it)
      if (idx < n) {
        x[u] = xyz[3 * idx + 0];
        y[u] = xyz[3 * idx + 1];
        if (zall) z[u] = xyz[3 * idx + 2];  // (the same cache lines)
      }
    }
    // (the range AFTER the loads: with fmin / fmax next to each load the four waited for one
    // another -- 0.245 ms against the 0.228 of the pass without heights.  Only BINNED points
    // count: a stray height off the map -- a sentinel, an outlier -- would move zref away from
    // the terrain, every record offset would be large and every tile's single-precision budget
    // gone)
    unsigned look = 0;  // kHalo: which of the four may have to travel
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const size_t idx = base + u * stride;
      if (idx < n) {
        const double px = x[u] - p.sub_x;  // dsm.cc:42
        const double py = y[u] - p.sub_y;  // dsm.cc:43
        int k1, k2;
        if (p3_keys(p, px, py, &k1, &k2)) {
          atomicAdd(&s_hist[k1 * p.p3_n2 + k2], 1u);
          if (zall) {
            zlo = fmin(zlo, z[u]);
            zhi = fmax(zhi, z[u]);
          }
        }
        if (kHalo && halo_candidate(hp, x[u], y[u])) look |= 1u << u;
      }
    }
    if (kHalo) {
#pragma nounroll
      while (look) {
        const int u = __builtin_ctz(look);
        look &= look - 1;
        halo_emit(stage, hp, xyz, base + (size_t)u * stride, halo_out, halo_counts);
      }
    }
  }
  if (kHalo) halo_flush(stage, hp, xyz, halo_out, halo_counts);
  if (zall)
    range_commit_wave(zlo, zhi, zall, (size_t)blockIdx.x * (kP3CountThreads / 64) + (threadIdx.x >> 6));
  __syncthreads();
  uint32_t* row = hist_rows + (size_t)blockIdx.x * nk;
  for (int k = threadIdx.x; k < nk; k += kP3CountThreads) row[k] = s_hist[k];
}

// What the single-workgroup kernel in front of the first scatter pass (k_dsm_p3_reduce_scan,
// k_scan_top) resets on its way, so that no launch of its own
// is spent on it: the call's own height range (k_range_reduce folds the scatter waves' partials
// into it later) and the counters of the gather's tile lists (amhip_dsm.hip: dsm_run).
struct SortAux {
  unsigned long long* call_range;  // may be null
  uint32_t* zero_words;            // may be null
  int nzero;
};
__device__ __forceinline__ void aux_reset(const SortAux& a) {
  if (threadIdx.x == 0 && a.call_range) {
    a.call_range[0] = kOrderedPlusInf;
    a.call_range[1] = kOrderedMinusInf;
  }
  if (a.zero_words && (int)threadIdx.x < a.nzero) a.zero_words[threadIdx.x] = 0u;
}

// One block.  start2 = exclusive scan of the (k1, k2) counts (+ total) and a
// copy as the pass-2 append cursors; start1 / cursor1 for pass 1; blk2 = first
// pass-2 workgroup of every k1 partition (partitions are cut into chunks).
// (the body: one workgroup of 1024 threads)
__device__ __forceinline__ void
p3_scan_body(const uint32_t* __restrict__ cnt, int n1, int n2,
              uint32_t* __restrict__ start2, uint32_t* __restrict__ cursor2,
              uint32_t* __restrict__ start1, uint32_t* __restrict__ cursor1,
              uint32_t* __restrict__ blk2, unsigned cap_small, unsigned cap_big,
              uint32_t* __restrict__ big_list, unsigned chunk,
              unsigned* __restrict__ host_big /* may be null: pinned mirror of the big list's length */) {
  __shared__ unsigned lds[1024 / 64 + 1];
  __shared__ unsigned s_start1[kP3MaxKeys + 1];
  const int nk = n1 * n2;
  unsigned carry = 0;
  if (threadIdx.x == 0) big_list[0] = 0;
  __syncthreads();
  // A wave owns a contiguous segment of the counters and walks it 64 at a time (coalesced; all
  // its loads in flight at once, wave scans without barriers), the workgroup scans the 16 segment
  // totals ONCE.  (Rounds of 1024 counters with a block scan each: 26 us for the 30 240
  // sub-partitions of a 50 M-point call; a thread per run of consecutive counters: 40 us, every
  // load instruction touches 64 lines; this: 17 us.)
  constexpr int kMaxIt = 32;  // (amhip_api.hip: n1 * n2 <= 32768 = 16 waves x 32 x 64)
  {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int seg = ((nk + 16 * 64 - 1) / (16 * 64)) * 64;  // counters per wave, a multiple of 64
    const int iters = seg / 64;
    const int w0 = wid * seg;
    unsigned v[kMaxIt];  // the counter, then its exclusive prefix inside the wave's segment
#pragma unroll
    for (int q = 0; q < kMaxIt; ++q) {
      const int i = w0 + q * 64 + lane;
      v[q] = (q < iters && i < nk) ? cnt[i] : 0u;
    }
    unsigned run = 0;   // (wave-uniform)
#pragma unroll
    for (int q = 0; q < kMaxIt; ++q) {
      if (q < iters) {
        // sub-partitions too full for k_dsm_p3_place's registers but not for a whole
        // CU's LDS (denser parts of a non-uniform cloud): k_dsm_p3_place_big's list
        // (no upper bound: beyond a CU's LDS the big kernel places in several rounds)
        if (v[q] > cap_small)
          big_list[1 + atomicAdd(&big_list[0], 1u)] = (uint32_t)(w0 + q * 64 + lane);
        const unsigned incl = wave_incl_scan(v[q], lane);
        v[q] = run + incl - v[q];
        run += __shfl(incl, 63, 64);
      }
    }
    if (lane == 0) lds[wid] = run;
    __syncthreads();
    unsigned base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const unsigned t = lds[w];
      if (w < wid) base += t;
      total += t;
    }
#pragma unroll
    for (int q = 0; q < kMaxIt; ++q) {
      const int i = w0 + q * 64 + lane;
      if (q < iters && i < nk) {
        const unsigned st = base + v[q];
        start2[i] = st;
        cursor2[i] = st;
        if (i % n2 == 0) s_start1[i / n2] = st;
      }
    }
    carry = total;
  }
  __syncthreads();
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
    nblk = (s_start1[k + 1] - s_start1[k] + chunk - 1) / chunk;
  }
  if (k == 0) start1[n1] = carry;
  unsigned total;
  const unsigned ex = block_excl_scan<1024>(nblk, &total, lds);
  if (k < n1) blk2[k] = ex;
  if (k == 0) blk2[n1] = total;
  // (how many sub-partitions the big placement kernel has to take: the NEXT call of this geometry
  // launches it only if this is not zero -- a device store into pinned host memory, never waited for)
  if (k == 0 && host_big) host_big[0] = big_list[0];
}

// Count pass, second half: the reduction of the count workgroups' histogram rows (64 counters per
// workgroup, the rows dealt to its 16 waves) and -- by the workgroup that finishes LAST (a ticket)
// -- the scan above: one launch instead of two, and the scan starts the moment the last counter is
// written.  `ticket` is zero at launch and left zero.
__global__ void __launch_bounds__(1024)
k_dsm_p3_reduce_scan(const uint32_t* __restrict__ hist_rows, int nrows, uint32_t* __restrict__ cnt, int n1, int n2,
                     uint32_t* __restrict__ start2, uint32_t* __restrict__ cursor2,
                     uint32_t* __restrict__ start1, uint32_t* __restrict__ cursor1,
                     uint32_t* __restrict__ blk2, unsigned cap_small, unsigned cap_big,
                     uint32_t* __restrict__ big_list, unsigned chunk,
                     SortAux aux, unsigned* __restrict__ host_big, unsigned* __restrict__ ticket) {
  __shared__ uint32_t s_part[16][64];
  __shared__ unsigned s_last;
  const int nk = n1 * n2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  uint32_t sum = 0;
  if (k < nk)
    for (int r = wid; r < nrows; r += 16) sum += hist_rows[(size_t)r * nk + k];
  s_part[wid][lane] = sum;
  __syncthreads();
  if (wid == 0 && k < nk) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += s_part[w][lane];
    cnt[k] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();  // (this workgroup's counters before its ticket)
    s_last = atomicAdd(ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();    // (the others' counters after the last ticket)
  if (threadIdx.x == 0) *ticket = 0u;
  aux_reset(aux);
  p3_scan_body(cnt, n1, n2, start2, cursor2, start1, cursor1, blk2, cap_small, cap_big, big_list, chunk, host_big);
}


// Passes 1 and 2.  kFirst: chunk of the input cloud, key k1, values/centre
// handling of the reference; else: chunk of one k1 partition, key k2.
// vb: the chunk.
template <bool kFirst>
__device__ __forceinline__ void p3_scatter_body(const unsigned vb, const double* __restrict__ src,
                                                const int32_t* __restrict__ values, size_t n,
                                                const DsmParams& p, const uint32_t* __restrict__ start1,
                                                const uint32_t* __restrict__ blk2,
                                                uint32_t* __restrict__ cursor, double* __restrict__ dst,
                                                double* __restrict__ zpart) {
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
    c0 = (size_t)vb * kP3Chunk;
    c1 = min(c0 + (size_t)kP3Chunk, n);
    nkeys = p.p3_n1;
  } else {
    const int n1 = p.p3_n1;
    const uint32_t b = vb;
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
  double zlo = __builtin_huge_val(), zhi = -__builtin_huge_val();
  uint32_t slot[kP3PerThread];  // key << 13 | rank in the chunk's run of that key
  // (all loads first, branch-free: see k_dsm_p3_scatter_rec)
  if (c1 <= c0) return;
#pragma unroll
  for (int k = 0; k < kP3PerThread; ++k) {
    const size_t ld = min(c0 + tid + (size_t)k * kP3Threads, c1 - 1);
    px[k] = src[3 * ld + 0];
    py[k] = src[3 * ld + 1];
    pz[k] = src[3 * ld + 2];
  }
  if (kFirst && values) {  // (OrthoFromPcl: the intensities take the heights' place)
#pragma unroll
    for (int k = 0; k < kP3PerThread; ++k)
      pz[k] = (double)values[min(c0 + tid + (size_t)k * kP3Threads, c1 - 1)];
  }
#pragma unroll
  for (int k = 0; k < kP3PerThread; ++k) {
    const size_t idx = c0 + tid + (size_t)k * kP3Threads;
    slot[k] = 0xFFFFFFFFu;
    if (idx < c1) {
      double x = px[k], y = py[k];
      const double z = pz[k];
      if (kFirst) {
        x -= p.sub_x;
        y -= p.sub_y;
      }
      px[k] = x;
      py[k] = y;
      int k1, k2;
      if (p3_keys(p, x, y, &k1, &k2)) {
        const int key = kFirst ? k1 : k2;
        slot[k] = ((uint32_t)key << 13) | atomicAdd(&s_cnt[key], 1u);
        if (kFirst) {
          zlo = fmin(zlo, z);
          zhi = fmax(zhi, z);
        }
      }
    }
  }
  if (kFirst && zpart)
    range_commit_wave(zlo, zhi, zpart, (size_t)vb * (kP3Threads / 64) + (tid >> 6));
  __syncthreads();
  unsigned my_base = 0;  // first slot of key `tid`'s run in the destination
  {
    const unsigned c = (tid < nkeys) ? s_cnt[tid] : 0u;
    unsigned total;
    const unsigned ex = block_excl_scan<kP3Threads>(c, &total, s_scan);
    if (tid < nkeys) {
      s_off[tid] = ex;
      // (the reservation's round trip to the counter runs under the placement below: its
      // result is only stored -- and so only awaited -- after it)
      if (c) my_base = atomicAdd(&cursor[tid], c);
    }
    if (tid == 0) s_scan[23] = total;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kP3PerThread; ++k) {
    if (slot[k] != 0xFFFFFFFFu) {
      const uint32_t key = slot[k] >> 13, rank = slot[k] & 0x1FFFu;
      const uint32_t q = s_off[key] + rank;
      s_pts[3 * q + 0] = px[k];
      s_pts[3 * q + 1] = py[k];
      s_pts[3 * q + 2] = pz[k];
      s_dest[q] = slot[k];
    }
  }
  if (tid < nkeys) s_base[tid] = my_base;
  __syncthreads();
  const uint32_t ne = 3u * s_scan[23];
  for (uint32_t e = tid; e < ne; e += kP3Threads) {
    const uint32_t q = e / 3u;
    const uint32_t d = s_dest[q];
    const uint32_t b = s_base[d >> 13];
    dst[3 * (size_t)(b + (d & 0x1FFFu)) + (e - 3u * q)] = s_pts[e];
  }
}

template <bool kFirst>
__global__ void __launch_bounds__(kP3Threads)
k_dsm_p3_scatter(const double* __restrict__ src, const int32_t* __restrict__ values, size_t n,
                 DsmParams p, const uint32_t* __restrict__ start1,
                 const uint32_t* __restrict__ blk2, uint32_t* __restrict__ cursor,
                 double* __restrict__ dst, double* __restrict__ zpart) {
  p3_scatter_body<kFirst>(blockIdx.x, src, values, n, p, start1, blk2, cursor, dst, zpart);
}

// Pass 3: one workgroup per (k1, k2) sub-partition.
// bin_z (may be null): per bin the ordered keys (zkey_of) of its lowest / highest float-rounded
// height -- the occupancy pre-pass of the single-precision gather reads them (amhip_dsm.hip)
__device__ __forceinline__ uint32_t place_zkey(double z) {
  const uint32_t b = __float_as_uint((float)z);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
template <int THREADS, int PER>
__device__ __forceinline__ void place_subpartition(const double* __restrict__ src, const DsmParams& p,
                                                   int cap, const uint32_t* __restrict__ start2,
                                                   uint32_t* __restrict__ bin_start,
                                                   double* __restrict__ sorted, int sp,
                                                   unsigned skip_lo, unsigned skip_hi,
                                                   uint2* __restrict__ bin_z) {
  extern __shared__ double s_pts[];                                  // 3 * cap
  uint32_t* s_bins = reinterpret_cast<uint32_t*>(s_pts + 3 * cap);   // p3_w
  uint32_t* s_scan = s_bins + p.p3_w;                                // 24
  uint32_t* s_zlo = s_scan + 24;                                     // p3_w (bin_z only)
  uint32_t* s_zhi = s_zlo + p.p3_w;                                  // p3_w
  const int tid = threadIdx.x;
  const int k1 = sp / p.p3_n2, k2 = sp - k1 * p.p3_n2;
  const int rr = k2 / p.p3_c;
  const int row = k1 * p.p3_r1 + rr;
  const int bx0 = (k2 - rr * p.p3_c) * p.p3_w;
  const int nbw = min(p.p3_w, p.nbx - bx0);
  if (sp == 0 && tid == 0)
    bin_start[(size_t)p.nbx * p.nby] = start2[p.p3_n1 * p.p3_n2];
  if (row >= p.nby || nbw <= 0) return;  // no bins (and therefore no points)
  const uint32_t g0 = start2[sp], g1 = start2[sp + 1];
  for (int k = tid; k < nbw; k += THREADS) {
    s_bins[k] = 0;
    if (bin_z) {
      s_zlo[k] = 0xFFFFFFFFu;
      s_zhi[k] = 0u;
    }
  }
  __syncthreads();
  // (skip_lo, skip_hi]: sub-partitions the other launch takes
  if ((g1 - g0) > skip_lo && (g1 - g0) <= skip_hi) return;
  const bool in_lds = (int)(g1 - g0) <= cap && cap <= THREADS * PER;
  // the sub-partition is read ONCE: a thread keeps its points (<= 8) in
  // registers between the count and the placement
  double px[PER], py[PER], pz[PER];
  int pb[PER];
  if (in_lds) {
    // (all loads first, branch-free: see k_dsm_p3_scatter_rec)
    if (g1 > g0) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const size_t ld = min(g0 + tid + (uint32_t)k * THREADS, g1 - 1);
        px[k] = src[3 * ld + 0];
        py[k] = src[3 * ld + 1];
        pz[k] = src[3 * ld + 2];
      }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t idx = g0 + tid + (uint32_t)k * THREADS;
      pb[k] = -1;
      if (idx < g1) {
        int bx, by;
        point_bin_xy(p, px[k], py[k], &bx, &by);
        pb[k] = bx - bx0;
        atomicAdd(&s_bins[pb[k]], 1u);
        if (bin_z) {
          const uint32_t zk = place_zkey(pz[k]);
          atomicMin(&s_zlo[pb[k]], zk);
          atomicMax(&s_zhi[pb[k]], zk);
        }
      }
    }
  } else {
    for (uint32_t idx = g0 + tid; idx < g1; idx += THREADS) {
      int bx, by;
      point_bin_xy(p, src[3 * (size_t)idx + 0], src[3 * (size_t)idx + 1], &bx, &by);
      atomicAdd(&s_bins[bx - bx0], 1u);
    }
  }
  __syncthreads();
  {
    const int per = (nbw + THREADS - 1) / THREADS;
    const int lo = tid * per;
    const int hi = min(lo + per, nbw);
    unsigned sum = 0;
    for (int k = lo; k < hi; ++k) sum += s_bins[k];
    unsigned total;
    unsigned run = block_excl_scan<THREADS>(sum, &total, s_scan);
    for (int k = lo; k < hi; ++k) {
      const unsigned t = s_bins[k];
      s_bins[k] = run;
      run += t;
    }
  }
  __syncthreads();
  uint32_t* out_start = bin_start + (size_t)row * p.nbx + bx0;
  for (int k = tid; k < nbw; k += THREADS) out_start[k] = g0 + s_bins[k];
  if (bin_z && in_lds) {
    uint2* zrow = bin_z + (size_t)row * p.nbx + bx0;
    for (int k = tid; k < nbw; k += THREADS) zrow[k] = make_uint2(s_zlo[k], s_zhi[k]);
  }
  __syncthreads();
  if (!in_lds) {
    // over-full sub-partition (clustered cloud): second read, direct placement
    for (uint32_t idx = g0 + tid; idx < g1; idx += THREADS) {
      const double x = src[3 * (size_t)idx + 0];
      const double y = src[3 * (size_t)idx + 1];
      const double z = src[3 * (size_t)idx + 2];
      int bx, by;
      point_bin_xy(p, x, y, &bx, &by);
      const size_t o = (size_t)g0 + atomicAdd(&s_bins[bx - bx0], 1u);
      sorted[3 * o + 0] = x;
      sorted[3 * o + 1] = y;
      sorted[3 * o + 2] = z;
      if (bin_z) {
        const uint32_t zk = place_zkey(z);
        atomicMin(&s_zlo[bx - bx0], zk);
        atomicMax(&s_zhi[bx - bx0], zk);
      }
    }
    if (bin_z) {
      __syncthreads();
      uint2* zrow = bin_z + (size_t)row * p.nbx + bx0;
      for (int k = tid; k < nbw; k += THREADS) zrow[k] = make_uint2(s_zlo[k], s_zhi[k]);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
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
  for (uint32_t e = tid; e < ne; e += THREADS) out[e] = s_pts[e];
}


// ---------------------------------------------------------------------------
// Pass 3 for sub-partitions beyond one LDS image (contexts of more than ~130 M points: the
// plan's <= 32 K sub-partitions then hold more than kP3BigCap points each; also the densest
// parts of clustered clouds).  Until round 4 such a sub-partition was placed DIRECTLY: a second
// read and one scattered 24-byte store per point through LDS cursors -- 11 of the 45 ms of
// BASELINE configs[3] on one GPU.  Now: one read for the bins' histogram (-> bin_start), then
// ROUNDS: the next run of consecutive bins that fits the LDS image (chosen from the exact
// counts: never an overflow) is collected from a re-read of the sub-partition -- it sits in the
// L2 by then --, counting-sorted in LDS and written as ONE contiguous coalesced range.  A
// single bin larger than the image (hundreds of points per cell) is placed directly.
// kRec: 20-byte records in, 16-byte records + rows out (place_records' formats).
// ---------------------------------------------------------------------------
template <int THREADS, bool kRec, int PER>
__device__ __forceinline__ void place_rounds(const void* __restrict__ src_v, const DsmParams& p, int cap,
                                             const uint32_t* __restrict__ start2,
                                             uint32_t* __restrict__ bin_start, double* __restrict__ sorted,
                                             uint4* __restrict__ rec16, uint32_t* __restrict__ sidx,
                                             uint2* __restrict__ bin_z, int sp) {
  // PER > 0: the sub-partition holds at most THREADS * PER points and a thread keeps its PER of
  // them in REGISTERS from the one read to the last round (configs[3] on one GPU: 13.3 K points
  // per sub-partition, 14 per thread); PER == 0: any size, every round re-reads it (from the L2).
  extern __shared__ double s_pts_raw[];
  constexpr int kWordsPer = kRec ? 5 : 6;
  constexpr int kRegs = PER > 0 ? PER : 1;
  uint32_t* s_words = reinterpret_cast<uint32_t*>(s_pts_raw);          // kWordsPer * cap
  uint32_t* s_bins = s_words + (size_t)kWordsPer * cap;                 // p3_w + 1: starts (+ total)
  uint32_t* s_cur = s_bins + p.p3_w + 1;                                // p3_w: cursors of the round
  uint32_t* s_scan = s_cur + p.p3_w;                                    // 24
  uint32_t* s_zlo = s_scan + 24;                                        // p3_w (bin_z only)
  uint32_t* s_zhi = s_zlo + p.p3_w;                                     // p3_w
  const double* srcd = reinterpret_cast<const double*>(src_v);
  const uint32_t* srcw = reinterpret_cast<const uint32_t*>(src_v);
  const int tid = threadIdx.x;
  const int k1 = sp / p.p3_n2, k2 = sp - k1 * p.p3_n2;
  const int rr = k2 / p.p3_c;
  const int row = k1 * p.p3_r1 + rr;
  const int bx0 = (k2 - rr * p.p3_c) * p.p3_w;
  const int nbw = min(p.p3_w, p.nbx - bx0);
  if (row >= p.nby || nbw <= 0) return;
  const uint32_t g0 = start2[sp], g1 = start2[sp + 1];
  auto zkey = [](uint32_t fbits) { return (fbits >> 31) ? ~fbits : (fbits | 0x80000000u); };
  for (int k = tid; k <= nbw; k += THREADS) s_bins[k] = 0;
  if (bin_z)
    for (int k = tid; k < nbw; k += THREADS) {
      s_zlo[k] = 0xFFFFFFFFu;
      s_zhi[k] = 0u;
    }
  __syncthreads();
  // the bin of sorted point idx (PER == 0)
  auto bin_of = [&](uint32_t idx) -> int {
    if (kRec) return div_by((int)(srcw[(size_t)kRecWords * idx] & 0xFFFFu), p.B, p.mul_B) - bx0;
    int bx, by;
    point_bin_xy(p, srcd[3 * (size_t)idx + 0], srcd[3 * (size_t)idx + 1], &bx, &by);
    return bx - bx0;
  };
  // register-resident points (PER > 0): all loads first, branch-free (rows past the end re-read
  // the last row and get bin -1)
  double rx[kRegs], ry[kRegs], rz[kRegs];
  uint32_t rw[kRegs][kRecWords];
  int rb[kRegs];
  if (PER > 0) {
    if (g1 > g0) {
#pragma unroll
      for (int k = 0; k < kRegs; ++k) {
        const size_t ld = min(g0 + tid + (uint32_t)k * THREADS, g1 - 1);
        if (kRec) {
#pragma unroll
          for (int t = 0; t < kRecWords; ++t) rw[k][t] = srcw[(size_t)kRecWords * ld + t];
        } else {
          rx[k] = srcd[3 * ld + 0];
          ry[k] = srcd[3 * ld + 1];
          rz[k] = srcd[3 * ld + 2];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kRegs; ++k) {
      const uint32_t idx = g0 + tid + (uint32_t)k * THREADS;
      rb[k] = -1;
      if (idx < g1) {
        if (kRec) {
          rb[k] = div_by((int)(rw[k][0] & 0xFFFFu), p.B, p.mul_B) - bx0;
        } else {
          int bx, by;
          point_bin_xy(p, rx[k], ry[k], &bx, &by);
          rb[k] = bx - bx0;
        }
        atomicAdd(&s_bins[rb[k]], 1u);
        if (bin_z) {
          const uint32_t zk = kRec ? zkey(rw[k][3]) : place_zkey(rz[k]);
          atomicMin(&s_zlo[rb[k]], zk);
          atomicMax(&s_zhi[rb[k]], zk);
        }
      }
    }
  } else {
    // ---- histogram (first read) ----
    for (uint32_t idx = g0 + tid; idx < g1; idx += THREADS) {
      const int b = bin_of(idx);
      atomicAdd(&s_bins[b], 1u);
      if (bin_z) {
        const uint32_t zk = kRec ? zkey(srcw[(size_t)kRecWords * idx + 3]) : place_zkey(srcd[3 * (size_t)idx + 2]);
        atomicMin(&s_zlo[b], zk);
        atomicMax(&s_zhi[b], zk);
      }
    }
  }
  __syncthreads();
  {
    const int per = (nbw + THREADS - 1) / THREADS;
    const int lo = tid * per;
    const int hi = min(lo + per, nbw);
    unsigned sum = 0;
    for (int k = lo; k < hi; ++k) sum += s_bins[k];
    unsigned total;
    unsigned run = block_excl_scan<THREADS>(sum, &total, s_scan);
    for (int k = lo; k < hi; ++k) {
      const unsigned t = s_bins[k];
      s_bins[k] = run;
      run += t;
    }
    if (tid == 0) s_bins[nbw] = total;
  }
  __syncthreads();
  {
    uint32_t* out_start = bin_start + (size_t)row * p.nbx + bx0;
    for (int k = tid; k < nbw; k += THREADS) out_start[k] = g0 + s_bins[k];
    if (bin_z) {
      uint2* zrow = bin_z + (size_t)row * p.nbx + bx0;
      for (int k = tid; k < nbw; k += THREADS) zrow[k] = make_uint2(s_zlo[k], s_zhi[k]);
    }
  }
  // one point of the round: to its slot of the LDS image, or (a bin beyond the image) to memory
  auto put = [&](bool direct, uint32_t q, uint32_t base, const uint32_t* v, double x, double y, double z) {
    if (kRec) {
      if (direct) {
        rec16[(size_t)g0 + q] = make_uint4(v[0], v[1], v[2], v[3]);
        sidx[(size_t)g0 + q] = v[4];
      } else {
#pragma unroll
        for (int t = 0; t < kRecWords; ++t) s_words[(size_t)kRecWords * (q - base) + t] = v[t];
      }
    } else {
      double* o = direct ? sorted + 3 * ((size_t)g0 + q) : s_pts_raw + 3 * (size_t)(q - base);
      o[0] = x;
      o[1] = y;
      o[2] = z;
    }
  };
  // ---- rounds ----
  int lo_bin = 0;
  while (lo_bin < nbw) {
    // the longest run of bins [lo_bin, hi_bin) with at most cap points (binary search on the
    // starts, every thread the same); a bin larger than cap forms a run of its own (direct)
    const uint32_t base = s_bins[lo_bin];
    int a = lo_bin, b = nbw;  // s_bins[a] - base <= cap always; find the largest such index
    while (b - a > 0) {
      const int mid = (a + b + 1) >> 1;
      if (s_bins[mid] - base <= (uint32_t)cap) a = mid; else b = mid - 1;
    }
    const bool direct = a == lo_bin;
    const int hi_bin = direct ? lo_bin + 1 : a;
    const uint32_t cnt = s_bins[hi_bin] - base;
    for (int k = lo_bin + tid; k < hi_bin; k += THREADS) s_cur[k] = s_bins[k];
    __syncthreads();
    if (PER > 0) {
#pragma unroll
      for (int k = 0; k < kRegs; ++k) {
        if (rb[k] >= lo_bin && rb[k] < hi_bin) {
          const uint32_t q = atomicAdd(&s_cur[rb[k]], 1u);
          put(direct, q, base, rw[k], rx[k], ry[k], rz[k]);
        }
      }
    } else {
      for (uint32_t idx = g0 + tid; idx < g1; idx += THREADS) {
        const int bb = bin_of(idx);
        if (bb < lo_bin || bb >= hi_bin) continue;
        const uint32_t q = atomicAdd(&s_cur[bb], 1u);
        uint32_t v[kRecWords] = {0u, 0u, 0u, 0u, 0u};
        double x = 0.0, y = 0.0, z = 0.0;
        if (kRec) {
#pragma unroll
          for (int t = 0; t < kRecWords; ++t) v[t] = srcw[(size_t)kRecWords * idx + t];
        } else {
          x = srcd[3 * (size_t)idx + 0];
          y = srcd[3 * (size_t)idx + 1];
          z = srcd[3 * (size_t)idx + 2];
        }
        put(direct, q, base, v, x, y, z);
      }
    }
    __syncthreads();
    if (!direct) {
      if (kRec) {
        for (uint32_t q = tid; q < cnt; q += THREADS) {
          const uint32_t* r = s_words + (size_t)kRecWords * q;
          rec16[(size_t)g0 + base + q] = make_uint4(r[0], r[1], r[2], r[3]);
          sidx[(size_t)g0 + base + q] = r[4];
        }
      } else {
        double* out = sorted + 3 * ((size_t)g0 + base);
        for (uint32_t e = tid; e < 3u * cnt; e += THREADS) out[e] = s_pts_raw[e];
      }
    }
    __syncthreads();
    lo_bin = hi_bin;
  }
}

// cap points in LDS, <= kP3PlacePer per thread in registers; fuller
// sub-partitions are left to k_dsm_p3_place_big (skip_lo < count <= skip_hi) or,
// beyond a CU's LDS, placed directly with a second read.
__global__ void __launch_bounds__(kP3PlaceThreads)
k_dsm_p3_place(const double* __restrict__ src, DsmParams p, int cap,
               const uint32_t* __restrict__ start2, uint32_t* __restrict__ bin_start,
               double* __restrict__ sorted, unsigned skip_lo, unsigned skip_hi,
               uint2* __restrict__ bin_z,
               // (how many sub-partitions the big placement kernel has to take, left in a pinned host
               // word for the NEXT call's launch policy by one thread of this large kernel -- not by
               // the single-workgroup scan every pass waits for; may be null)
               const uint32_t* __restrict__ big_list, unsigned* __restrict__ host_big) {
  if (host_big && blockIdx.x == 0 && threadIdx.x == 0) host_big[0] = big_list[0];
  place_subpartition<kP3PlaceThreads, kP3PlacePer>(src, p, cap, start2, bin_start, sorted,
                                                   (int)blockIdx.x, skip_lo, skip_hi, bin_z);
}

// The same with 1024 threads and a whole CU's LDS (kP3BigCap points), walking
// the list the scan made of the sub-partitions in (skip_lo, skip_hi].
__global__ void __launch_bounds__(kP3BigThreads)
k_dsm_p3_place_big(const double* __restrict__ src, DsmParams p,
                   const uint32_t* __restrict__ start2, uint32_t* __restrict__ bin_start,
                   double* __restrict__ sorted, const uint32_t* __restrict__ big_list,
                   uint2* __restrict__ bin_z, int cap_rounds, unsigned rounds_above, unsigned reg_max) {
  const unsigned count = big_list[0];
  for (unsigned k = blockIdx.x; k < count; k += gridDim.x) {
    const int sp = (int)big_list[1 + k];
    const uint32_t cnt = start2[sp + 1] - start2[sp];
    if (cnt > rounds_above && cnt <= min(reg_max, (unsigned)(kP3BigThreads * kP3RoundsPer)))
      place_rounds<kP3BigThreads, false, kP3RoundsPer>(src, p, cap_rounds, start2, bin_start, sorted, nullptr,
                                                       nullptr, bin_z, sp);
    else if (cnt > rounds_above)
      place_rounds<kP3BigThreads, false, 0>(src, p, cap_rounds, start2, bin_start, sorted, nullptr, nullptr,
                                            bin_z, sp);
    else
      place_subpartition<kP3BigThreads, kP3BigPer>(src, p, kP3BigCap, start2, bin_start, sorted, sp, 0u,
                                                   0u, bin_z);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// the record pipeline of the single-precision gather (DsmParams::rec_mode)
// ---------------------------------------------------------------------------
// The single-precision gather turns every point into a cell, two fixed-point offsets from
// that cell's centre and an f32 height offset anyway (amhip_dsm.hip, gather_tile_f32) -- so the
// sort carries exactly that instead of the three doubles: 20-byte records
//     w0 = ix | iy << 16   cell of the point in map cells + margin M (< 65536 each way)
//     w1, w2               offsets from the cell centre, int32 in units of 2^-fx_S cells
//     w3                   f32 bits of z - zref   (zref: the middle of the cloud's height range)
//     w4                   row of the point in the caller's cloud
// through the two scatter passes (44 + 40 bytes per point instead of 48 + 48; the keys of
// passes 2 and 3 are integer arithmetic on w0, no FP64), and the placement pass splits them
// into 16-byte records (w0 .. w3: what the gather stages) and the rows (w4: what the few
// routines that redo a cell or a tile in the reference's doubles use to fetch them from the
// untouched cloud).  ~0.3 ms less sort and ~0.1 ms less staging per 50 M points.
// The reference's doubles are NOT lost: they stay where the caller put them.
// (20-byte records: 3072 points per scatter workgroup fit the LDS budget of two workgroups per
// CU that 2560 24-byte points use -- longer runs)
constexpr int kRecPerThread = 6;
constexpr int kRecChunk = kP3Threads * kRecPerThread;

// zref[0] = middle of [min, max] over the partials the count pass left, [1] / [2] the range
__global__ void __launch_bounds__(1024)
k_dsm_zref(const double* __restrict__ part, size_t nparts, double* __restrict__ zref) {
  __shared__ double s_pair[2 * 16];
  double lo = __builtin_huge_val(), hi = -__builtin_huge_val();
  for (size_t k = threadIdx.x; k < nparts; k += 1024) {
    lo = fmin(lo, part[2 * k]);
    hi = fmax(hi, part[2 * k + 1]);
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
    for (int w = 1; w < 16; ++w) {
      lo = fmin(lo, s_pair[2 * w]);
      hi = fmax(hi, s_pair[2 * w + 1]);
    }
    const bool ok = lo <= hi && lo > -1.0e300 && hi < 1.0e300;
    zref[0] = ok ? 0.5 * lo + 0.5 * hi : 0.0;
    zref[1] = lo;
    zref[2] = hi;
  }
}

// the record of a (centre-shifted) point; false: outside the binned area (like point_bin)
__device__ __forceinline__ bool make_record(const DsmParams& p, double px, double py, double z,
                                            double zref, uint32_t row, uint32_t* w) {
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
  const double scale = (double)(1u << p.fx_S);
  // offset from the cell's centre in cells, in [-0.5, 0.5] (the same value gather_tile_f32
  // formed from the doubles), rounded to the fixed-point grid
  const double fx = (cx + (double)p.M) - (double)ix;
  const double fy = (cy + (double)p.M) - (double)iy;
  w[0] = (uint32_t)ix | ((uint32_t)iy << 16);
  w[1] = (uint32_t)(int)rint(fx * scale);
  w[2] = (uint32_t)(int)rint(fy * scale);
  w[3] = __float_as_uint((float)(z - zref));
  w[4] = row;
  return true;
}

__device__ __forceinline__ void record_keys(const DsmParams& p, uint32_t w0, int* k1, int* k2,
                                            int* bx_out) {
  const int bx = div_by((int)(w0 & 0xFFFFu), p.B, p.mul_B);
  const int by = div_by((int)(w0 >> 16), p.B, p.mul_B);
  const int a = div_by(by, p.p3_r1, p.mul_r1);
  *k1 = a;
  *k2 = (by - a * p.p3_r1) * p.p3_c + div_by(bx, p.p3_w, p.mul_w);
  *bx_out = bx;
}

// Passes 1 and 2 on records.  kFirst: chunk of the caller's cloud -> records, key k1; else:
// chunk of one k1 partition of records, key k2.  Same structure as k_dsm_p3_scatter.
template <bool kFirst>
__global__ void __launch_bounds__(kP3Threads)
k_dsm_p3_scatter_rec(const double* __restrict__ cloud, const uint32_t* __restrict__ src, size_t n,
                     DsmParams p, const double* __restrict__ zref,
                     const uint32_t* __restrict__ start1, const uint32_t* __restrict__ blk2,
                     uint32_t* __restrict__ cursor, uint32_t* __restrict__ dst,
                     double* __restrict__ zpart) {
  extern __shared__ uint32_t s_words[];                       // kRecWords * kRecChunk
  uint32_t* s_dest = s_words + kRecWords * kRecChunk;          // kP3Chunk
  uint32_t* s_cnt = s_dest + kRecChunk;                        // kP3MaxKeys
  uint32_t* s_off = s_cnt + kP3MaxKeys;
  uint32_t* s_base = s_off + kP3MaxKeys;
  uint32_t* s_scan = s_base + kP3MaxKeys;  // 24
  const int tid = threadIdx.x;
  int nkeys;
  size_t c0, c1;
  if (kFirst) {
    c0 = (size_t)blockIdx.x * kRecChunk;
    c1 = min(c0 + (size_t)kRecChunk, n);
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
    c0 = (size_t)start1[lo] + (size_t)(b - blk2[lo]) * kRecChunk;
    c1 = min(c0 + (size_t)kRecChunk, (size_t)start1[lo + 1]);
    nkeys = p.p3_n2;
    cursor += (size_t)lo * p.p3_n2;
  }
  if (tid < kP3MaxKeys) s_cnt[tid] = 0;
  __syncthreads();
  const double zr = kFirst ? zref[0] : 0.0;
  uint32_t w[kRecPerThread][kRecWords];
  double zlo = __builtin_huge_val(), zhi = -__builtin_huge_val();
  uint32_t slot[kRecPerThread];  // key << 13 | rank in the chunk's run of that key
  // ALL the thread's loads first, branch-free (rows past the chunk's end re-read its last row):
  // with the loads inside the loop below -- whose key arithmetic branches -- the compiler
  // waited for each before issuing the next, six dependent round trips per workgroup.
  if (c1 <= c0) return;  // (workgroup-uniform; never for a launched block)
  double cx[kFirst ? kRecPerThread : 1], cy[kFirst ? kRecPerThread : 1], cz[kFirst ? kRecPerThread : 1];
#pragma unroll
  for (int k = 0; k < kRecPerThread; ++k) {
    const size_t ld = min(c0 + tid + (size_t)k * kP3Threads, c1 - 1);
    if (kFirst) {
      cx[k] = cloud[3 * ld + 0];
      cy[k] = cloud[3 * ld + 1];
      cz[k] = cloud[3 * ld + 2];
    } else {
#pragma unroll
      for (int q = 0; q < kRecWords; ++q) w[k][q] = src[kRecWords * ld + q];
    }
  }
#pragma unroll
  for (int k = 0; k < kRecPerThread; ++k) {
    const size_t idx = c0 + tid + (size_t)k * kP3Threads;
    slot[k] = 0xFFFFFFFFu;
    if (idx < c1) {
      bool in;
      if (kFirst) {
        const double x = cx[k] - p.sub_x;  // dsm.cc:42
        const double y = cy[k] - p.sub_y;  // dsm.cc:43
        const double z = cz[k];
        in = make_record(p, x, y, z, zr, (uint32_t)idx, w[k]);
        if (in) {
          zlo = fmin(zlo, z);
          zhi = fmax(zhi, z);
        }
      } else {
        in = true;
      }
      if (in) {
        int k1, k2, bx;
        record_keys(p, w[k][0], &k1, &k2, &bx);
        const int key = kFirst ? k1 : k2;
        slot[k] = ((uint32_t)key << 13) | atomicAdd(&s_cnt[key], 1u);
      }
    }
  }
  if (kFirst && zpart)
    range_commit_wave(zlo, zhi, zpart, (size_t)blockIdx.x * (kP3Threads / 64) + (tid >> 6));
  __syncthreads();
  unsigned my_base = 0;  // first slot of key `tid`'s run in the destination
  {
    const unsigned c = (tid < nkeys) ? s_cnt[tid] : 0u;
    unsigned total;
    const unsigned ex = block_excl_scan<kP3Threads>(c, &total, s_scan);
    if (tid < nkeys) {
      s_off[tid] = ex;
      // (the reservation's round trip to the counter runs under the placement below: its
      // result is only stored -- and so only awaited -- after it)
      if (c) my_base = atomicAdd(&cursor[tid], c);
    }
    if (tid == 0) s_scan[23] = total;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kRecPerThread; ++k) {
    if (slot[k] != 0xFFFFFFFFu) {
      const uint32_t key = slot[k] >> 13, rank = slot[k] & 0x1FFFu;
      const uint32_t q = s_off[key] + rank;
#pragma unroll
      for (int t = 0; t < kRecWords; ++t) s_words[kRecWords * q + t] = w[k][t];
      s_dest[q] = slot[k];
    }
  }
  if (tid < nkeys) s_base[tid] = my_base;
  __syncthreads();
  // copy-out: a lane per record (16 + 4 bytes; consecutive lanes of a run on consecutive records)
  const uint32_t cnt = s_scan[23];
  for (uint32_t q = tid; q < cnt; q += kP3Threads) {
    const uint32_t d = s_dest[q];
    uint32_t* o = dst + (size_t)kRecWords * (s_base[d >> 13] + (d & 0x1FFFu));
    const uint32_t* r = s_words + kRecWords * q;
    const uint32_t a0 = r[0], a1 = r[1], a2 = r[2], a3 = r[3], a4 = r[4];
    o[0] = a0;
    o[1] = a1;
    o[2] = a2;
    o[3] = a3;
    o[4] = a4;
  }
}

// Pass 3 on records: one workgroup per (k1, k2) sub-partition; writes the 16-byte records and
// the rows in bin order, bin_start and the bins' height ranges (keys of the f32 offsets: the
// budget of a tile only needs the RANGE, and the offsets are what the gather sums).
template <int THREADS, int PER>
__device__ __forceinline__ void place_records(const uint32_t* __restrict__ src, const DsmParams& p,
                                              int cap, const uint32_t* __restrict__ start2,
                                              uint32_t* __restrict__ bin_start,
                                              uint4* __restrict__ rec16, uint32_t* __restrict__ sidx,
                                              uint2* __restrict__ bin_z, int sp, unsigned skip_lo,
                                              unsigned skip_hi) {
  extern __shared__ uint32_t s_words[];                 // kRecWords * cap
  uint32_t* s_bins = s_words + (size_t)kRecWords * cap; // p3_w
  uint32_t* s_scan = s_bins + p.p3_w;                   // 24
  uint32_t* s_zlo = s_scan + 24;                        // p3_w
  uint32_t* s_zhi = s_zlo + p.p3_w;                     // p3_w
  const int tid = threadIdx.x;
  const int k1 = sp / p.p3_n2, k2 = sp - k1 * p.p3_n2;
  const int rr = k2 / p.p3_c;
  const int row = k1 * p.p3_r1 + rr;
  const int bx0 = (k2 - rr * p.p3_c) * p.p3_w;
  const int nbw = min(p.p3_w, p.nbx - bx0);
  if (sp == 0 && tid == 0)
    bin_start[(size_t)p.nbx * p.nby] = start2[p.p3_n1 * p.p3_n2];
  if (row >= p.nby || nbw <= 0) return;  // no bins (and therefore no points)
  const uint32_t g0 = start2[sp], g1 = start2[sp + 1];
  for (int k = tid; k < nbw; k += THREADS) {
    s_bins[k] = 0;
    s_zlo[k] = 0xFFFFFFFFu;
    s_zhi[k] = 0u;
  }
  __syncthreads();
  if ((g1 - g0) > skip_lo && (g1 - g0) <= skip_hi) return;  // the other launch's
  const bool in_lds = (int)(g1 - g0) <= cap && cap <= THREADS * PER;
  auto zkey = [](uint32_t fbits) { return (fbits >> 31) ? ~fbits : (fbits | 0x80000000u); };
  uint32_t w[PER][kRecWords];
  int pb[PER];
  if (in_lds) {
    // (all loads first, branch-free: see k_dsm_p3_scatter_rec)
    if (g1 > g0) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const uint32_t ld = min(g0 + tid + (uint32_t)k * THREADS, g1 - 1);
#pragma unroll
        for (int t = 0; t < kRecWords; ++t) w[k][t] = src[(size_t)kRecWords * ld + t];
      }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t idx = g0 + tid + (uint32_t)k * THREADS;
      pb[k] = -1;
      if (idx < g1) {
        pb[k] = div_by((int)(w[k][0] & 0xFFFFu), p.B, p.mul_B) - bx0;
        atomicAdd(&s_bins[pb[k]], 1u);
        // (the bins' height ranges: read off the placed records below -- two more LDS atomics
        // per point here cost more than one pass of a thread per bin there)
      }
    }
  } else {
    for (uint32_t idx = g0 + tid; idx < g1; idx += THREADS) {
      const uint32_t w0 = src[(size_t)kRecWords * idx], w3 = src[(size_t)kRecWords * idx + 3];
      const int b = div_by((int)(w0 & 0xFFFFu), p.B, p.mul_B) - bx0;
      atomicAdd(&s_bins[b], 1u);
      const uint32_t zk = zkey(w3);
      atomicMin(&s_zlo[b], zk);
      atomicMax(&s_zhi[b], zk);
    }
  }
  __syncthreads();
  {
    const int per = (nbw + THREADS - 1) / THREADS;
    const int lo = tid * per;
    const int hi = min(lo + per, nbw);
    unsigned sum = 0;
    for (int k = lo; k < hi; ++k) sum += s_bins[k];
    unsigned total;
    unsigned run = block_excl_scan<THREADS>(sum, &total, s_scan);
    for (int k = lo; k < hi; ++k) {
      const unsigned t = s_bins[k];
      s_bins[k] = run;
      run += t;
    }
  }
  __syncthreads();
  uint32_t* out_start = bin_start + (size_t)row * p.nbx + bx0;
  uint2* zrow = bin_z + (size_t)row * p.nbx + bx0;
  for (int k = tid; k < nbw; k += THREADS) {
    out_start[k] = g0 + s_bins[k];
    if (!in_lds) zrow[k] = make_uint2(s_zlo[k], s_zhi[k]);
  }
  __syncthreads();
  if (!in_lds) {
    // over-full sub-partition (clustered cloud): second read, direct placement
    for (uint32_t idx = g0 + tid; idx < g1; idx += THREADS) {
      uint32_t v[kRecWords];
#pragma unroll
      for (int t = 0; t < kRecWords; ++t) v[t] = src[(size_t)kRecWords * idx + t];
      const int b = div_by((int)(v[0] & 0xFFFFu), p.B, p.mul_B) - bx0;
      const size_t o = (size_t)g0 + atomicAdd(&s_bins[b], 1u);
      rec16[o] = make_uint4(v[0], v[1], v[2], v[3]);
      sidx[o] = v[4];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (pb[k] >= 0) {
      const uint32_t q = atomicAdd(&s_bins[pb[k]], 1u);
#pragma unroll
      for (int t = 0; t < kRecWords; ++t) s_words[kRecWords * q + t] = w[k][t];
    }
  }
  __syncthreads();
  // (after the placement s_bins[b] is the END of bin b: a thread per bin walks its records)
  for (int b = tid; b < nbw; b += THREADS) {
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t q = b ? s_bins[b - 1] : 0u; q < s_bins[b]; ++q) {
      const uint32_t zk = zkey(s_words[kRecWords * q + 3]);
      lo = min(lo, zk);
      hi = max(hi, zk);
    }
    zrow[b] = make_uint2(lo, hi);
  }
  const uint32_t cnt = g1 - g0;
  for (uint32_t q = tid; q < cnt; q += THREADS) {
    rec16[(size_t)g0 + q] = make_uint4(s_words[kRecWords * q], s_words[kRecWords * q + 1],
                                      s_words[kRecWords * q + 2], s_words[kRecWords * q + 3]);
    sidx[(size_t)g0 + q] = s_words[kRecWords * q + 4];
  }
}

__global__ void __launch_bounds__(kP3PlaceThreads)
k_dsm_p3_place_rec(const uint32_t* __restrict__ src, DsmParams p, int cap,
                   const uint32_t* __restrict__ start2, uint32_t* __restrict__ bin_start,
                   uint4* __restrict__ rec16, uint32_t* __restrict__ sidx, uint2* __restrict__ bin_z,
                   unsigned skip_lo, unsigned skip_hi) {
  place_records<kP3PlaceThreads, kP3PlacePer>(src, p, cap, start2, bin_start, rec16, sidx, bin_z,
                                              (int)blockIdx.x, skip_lo, skip_hi);
}

__global__ void __launch_bounds__(kP3BigThreads)
k_dsm_p3_place_rec_big(const uint32_t* __restrict__ src, DsmParams p,
                       const uint32_t* __restrict__ start2, uint32_t* __restrict__ bin_start,
                       uint4* __restrict__ rec16, uint32_t* __restrict__ sidx,
                       uint2* __restrict__ bin_z, const uint32_t* __restrict__ big_list, int cap_rounds,
                       unsigned rounds_above, unsigned reg_max) {
  const unsigned count = big_list[0];
  for (unsigned k = blockIdx.x; k < count; k += gridDim.x) {
    const int sp = (int)big_list[1 + k];
    const uint32_t cnt = start2[sp + 1] - start2[sp];
    if (cnt > rounds_above && cnt <= min(reg_max, (unsigned)(kP3BigThreads * kP3RoundsPerRec)))
      place_rounds<kP3BigThreads, true, kP3RoundsPerRec>(src, p, cap_rounds, start2, bin_start, nullptr, rec16,
                                                         sidx, bin_z, sp);
    else if (cnt > rounds_above)
      place_rounds<kP3BigThreads, true, 0>(src, p, cap_rounds, start2, bin_start, nullptr, rec16, sidx, bin_z,
                                           sp);
    else
      place_records<kP3BigThreads, kP3BigPer>(src, p, kP3BigCap, start2, bin_start, rec16, sidx, bin_z, sp,
                                              0u, 0u);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// multi-GPU: compact the points other windows need (their halo)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_halo_select(const double* __restrict__ xyz, size_t n, HaloParams hp,
              double* __restrict__ out, unsigned long long* __restrict__ counts) {
  __shared__ HaloStage stage;
  halo_stage_init(&stage, hp);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += stride) {
    if (halo_candidate(hp, xyz[3 * idx + 0], xyz[3 * idx + 1]))
      halo_emit(&stage, hp, xyz, idx, out, counts);
  }
  halo_flush(&stage, hp, xyz, out, counts);
}

int halo_select_run(Ctx* c, const double* dev_xyz, size_t n, const HaloParams& hp,
                    double* dev_out, unsigned long long* dev_counts) {
  ScopedTimer t(c, AMHIP_K_HALO_SELECT);
  AMHIP_TRY(hipMemsetAsync(dev_counts, 0, sizeof(unsigned long long) * hp.nd, c->stream));
  if (n == 0) return AMHIP_OK;
  size_t grid = (n + 255) / 256;
  if (grid > 256 * 8) grid = 256 * 8;
  hipLaunchKernelGGL(k_halo_select, dim3((unsigned)grid), dim3(256), 0, c->stream,
                     dev_xyz, n, hp, dev_out, dev_counts);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

// ---------------------------------------------------------------------------
// bounding box, in cells of p's window, of the points point_bin() would bin (small clouds onto
// large maps: the call then runs on a sub-window, amhip_api.hip)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_dsm_bbox(const double* __restrict__ xyz, size_t n, DsmParams p, int* __restrict__ out5) {
  int lo_i = 0x7FFFFFFF, hi_i = -0x7FFFFFFF, lo_j = 0x7FFFFFFF, hi_j = -0x7FFFFFFF;
  unsigned cnt = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += stride) {
    const double px = xyz[3 * idx + 0] - p.sub_x;  // dsm.cc:42
    const double py = xyz[3 * idx + 1] - p.sub_y;  // dsm.cc:43
    // (point_bin's own test and rounding)
    const double cx = (p.base_x - px) * p.inv_res - (double)p.i_off;
    const double cy = (p.base_y - py) * p.inv_res - (double)p.j_off;
    const double lo = -(double)p.M - 0.5;
    const double hx = (double)(p.rows + p.M) - 0.5;
    const double hy = (double)(p.cols + p.M) - 0.5;
    if (!(cx >= lo && cx < hx && cy >= lo && cy < hy)) continue;
    const int ix = (int)floor(cx + 0.5), iy = (int)floor(cy + 0.5);
    lo_i = min(lo_i, ix);
    hi_i = max(hi_i, ix);
    lo_j = min(lo_j, iy);
    hi_j = max(hi_j, iy);
    ++cnt;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    lo_i = min(lo_i, __shfl_xor(lo_i, d, 64));
    hi_i = max(hi_i, __shfl_xor(hi_i, d, 64));
    lo_j = min(lo_j, __shfl_xor(lo_j, d, 64));
    hi_j = max(hi_j, __shfl_xor(hi_j, d, 64));
    cnt += __shfl_xor(cnt, d, 64);
  }
  // (one set of atomics per WORKGROUP, a hundred workgroups: thousands of atomics on the same five
  // words serialise in the L2 -- 0.26 ms for a 360 K-point cloud when every wave sent its own)
  __shared__ int s_box[4][5];
  if ((threadIdx.x & 63) == 0) {
    int* w = s_box[threadIdx.x >> 6];
    w[0] = lo_i;
    w[1] = hi_i;
    w[2] = lo_j;
    w[3] = hi_j;
    w[4] = (int)cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned total = 0;
    for (int k = 0; k < 4; ++k) {
      lo_i = min(lo_i, s_box[k][0]);
      hi_i = max(hi_i, s_box[k][1]);
      lo_j = min(lo_j, s_box[k][2]);
      hi_j = max(hi_j, s_box[k][3]);
      total += (unsigned)s_box[k][4];
    }
    if (total) {
      // (biased so that an all-zero buffer is the empty box: the reset is a memset, not a launch)
      atomicMax(&out5[0], kBboxBias - lo_i);
      atomicMax(&out5[1], hi_i + kBboxBias);
      atomicMax(&out5[2], kBboxBias - lo_j);
      atomicMax(&out5[3], hi_j + kBboxBias);
      atomicAdd(reinterpret_cast<unsigned*>(&out5[4]), total);
    }
  }
}

int dsm_bbox_run(Ctx* c, const double* dev_xyz, size_t n, const DsmParams& p, int* dev_bbox5) {
  ScopedTimer t(c, AMHIP_K_DSM_BIN_COUNT);
  AMHIP_TRY(hipMemsetAsync(dev_bbox5, 0, 5 * sizeof(int), c->stream));
  size_t grid = std::min<size_t>((n + 2047) / 2048, 128);
  hipLaunchKernelGGL(k_dsm_bbox, dim3((unsigned)grid), dim3(256), 0, c->stream, dev_xyz, n, p, dev_bbox5);
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
           uint32_t* __restrict__ total_out, SortAux aux) {
  __shared__ unsigned lds[1024 / 64 + 1];
  aux_reset(aux);
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


// tuning knob p3_rounds_cap=n (tests): sub-partitions above n points are placed in rounds over an
// image of n points -- exercises place_rounds (and its one-bin-beyond-the-image direct case)
// on clouds of test size; normally only contexts beyond ~130 M points get there
// tuning knob p3_rounds_reread: the rounds re-read the sub-partition instead of keeping it in registers
// (the path of sub-partitions beyond 14 K points)
static void p3_rounds_knob(int* cap_rounds, unsigned* rounds_above, unsigned* reg_max) {
  const int knob = (int)tuning("p3_rounds_cap", 0.0);
  const bool reread = tuning_on("p3_rounds_reread");
  if (knob >= 16 && knob < *cap_rounds) {
    *cap_rounds = knob;
    *rounds_above = (unsigned)knob;
  }
  if (reread) *reg_max = 0u;
}

// tuning knob no_launch_skips (tests, A-B): every capacity-class / big-list launch is made whatever the
// previous call's counters say (amhip_dsm.hip: dsm_run uses the same switch)
bool no_launch_skips() {
  return tuning_on("no_launch_skips");
}

// what a call's pinned launch-policy counters are counters OF: the window's geometry and the sort's plan
static unsigned long long sort_geometry_signature(const DsmParams& p) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](const void* v, size_t bytes) {
    const unsigned char* b = static_cast<const unsigned char*>(v);
    for (size_t k = 0; k < bytes; ++k) h = (h ^ b[k]) * 1099511628211ull;
  };
  const int ints[] = {p.rows, p.cols, p.M, p.B, p.nbx, p.nby, p.i_off, p.j_off, p.p3_r1, p.p3_c,
                      p.p3_w, p.p3_n1, p.p3_n2, p.p3_cap};
  const double dbls[] = {p.base_x, p.base_y, p.res, p.inv_res, p.sub_x, p.sub_y};
  mix(ints, sizeof(ints));
  mix(dbls, sizeof(dbls));
  return h ? h : 1ull;
}

// ---------------------------------------------------------------------------
// host driver: sort `n` points into c->sorted / c->bin_start
// ---------------------------------------------------------------------------

static int dsm_sort_impl(Ctx* c, const double* dev_xyz, const int32_t* dev_values, size_t n,
                         const DsmParams& p, unsigned long long* zrange, const SortSplit* split);

int dsm_sort(Ctx* c, const double* dev_xyz, const int32_t* dev_values, size_t n,
             const DsmParams& p, unsigned long long* zrange, const SortSplit* split) {
  const int rc = dsm_sort_impl(c, dev_xyz, dev_values, n, p, zrange, split);
  if (rc) return rc;
  // (every sort hands SortAux to one of its single-workgroup kernels)
  if (!c->aux_done && !(split && split->phase == 1)) return arg_failure("dsm_sort: the call's range was not reset");
  return AMHIP_OK;
}

static int dsm_sort_impl(Ctx* c, const double* dev_xyz, const int32_t* dev_values, size_t n,
                         const DsmParams& p, unsigned long long* zrange, const SortSplit* split) {
  // [min z, max z] of the binned points (for the mosaic's coarse cull): every
  // workgroup of the first scatter pass -- it loads z anyway -- writes a
  // partial, k_range_reduce folds them into *zrange
  // (always: the call's own range -- c->dev_zrange[2], [3] -- bounds max |z| for the gather's
  // rounding guard; `zrange`, the running range, may be null)
  double* zpart = nullptr;
  {
    const size_t max_waves = 8 * std::max<size_t>((n + 2047) / 2048 + 1, 256 * 16);
    const int rc = ensure_capacity(&c->zpart, &c->zpart_cap, 2 * max_waves + 16);
    if (rc) return rc;
    zpart = c->zpart;
  }
  // What the single-workgroup kernel in front of the first scatter resets (SortAux): the call's own
  // range and -- when dsm_run asked for it -- the counters of the gather's tile lists.
  // k_range_reduce (amhip_dsm.hip: the gather's prologue) folds zpart[0 .. range_parts) afterwards.
  const SortAux aux = {c->dev_zrange + 2, c->aux_zero_words, c->aux_zero_words ? c->aux_nzero : 0};
  c->aux_done = false;
  c->range_parts = 0;
  c->range_running = zrange;
  const size_t nbins = (size_t)p.nbx * (size_t)p.nby;
  const size_t nblocks_scan = (nbins + kScanE - 1) / kScanE;
  const bool force_one_level_ = tuning_on("sort_one_level");
  const unsigned long long geo_sig = sort_geometry_signature(p);   // (window geometry + sort plan)
  {
    int rc;
    if ((rc = ensure_capacity(&c->sorted, &c->sorted_cap, 3 * n))) return rc;
    if ((rc = ensure_capacity(&c->bin_start, &c->bin_cap, nbins + 4))) return rc;
  }
  c->last_num_bins = (int64_t)nbins;
  c->last_bin_cells = p.B;
  c->bin_z_valid = false;
  c->pts = PtsView{c->sorted, nullptr, nullptr, dev_xyz, nullptr, p.sub_x, p.sub_y};
  // the record pipeline (the single-precision gather's mode): 16-byte records + rows + zref
  const bool rec = p.fx_ok && !p.pcl_mode && !dev_values;
  if (rec) {
    int rc;
    if ((rc = ensure_capacity(&c->rec16, &c->rec16_cap, 4 * n + 16))) return rc;
    if ((rc = ensure_capacity(&c->sidx, &c->sidx_cap, n + 16))) return rc;
    if ((rc = ensure_capacity(&c->zref, &c->zref_cap, (size_t)8))) return rc;
  }

  const bool force_one_level = force_one_level_;
  const bool three_pass = p.p3_n1 > 0 && !force_one_level;
  if (split && split->phase == 1 && !three_pass)  // small clouds: selection in a pass of its own
    return halo_select_run(c, dev_xyz, split->n_prefix, split->hp, split->halo_out,
                           split->halo_counts);
  if (split && !three_pass) split = nullptr;
  if (three_pass) {
    // ---- three-pass partition sort ---------------------------------------------
    const int n1 = p.p3_n1, n2 = p.p3_n2, nk = n1 * n2;
    auto count_grid = [](size_t len) {
      return std::min<size_t>(std::max<size_t>((len + 8191) / 8192, 1), 256);
    };
    // (tiled call: the prefix is counted -- and its halo selected -- before the caller's
    // exchange, the received rows after it; each part writes its own histogram rows)
    const size_t n_a = split ? split->n_prefix : n;
    const size_t g_a = count_grid(n_a), g_b = n > n_a ? count_grid(n - n_a) : 0;
    const size_t gcount = g_a + g_b;
    int rc;
    if (rec) {
      if ((rc = ensure_capacity(&c->rec_a, &c->rec_a_cap, (size_t)kRecWords * n + 16))) return rc;
      if ((rc = ensure_capacity(&c->rec_b, &c->rec_b_cap, (size_t)kRecWords * n + 16))) return rc;
      if ((rc = ensure_capacity(&c->zall, &c->zall_cap, 2 * gcount * (kP3CountThreads / 64) + 16))) return rc;
    } else if ((rc = ensure_capacity(&c->tmp_points, &c->tmp_points_cap, 3 * n))) {
      return rc;
    }
    double* const zall = rec ? c->zall : nullptr;
    const size_t ws_words = gcount * (size_t)nk + 4 * (size_t)nk + 4 * (size_t)n1 + 64;
    if ((rc = ensure_capacity(&c->stripe_ws, &c->stripe_ws_cap, ws_words))) return rc;
    uint32_t* hist_rows = c->stripe_ws;
    uint32_t* cnt = hist_rows + gcount * (size_t)nk;
    uint32_t* start2 = cnt + nk;       // nk + 1
    uint32_t* cursor2 = start2 + nk + 1;
    uint32_t* start1 = cursor2 + nk;   // n1 + 1
    uint32_t* cursor1 = start1 + n1 + 1;
    uint32_t* blk2 = cursor1 + n1;     // n1 + 1
    uint32_t* big_list = blk2 + n1 + 1;  // [count] + up to nk sub-partition ids
    {
      ScopedTimer t(c, AMHIP_K_DSM_BIN_COUNT);
      const size_t lds = (size_t)nk * sizeof(uint32_t);
      const size_t lds_halo = (size_t)((nk + 1) & ~1) * sizeof(uint32_t) + sizeof(HaloStage);
      AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_count<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_count<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_halo));
      const HaloParams no_halo = {};
      if (split && split->phase == 1) {
        AMHIP_TRY(hipMemsetAsync(split->halo_counts, 0,
                                 sizeof(unsigned long long) * split->hp.nd, c->stream));
        hipLaunchKernelGGL(k_dsm_p3_count<true>, dim3((unsigned)g_a), dim3(kP3CountThreads),
                           lds_halo, c->stream, dev_xyz, n_a, p, hist_rows, split->hp, split->halo_out,
                           split->halo_counts, zall);
        AMHIP_TRY(hipGetLastError());
        return AMHIP_OK;  // amhip_dsm_tiled_finish_dev comes back with phase 2
      }
      if (!split)
        hipLaunchKernelGGL(k_dsm_p3_count<false>, dim3((unsigned)g_a), dim3(kP3CountThreads), lds,
                           c->stream, dev_xyz, n_a, p, hist_rows, no_halo, (double*)nullptr,
                           (unsigned long long*)nullptr, zall);
      else if (g_b)
        hipLaunchKernelGGL(k_dsm_p3_count<false>, dim3((unsigned)g_b), dim3(kP3CountThreads), lds,
                           c->stream, dev_xyz + 3 * n_a, n - n_a, p, hist_rows + g_a * (size_t)nk,
                           no_halo, (double*)nullptr, (unsigned long long*)nullptr,
                           zall ? zall + 2 * g_a * (kP3CountThreads / 64) : nullptr);
      // (the records' reference height: the middle of the range every count workgroup left --
      // in a tiled call both parts', the second of which may be absent: its rows hold the first
      // call's partials or the initial "empty" pairs)
      if (rec)
        hipLaunchKernelGGL(k_dsm_zref, dim3(1), dim3(1024), 0, c->stream, zall,
                           (split && !g_b ? g_a : gcount) * (size_t)(kP3CountThreads / 64), c->zref);
      // (reduction of the count workgroups' rows + the scan by the workgroup that finishes last: one launch)
      hipLaunchKernelGGL(k_dsm_p3_reduce_scan, dim3((unsigned)((nk + 63) / 64)), dim3(1024), 0, c->stream,
                         hist_rows, (int)gcount, cnt, n1, n2, start2, cursor2, start1, cursor1, blk2,
                         (unsigned)p.p3_cap, (unsigned)kP3BigCap, big_list, (unsigned)(rec ? kRecChunk : kP3Chunk),
                         aux, (unsigned*)nullptr, c->dev_tickets);
      c->aux_done = true;
      AMHIP_TRY(hipGetLastError());
    }
    if (rec) {
      {
        ScopedTimer t(c, AMHIP_K_DSM_SCATTER);
        const size_t lds = (size_t)kRecChunk * (kRecWords + 1) * 4 + (3 * kP3MaxKeys + 32) * sizeof(uint32_t);
        AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_scatter_rec<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_scatter_rec<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const size_t g1 = (n + kRecChunk - 1) / kRecChunk;
        hipLaunchKernelGGL(k_dsm_p3_scatter_rec<true>, dim3((unsigned)g1), dim3(kP3Threads), lds,
                           c->stream, dev_xyz, (const uint32_t*)nullptr, n, p, c->zref, start1, blk2,
                           cursor1, c->rec_a, zpart);
        c->range_parts = (size_t)g1 * (kP3Threads / 64);
        hipLaunchKernelGGL(k_dsm_p3_scatter_rec<false>, dim3((unsigned)(g1 + n1)), dim3(kP3Threads),
                           lds, c->stream, (const double*)nullptr, c->rec_a, n, p, c->zref, start1, blk2,
                           cursor2, c->rec_b, (double*)nullptr);
        AMHIP_TRY(hipGetLastError());
      }
      {
        ScopedTimer t(c, AMHIP_K_DSM_SCAN);
        if ((rc = ensure_capacity(&c->bin_z, &c->bin_z_cap, 2 * (nbins + 4)))) return rc;
        uint2* bin_z = reinterpret_cast<uint2*>(c->bin_z);
        uint4* rec16 = reinterpret_cast<uint4*>(c->rec16);
        const size_t lds = (size_t)p.p3_cap * kRecWords * 4 + (3 * (size_t)p.p3_w + 32) * sizeof(uint32_t);
        AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_place_rec),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_dsm_p3_place_rec, dim3((unsigned)nk), dim3(kP3PlaceThreads), lds,
                           c->stream, c->rec_b, p, p.p3_cap, start2, c->bin_start, rec16, c->sidx, bin_z,
                           (unsigned)p.p3_cap, 0xFFFFFFFFu);
        // (sub-partitions beyond one image: rounds over an image of cap_rounds points next to
        // the rounds' tables -- place_rounds)
        const size_t tables = (4 * (size_t)p.p3_w + 64) * sizeof(uint32_t);
        int cap_rounds = (int)std::min<size_t>(kP3BigCap, (kLdsMaxBytes - tables) / (kRecWords * 4));
        unsigned rounds_above = kP3BigCap, reg_max = 0xFFFFFFFFu;
        p3_rounds_knob(&cap_rounds, &rounds_above, &reg_max);
        const size_t lds_big = std::max((size_t)kP3BigCap * kRecWords * 4 + (3 * (size_t)p.p3_w + 32) * sizeof(uint32_t),
                                        (size_t)cap_rounds * kRecWords * 4 + tables);
        AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_place_rec_big),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
        hipLaunchKernelGGL(k_dsm_p3_place_rec_big, dim3(256), dim3(kP3BigThreads), lds_big, c->stream,
                           c->rec_b, p, start2, c->bin_start, rec16, c->sidx, bin_z, big_list, cap_rounds, rounds_above, reg_max);
        AMHIP_TRY(hipGetLastError());
        c->bin_z_valid = true;
        c->pts = PtsView{nullptr, rec16, c->sidx, dev_xyz, c->zref, p.sub_x, p.sub_y};
      }
      return AMHIP_OK;
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
                         c->stream, dev_xyz, dev_values, n, p, start1, blk2, cursor1, c->sorted,
                         zpart);
      c->range_parts = (size_t)g1 * (kP3Threads / 64);
      hipLaunchKernelGGL(k_dsm_p3_scatter<false>, dim3((unsigned)(g1 + n1)), dim3(kP3Threads),
                         lds, c->stream, c->sorted, (const int32_t*)nullptr, n, p, start1, blk2,
                         cursor2, c->tmp_points, (double*)nullptr);
      AMHIP_TRY(hipGetLastError());
    }
    {
      ScopedTimer t(c, AMHIP_K_DSM_SCAN);
      // single-precision gather: the placement pass also leaves every bin's height range
      // (the occupancy pre-pass sorts tiles without room under the error bound onto the FP64
      // lists before anything is staged); bins no sub-partition covers keep "empty"
      uint2* bin_z = nullptr;  // (the record pipeline returned above; the doubles pipeline needs none)
      const size_t zlds = bin_z ? 2 * (size_t)p.p3_w * sizeof(uint32_t) : 0;
      const size_t lds = (size_t)p.p3_cap * 24 + ((size_t)p.p3_w + 32) * sizeof(uint32_t) + zlds;
      AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_place),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      // Sub-partitions beyond the placement workgroup's registers (denser parts of a non-uniform
      // cloud) are k_dsm_p3_place_big's.  Its launch is skipped when the PREVIOUS three-pass call of
      // this geometry had none (the scan leaves their number in a pinned word, never waited for):
      // k_dsm_p3_place then keeps nothing back and places a stray over-full sub-partition itself,
      // directly (slower, same result) -- the word it leaves brings the big kernel back next call.
      const bool want_big = !(c->sort_stats_sig == geo_sig && c->host_sort_stats && c->host_sort_stats[0] == 0u) ||
                            no_launch_skips();
      c->sort_stats_sig = geo_sig;
      hipLaunchKernelGGL(k_dsm_p3_place, dim3((unsigned)nk), dim3(kP3PlaceThreads), lds,
                         c->stream, c->tmp_points, p, p.p3_cap, start2, c->bin_start, c->sorted,
                         want_big ? (unsigned)p.p3_cap : 0xFFFFFFFFu, 0xFFFFFFFFu, bin_z,
                         (const uint32_t*)big_list, c->host_sort_stats);
      const size_t tables = (4 * (size_t)p.p3_w + 64) * sizeof(uint32_t);
      int cap_rounds = (int)std::min<size_t>(kP3BigCap, (kLdsMaxBytes - tables) / 24);
      unsigned rounds_above = kP3BigCap, reg_max = 0xFFFFFFFFu;
      p3_rounds_knob(&cap_rounds, &rounds_above, &reg_max);
      const size_t lds_big = std::max((size_t)kP3BigCap * 24 + ((size_t)p.p3_w + 32) * sizeof(uint32_t) + zlds,
                                      (size_t)cap_rounds * 24 + tables);
      AMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dsm_p3_place_big),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
      if (want_big)
        hipLaunchKernelGGL(k_dsm_p3_place_big, dim3(256), dim3(kP3BigThreads), lds_big, c->stream,
                           c->tmp_points, p, start2, c->bin_start, c->sorted, big_list, bin_z, cap_rounds, rounds_above, reg_max);
      c->bin_z_valid = bin_z != nullptr;
      AMHIP_TRY(hipGetLastError());
    }
  } else {
    // ---- one-level counting sort (clouds below the partition sort's threshold) --------
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
      double* zall = nullptr;
      if (rec) {
        int rc;
        if ((rc = ensure_capacity(&c->zall, &c->zall_cap, 2 * grid_pts * 4 + 16))) return rc;
        zall = c->zall;
      }
      hipLaunchKernelGGL(k_dsm_bin_count, dim3((unsigned)grid_pts), dim3(block), 0, c->stream,
                         dev_xyz, n, p, c->bin_start, c->rank, zall);
      if (rec)
        hipLaunchKernelGGL(k_dsm_zref, dim3(1), dim3(1024), 0, c->stream, zall, grid_pts * 4, c->zref);
      AMHIP_TRY(hipGetLastError());
    }
    {
      ScopedTimer t(c, AMHIP_K_DSM_SCAN);
      hipLaunchKernelGGL(k_scan_partials, dim3((unsigned)nblocks_scan), dim3(kScanT), 0,
                         c->stream, c->bin_start, nbins, c->scan_partials);
      hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, c->scan_partials,
                         nblocks_scan, c->bin_start + nbins, aux);
      c->aux_done = true;
      hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nblocks_scan), dim3(kScanT), 0, c->stream,
                         c->bin_start, nbins, c->scan_partials);
      AMHIP_TRY(hipGetLastError());
    }
    {
      ScopedTimer t(c, AMHIP_K_DSM_SCATTER);
      hipLaunchKernelGGL(k_dsm_scatter, dim3((unsigned)grid_pts), dim3(block), 0, c->stream,
                         dev_xyz, dev_values, n, p, c->bin_start, c->rank, c->sorted, zpart,
                         rec ? reinterpret_cast<uint4*>(c->rec16) : (uint4*)nullptr,
                         rec ? c->sidx : (uint32_t*)nullptr, rec ? c->zref : (const double*)nullptr);
      if (rec)
        c->pts = PtsView{c->sorted, reinterpret_cast<const uint4*>(c->rec16), c->sidx, dev_xyz, c->zref,
                         p.sub_x, p.sub_y};
      c->range_parts = (size_t)grid_pts * 4;
      AMHIP_TRY(hipGetLastError());
    }
  }
  return AMHIP_OK;
}

}  // namespace amhip
