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
// The sort lives in amhip_sort.hip (dsm_sort); this file holds the gather:
//   k_dsm_gather_tiled   one workgroup per 64 x 16 (or 64 x 32) cells: stage the
//                        tile's points in LDS once, re-bin them at cell
//                        granularity, per-lane radius search + division-free
//                        IDW; fallback ladder / overflow / numerically extreme
//                        cells go to
//   cell_global / cell_fallback_global / k_dsm_gather
//                        the same search on the global bins (also used for grids
//                        finer than 16 cells per radius and the adaptive
//                        OrthoFromPcl passes)
// (integer / FP64 streaming work, no MFMA)
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "amhip_common.h"
#include "amhip_device.h"

#ifndef AMHIP_GATHER_UNROLL
#define AMHIP_GATHER_UNROLL 2
#endif

namespace amhip {

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
// gather
// ---------------------------------------------------------------------------
// The binned cloud as the gather's routines see it (amhip_common.h: PtsView).  Two pipelines
// fill it (amhip_sort.hip):
//   doubles    sorted != null: (px, py, z) of sorted point g at sorted[3 g ..], centre offsets of
//              dsm.cc:42-43 applied -- the FP64 mode, OrthoFromPcl, small clouds;
//   records    rec != null: 16-byte records of the single-precision gather (cell, fixed-point
//              offsets from the cell centre, f32 height offset from zref[0]); the reference's
//              doubles of sorted point g are then those of row sidx[g] of the caller's UNTOUCHED
//              cloud -- fetched only by the routines that redo a cell or a tile in FP64 (a few
//              per tile).  Small clouds (one-level sort) carry both.
using Pts = PtsView;
__device__ __forceinline__ double pts_x(const Pts& P, size_t g) {
  return P.sorted ? P.sorted[3 * g + 0] : P.cloud[3 * (size_t)P.sidx[g] + 0] - P.sub_x;
}
__device__ __forceinline__ double pts_y(const Pts& P, size_t g) {
  return P.sorted ? P.sorted[3 * g + 1] : P.cloud[3 * (size_t)P.sidx[g] + 1] - P.sub_y;
}
__device__ __forceinline__ double pts_z(const Pts& P, size_t g) {
  return P.sorted ? P.sorted[3 * g + 2] : P.cloud[3 * (size_t)P.sidx[g] + 2];
}

struct Accum {
  double num, den;
  unsigned cnt;
  bool exact;
  double exact_z;  // value of (one of) the point(s) with d2 == 0
};

// ---------------------------------------------------------------------------
// Run-to-run reproducibility of the stored floats (round 4)
// ---------------------------------------------------------------------------
// The order in which a cell's neighbours are summed is not fixed: the binning's atomics leave
// the points of a bin -- and the staging's LDS atomics the points of a cell -- in whatever order
// the hardware served them.  The double sums then differ in their last bits from run to run, and
// a quotient that lies on a float rounding boundary rounds either way (observed: 1 cell in 1e8).
// The reference's kd-tree order is fixed (nanoflann.hpp:929-946), its floats are the same every
// run.  So are these, without sorting anything: every routine bounds the distance between its
// quotient and the EXACT inverse-distance average of the same doubles,
//     |h - h_exact| <= (2 n + 3) 2^-53 max|z|      (a term passes n + 1 roundings on its way into
//                                                   either sum; the weights are all positive;
//                                                   max|z| over the CALL's binned points: the
//                                                   sort leaves their range, call_zmax()),
// -- (4 n + 8) 2^-53 max|z| is used: it also covers canonical_search()'s own 2 x 2^-53 --
// and stores (float)h only when every value within that distance rounds to the same float.
// The rare cell that fails the test (a few per 1e8) is redone by canonical_search(): the
// reference's own terms fl(z/d2) and fl(1/d2) -- true divisions -- summed in double-double,
// where the order of the additions does not reach the 53rd bit, let alone the 24th.  Any two
// runs therefore store the same float in every cell.
__device__ __forceinline__ bool round_is_certain(double h, double err) {
  return (float)(h - err) == (float)(h + err);
}
// (p.canon_all -- tuning knob dsm_canon_all, tests: every quotient counts as uncertain, so every cell
// with a hit is stored by canonical_search())
__device__ __forceinline__ double idw_err_bound(const DsmParams& p, unsigned n, double zmax) {
  return p.canon_all ? __builtin_huge_val() : ((double)(4u * n + 8u) * 0x1p-53) * zmax;
}

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
                                            const Pts P,
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
      const double px = pts_x(P, (size_t)k);
      const double py = pts_y(P, (size_t)k);
      // L2_Adaptor with size == 2 (nanoflann.hpp:319-322): 0 + dx*dx, + dy*dy
      const double dx = qx - px;
      const double dy = qy - py;
      double d2 = dx * dx;
      d2 = d2 + dy * dy;
      if (MODE == 0) {
        if (d2 < T) {  // RadiusResultSet::addPoint, strict (nanoflann.hpp:157)
          if (d2 > 0.0) {
            idw_add(d2, pts_z(P, (size_t)k), &acc->num, &acc->den);
          } else {
            acc->exact = true;  // dsm.cc:165 CHECK / ortho-from-pcl.cc:91-96
            acc->exact_z = pts_z(P, (size_t)k);
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
  // [min, max] (ordered keys) of the heights / values of the points this call binned: the sort's
  // first scatter pass leaves them (c->dev_zrange + 2).  max |z| of round_is_certain's bound.
  const unsigned long long* __restrict__ zcall;
};

// max |z| over the call's binned points (0 for an empty call); wave-uniform, two cached loads
__device__ __forceinline__ double call_zmax(const CellOut& o) {
  const double lo = from_ordered_key(o.zcall[0]), hi = from_ordered_key(o.zcall[1]);
  return lo <= hi ? fmax(fabs(lo), fabs(hi)) : 0.0;
}

// cell (i, j) of the call's window in the layer (and mask) it writes: DsmParams::out_*
__device__ __forceinline__ size_t cell_at(const DsmParams& p, int i, int j) {
  return (size_t)(p.out_i0 + i) + (size_t)(p.out_j0 + j) * (size_t)p.out_pitch;
}

__device__ __forceinline__ void leave_untouched(const DsmParams& p, const CellOut& o, int i,
                                                int j) {
  if (o.fill_untouched) o.layer[cell_at(p, i, j)] = o.init_value;
}

__device__ __forceinline__ void emit_value(const DsmParams& p, const CellOut& o, int i, int j,
                                           double v) {
  const size_t at = cell_at(p, i, j);
  o.layer[at] = (float)v;
  if (o.mask) o.mask[at] = 1;
}

// s + e == a + b exactly (Knuth)
__device__ __forceinline__ void two_sum(double a, double b, double* s, double* e) {
  const double t = a + b;
  const double bb = t - a;
  *e = (a - (t - bb)) + (b - bb);
  *s = t;
}

// One search (window half-width w cells, squared radius T) of cell (i, j) with at least one hit,
// in the arithmetic every summation order agrees on (see round_is_certain): the reference's
// terms z / d2 and 1 / d2 (dsm.cc:166-168, true divisions) summed in double-double, one
// division of the two sums, stored as float.  One lane per cell on the global bins; a few cells
// per 1e8 come here.
__device__ __forceinline__ void canonical_search(const DsmParams& p, const uint32_t* __restrict__ start,
                                              const Pts P, double qx, double qy, int i, int j, int w,
                                              double T, const CellOut& o) {
  const int bx0 = (i - w + p.M) / p.B, bx1 = (i + w + p.M) / p.B;
  const int by0 = (j - w + p.M) / p.B, by1 = (j + w + p.M) / p.B;
  double nh = 0.0, nl = 0.0, dh = 0.0, dl = 0.0;
  bool exact = false;
  double exact_z = 0.0;
  for (int by = by0; by <= by1; ++by) {
    const uint32_t* row = start + (size_t)by * p.nbx;
    const uint32_t e = row[bx1 + 1];
    for (uint32_t k = row[bx0]; k < e; ++k) {
      const double dx = qx - pts_x(P, (size_t)k);
      const double dy = qy - pts_y(P, (size_t)k);
      double d2 = dx * dx;
      d2 = d2 + dy * dy;  // L2_Adaptor (nanoflann.hpp:319-322)
      if (!(d2 < T)) continue;
      const double z = pts_z(P, (size_t)k);
      if (!(d2 > 0.0)) {
        exact = true;
        exact_z = z;
        continue;
      }
      double t, err;
      two_sum(nh, z / d2, &t, &err);
      nh = t;
      nl += err;
      two_sum(dh, 1.0 / d2, &t, &err);
      dh = t;
      dl += err;
    }
  }
  if (exact) {
    if (p.pcl_mode)
      emit_value(p, o, i, j, exact_z);       // ortho-from-pcl.cc:91-96 perfect match
    else {
      atomicOr(o.dev_err, kDevErrExactHit);  // dsm.cc:165 CHECK(distances[i] > 0.0)
      leave_untouched(p, o, i, j);
    }
    return;
  }
  if (!(dh > 0.0)) {  // (never: the caller saw hits)
    leave_untouched(p, o, i, j);
    return;
  }
  // (nh + nl) / (dh + dl): quotient of the leading parts, corrected by the exact remainder
  const double q = nh / dh;
  const double r = (fma(-q, dh, nh) + nl) - q * dl;
  emit_value(p, o, i, j, q + r / dh);
}

// Turns an accumulated search into the cell's value.  0: no hit (the cell is not done), 1: done,
// 2: the quotient is too close to a float rounding boundary to be the same in every summation
// order -- the caller redoes that search with canonical_search().
template <bool kCanon = true>
__device__ __forceinline__ int finish_accum(const DsmParams& p, const CellOut& o, int i, int j,
                                            const Accum& acc) {
  if (acc.exact) {
    if (p.pcl_mode)
      emit_value(p, o, i, j, acc.exact_z);  // ortho-from-pcl.cc:91-96 perfect match
    else {
      atomicOr(o.dev_err, kDevErrExactHit);  // dsm.cc:165 CHECK(distances[i] > 0.0)
      leave_untouched(p, o, i, j);           // (the reference aborts; never uninitialised memory)
    }
    return 1;
  }
  if (acc.cnt > 0) {
    const double h = acc.num / acc.den;
    if (kCanon && !round_is_certain(h, idw_err_bound(p, acc.cnt, call_zmax(o)))) return 2;
    emit_value(p, o, i, j, h);
    return 1;
  }
  return 0;
}

// Expanding-radius fallback for a cell whose first search (T[0]) was empty
// (dsm.cc:133-144): the reference retries with T[1], T[2], ... until a search
// returns something.  Equivalent: find the nearest point within the LAST
// radius, pick the first level whose threshold exceeds its d2, gather with
// that threshold.  Works on the global bin structure.
// kCanon = false (the single-precision mode's kernels: their floats move with the summation
// order anyway): an ambiguous quotient is stored as it is, no canonical_search in the kernel
template <bool kCanon = true>
__device__ __forceinline__ bool cell_fallback_global(const DsmParams& p,
                                                     const uint32_t* __restrict__ start,
                                                     const Pts P, int i,
                                                     int j, double qx, double qy,
                                                     const CellOut& o) {
  if (p.nlevels <= 1) return false;
  Accum acc = {0.0, 0.0, 0u, false, 0.0};
  const int last = p.nlevels - 1;
  double dmin = __builtin_huge_val();
  scan_window<1>(p, start, P, qx, qy, i, j, p.w[last], 0.0, &acc, &dmin);
  int level = -1;
  for (int k = 1; k <= last; ++k) {
    if (dmin < p.T[k]) {
      level = k;
      break;
    }
  }
  if (level < 0) return false;  // nothing within the last radius: cell untouched
  scan_window<0>(p, start, P, qx, qy, i, j, p.w[level], p.T[level], &acc, &dmin);
  const int fin = finish_accum<kCanon>(p, o, i, j, acc);
  if (kCanon && fin == 2) canonical_search(p, start, P, qx, qy, i, j, p.w[level], p.T[level], o);
  return fin != 0;
}

// ---- OPTIONAL capped mode (amhip_ctx_set_dsm_knn; not a reference code path) ----------
// The k nearest points of the search result, kept sorted by ascending d2 in registers
// (nanoflann::KNNResultSet::addPoint, nanoflann.hpp:100-125: a later arrival never
// displaces an equal distance).
constexpr int kMaxKnn = 8;
template <int K>
struct KnnSetT {
  double d2[K], z[K];
  double worst;   // d2 of the k-th entry once the set is full, +inf before: what a candidate has to beat
  int n;          // (a scalar of its own: s->d2[k - 1] with a run-time k would move the arrays to scratch memory)
};
typedef KnnSetT<kMaxKnn> KnnSet;

// (K: the registers the set takes, k <= K the cap in force)
template <int K>
__device__ __forceinline__ void knn_add(KnnSetT<K>* s, int k, double d2, double z) {
  if (!(d2 < s->worst)) return;
  // shift the strictly greater entries up (static indices only: the set lives in registers)
  double cd = d2, cz = z;
  bool inserted = false;  // from the insertion point on every entry moves up by one
#pragma unroll
  for (int q = 0; q < K; ++q) {
    if (q < k) {
      const bool here = inserted || q >= s->n || cd < s->d2[q];
      const double od = s->d2[q], oz = s->z[q];
      if (here) {
        s->d2[q] = cd;
        s->z[q] = cz;
        cd = od;
        cz = oz;
        inserted = true;
      }
    }
  }
  if (s->n < k) ++s->n;
  if (s->n == k) {
#pragma unroll
    for (int q = 0; q < K; ++q)
      if (q == k - 1) s->worst = s->d2[q];
  }
}

template <int K>
__device__ __forceinline__ void knn_init(KnnSetT<K>* s) {
  s->n = 0;
  s->worst = __builtin_huge_val();
#pragma unroll
  for (int q = 0; q < K; ++q) s->d2[q] = s->z[q] = 0.0;
}

__device__ __forceinline__ void knn_scan(const DsmParams& p, const uint32_t* __restrict__ start,
                                         const Pts P, double qx, double qy,
                                         int i, int j, int w, double T, KnnSet* s) {
  const int bx0 = (i - w + p.M) / p.B, bx1 = (i + w + p.M) / p.B;
  const int by0 = (j - w + p.M) / p.B, by1 = (j + w + p.M) / p.B;
  for (int by = by0; by <= by1; ++by) {
    const uint32_t* row = start + (size_t)by * p.nbx;
    const uint32_t e = row[bx1 + 1];
    for (uint32_t k = row[bx0]; k < e; ++k) {
      const double dx = qx - pts_x(P, (size_t)k);
      const double dy = qy - pts_y(P, (size_t)k);
      double d2 = dx * dx;
      d2 = d2 + dy * dy;
      if (d2 < T) knn_add(s, p.knn_k, d2, pts_z(P, (size_t)k));
    }
  }
}

__device__ __forceinline__ void cell_global_knn(const DsmParams& p,
                                                const uint32_t* __restrict__ start,
                                                const Pts P, int i, int j,
                                                const CellOut& o) {
  const double qx = p.base_x + p.res * (-(double)(i + p.i_off));
  const double qy = p.base_y + p.res * (-(double)(j + p.j_off));
  KnnSet s;
  knn_init(&s);
  knn_scan(p, start, P, qx, qy, i, j, p.w[0], p.T[0], &s);
  if (s.n == 0 && p.nlevels > 1) {  // the ladder of dsm.cc:133-144, as in cell_fallback_global
    Accum acc = {0.0, 0.0, 0u, false, 0.0};
    const int last = p.nlevels - 1;
    double dmin = __builtin_huge_val();
    scan_window<1>(p, start, P, qx, qy, i, j, p.w[last], 0.0, &acc, &dmin);
    for (int k = 1; k <= last; ++k)
      if (dmin < p.T[k]) {
        knn_scan(p, start, P, qx, qy, i, j, p.w[k], p.T[k], &s);
        break;
      }
  }
  if (s.n == 0) {
    leave_untouched(p, o, i, j);
    return;
  }
  // the oracle's arithmetic: true divisions, ascending distance
  double num = 0.0, den = 0.0;
  bool exact = false;
#pragma unroll
  for (int q = 0; q < kMaxKnn; ++q)
    if (q < s.n) {
      if (!(s.d2[q] > 0.0)) exact = true;
      num += s.z[q] / s.d2[q];
      den += 1.0 / s.d2[q];
    }
  if (exact) {
    atomicOr(o.dev_err, kDevErrExactHit);  // dsm.cc:165 CHECK(distances[i] > 0.0)
    leave_untouched(p, o, i, j);
    return;
  }
  emit_value(p, o, i, j, num / den);
}

// Whole cell through the global bins (first level + fallback).
template <bool kCanon = true>
__device__ __forceinline__ void cell_global(const DsmParams& p,
                                            const uint32_t* __restrict__ start,
                                            const Pts P, int i, int j,
                                            const CellOut& o) {
  if (p.only_unfilled && o.mask[cell_at(p, i, j)]) return;
  // grid_map_core getPosition (oracle/amo_compat.h cell_position)
  const double qx = p.base_x + p.res * (-(double)(i + p.i_off));
  const double qy = p.base_y + p.res * (-(double)(j + p.j_off));
  Accum acc = {0.0, 0.0, 0u, false, 0.0};
  double dmin = 0.0;
  scan_window<0>(p, start, P, qx, qy, i, j, p.w[0], p.T[0], &acc, &dmin);
  const int fin = finish_accum<kCanon>(p, o, i, j, acc);
  if (kCanon && fin == 2) canonical_search(p, start, P, qx, qy, i, j, p.w[0], p.T[0], o);
  bool done = fin != 0;
  if (!done) done = cell_fallback_global<kCanon>(p, start, P, i, j, qx, qy, o);
  if (!done) {
    leave_untouched(p, o, i, j);
    if (o.unfilled) atomicAdd(o.unfilled, 1u);
  }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// A block of up to 4 x 4 cells per WAVE, the lanes over the candidates: for tiles
// denser than any LDS image can hold (dense stereo clouds: tens of points per
// cell, hundreds of neighbours per search disc).  The cells of a block lie in
// one bin, so they share the window of bins; the window's bin rows are
// contiguous spans of the sorted cloud, read coalesced -- every lane loads a
// candidate once and tests it against all 16 cells, keeping partial
// inverse-distance sums per cell; one butterfly reduction per block.  Same
// neighbour sets as cell_global(), the order of the double sums differs (as it
// does from the kd-tree's anyway).  Must be called by all 64 lanes of a wave
// with the same block; bi0..bi1 x bj0..bj1 inclusive, <= 4 each way.
__device__ __forceinline__ void block_wave(const DsmParams& p, const uint32_t* __restrict__ start,
                                           const Pts P, int bi0, int bi1,
                                           int bj0, int bj1, const CellOut& o) {
  const int lane = threadIdx.x & 63;
  const int w = p.w[0];
  const double T = p.T[0];
  double qx[4], qy[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    // (cells beyond the block repeat its last one: computed, never written)
    qx[a] = p.base_x + p.res * (-(double)(min(bi0 + a, bi1) + p.i_off));
    qy[a] = p.base_y + p.res * (-(double)(min(bj0 + a, bj1) + p.j_off));
  }
  const int bx0 = (bi0 - w + p.M) / p.B, bx1 = (bi1 + w + p.M) / p.B;
  const int by0 = (bj0 - w + p.M) / p.B, by1 = (bj1 + w + p.M) / p.B;
  double num[16], den[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) num[c] = den[c] = 0.0;
  unsigned exact = 0;  // bit c: some neighbour of cell c at distance 0
  unsigned ncand = 0;  // candidates of the wave (round_is_certain's n)
  for (int by = by0; by <= by1; ++by) {
    const uint32_t* row = start + (size_t)by * p.nbx;
    const uint32_t s0 = row[bx0], e0 = row[bx1 + 1];
    ncand += e0 - s0;
    // (the next candidate is on its way while this one is worked on: ~160 FP64 instructions per
    // candidate against a memory round trip, three waves per SIMD to cover it)
    double nx = 0.0, ny = 0.0, nz = 0.0;
    if (s0 + lane < e0) {
      nx = pts_x(P, (size_t)(s0 + lane));
      ny = pts_y(P, (size_t)(s0 + lane));
      nz = pts_z(P, (size_t)(s0 + lane));
    }
    for (uint32_t k = s0 + lane; k < e0; k += 64) {
      const double px = nx, py = ny, pz = nz;
      if (k + 64 < e0) {
        nx = pts_x(P, (size_t)k + 64);
        ny = pts_y(P, (size_t)k + 64);
        nz = pts_z(P, (size_t)k + 64);
      }
      double dx2[4], dy2[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const double dx = qx[a] - px, dy = qy[a] - py;
        dx2[a] = dx * dx;
        dy2[a] = dy * dy;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double d2 = dx2[c & 3] + dy2[c >> 2];  // L2_Adaptor (nanoflann.hpp:319-322)
        const bool in = d2 < T;                      // strict (nanoflann.hpp:157)
        // 1 / d2 for every candidate (no branch per cell), kept only for the hits
        double r = __builtin_amdgcn_rcp(d2);
        double e = fma(-d2, r, 1.0);
        r = fma(r, e, r);
        e = fma(-d2, r, 1.0);
        r = fma(r, e, r);
        const bool hit = in & (d2 > 0.0);
        r = hit ? r : 0.0;
        num[c] = fma(pz, r, num[c]);
        den[c] += r;
        exact |= (in & !hit) ? (1u << c) : 0u;
      }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) exact |= __shfl_xor(exact, d, 64);
  // Butterfly with halving: 32 partial sums per lane (16 x num, 16 x den) -> after
  // the exchange over lane bit 5 a lane keeps 16 of them, then 8, 4, 2, 1; the
  // last exchange over bit 0 completes the sums.  Lane l ends with the total of
  // value l >> 1 (num of cell c at lane 2c, den at lane 32 + 2c).
  double v[32];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    v[c] = num[c];
    v[16 + c] = den[c];
  }
#pragma unroll
  for (int half = 16, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
    const bool up = (lane & bit) != 0;
#pragma unroll
    for (int k = 0; k < half; ++k) {
      const double send = up ? v[k] : v[k + half];
      const double keep = up ? v[k + half] : v[k];
      v[k] = keep + __shfl_xor(send, bit, 64);
    }
  }
  v[0] += __shfl_xor(v[0], 1, 64);
  const double my_num = v[0];
  const double my_den = __shfl(v[0], (lane & 31) | 32, 64);
  const int cidx = (lane & 31) >> 1;
  const int a = cidx & 3, bq = cidx >> 2;
  if (lane >= 32 || (lane & 1) || bi0 + a > bi1 || bj0 + bq > bj1) return;
  const int i = bi0 + a, j = bj0 + bq;
  if (p.only_unfilled && o.mask[cell_at(p, i, j)]) return;
  if (((exact >> cidx) & 1u) || !(my_den > 0.0)) {
    // exact hit (CHECK failure / OrthoFromPcl's perfect match, which depends on
    // the scan order) or an empty first search (the ladder): the scalar routine
    cell_global(p, start, P, i, j, o);
    return;
  }
  const double hq = my_num / my_den;
  if (round_is_certain(hq, idw_err_bound(p, ncand, call_zmax(o))))
    emit_value(p, o, i, j, hq);
  else  // (see round_is_certain: the order-independent arithmetic, a few cells per 1e8)
    canonical_search(p, start, P, p.base_x + p.res * (-(double)(i + p.i_off)),
                     p.base_y + p.res * (-(double)(j + p.j_off)), i, j, w, T, o);
}

// Pure global-memory gather: used when the first-level window is too wide for
// the LDS image (very fine grids) and by the adaptive OrthoFromPcl passes.
__global__ void __launch_bounds__(256)
k_dsm_gather(DsmParams p, const uint32_t* __restrict__ start,
             const Pts P, CellOut o) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= p.rows || j >= p.cols) return;
  cell_global(p, start, P, i, j, o);
}

// The optional capped mode has a kernel of its own: its register-resident result set
// would otherwise set the register budget (and with it the occupancy) of every
// kernel that can reach cell_global.
__global__ void __launch_bounds__(256)
k_dsm_gather_knn(DsmParams p, const uint32_t* __restrict__ start,
                 const Pts P, CellOut o) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= p.rows || j >= p.cols) return;
  if (p.only_unfilled && o.mask[cell_at(p, i, j)]) return;
  cell_global_knn(p, start, P, i, j, o);
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
constexpr int kListHdr = 8;         // counters in front of the tile lists
constexpr int kNumLists = 7;
// s_setreg operand: HW_REG_MODE (id 1), offset 6, width 2 = FP_DENORM for f64 / f16
constexpr int kHwRegModeFpDenormF64 = 1 | (6 << 6) | ((2 - 1) << 11);

// Ordered 32-bit keys of floats (monotone: a < b  <=>  key(a) < key(b)); LDS / global atomicMin /
// atomicMax on heights.
__device__ __forceinline__ uint32_t zkey_of(float v) {
  const uint32_t b = __float_as_uint(v);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float zkey_to_float(uint32_t k) {
  return __uint_as_float((k >> 31) ? (k & 0x7FFFFFFFu) : ~k);
}

// The single-precision gather's error budget for a region whose heights span [zmin_f, zmax_f]
// (floats: the rounded extremes of the staged points): how many f32 additions may accumulate
// between two flushes of the running sums so that
//   |dh| <= 2 (eps_w + (n + 2) u) S,  S = half the height range, u = 2^-24,
// stays below 0.8 x (1e-4 m - one spacing of the stored float) -- above 1024 m, where a float
// spacing alone exceeds 1e-4 m, "1 LSB" is the bar and a quarter of a spacing the budget.
// Returns the allowed additions (-1: no finite range / no room at all, 1 << 20: flat), the middle
// of the range in *z0w (relative to zoff).
// ONE definition for the gather kernel and for the occupancy pre-pass that sorts tiles without
// room (< 48 additions) straight onto the FP64 lists.
__device__ __forceinline__ int fx_allowed_additions(float zmin_f, float zmax_f, float epsw, double zoff,
                                                   double* z0w) {
  // (record pipeline: zmin_f / zmax_f are f32 OFFSETS from zoff = zref; the stored float is the
  // height zoff + offset, and every offset carries the rounding of its own conversion)
  const double zmin = (double)zmin_f, zmax = (double)zmax_f;
  // the middle, as a float: the staging subtracts it from the f32 offsets in single precision
  *z0w = (double)(float)(0.5 * zmin + 0.5 * zmax);
  // max |z - z0| over the region: half the f32 range, plus what the conversions lost
  const float S_half = (float)(0.5 * zmax - 0.5 * zmin) * 1.000001f +
                       (fabsf(zmin_f) + fabsf(zmax_f)) * 1.2e-7f;
  // spacing of the stored floats: at the largest height of the region for the 1e-4 m regime,
  // at the smallest for the "1 LSB" regime (above 1024 m) -- conservative both ways
  const double alo = zoff + zmin, ahi = zoff + zmax;
  const float zabs_hi = (float)fmax(fabs(alo), fabs(ahi));
  const float zabs_lo = (alo <= 0.0 && ahi >= 0.0) ? 0.0f : (float)fmin(fabs(alo), fabs(ahi));
  const float ulp_hi = __uint_as_float((__float_as_uint(fmaxf(zabs_hi, 1e-30f)) & 0x7F800000u)) * 1.1920929e-7f;
  const float ulp_lo = __uint_as_float((__float_as_uint(fmaxf(zabs_lo, 1e-30f)) & 0x7F800000u)) * 1.1920929e-7f;
  float allowed = 0.8f * (ulp_hi < 1e-4f * 0.6f ? 1e-4f - ulp_hi : 0.25f * ulp_lo);
  // the records' own rounding: every offset is (float)(z - zref), half a spacing at most
  allowed -= fmaxf(fabsf(zmin_f), fabsf(zmax_f)) * 6.0e-8f;
  if (!(S_half <= 3.0e38f) || !(allowed > 0.0f)) return -1;
  if (S_half == 0.0f) return 1 << 20;
  // (n + 3: the hits of a trip, the add of the trip's sum, and the staging's f32 subtraction)
  const float room = (0.5f * allowed / S_half - epsw) * 16777216.0f - 3.0f;
  return room > 1.0e6f ? (1 << 20) : (int)room;
}
constexpr int kSortTrips = 6;       // gather: windows of up to this many row pairs take their spans longest first
#ifndef AMHIP_GATHER_PLAIN_TRIPS
#define AMHIP_GATHER_PLAIN_TRIPS 0  // (1: A-B build with the spans top to bottom)
#endif
constexpr int kFxMinAdditions = 48;  // a trip of a dense tile brings up to ~20 candidates; below: FP64

// One lane per gather tile (tile numbering: ti + tj * tiles_i, as in the gather):
//   occ[tile] = 0          no binned point within the LAST fallback radius of the tile
//             = 1 + class  otherwise; class = which LDS capacity the points of the
//                          tile's first-level region (the `np` gather_tile stages)
//                          fit: 0: cap0 (the launch's own), 1: <= cap1, 2: <= cap2,
//                          3: more than any LDS image holds (wave-per-cell path)
// Clouds are not uniform (overlapping strips, partial coverage): the capacity of
// the main launch follows the MEAN density, denser tiles go to launches with more
// LDS per workgroup instead of falling back to the global-memory path.
//   lists (may be null): [kListHdr counters] then arrays of ntiles ids:
//     list 0  occupied class-0 tiles (only filled when `list0` -- sparse calls)
//     list k  class-k tiles (k = 1, 2, 3)
//     list 4  class-0 tiles the single-precision gather hands to the FP64 kernel
//             (filled by k_dsm_gather_f32, not here)
//     list 5  class-1 / class-2 tiles of the single-precision list launches handed to the
//             FP64 kernel's largest LDS image, list 6: those beyond it (wave-per-block kernel)
//   bin_z (may be null; single-precision mode after the three-pass sort): per bin the ordered
//     keys of its lowest / highest (float-rounded) height, written by the placement pass.  The
//     tile's region then has a known height range BEFORE anything is staged: a tile whose error
//     budget has no room (rough terrain: walls, canopy) gets occ = 1 + 4 -- no single-precision
//     launch takes it -- and goes straight onto the FP64 list the kernel itself would have handed
//     it to (rej_own: list 4, else 5, or 6 beyond rej_big_np points), instead of being staged,
//     rejected and staged again (bench.py rough_terrain: 5.7 -> 4.8 ms per DSM call).
// Append `tile` to list `lst` (-1: none) of every lane of the wave with ONE returning atomic per
// (wave, list): tens of thousands of returning atomics on one counter serialise in L2 (0.86 ms
// for the 73 K tiles of a rough scene, measured; 0.02 ms this way).
__device__ __forceinline__ void wave_append_tile(int* __restrict__ lists, int ntiles, int lst, int tile) {
  unsigned* cnt = reinterpret_cast<unsigned*>(lists);
  const int lane = threadIdx.x & 63;
  for (int L = 0; L < kNumLists; ++L) {
    const unsigned long long m = __ballot(lst == L);
    if (!m) continue;
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(&cnt[L], (unsigned)__popcll(m));
    base = __shfl(base, leader, 64);
    if (lst == L)
      lists[kListHdr + (size_t)L * ntiles + base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = tile;
  }
}

__device__ __forceinline__ void
tile_occupancy_block(const DsmParams& p, int tile_j, const uint32_t* __restrict__ start,
                     uint8_t* __restrict__ occ, int* __restrict__ lists, int list0, int cap0,
                     int cap1, int cap2, const uint2* __restrict__ bin_z, int rej_own,
                     int rej_big_np, int rej_dense, const double* __restrict__ zref_dev) {
  const int ntiles = p.tiles_i * p.tiles_j;
  const double zref = bin_z ? zref_dev[0] : 0.0;
  const int tile = blockIdx.x * 256 + threadIdx.x;
  int lst = -1;       // list this tile is appended to
  int rejected = 0;   // pre-classified for the FP64 kernel
  if (tile < ntiles) {
    const int ti = tile % p.tiles_i, tj = tile / p.tiles_i;
    const int i0 = ti * kTileI, j0 = tj * tile_j;
    const int i_hi = min(i0 + kTileI, p.rows) - 1;
    const int j_hi = min(j0 + tile_j, p.cols) - 1;
    const int wl = p.w[p.nlevels - 1];
    const int ex0 = (i0 - wl + p.M) / p.B, ex1 = (i_hi + wl + p.M) / p.B;
    const int ey0 = (j0 - wl + p.M) / p.B, ey1 = (j_hi + wl + p.M) / p.B;
    uint32_t tot = 0;
    for (int by = ey0; by <= ey1; ++by) {
      const uint32_t* row = start + (size_t)by * p.nbx;
      tot += row[ex1 + 1] - row[ex0];
    }
    if (tot == 0) {
      occ[tile] = 0;
    } else {
      // the first-level region, exactly as gather_tile() sums it
      const int w0 = p.w[0];
      const int rbx0 = (i0 - w0 + p.M) / p.B, rbx1 = (i_hi + w0 + p.M) / p.B;
      const int rby0 = (j0 - w0 + p.M) / p.B, rby1 = (j_hi + w0 + p.M) / p.B;
      uint32_t np = 0;
      for (int by = rby0; by <= rby1; ++by) {
        const uint32_t* row = start + (size_t)by * p.nbx;
        np += row[rbx1 + 1] - row[rbx0];
      }
      const int cls = np <= (uint32_t)cap0 ? 0 : (np <= (uint32_t)cap1 ? 1 : (np <= (uint32_t)cap2 ? 2 : 3));
      if (bin_z && lists && cls < 3 && np > 0) {
        uint32_t klo = 0xFFFFFFFFu, khi = 0u;
        // (six independent loads in flight per step: one lane walks ~108 bins, and a serial
        // chain of that many L2 round trips was 0.045 ms of the pre-pass)
        const int nbxr = rbx1 - rbx0 + 1;
        for (int by = rby0; by <= rby1; ++by) {
          const uint2* row = bin_z + (size_t)by * p.nbx + rbx0;
          int k = 0;
          for (; k + 6 <= nbxr; k += 6) {
            const uint2 v0 = row[k], v1 = row[k + 1], v2 = row[k + 2], v3 = row[k + 3], v4 = row[k + 4],
                        v5 = row[k + 5];
            klo = min(min(min(klo, v0.x), min(v1.x, v2.x)), min(min(v3.x, v4.x), v5.x));
            khi = max(max(max(khi, v0.y), max(v1.y, v2.y)), max(max(v3.y, v4.y), v5.y));
          }
          for (; k < nbxr; ++k) {
            const uint2 v = row[k];
            klo = min(klo, v.x);
            khi = max(khi, v.y);
          }
        }
        if (klo <= khi) {
          double z0w;
          const int na = fx_allowed_additions(zkey_to_float(klo), zkey_to_float(khi), p.fx_epsw, zref, &z0w);
          if (na < kFxMinAdditions) {
            rejected = 1;
            const int own = (cls == 0) && rej_own;
            // (rej_dense: the own-image tiles are taken by a dense launch that filters on occ
            // instead of walking list 4 -- dsm_run decides, from the previous call's counts)
            lst = own ? (rej_dense ? -1 : 4) : ((int)np > rej_big_np ? 6 : 5);
            occ[tile] = (uint8_t)(own ? 1 + 4 : 1 + 5);
          }
        }
      }
      if (!rejected) {
        occ[tile] = (uint8_t)(1 + cls);
        if (lists && (cls > 0 || list0)) lst = cls;
      }
    }
  }
  if (lists) {
    wave_append_tile(lists, ntiles, lst, tile);
    // [7]: pre-classified tiles that are on NO list (the dense launch takes them)
    const unsigned long long m = __ballot(rejected != 0 && lst < 0);
    if (m && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1)
      atomicAdd(reinterpret_cast<unsigned*>(lists) + 7, (unsigned)__popcll(m));
  }
}

// The gather's PROLOGUE (round 5: one launch for what were four).  Workgroups [0, nocc) classify the
// tiles as described above; workgroups [nocc, nocc + nrange) fold the scatter waves' height
// partials into the running range and the call's own (range_reduce_block: what k_range_reset +
// k_range_reduce did); the workgroup that finishes LAST (a ticket) leaves the list counters in a
// pinned host word array for the next call's launch policy (what a hipMemcpyAsync did).
struct ProloguePart {
  const double* zpart;              // the scatter waves' partial [min z, max z] pairs (may be null)
  size_t nparts;
  unsigned long long* range;        // running range (may be null)
  unsigned long long* call_range;
  unsigned nocc, nrange;
  unsigned* ticket;                 // zero at launch, left zero
  unsigned* host_stats;             // pinned: kListHdr words (may be null: the caller copies later)
};

__global__ void __launch_bounds__(256)
k_dsm_tile_occupancy(DsmParams p, int tile_j, const uint32_t* __restrict__ start,
                     uint8_t* __restrict__ occ, int* __restrict__ lists, int list0, int cap0,
                     int cap1, int cap2, const uint2* __restrict__ bin_z, int rej_own,
                     int rej_big_np, int rej_dense, const double* __restrict__ zref_dev,
                     ProloguePart pp) {
  __shared__ double s_pair[32];
  __shared__ unsigned s_last;
  if (blockIdx.x >= pp.nocc) {
    if (pp.zpart && pp.nparts)
      range_reduce_block(pp.zpart, pp.nparts, pp.range, pp.call_range, blockIdx.x - pp.nocc, pp.nrange, s_pair);
  } else {
    tile_occupancy_block(p, tile_j, start, occ, lists, list0, cap0, cap1, cap2, bin_z, rej_own, rej_big_np,
                         rej_dense, zref_dev);
  }
  if (!pp.host_stats || !lists) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(pp.ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x == 0) *pp.ticket = 0u;
  if (threadIdx.x < (unsigned)kListHdr)
    pp.host_stats[threadIdx.x] = __hip_atomic_load(reinterpret_cast<unsigned*>(lists) + threadIdx.x,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// kKnn: the OPTIONAL capped mode (amhip_ctx_set_dsm_knn) on the same LDS image -- per cell the k
// nearest of the first search's points, summed in ascending distance with true divisions like
// cell_global_knn(); cells without a first-level neighbour, tiles beyond the image and exact hits
// take cell_global_knn() itself.
template <int NT, int kTileJ, int kCap, int kKnn = 0 /* 0: off; else the capped mode's set size, 4 or 8 */>
__device__ __forceinline__ void gather_tile(const DsmParams& p, const uint32_t* __restrict__ start,
                                            const Pts P,
                                            const uint8_t* __restrict__ tile_occ, const CellOut& o,
                                            const int tile, unsigned char* smem, int my_class) {
  constexpr int kWaves = NT / 64;
  constexpr int kCellsPerLane = kTileJ / kWaves;
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

  const int ti = tile % p.tiles_i;
  const int tj = tile / p.tiles_i;
  const int i0 = ti * kTileI;
  const int j0 = tj * kTileJ;
  const int i_hi = min(i0 + kTileI, p.rows) - 1;
  const int j_hi = min(j0 + kTileJ, p.cols) - 1;
  const int w0 = p.w[0];

  // No point within the LAST fallback radius of the tile (k_dsm_tile_occupancy):
  // every cell stays untouched.  In incremental mapping this is most of the map,
  // so the test is one byte, read before anything else.
  const int occ = tile_occ[tile];
  // (dense launch: tiles of another capacity class belong to another launch; my_class 4 = the
  // dense launch over the tiles the occupancy pre-pass classified for this kernel: nothing else)
  if (my_class >= 0 && occ != 0 && occ - 1 != my_class) return;
  if (occ == 0) {
    if (my_class == 4) return;
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

  // (as in gather_tile_f32: the cell table is cleared / scanned / rewritten in whole quads while
  // wave 0 reads the region's bin rows and prefixes their lengths across its lanes)
  const bool geom_ok = nrb <= 64 && ncell + 3 <= p.lds_cells;
  {
    uint4* q = reinterpret_cast<uint4*>(s_off);
    const int nq0 = geom_ok ? (ncell + 4) >> 2 : 0;
    for (int k = tid; k < nq0; k += NT) q[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (wid == 0) {
    uint32_t gs = 0, cnt = 0;
    if (geom_ok && lane < nrb) {
      const uint32_t* row = start + (size_t)(rby0 + lane) * p.nbx;
      gs = row[rbx0];
      cnt = row[rbx1 + 1] - gs;
    }
    const uint32_t incl = wave_incl_scan(cnt, lane);
    if (lane < nrb && geom_ok) {
      s_rowg[lane] = gs;
      s_rowp[lane + 1] = incl;
    }
    if (lane == 63) s_ctl[0] = incl;
    if (lane == 0) {
      s_rowp[0] = 0;
      s_ctl[1] = 0;
    }
  }
  __syncthreads();
  const int np = (int)s_ctl[0];
  const bool use_lds = geom_ok && np <= p.lds_cap;
  if (!use_lds) {
    // region too tall for the row tables (very fine grids) or a tile that does
    // not fit this launch's LDS image: one lane per cell on the global bins
    for (int c = 0; c < kCellsPerLane; ++c) {
      const int i = i0 + lane, j = j0 + wid * kCellsPerLane + c;
      if (i <= i_hi && j <= j_hi) {
        if constexpr (kKnn > 0) cell_global_knn(p, start, P, i, j, o);
        else cell_global(p, start, P, i, j, o);
      }
    }
    return;
  }

  // ---- stage + cell-bin the region's points in LDS --------------------------
  // pass 1: count per cell (LDS atomics), remember (cell, rank) per point
  static_assert(kCellsPerLane % 2 == 0, "cell pairs must start on even rows of the tile");
  constexpr int kMaxK = (kCap + NT - 1) / NT;  // p.lds_cap == kCap
  uint32_t pslot[kMaxK];                       // cell << 13 | rank  (rank < kCap <= 8192)
  double ppx[kMaxK], ppy[kMaxK], ppz[kMaxK];   // the thread's points (placed after the scan)
  // (all loads first -- threads past the region's end re-read its last point: inside the loop
  // below each waited for the one before, see gather_tile_f32)
  if (np > 0) {
    size_t pg[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      const int idx = min(tid + k * NT, np - 1);
      int r = 0;
      while (idx >= (int)s_rowp[r + 1]) ++r;
      pg[k] = (size_t)s_rowg[r] + (size_t)(idx - (int)s_rowp[r]);
    }
    // (the doubles / records choice of pts_x() outside the loop: inside, it is a branch per
    // point and the loads of one point wait for the previous point's)
    if (P.sorted) {
#pragma unroll
      for (int k = 0; k < kMaxK; ++k) {
        ppx[k] = P.sorted[3 * pg[k] + 0];
        ppy[k] = P.sorted[3 * pg[k] + 1];
        ppz[k] = P.sorted[3 * pg[k] + 2];
      }
    } else {
      const double* q[kMaxK];
#pragma unroll
      for (int k = 0; k < kMaxK; ++k) q[k] = P.cloud + 3 * (size_t)P.sidx[pg[k]];
#pragma unroll
      for (int k = 0; k < kMaxK; ++k) {
        ppx[k] = q[k][0] - P.sub_x;
        ppy[k] = q[k][1] - P.sub_y;
        ppz[k] = q[k][2];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    const int idx = tid + k * NT;
    pslot[k] = 0xFFFFFFFFu;
    if (idx < np) {
      const double px = ppx[k];
      const double py = ppy[k];
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
      pslot[k] = (cell << 13) | atomicAdd(&s_off[cell], 1u);
    }
  }
  __syncthreads();
  {
    // exclusive scan of the counters, in place: a thread owns consecutive whole quads, a wave
    // their prefix (DPP scan), the wave totals meet in LDS
    const int nq = (ncell + 4) >> 2;
    const int qper = (nq + NT - 1) / NT;
    const int q0 = tid * qper, q1 = min(q0 + qper, nq);
    uint4* const qoff = reinterpret_cast<uint4*>(s_off);
    unsigned qsum = 0;
    for (int k = q0; k < q1; ++k) {
      const uint4 v = qoff[k];
      qsum += (v.x + v.y) + (v.z + v.w);
    }
    const unsigned qincl = wave_incl_scan(qsum, lane);
    if (lane == 63) s_scan[wid] = qincl;
    __syncthreads();
    const int mywave = __builtin_amdgcn_readfirstlane(wid);
    unsigned run = qincl - qsum;
#pragma unroll
    for (int w = 0; w < kWaves - 1; ++w)
      if (w < mywave) run += s_scan[w];
    for (int k = q0; k < q1; ++k) {
      const uint4 v = qoff[k];
      uint4 e;
      e.x = run;
      e.y = run + v.x;
      e.z = e.y + v.y;
      e.w = e.z + v.z;
      run = e.w + v.w;
      qoff[k] = e;  // (entry ncell, an empty count, receives the total)
    }
  }
  __syncthreads();
  // pass 2: drop the points into their sorted slot
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    if (pslot[k] != 0xFFFFFFFFu) {
      const uint32_t pos = s_off[pslot[k] >> 13] + (pslot[k] & 0x1FFFu);
      s_xy[pos] = make_double2(ppx[k], ppy[k]);
      s_z[pos] = ppz[k];
    }
  }
  __syncthreads();
  if constexpr (kKnn > 0) {
    // ---- capped mode: lane = row index i, one cell at a time (the set of k lives in registers) ----
    const int i = i0 + lane;
    if (i <= i_hi) {
      const double qx = p.base_x + p.res * (-(double)(i + p.i_off));
      const int ci = i + p.M - ox;
      const double T0k = p.T[0];
      for (int c = 0; c < kCellsPerLane; ++c) {
        const int j = j0 + wid * kCellsPerLane + c;
        if (j > j_hi) break;
        const double qy = p.base_y + p.res * (-(double)(j + p.j_off));
        // (the row pairs of the cell PAIR this cell belongs to: they cover both cells' windows, the
        // exact test drops what lies beyond this one's)
        const int cjp = (j0 + wid * kCellsPerLane + (c & ~1)) + p.M - oy;
        const uint32_t* orow = s_off + ((cjp - w0 + sh) >> 1) * RW2 + 2 * ci;
        KnnSetT<(kKnn > 0 ? kKnn : 1)> ks;
        knn_init(&ks);
        for (int r = 0; r <= w0; ++r) {
          const int w = p.wrp[r];
          const uint32_t kb = orow[-2 * w], ke = orow[2 * w + 2];
          orow += RW2;
          uint32_t k = kb;
          for (; k < ke; ++k) {
            const double2 xy = s_xy[k];
            const double dx = qx - xy.x;
            const double dy = qy - xy.y;
            double d2 = dx * dx;
            d2 = d2 + dy * dy;
            if (d2 < T0k) knn_add(&ks, p.knn_k, d2, s_z[k]);
          }
        }
        bool again = ks.n == 0;      // no first-level neighbour: the ladder, on the global bins
        double num = 0.0, den = 0.0;
#pragma unroll
        for (int q = 0; q < kKnn; ++q)
          if (q < ks.n) {
            if (!(ks.d2[q] > 0.0)) again = true;   // (exact hit: cell_global_knn raises the CHECK)
            num += ks.z[q] / ks.d2[q];
            den += 1.0 / ks.d2[q];
          }
        if (again) {
          const uint32_t slot = atomicAdd(&s_ctl[1], 1u);
          s_flag[slot] = (uint16_t)((wid * kCellsPerLane + c) * kTileI + lane);
        } else {
          emit_value(p, o, i, j, num / den);
        }
      }
    }
    __syncthreads();
    const int nflag_k = (int)s_ctl[1];
    for (int f = tid; f < nflag_k; f += NT) {
      const int code = s_flag[f] & 0x3FFF;
      cell_global_knn(p, start, P, i0 + (code % kTileI), j0 + (code / kTileI), o);
    }
    return;
  }
  // (max |z| of the call's points: round_is_certain's bound, loaded here, used after the loop.
  // The bound's n is the lane's own candidate count: a tile-wide n -- the image's 860 points,
  // a wave-uniform bound in scalar registers -- sends 1000 cells per 1e8 to the redo instead of
  // 60, and a redo is a serial walk of global memory as long as a whole tile: measured slower.)
  // err = (4 n + 8) 2^-53 max|z| as ONE fma per cell pair: c1 n + c0 (wave-uniform constants;
  // tuning knob dsm_canon_all makes them infinite)
  const double zmax_tile = p.canon_all ? __builtin_huge_val() : call_zmax(o);
  const double err_c1 = 0x1p-51 * zmax_tile, err_c0 = 0x1p-50 * zmax_tile;

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
      unsigned ncand = 0;  // candidates tested (>= hits of either cell): round_is_certain's n
      // rows jA-w0 .. jA+1+w0 (the last one only matters for cell B), one row
      // pair = one contiguous span per trip (lanes wait for each other per
      // trip, and the spread of a two-row candidate count is relatively smaller).
      const uint32_t* orow = s_off + ((cj - w0 + sh) >> 1) * RW2 + 2 * ci;
      // FP64 denormals are FLUSHED while the products run (MODE.FP_DENORM[7:6] = 0):
      // a product that dips below 2^-1022 inside one trip -- hundreds of points
      // within centimetres of a centre, in a tile of a dense capacity class -- then
      // becomes exactly 0, stays 0 and sends the cell to the reciprocal-weight
      // routine below, instead of silently losing bits as a denormal and climbing
      // back into the normal range before the end of the trip.  (Overflow is
      // sticky anyway.  A d2 itself is never denormal: coordinates are doubles of
      // magnitude >= 1e-3, their differences multiples of ~1e-19.)
      __builtin_amdgcn_s_setreg(kHwRegModeFpDenormF64, 0);
      // The lanes of a wave wait for each other per trip, and a trip lasts as long as its
      // busiest lane: 70 iterations per wave for a mean of 41 candidates per lane when every lane
      // takes its row pairs top to bottom (5 trips at cfg2's density; max over 64 lanes of a
      // Poisson count, five times).  Every lane therefore takes its spans LONGEST FIRST: the
      // s-th trip of the wave then meets every lane's s-th longest span, and the sum over s of
      // the maxima is 60 (simulated; the sums are order-free, the rounding guard below covers
      // the order).  Spans as (length << 16 | first) in a 6-key sorting network of v_max / v_min
      // pairs; windows of more than 6 row pairs keep the plain order.
      uint32_t key0 = 0, key1 = 0, key2 = 0, key3 = 0, key4 = 0, key5 = 0;
      const bool sorted_trips = w0 < kSortTrips && !AMHIP_GATHER_PLAIN_TRIPS;
      if (sorted_trips) {
        // (branch-free: a row pair beyond the window re-reads the last one and gets length 0, so
        // that all twelve offsets are in flight before the first wait)
        uint32_t kb_[kSortTrips], ke_[kSortTrips];
#pragma unroll
        for (int r = 0; r < kSortTrips; ++r) {
          // (static index: the six widths arrive with one scalar load, not one load and wait each)
          const int rr = min(r, w0);
          const int w = r <= w0 ? p.wrp[r] : 0;
          kb_[r] = orow[rr * RW2 - 2 * w];
          ke_[r] = orow[rr * RW2 + 2 * w + 2];
        }
        auto span = [&](int r) __attribute__((always_inline)) -> uint32_t {
          const uint32_t len = r <= w0 ? ke_[r] - kb_[r] : 0u;
          ncand += len;
          return (len << 16) | kb_[r];   // (both below 2^16: the LDS image holds <= 7680 points)
        };
        key0 = span(0), key1 = span(1), key2 = span(2), key3 = span(3), key4 = span(4), key5 = span(5);
        auto cx = [](uint32_t& a, uint32_t& b) __attribute__((always_inline)) {
          const uint32_t hi = max(a, b), lo = min(a, b);
          a = hi;
          b = lo;
        };
        // (12 compare-exchanges, 5 layers)
        cx(key0, key5), cx(key1, key3), cx(key2, key4);
        cx(key1, key2), cx(key3, key4);
        cx(key0, key3), cx(key2, key5);
        cx(key0, key1), cx(key2, key3), cx(key4, key5);
        cx(key1, key2), cx(key3, key4);
      }
      for (int r = 0; r <= w0; ++r) {
        uint32_t kb, ke;
        if (sorted_trips) {
          kb = key0 & 0xFFFFu;
          ke = kb + (key0 >> 16);
          key0 = key1, key1 = key2, key2 = key3, key3 = key4, key4 = key5;
        } else {
          const int w = p.wrp[r];
          kb = orow[-2 * w];
          ke = orow[2 * w + 2];
          orow += RW2;
          ncand += ke - kb;
        }
        auto candidate = [&](const double2* __restrict__ cxy, const double* __restrict__ cz) __attribute__((always_inline)) {
          const double2 xy = *cxy;
          const double z = *cz;
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
        };
#if AMHIP_GATHER_UNROLL > 1
        // (unrolled by hand: the loop bookkeeping is a quarter of the loop's
        // instructions otherwise)
        // (pointers, not indices: the latch is then two adds and a compare instead of six VALU
        // instructions per pair of candidates)
        const double2* pxy = s_xy + kb;
        const double* pz = s_z + kb;
        const double2* const pend = s_xy + ke;
        if (kb + 1u < ke) {                     // (at least one pair: ke >= 2, so pend - 1 is inside the image)
          const double2* const plim = pend - 1;  // (a pair starts below it)
          do {
            candidate(pxy, pz);
            candidate(pxy + 1, pz + 1);
            pxy += 2;
            pz += 2;
          } while (pxy < plim);
        }
        if (pxy < pend) candidate(pxy, pz);
#else
        for (uint32_t k = kb; k < ke; ++k) candidate(s_xy + k, s_z + k);
#endif
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
      __builtin_amdgcn_s_setreg(kHwRegModeFpDenormF64, 3);  // denormals allowed again
      // P == 0 <=> a hit with d2 == 0 -- or a product that underflowed inside
      // one trip.  Either way (and whenever the running values came close to
      // the ends of the double range) the cell is re-done by the global
      // routine, which weights with reciprocals and tracks exact hits
      // explicitly: DSM -> CHECK failure (dsm.cc:165), OrthoFromPcl -> that
      // point's value (ortho-from-pcl.cc:91-96).
      const double err_pair = fma((double)ncand, err_c1, err_c0);  // (both cells saw the same candidates)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && !haveB) break;
        const double Nn = h ? NB : NA, Dd = h ? DB : DA, Pp = h ? PB : PA;
        const bool suspect = h ? suspectB : suspectA;
        const int jj = jA + h;
        int queue = 0;  // 1: no first-level neighbour (ladder), 2: redo the whole cell,
                        // 3: the quotient sits on a float rounding boundary (canonical_search)
        if (Pp == 0.0 || suspect || !(Dd < 0x1p+1000) || !(fabs(Nn) < 0x1p+1000)) {
          queue = 2;
        } else if (Dd > 0.0) {
          const double hq = Nn / Dd;
          // (round_is_certain, with its lower float stored: the same value as (float)hq then)
          const float f_lo = (float)(hq - err_pair), f_hi = (float)(hq + err_pair);
          if (f_lo == f_hi) {
            const size_t at = cell_at(p, i, jj);
            o.layer[at] = f_lo;
            if (o.mask) o.mask[at] = 1;
          } else {
            queue = 3;
          }
        } else {
          queue = 1;
        }
        if (queue) {
          const uint32_t slot = atomicAdd(&s_ctl[1], 1u);
          s_flag[slot] = (uint16_t)(((wid * kCellsPerLane + c + h) * kTileI + lane) |
                                    (queue == 2 ? 0x8000 : (queue == 3 ? 0x4000 : 0)));
        }
      }
    }
  }
  __syncthreads();

  // ---- queued cells (dense over the workgroup): the fallback ladder, or for
  // OrthoFromPcl the full global routine (several coincident exact hits) ------
  const int nflag = (int)s_ctl[1];
  for (int f = tid; f < nflag; f += NT) {
    const int code = s_flag[f] & 0x3FFF;
    const bool redo = (s_flag[f] & 0x8000) != 0;
    const bool canon = (s_flag[f] & 0x4000) != 0;
    const int fi = i0 + (code % kTileI);
    const int fj = j0 + (code / kTileI);
    if (canon) {
      canonical_search(p, start, P, p.base_x + p.res * (-(double)(fi + p.i_off)),
                       p.base_y + p.res * (-(double)(fj + p.j_off)), fi, fj, p.w[0], p.T[0], o);
    } else if (p.pcl_mode || redo) {
      cell_global(p, start, P, fi, fj, o);
    } else {
      const double fqx = p.base_x + p.res * (-(double)(fi + p.i_off));
      const double fqy = p.base_y + p.res * (-(double)(fj + p.j_off));
      const bool done = cell_fallback_global(p, start, P, fi, fj, fqx, fqy, o);
      if (!done) {
        leave_untouched(p, o, fi, fj);
        if (o.unfilled) atomicAdd(o.unfilled, 1u);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// LDS-tiled gather in single precision with exact guards (dsm::Dsm only)
// ---------------------------------------------------------------------------
// The FP64 gather above is bound by FP64 VALU issue (5 + 4 FP64 instructions per
// candidate test + hit).  The contract asks for the reference's neighbour SETS and
// heights within 1e-4 m, not for its doubles, so this variant does the per-pair work
// at the f32 rate:
//   * staging converts a point to 32-bit fixed point in units of 2^-S cells relative
//     to the tile region (S = 28 for a 4-cell radius; the integers wrap, only
//     differences of <= w0 + 2 cells are formed and those are exact) and its height to
//     an f32 offset from the middle z0 of the region's height range;
//   * a test is  dx = cvt(Ui - U), dy = cvt(Vj - V), d2 = dx*dx + dy*dy  in f32: within
//     3e-7 (relative) of the reference's double at the search radius.  d2 < T(1 + 2e-6)
//     counts as a hit; a hit with d2 >= T(1 - 2e-6) marks the cell AMBIGUOUS;
//   * a hit adds  w = v_rcp_f32(d2)  to the weight sum and  w * (z - z0)  to the
//     numerator; the cell's height is z0 + N / D.
// Guards (any of them sends the CELL to the FP64 routine, which decides with the
// reference's own doubles): an ambiguous hit; a weight sum that a hit nearer than
// fx_theta cells would produce (the fixed-point quantum 2^-(S+1) cells is then no
// longer small against the distance; also catches exact hits: rcp(0) = inf); more
// hits than the error bound allows for the tile's height range.  A TILE whose height
// range leaves no room under the bound (|dh| <= 2 (eps_w + (n + 2) 2^-24) (zmax - zmin)/2
// must stay below 1e-4 m minus one float spacing of the stored height) is appended to
// a list and done by the FP64 kernel afterwards.  Empty first searches take the FP64
// ladder as before, so the NaN pattern is the reference's by construction.
__device__ __forceinline__ void wave_minmax_d(double* lo, double* hi) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    *lo = fmin(*lo, __shfl_xor(*lo, d, 64));
    *hi = fmax(*hi, __shfl_xor(*hi, d, 64));
  }
}

// One cell by a whole WAVE on the global bins, the lanes over the candidates of the
// first-level window (FP64, reciprocal weights): the guard path of the f32 gather.
// Same decisions as cell_global(); all 64 lanes must call it with the same cell.
__device__ __forceinline__ void cell_wave_exact(const DsmParams& p, const uint32_t* __restrict__ start,
                                                const Pts P, int i, int j,
                                                const CellOut& o) {
  const int lane = threadIdx.x & 63;
  const double qx = p.base_x + p.res * (-(double)(i + p.i_off));
  const double qy = p.base_y + p.res * (-(double)(j + p.j_off));
  const int w = p.w[0];
  const double T = p.T[0];
  const int bx0 = (i - w + p.M) / p.B, bx1 = (i + w + p.M) / p.B;
  const int by0 = (j - w + p.M) / p.B, by1 = (j + w + p.M) / p.B;
  double num = 0.0, den = 0.0;
  unsigned exact = 0;
  // The routine is a chain of dependent memory round trips, and it runs at the END of a tile
  // while the workgroup's other waves have left (with one workgroup per CU -- the 7680-point
  // images of dense clouds -- nothing else covers it).  Bin row after bin row, each with its
  // own span -> (row index ->) point -> height chain, it was 9 round trips on sorted doubles
  // and 12 through the records' row indices (+ 0.4 ms on 12 000 tiles at 4 points per cell).
  // Now: the spans of ALL bin rows in one trip (one lane each), then the candidates of all rows
  // as one range, four batches of 64 in flight, x, y AND z fetched together: 3 round trips.
  constexpr int kRows = 8;  // (the first search radius spans <= 3 bin rows: B = that radius)
  const int nrow = by1 - by0 + 1;
  if (nrow <= kRows) {
    uint32_t s0 = 0, len = 0;
    if (lane < nrow) {
      const uint32_t* row = start + (size_t)(by0 + lane) * p.nbx;
      s0 = row[bx0];
      len = row[bx1 + 1] - s0;
    }
    const uint32_t incl = wave_incl_scan(len, lane);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t rs[kRows], re[kRows];  // (wave-uniform: first point and exclusive prefix per row)
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      rs[r] = (uint32_t)__builtin_amdgcn_readlane((int)s0, r);
      re[r] = (uint32_t)__builtin_amdgcn_readlane((int)(incl - len), r);
    }
    constexpr int kU = 4;
    for (uint32_t c0 = 0; c0 < total; c0 += 64 * kU) {
      size_t g[kU];
      bool ok[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t c = c0 + 64u * u + lane;
        ok[u] = c < total;
        uint32_t gg = rs[0] + c;
#pragma unroll
        for (int r = 1; r < kRows; ++r)
          if (r < nrow && c >= re[r]) gg = rs[r] + (c - re[r]);
        g[u] = gg;
      }
      const double* q[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        q[u] = nullptr;
        if (ok[u]) q[u] = P.sorted ? P.sorted + 3 * g[u] : P.cloud + 3 * (size_t)P.sidx[g[u]];
      }
      double x[kU], y[kU], z[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        x[u] = y[u] = z[u] = 0.0;
        if (ok[u]) {
          x[u] = q[u][0];
          y[u] = q[u][1];
          z[u] = q[u][2];
        }
      }
      const double sx = P.sorted ? 0.0 : P.sub_x, sy = P.sorted ? 0.0 : P.sub_y;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (ok[u]) {
          const double dx = qx - (x[u] - sx);
          const double dy = qy - (y[u] - sy);
          double d2 = dx * dx;
          d2 = d2 + dy * dy;       // L2_Adaptor (nanoflann.hpp:319-322)
          if (d2 < T) {            // strict (nanoflann.hpp:157)
            if (d2 > 0.0) idw_add(d2, z[u], &num, &den);
            else exact = 1;
          }
        }
      }
    }
  } else {
    for (int by = by0; by <= by1; ++by) {
      const uint32_t* row = start + (size_t)by * p.nbx;
      const uint32_t s0 = row[bx0], e0 = row[bx1 + 1];
      for (uint32_t k = s0 + lane; k < e0; k += 64) {
        const double dx = qx - pts_x(P, (size_t)k);
        const double dy = qy - pts_y(P, (size_t)k);
        double d2 = dx * dx;
        d2 = d2 + dy * dy;
        if (d2 < T) {
          if (d2 > 0.0) idw_add(d2, pts_z(P, (size_t)k), &num, &den);
          else exact = 1;
        }
      }
    }
  }
  num = wave_sum_d(num);
  den = wave_sum_d(den);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) exact |= __shfl_xor(exact, d, 64);
  if (lane != 0) return;
  if (exact || !(den > 0.0)) {
    // exact hit (dsm.cc:165 CHECK) or an empty first search (the ladder): scalar routine
    cell_global<false>(p, start, P, i, j, o);
    return;
  }
  emit_value(p, o, i, j, num / den);
}

// kVar: 0 = the product kernel.  1, 2, 5, 6 = TIMING PROBES (1: candidate loop without the
// hit updates, 2: no candidate loop -- both give wrong heights; 5 / 6: leave after the staging /
// after the loads): instantiated and selectable (tuning knob f32_variant) ONLY in a build with
// -DAMHIP_TIMING_PROBES (AMHIP_BUILD_DEFINES=-DAMHIP_TIMING_PROBES python -m
// aerial_mapper_amd.build --force); the shipped library holds kVar = 0 alone.
template <int NT, int kTileJ, int kCap, int kVar = 0>
__device__ __forceinline__ void gather_tile_f32(const DsmParams& p, const uint32_t* __restrict__ start,
                                                const Pts P,
                                                const uint8_t* __restrict__ tile_occ, const CellOut& o,
                                                const int tile, unsigned char* smem, int my_class,
                                                int* __restrict__ exact_list,
                                                unsigned* __restrict__ exact_count,
                                                int* __restrict__ big_list = nullptr,
                                                unsigned* __restrict__ big_count = nullptr,
                                                int big_np = 0x7FFFFFFF) {
#ifndef AMHIP_TIMING_PROBES
  static_assert(kVar == 0, "timing probes need -DAMHIP_TIMING_PROBES");
#endif
  constexpr int kWaves = NT / 64;
  constexpr int kCellsPerLane = kTileJ / kWaves;
  // (the 1024-point instance serves ~0.5 points per cell: a trip is ~10 candidates, one run)
  constexpr bool kChunked = kCap > 1024 && kVar == 0;
  constexpr int kFlush = 32;
  // [rec: cap+1 uint4 (U, V, dz, -)][cell offsets][rows][scan][ctl][z range][flags]
  uint4* s_rec = reinterpret_cast<uint4*>(smem);
  uint32_t* s_off = reinterpret_cast<uint32_t*>(smem + (size_t)(p.lds_cap + 2) * 16);
  uint32_t* s_rowg = s_off + p.lds_cells + 1;
  uint32_t* s_rowp = s_rowg + kMaxRegionRows;
  uint32_t* s_scan = s_rowp + kMaxRegionRows + 1;
  uint32_t* s_ctl = s_scan + 24;  // [0] np, [1] nflag, [2..3] pad, [4..7] zmin / zmax keys
  uint32_t* s_zkey = s_ctl + 4;  // [0] min, [1] max of the region's heights (ordered f32 keys)
  uint16_t* s_flag = reinterpret_cast<uint16_t*>(s_ctl + 8);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int ti = tile % p.tiles_i;
  const int tj = tile / p.tiles_i;
  const int i0 = ti * kTileI;
  const int j0 = tj * kTileJ;
  const int i_hi = min(i0 + kTileI, p.rows) - 1;
  const int j_hi = min(j0 + kTileJ, p.cols) - 1;
  const int w0 = p.w[0];

  const int occ = tile_occ[tile];
  if (my_class >= 0 && occ != 0 && occ - 1 != my_class) return;
  if (occ == 0) {
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

  const int rbx0 = (i0 - w0 + p.M) / p.B;
  const int rbx1 = (i_hi + w0 + p.M) / p.B;
  const int rby0 = (j0 - w0 + p.M) / p.B;
  const int rby1 = (j_hi + w0 + p.M) / p.B;
  const int nrb = rby1 - rby0 + 1;
  const int RW = (rbx1 - rbx0 + 1) * p.B;
  const int RH = nrb * p.B;
  const int sh = (j0 - w0 + p.M - (rby0 * p.B)) & 1;
  const int RW2 = 2 * RW;
  const int ncell = (RH / 2 + 2) * RW2;
  const int ox = rbx0 * p.B;
  const int oy = rby0 * p.B;

  // (the cell table is cleared, scanned and rewritten in whole quads: entries 0 .. ncell + 3)
  const bool geom_ok = nrb <= 64 && ncell + 3 <= p.lds_cells;
  // ONE phase: the cell table is cleared (16 bytes per lane) while wave 0 reads the region's bin
  // rows and prefixes their lengths across its lanes (no serial pass, no barrier in between)
  {
    uint4* q = reinterpret_cast<uint4*>(s_off);
    const int nq = geom_ok ? (ncell + 4) >> 2 : 0;
    for (int k = tid; k < nq; k += NT) q[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (wid == 0) {
    uint32_t gs = 0, cnt = 0;
    if (geom_ok && lane < nrb) {
      const uint32_t* row = start + (size_t)(rby0 + lane) * p.nbx;
      gs = row[rbx0];
      cnt = row[rbx1 + 1] - gs;
    }
    const uint32_t incl = wave_incl_scan(cnt, lane);
    if (lane < nrb && geom_ok) {
      s_rowg[lane] = gs;
      s_rowp[lane + 1] = incl;
    }
    if (lane == 63) s_ctl[0] = incl;
    if (lane == 0) {
      s_rowp[0] = 0;
      s_ctl[1] = 0;
      s_zkey[0] = 0xFFFFFFFFu;  // running min of the heights (ordered float keys)
      s_zkey[1] = 0u;           // running max
    }
  }
  __syncthreads();
  const int np = (int)s_ctl[0];
  if (!(geom_ok && np <= p.lds_cap)) {
    // (cannot happen for the class this launch serves; kept for safety) FP64 global path
    for (int c = 0; c < kCellsPerLane; ++c) {
      const int i = i0 + lane, j = j0 + wid * kCellsPerLane + c;
      if (i <= i_hi && j <= j_hi) cell_global<false>(p, start, P, i, j, o);
    }
    return;
  }

  // ---- stage: load the region's 16-byte records (amhip_sort.hip: cell, fixed-point offsets
  // from the cell centre, f32 height offset from zref), count per cell, height range ----------
  static_assert(kCellsPerLane % 2 == 0, "cell pairs must start on even rows of the tile");
  constexpr int kMaxK = (kCap + NT - 1) / NT;
  uint32_t pslot[kMaxK];
  uint32_t pU[kMaxK], pV[kMaxK];
  float pdz[kMaxK];
  float zlo = __builtin_huge_valf(), zhi = -__builtin_huge_valf();
  // ALL the thread's records first (threads past the region's end re-read its last record): with
  // the load inside the loop below, between a row search in LDS and a returning LDS atomic, the
  // compiler waited for each record before it asked for the next -- up to 15 dependent memory
  // round trips per tile in the 7680-point instance.
  uint4 recs[kMaxK];
  if (np > 0) {
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      const int idx = min(tid + k * NT, np - 1);
      int r = 0;
      while (idx >= (int)s_rowp[r + 1]) ++r;
      recs[k] = P.rec[(size_t)s_rowg[r] + (size_t)(idx - (int)s_rowp[r])];
    }
  }
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    const int idx = tid + k * NT;
    pslot[k] = 0xFFFFFFFFu;
    if (idx < np) {
      const uint4 rec = recs[k];
      const float dz = __uint_as_float(rec.w);
      pdz[k] = dz;
      zlo = fminf(zlo, dz);
      zhi = fmaxf(zhi, dz);
      // position in the region's frame, 32-bit fixed point (wrapping: only differences count)
      const int rx = (int)(rec.x & 0xFFFFu) - ox, ry = (int)(rec.x >> 16) - oy;
      pU[k] = ((uint32_t)rx << p.fx_S) + rec.y;
      pV[k] = ((uint32_t)ry << p.fx_S) + rec.z;
      const int ix = min(max(rx, 0), RW - 1);
      int iy = min(max(ry, 0), RH - 1);
      iy += sh;
      const uint32_t cell = (uint32_t)((iy >> 1) * RW2 + 2 * ix + (iy & 1));
      pslot[k] = (cell << 13) | atomicAdd(&s_off[cell], 1u);
    }
  }
  if (kVar == 6) return;
  // height range of the region, in f32 (the budget below widens it by the conversion's error):
  // a wave reduction of floats (6 + 6 DPP min / max instead of 24 ds_bpermute + 34 FP64 ops), one
  // LDS atomic pair per wave on order-preserving keys.  (An atomic pair per THREAD on the one
  // address serialises: measured +0.5 ms.)
  {
    // (+-inf where the thread had no point; six DPP steps each, result in lane 63)
    const float flo = wave_min_to_lane63(zlo), fhi = wave_max_to_lane63(zhi);
    if (lane == 63 && flo <= fhi) {
      atomicMin(&s_zkey[0], zkey_of(flo));
      atomicMax(&s_zkey[1], zkey_of(fhi));
    }
  }
  __syncthreads();
  // ---- cell-table scan: a thread owns consecutive whole quads (one at the usual sizes), a wave
  // their prefix; the last wave works out the tile's error budget while the others wait ----
  const int nq = (ncell + 4) >> 2;  // quads that cover entries 0 .. ncell
  const int qper = (nq + NT - 1) / NT;
  const int q0 = tid * qper, q1 = min(q0 + qper, nq);
  uint4* const qoff = reinterpret_cast<uint4*>(s_off);
  unsigned qsum = 0;
  // (one quad per thread at the usual sizes: unrolled by eight the loop costs the kernel a spill)
#pragma unroll 1
  for (int k = q0; k < q1; ++k) {
    const uint4 v = qoff[k];
    qsum += (v.x + v.y) + (v.z + v.w);
  }
  const unsigned qincl = wave_incl_scan(qsum, lane);
  if (lane == 63) s_scan[wid] = qincl;
  if (wid == kWaves - 1) {
    // ---- the tile's error budget ----------------------------------------------------
    //   (fx_allowed_additions(): the bound the occupancy pre-pass applies to the same heights)
    float zmin_f = 0.f, zmax_f = 0.f;
    if (np) {
      zmin_f = zkey_to_float(s_zkey[0]);
      zmax_f = zkey_to_float(s_zkey[1]);
    }
    double z0w;
    const int na = fx_allowed_additions(zmin_f, zmax_f, p.fx_epsw, P.zref[0], &z0w);
    if (lane == 0) {
      s_ctl[2] = (uint32_t)na;
      s_zkey[2] = (uint32_t)__double2loint(z0w);
      s_zkey[3] = (uint32_t)__double2hiint(z0w);
    }
  }
  __syncthreads();
  {
    const int mywave = __builtin_amdgcn_readfirstlane(wid);
    unsigned run = qincl - qsum;
#pragma unroll
    for (int w = 0; w < kWaves - 1; ++w)
      if (w < mywave) run += s_scan[w];
    for (int k = q0; k < q1; ++k) {
      const uint4 v = qoff[k];
      uint4 e;
      e.x = run;
      e.y = run + v.x;
      e.z = e.y + v.y;
      e.w = e.z + v.z;
      run = e.w + v.w;
      qoff[k] = e;  // (entry ncell, an empty count, receives the total)
    }
  }
  const int n_allowed = (int)s_ctl[2];
  // (z0: the middle of the region's OFFSETS, exactly a float; heights = zref + z0 + N / D)
  const double z0rel = __hiloint2double((int)s_zkey[3], (int)s_zkey[2]);
  const float z0f = (float)z0rel;
  const double z0 = P.zref[0] + z0rel;
  // (a trip of a dense tile brings up to ~20 candidates: below that the FP64 kernel takes
  // the whole tile -- staging it twice is cheaper than redoing most of its cells)
  if (n_allowed < kFxMinAdditions) {
    // (a tile of a larger capacity class may hold more points than the FP64 kernel's LDS
    // image takes: those go to the wave-per-block kernel's list)
    if (tid == 0) {
      if (np > big_np) big_list[atomicAdd(big_count, 1u)] = tile;
      else exact_list[atomicAdd(exact_count, 1u)] = tile;
    }
    return;
  }
  __syncthreads();
  // pass 2: drop the points into their sorted slot
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    if (pslot[k] != 0xFFFFFFFFu) {
      const uint32_t pos = s_off[pslot[k] >> 13] + (pslot[k] & 0x1FFFu);
      s_rec[pos] = make_uint4(pU[k], pV[k], __float_as_uint(pdz[k] - z0f), 0u);
    }
  }
  __syncthreads();

  if (kVar == 5) return;
  // ---- gather -------------------------------------------------------------------
  const int i = i0 + lane;
  const float thi = p.fx_thi, tlo = p.fx_tlo;
  const float one_cell = (float)(1u << p.fx_S);
  if (i <= i_hi) {
    const int ci = i + p.M - ox;
    const uint32_t Ui = (uint32_t)ci << p.fx_S;
    for (int c = 0; c < kCellsPerLane; c += 2) {
      const int jA = j0 + wid * kCellsPerLane + c;
      if (jA > j_hi) break;
      const bool haveB = (jA + 1 <= j_hi) && (c + 1 < kCellsPerLane);
      const float thiB = haveB ? thi : -1.0f;
      const int cj = jA + p.M - oy;
      const uint32_t VjA = (uint32_t)cj << p.fx_S;
      float NA = 0.f, DA = 0.f, NB = 0.f, DB = 0.f;      // totals
      float mA = 0.f, mB = 0.f;                          // largest d2 among the hits
      int nmax = 0;                                      // most candidates of one trip
      const uint32_t* orow = s_off + ((cj - w0 + sh) >> 1) * RW2 + 2 * ci;
      // (every lane takes its spans longest first: see gather_tile)
      uint32_t key0 = 0, key1 = 0, key2 = 0, key3 = 0, key4 = 0, key5 = 0;
      const bool sorted_trips = w0 < kSortTrips && !AMHIP_GATHER_PLAIN_TRIPS;
      if (sorted_trips) {
        uint32_t kb_[kSortTrips], ke_[kSortTrips];   // (all twelve offsets in flight: see gather_tile)
#pragma unroll
        for (int r = 0; r < kSortTrips; ++r) {
          // (static index: the six widths arrive with one scalar load, not one load and wait each)
          const int rr = min(r, w0);
          const int w = r <= w0 ? p.wrp[r] : 0;
          kb_[r] = orow[rr * RW2 - 2 * w];
          ke_[r] = orow[rr * RW2 + 2 * w + 2];
        }
        auto span = [&](int r) __attribute__((always_inline)) -> uint32_t {
          const uint32_t len = r <= w0 ? ke_[r] - kb_[r] : 0u;
          return (len << 16) | kb_[r];
        };
        key0 = span(0), key1 = span(1), key2 = span(2), key3 = span(3), key4 = span(4), key5 = span(5);
        auto cx = [](uint32_t& a, uint32_t& b) __attribute__((always_inline)) {
          const uint32_t hi = max(a, b), lo = min(a, b);
          a = hi;
          b = lo;
        };
        cx(key0, key5), cx(key1, key3), cx(key2, key4);
        cx(key1, key2), cx(key3, key4);
        cx(key0, key3), cx(key2, key5);
        cx(key0, key1), cx(key2, key3), cx(key4, key5);
        cx(key1, key2), cx(key3, key4);
      }
      for (int r = 0; r <= w0; ++r) {
        uint32_t kb, ke;
        if (sorted_trips) {
          kb = key0 & 0xFFFFu;
          ke = kb + (key0 >> 16);
          key0 = key1, key1 = key2, key2 = key3, key3 = key4, key4 = key5;
        } else {
          const int w = p.wrp[r];
          kb = orow[-2 * w];
          ke = orow[2 * w + 2];
          orow += RW2;
        }
        nmax = max(nmax, (int)(ke - kb));
        float nA = 0.f, dA = 0.f, nB = 0.f, dB = 0.f;  // this trip's sums
        // One candidate = 18 f32-rate VALU instructions for the two cells: exact integer
        // differences, conversions, d2 for A and B, and per cell the hit update under the
        // EXEC mask of the test (v_cmpx): w = rcp(d2); m = max(m, d2); d += w; n += w dz.
        // (Left to the compiler the two hits become 8 selects + 2 canonicalising max.)
        // v_max sits between v_rcp and the first use of its result: gfx950 needs one
        // wait state after a transcendental.
        const uint4* pr = s_rec + kb;
        const uint4* const pe = s_rec + (kVar == 2 ? kb : ke);
        auto cand = [&](const uint4 rec) __attribute__((always_inline)) {
          if (kVar == 1) {
            float t0, t1, t3;
            asm volatile(
                "v_sub_u32 %[t0], %[Ui], %[x]\n\t"
                "v_sub_u32 %[t1], %[Vj], %[y]\n\t"
                "v_cvt_f32_i32 %[t0], %[t0]\n\t"
                "v_cvt_f32_i32 %[t1], %[t1]\n\t"
                "v_mul_f32 %[t0], %[t0], %[t0]\n\t"
                "v_add_f32 %[t3], %[one], %[t1]\n\t"
                "v_fma_f32 %[t1], %[t1], %[t1], %[t0]\n\t"
                "v_fma_f32 %[t3], %[t3], %[t3], %[t0]\n\t"
                "v_max_f32 %[mA], %[mA], %[t1]\n\t"
                "v_max_f32 %[mB], %[mB], %[t3]"
                : [t0] "=&v"(t0), [t1] "=&v"(t1), [t3] "=&v"(t3), [mA] "+v"(nA), [mB] "+v"(nB)
                : [Ui] "v"(Ui), [Vj] "v"(VjA), [x] "v"(rec.x), [y] "v"(rec.y), [z] "v"(rec.z),
                  [pad] "v"(rec.w), [one] "v"(one_cell));
            dA = 1.f;
            dB = 1.f;
            return;
          }
          float t0, t1, t3;
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
              : [t0] "=&v"(t0), [t1] "=&v"(t1), [t3] "=&v"(t3), [sv] "=&s"(sv), [mA] "+v"(mA),
                [mB] "+v"(mB), [nA] "+v"(nA), [dA] "+v"(dA), [nB] "+v"(nB), [dB] "+v"(dB)
              : [Ui] "v"(Ui), [Vj] "v"(VjA), [x] "v"(rec.x), [y] "v"(rec.y), [z] "v"(rec.z),
                [pad] "v"(rec.w), [one] "v"(one_cell), [thi] "v"(thi), [thiB] "v"(thiB)
              : "vcc");
        };
        if (kChunked) {
          // denser clouds (the instances with larger LDS images): a trip brings 70 .. 400
          // candidates, and that many additions into one f32 partial sum would eat the tile's
          // error budget -- cells of steep tiles then went to the FP64 routine one by one.
          // Runs of <= kFlush candidates, each flushed into the totals: the additions per
          // accumulator are bounded by n_eff below whatever the density.
          do {
            const uint4* const pc = (pe - pr > kFlush) ? pr + kFlush : pe;
            nA = dA = nB = dB = 0.f;
            for (; pr < pc; ++pr) cand(*pr);
            NA += nA;
            DA += dA;
            NB += nB;
            DB += dB;
          } while (pr < pe);
        } else {
          for (; pr < pe; ++pr) cand(*pr);
          if (kVar == 2) dA = dB = 1.f;
          NA += nA;
          DA += dA;
          NB += nB;
          DB += dB;
        }
      }
      // additions one accumulator has seen at most: a run, then the flushes into the total
      const int n_eff = kChunked ? min(nmax, kFlush) + (w0 + 1) * ((nmax + kFlush - 1) / kFlush)
                                 : nmax + w0 + 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && !haveB) break;
        const float Nn = h ? NB : NA, Dd = h ? DB : DA, mm = h ? mB : mA;
        const int jj = jA + h;
        int queue = 0;  // 1: no first-level neighbour (ladder), 2: redo the cell in FP64
        if (Dd > 0.0f) {
          // (NaN / inf sums fail the first comparison)
          if (kVar != 1 && kVar != 2 &&
              (!(Dd < p.fx_denmax) || mm >= tlo || n_eff > n_allowed)) queue = 2;
          else emit_value(p, o, i, jj, z0 + (double)(Nn * __builtin_amdgcn_rcpf(Dd)));
        } else if (Dd == 0.0f) {
          queue = 1;
        } else {
          queue = 2;
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

  // ---- queued cells: the ladder lane-parallel, the FP64 redos wave by wave ----------
  const int nflag = (int)s_ctl[1];
  for (int f = tid; f < nflag; f += NT) {
    if (s_flag[f] & 0x8000) continue;
    const int code = s_flag[f] & 0x7FFF;
    const int fi = i0 + (code % kTileI);
    const int fj = j0 + (code / kTileI);
    const double fqx = p.base_x + p.res * (-(double)(fi + p.i_off));
    const double fqy = p.base_y + p.res * (-(double)(fj + p.j_off));
    const bool done = cell_fallback_global<false>(p, start, P, fi, fj, fqx, fqy, o);
    if (!done) {
      leave_untouched(p, o, fi, fj);
      if (o.unfilled) atomicAdd(o.unfilled, 1u);
    }
  }
  for (int f = wid; f < nflag; f += kWaves) {
    if (!(s_flag[f] & 0x8000)) continue;
    const int code = s_flag[f] & 0x7FFF;
    cell_wave_exact(p, start, P, i0 + (code % kTileI), j0 + (code / kTileI), o);
  }
}

#ifndef AMHIP_F32_WAVES
#define AMHIP_F32_WAVES 8
#endif
template <int NT, int kTileJ, int kCap, int kVar = 0>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(AMHIP_F32_WAVES, AMHIP_F32_WAVES)))
k_dsm_gather_f32(DsmParams p, const uint32_t* __restrict__ start,
                 const Pts P, const uint8_t* __restrict__ tile_occ,
                 CellOut o, int* __restrict__ exact_list, unsigned* __restrict__ exact_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ntiles = p.tiles_i * p.tiles_j;
  const int b = blockIdx.x;
  const int xcd = b & 7, k = b >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  gather_tile_f32<NT, kTileJ, kCap, kVar>(p, start, P, tile_occ, o, tile, smem, 0, exact_list,
                                          exact_count);
}

// The same without the 64-register ceiling: instances whose threads keep more than two staged
// points each (4096-point images: two workgroups per CU anyway).
template <int NT, int kTileJ, int kCap>
__global__ void __launch_bounds__(NT)
k_dsm_gather_f32_wide(DsmParams p, const uint32_t* __restrict__ start,
                      const Pts P, const uint8_t* __restrict__ tile_occ,
                      CellOut o, int* __restrict__ exact_list, unsigned* __restrict__ exact_count,
                      int* __restrict__ big_list, unsigned* __restrict__ big_count, int big_np) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ntiles = p.tiles_i * p.tiles_j;
  const int b = blockIdx.x;
  const int xcd = b & 7, k = b >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  gather_tile_f32<NT, kTileJ, kCap>(p, start, P, tile_occ, o, tile, smem, 0, exact_list,
                                    exact_count, big_list, big_count, big_np);
}

template <int NT, int kTileJ, int kCap>
__global__ void __launch_bounds__(NT)
k_dsm_gather_f32_list(DsmParams p, const uint32_t* __restrict__ start,
                      const Pts P, const uint8_t* __restrict__ tile_occ,
                      const int* __restrict__ tile_list, const unsigned* __restrict__ tile_count,
                      CellOut o, int* __restrict__ exact_list,
                      unsigned* __restrict__ exact_count, int* __restrict__ big_list,
                      unsigned* __restrict__ big_count, int big_np) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned count = *tile_count;
  for (unsigned k = blockIdx.x; k < count; k += gridDim.x) {
    gather_tile_f32<NT, kTileJ, kCap>(p, start, P, tile_occ, o, tile_list[k], smem, -1,
                                      exact_list, exact_count, big_list, big_count, big_np);
    __syncthreads();
  }
}

// Dense launch: one workgroup per tile.  XCD-aware tile order: consecutive tiles
// (which share halo points) go to the same XCD's L2.  Blocks are dealt
// round-robin to the 8 XCDs, so XCD x gets the x-th contiguous chunk of the tile
// list (bijective for any count).
// 8 waves per SIMD (64 VGPRs; left to itself the compiler takes 65 = 7 waves, i.e. THREE 512-thread
// workgroups per CU): the 64 x 16 / 1024-point image is sized so that FOUR fit a CU's LDS, and the
// fourth hides the other three's staging and epilogue phases -- same box 2.67 -> 2.54 ms per 1e8
// cells (round 5; the loop itself is FP64-issue bound either way).
// (Only that instantiation: the larger images' occupancy is limited by their LDS anyway, and the
// 64-register cap cost them spills -- ADVICE r5.)
template <int NT, int kTileJ, int kCap>
__global__ void __launch_bounds__(NT)
    __attribute__((amdgpu_waves_per_eu((NT == 512 && kTileJ == 16 && kCap == 1024) ? 8 : 1,
                                       (NT == 512 && kTileJ == 16 && kCap == 1024) ? 8 : 10)))
k_dsm_gather_tiled(DsmParams p, const uint32_t* __restrict__ start,
                   const Pts P, const uint8_t* __restrict__ tile_occ,
                   CellOut o, int my_class) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ntiles = p.tiles_i * p.tiles_j;
  const int b = blockIdx.x;
  const int xcd = b & 7, k = b >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  gather_tile<NT, kTileJ, kCap>(p, start, P, tile_occ, o, tile, smem, my_class);
}

// The capped mode's dense launch (every capacity class: a tile beyond the image takes the global bins).
template <int NT, int kTileJ, int kCap, int kSet>
__global__ void __launch_bounds__(NT)
k_dsm_gather_tiled_knn(DsmParams p, const uint32_t* __restrict__ start, const Pts P,
                       const uint8_t* __restrict__ tile_occ, CellOut o) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ntiles = p.tiles_i * p.tiles_j;
  const int b = blockIdx.x;
  const int xcd = b & 7, k = b >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  gather_tile<NT, kTileJ, kCap, kSet>(p, start, P, tile_occ, o, tile, smem, -1);
}

// List launch: a fixed grid walks a list of tiles -- the occupied tiles of a
// sparse call (a small cloud on a large map, e.g. one stereo pair of an
// incremental mapping run) or the tiles of one capacity class.
template <int NT, int kTileJ, int kCap>
__global__ void __launch_bounds__(NT)
k_dsm_gather_tiled_sparse(DsmParams p, const uint32_t* __restrict__ start,
                          const Pts P,
                          const uint8_t* __restrict__ tile_occ, const int* __restrict__ tile_list,
                          const unsigned* __restrict__ tile_count, CellOut o) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned count = *tile_count;
  for (unsigned k = blockIdx.x; k < count; k += gridDim.x) {
    gather_tile<NT, kTileJ, kCap>(p, start, P, tile_occ, o, tile_list[k], smem, -1);
    __syncthreads();
  }
}

// block_wave() in single precision under the guards of gather_tile_f32: the
// candidate's position becomes 32-bit fixed point relative to the block's first cell (units of
// 2^-(fx_S - 1) cells: one bit coarser than the tile kernel's, because the differences formed
// here reach w0 + 4.5 cells and must stay clear of the 32-bit wrap), its height an f32 offset
// from the first candidate's; per (cell, candidate): exact integer difference -> f32 -> d2, the
// reciprocal weight, sums per lane, one butterfly per block.  Cells with an ambiguous hit (d2
// within 2e-6 of the radius), a point nearer than theta, no neighbour at all or non-finite sums
// take cell_global() (the reference's doubles); a block whose height spread leaves no room for
// its additions under the 1e-4 m budget is redone by block_wave() as a whole.
__device__ __forceinline__ void block_wave_f32(const DsmParams& p, const uint32_t* __restrict__ start,
                                               const Pts P, int bi0, int bi1,
                                               int bj0, int bj1, const CellOut& o) {
  const int lane = threadIdx.x & 63;
  const int w = p.w[0];
  const int S = p.fx_S - 1;
  const float thi = p.fx_thi * 0.25f, tlo = p.fx_tlo * 0.25f;  // (squared scale: one bit -> 1/4)
  const float denmax = p.fx_denmax * 4.0f;
  const int bx0 = (bi0 - w + p.M) / p.B, bx1 = (bi1 + w + p.M) / p.B;
  const int by0 = (bj0 - w + p.M) / p.B, by1 = (bj1 + w + p.M) / p.B;
  // reference height: the first candidate's f32 offset from zref (wave-uniform)
  float z0f = 0.0f;
  unsigned ncand = 0;
  for (int by = by0; by <= by1; ++by) {
    const uint32_t* row = start + (size_t)by * p.nbx;
    const uint32_t s0 = row[bx0], e0 = row[bx1 + 1];
    if (e0 > s0 && ncand == 0) z0f = __uint_as_float(P.rec[s0].w);
    ncand += e0 - s0;
  }
  const double z0 = P.zref[0] + (double)z0f;
  float num[16], den[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) num[c] = den[c] = 0.0f;
  unsigned amb = 0;      // bit c: cell c has a hit inside the band around the radius
  float zspread = 0.0f;  // max |z - z0| over the lane's candidates
  // block origin in the records' cell frame (window cells + margin M)
  const int ox = bi0 + p.M, oy = bj0 + p.M;
  for (int by = by0; by <= by1; ++by) {
    const uint32_t* row = start + (size_t)by * p.nbx;
    const uint32_t s0 = row[bx0], e0 = row[bx1 + 1];
    // (no prefetch of the next candidate as in block_wave: four more live registers cost this
    // instance more in spills than the overlap gains -- 3.27 -> 3.41 ms at 8 points per cell)
    for (uint32_t k = s0 + lane; k < e0; k += 64) {
      const uint4 rec = P.rec[k];
      const float zf = __uint_as_float(rec.w) - z0f;
      zspread = fmaxf(zspread, fabsf(zf));
      // position relative to the block's first cell in units of 2^-S cells (S = fx_S - 1: the
      // record's offset loses its last bit, rounded to nearest)
      const uint32_t U = ((uint32_t)((int)(rec.x & 0xFFFFu) - ox) << S) + (uint32_t)(((int)rec.y + 1) >> 1);
      const uint32_t V = ((uint32_t)((int)(rec.x >> 16) - oy) << S) + (uint32_t)(((int)rec.z + 1) >> 1);
      float dx2[4], dy2[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        // (cells beyond the block repeat nothing: they are computed and never written)
        const float dx = (float)(int)(((uint32_t)a << S) - U);
        const float dy = (float)(int)(((uint32_t)a << S) - V);
        dx2[a] = dx * dx;
        dy2[a] = dy * dy;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float d2 = dx2[c & 3] + dy2[c >> 2];
        const bool hit = d2 < thi;
        const float r = hit ? __builtin_amdgcn_rcpf(d2) : 0.0f;
        den[c] += r;
        num[c] = fmaf(r, zf, num[c]);
        amb |= (hit & (d2 >= tlo)) ? (1u << c) : 0u;
      }
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    amb |= __shfl_xor(amb, d, 64);
    zspread = fmaxf(zspread, __shfl_xor(zspread, d, 64));
  }
  // ---- the block's error budget (as in gather_tile_f32): additions per accumulator = the lane's
  // candidates + the six levels of the butterfly
  {
    const float zabs = fabsf((float)z0) + zspread;
    const float zabs_lo = fmaxf(fabsf((float)z0) - zspread, 0.0f);
    const float ulp = __uint_as_float((__float_as_uint(fmaxf(zabs, 1e-30f)) & 0x7F800000u)) * 1.1920929e-7f;
    const float ulp_lo = __uint_as_float((__float_as_uint(fmaxf(zabs_lo, 1e-30f)) & 0x7F800000u)) * 1.1920929e-7f;
    // (minus the records' own rounding of z - zref: half a spacing of the offset at most)
    const float allowed = 0.8f * (ulp < 1e-4f * 0.6f ? 1e-4f - ulp : 0.25f * ulp_lo) -
                          (fabsf(z0f) + zspread) * 6.0e-8f;
    const float S_half = zspread * 1.000001f + (fabsf(z0f) + zspread) * 1.2e-7f;
    // (positions one bit coarser, and rounded twice: three times the tile kernel's quantum term)
    const float epsw = 3.0f * (p.fx_epsw - 4e-7f) + 4e-7f;
    const float n_add = (float)((ncand + 63u) / 64u + 8u);
    const bool ok = (S_half <= 3.0e38f) && allowed > 0.0f &&
                    (S_half == 0.0f || 2.0f * (epsw + (n_add + 3.0f) * 5.9604645e-8f) * S_half <= allowed);
    if (!ok) {  // (wave-uniform)
      block_wave(p, start, P, bi0, bi1, bj0, bj1, o);
      return;
    }
  }
  // butterfly with halving (see block_wave): lane l ends with value l >> 1
  float v[32];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    v[c] = num[c];
    v[16 + c] = den[c];
  }
#pragma unroll
  for (int half = 16, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
    const bool up = (lane & bit) != 0;
#pragma unroll
    for (int k = 0; k < half; ++k) {
      const float send = up ? v[k] : v[k + half];
      const float keep = up ? v[k + half] : v[k];
      v[k] = keep + __shfl_xor(send, bit, 64);
    }
  }
  v[0] += __shfl_xor(v[0], 1, 64);
  const float my_num = v[0];
  const float my_den = __shfl(v[0], (lane & 31) | 32, 64);
  const int cidx = (lane & 31) >> 1;
  const int a = cidx & 3, bq = cidx >> 2;
  const bool owner = lane < 32 && !(lane & 1) && bi0 + a <= bi1 && bj0 + bq <= bj1;
  // an ambiguous hit, a point nearer than theta (the weight sum says so; rcp(0) = inf), no
  // neighbour (the ladder) or non-finite sums: the reference's doubles decide -- a whole wave
  // per such cell (cell_wave_exact), not one lane over hundreds of candidates
  const bool redo = owner && (((amb >> cidx) & 1u) || !(my_den > 0.0f) || !(my_den < denmax) ||
                              !(my_num == my_num));
  if (owner && !redo)
    emit_value(p, o, bi0 + a, bj0 + bq, z0 + (double)(my_num * __builtin_amdgcn_rcpf(my_den)));
  unsigned long long todo = __ballot(redo);
  while (todo) {  // (wave-uniform)
    const int l = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int cc = (l & 31) >> 1;
    cell_wave_exact(p, start, P, bi0 + (cc & 3), bj0 + (cc >> 2), o);
  }
}

// Class-3 tiles (more points than any LDS image holds): a fixed grid walks their
// list; the four waves of a workgroup share a tile's blocks of 4 x 4 cells
// (block_wave).  No LDS, its own register budget: three waves per SIMD.  (The single-precision
// instance carries the FP64 routines for its redos besides its own 16 x 2 accumulators and
// spills 18 registers at three waves, all outside the candidate loop; at two waves it spills
// nothing and is 18 % slower -- 8 / 16 points per cell: 4.12 / 4.52 ms against 3.48 / 3.70.)
#ifndef AMHIP_DENSE_F32_WAVES
#define AMHIP_DENSE_F32_WAVES 3
#endif
template <bool kF32>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kF32 ? AMHIP_DENSE_F32_WAVES : 3)))
k_dsm_gather_dense(DsmParams p, int tile_j, const uint32_t* __restrict__ start,
                   const Pts P, const int* __restrict__ tile_list,
                   const unsigned* __restrict__ tile_count, CellOut o) {
  const unsigned count = *tile_count;
  const int wid = threadIdx.x >> 6;
  for (unsigned t = blockIdx.x; t < count; t += gridDim.x) {
    const int tile = tile_list[t];
    const int ti = tile % p.tiles_i, tj = tile / p.tiles_i;
    const int i0 = ti * kTileI, j0 = tj * tile_j;
    const int i_hi = min(i0 + kTileI, p.rows) - 1;
    const int j_hi = min(j0 + tile_j, p.cols) - 1;
    // blocks = cells of one bin, cut into pieces of <= 4 x 4, clipped to the tile
    const int gb = min(p.B, 4);
    const int nbi = (i_hi - i0) / gb + 2, nbj = (j_hi - j0) / gb + 2;  // (upper bounds)
    int idx = 0;
    for (int bj = ((j0 + p.M) / p.B) * p.B - p.M; bj <= j_hi; bj += p.B)
      for (int sj = 0; sj < p.B; sj += gb)
        for (int bi = ((i0 + p.M) / p.B) * p.B - p.M; bi <= i_hi; bi += p.B)
          for (int si = 0; si < p.B; si += gb, ++idx) {
            if ((idx & 3) != wid) continue;
            const int a0 = max(bi + si, i0), a1 = min(min(bi + si + gb - 1, bi + p.B - 1), i_hi);
            const int b0 = max(bj + sj, j0), b1 = min(min(bj + sj + gb - 1, bj + p.B - 1), j_hi);
            if (a0 > a1 || b0 > b1) continue;
            if (kF32) block_wave_f32(p, start, P, a0, a1, b0, b1, o);
            else block_wave(p, start, P, a0, a1, b0, b1, o);
          }
    (void)nbi;
    (void)nbj;
  }
}

// (calls without the LDS-tiled gather: no prologue to ride on)
__global__ void __launch_bounds__(256)
k_range_reduce(const double* __restrict__ part, size_t nparts, unsigned long long* __restrict__ range,
               unsigned long long* __restrict__ call_range) {
  __shared__ double s_pair[32];
  range_reduce_block(part, nparts, range, call_range, blockIdx.x, gridDim.x, s_pair);
}

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
int dsm_run(Ctx* c, const double* dev_xyz, const int32_t* dev_values, size_t n,
            const DsmParams& p, float* out, unsigned char* mask, unsigned* unfilled,
            bool fill_untouched, float init_value, unsigned long long* zrange,
            const SortSplit* split) {
  const CellOut cell_out = {out, mask, unfilled, c->dev_err, fill_untouched ? 1 : 0, init_value,
                            c->dev_zrange + 2};
  // any other sort on this context overwrites the histogram rows a pending
  // amhip_dsm_tiled_begin_dev left behind: its finish call then fails instead of mis-binning
  if (!split) c->tiled_pending = false;
  // (the counters of the gather's tile lists are zeroed by the sort's single-workgroup kernel on
  // its way -- SortAux --, so the list buffer has to exist before the sort is enqueued)
  c->aux_zero_words = nullptr;
  c->aux_nzero = 0;
  if (p.lds_ok) {
    const size_t ntiles0 = (size_t)p.tiles_i * (size_t)p.tiles_j;
    int rc;
    if ((rc = ensure_capacity(&c->tile_list, &c->tile_list_cap, kNumLists * ntiles0 + kListHdr))) return rc;
    c->aux_zero_words = reinterpret_cast<uint32_t*>(c->tile_list);
    c->aux_nzero = kListHdr;
  }
  {
    const int rc = dsm_sort(c, dev_xyz, dev_values, n, p, zrange, split);
    if (rc) return rc;
  }
  if (split && split->phase == 1) return AMHIP_OK;  // (tiled call: the rest follows the exchange)
  {
    ScopedTimer t(c, AMHIP_K_DSM_GATHER);
    const PtsView pts_view = c->pts;   // what dsm_sort left: doubles, records or both
    const int64_t prev_ntiles = c->last_ntiles;
    c->last_ntiles = 0;
    c->last_call_f32 = false;
    if (p.lds_ok) {
      const unsigned ntiles = (unsigned)p.tiles_i * (unsigned)p.tiles_j;
      c->last_ntiles = ntiles;
      {
        int rc;
        if ((rc = ensure_capacity(&c->tile_occ, &c->tile_occ_cap, (size_t)ntiles + 16))) return rc;
      }
      // Sparse call (few points per map cell, the layer already materialized):
      // only the occupied tiles are visited, by a fixed grid walking their list.
      const bool sparse = !fill_untouched && ntiles > 8192 &&
                          (double)n * 16.0 < (double)p.rows * (double)p.cols;
      // Capacity classes (k_dsm_tile_occupancy): tiles whose first-level region
      // holds more points than the main launch's LDS image go to list launches
      // with 2x / 4x the capacity (fewer workgroups per CU, still LDS-resident);
      // beyond that gather_tile() takes the global-memory path, as before.
      // Capacities follow the workgroups a CU's 160 KB of LDS can hold: 1024
      // points -> 4 per CU; class 1 = the most that still leaves two per CU;
      // class 2 = everything one workgroup can take.
      const int cap0 = p.lds_cap;
      // (the rest of the LDS image -- cell table, row tables, flags -- grows with the
      // search window: wide radii / coarse grids leave less room for points; a class
      // that cannot hold more than the one below it stays empty)
      const long fixed_bytes = (long)p.lds_bytes - ((long)cap0 + 2) * 24;
      // (even: the cell table behind the points is read and written in 16-byte quads)
      auto cap_fit = [&](long limit) { return (int)((limit - fixed_bytes) / 24 - 2) & ~1; };
      // (cap0 > 2048: a single-precision main launch -- the FP64 images keep their own sizes)
      const int cap0_64 = std::min(cap0, 2048);
      const int cap1 = std::max(cap0_64, std::min(p.tile_j == 16 ? 2752 : 2432, cap_fit(80 * 1024)));
      const int cap2 = std::max(cap1, std::min(p.tile_j == 16 ? 5600 : 5200, cap_fit(150 * 1024)));
      {
        int rc;
        if ((rc = ensure_capacity(&c->tile_list, &c->tile_list_cap,
                                  kNumLists * (size_t)ntiles + kListHdr)))
          return rc;
      }
      unsigned* tile_count = reinterpret_cast<unsigned*>(c->tile_list);
      int* const lists = c->tile_list;
      const bool f32 = p.fx_ok && !p.pcl_mode && !p.only_unfilled && !mask && !unfilled;
      // Single-precision mode: its records take 16 bytes against the FP64 kernel's 24, so the
      // same two LDS budgets (two workgroups per CU / one) hold more points: classes 1 and 2
      // are cut at capf1 / capf2 and walked by list launches of the single-precision kernel
      // (4096- / 7680-point register instances); what those hand back goes to the FP64
      // kernel's largest image (list 5) or, beyond it, to the wave-per-block kernel (list 6).
      const long fixed32 = (long)p.lds_bytes_f32 - ((long)cap0 + 2) * 16;
      auto cap_fit32 = [&](long limit) { return (int)((limit - fixed32) / 16 - 2) & ~1; };
      const int capf1 = std::max(cap0, std::min(4096, cap_fit32(80 * 1024)));
      const int capf2 = std::max(capf1, std::min(7680, cap_fit32(150 * 1024)));
      int ccap0 = cap0, ccap1 = f32 ? capf1 : cap1, ccap2 = f32 ? capf2 : cap2;  // classification
#ifdef AMHIP_TIMING_PROBES
      ccap0 = (int)tuning("gather_class_cap0", ccap0);   // (debugging)
      ccap1 = (int)tuning("gather_class_cap1", ccap1);
      ccap2 = (int)tuning("gather_class_cap2", ccap2);
#endif
      // single-precision mode after the three-pass sort: tiles whose height range leaves no room
      // under the error bound go straight onto the FP64 list the kernel would hand them to --
      // list 4 while the FP64 kernel's image of cap0 points fits a CU (rej_own), else list 5, or
      // list 6 beyond its largest image
      const bool rej_own = cap0 <= 2048 || ((long)cap0 + 2) * 24 + fixed_bytes <= 150 * 1024;
      const uint2* bin_z = (f32 && c->bin_z_valid) ? reinterpret_cast<const uint2*>(c->bin_z) : nullptr;
      c->last_call_f32 = bin_z != nullptr;  // (its tile counters say how rough the scene is)
      // Rough scenes: when the PREVIOUS call on this context pre-classified more than a tenth
      // of its tiles for the FP64 kernel, that kernel is launched densely over the tiles (one
      // workgroup each, filtering on occ) instead of walking a list of tens of thousands with a
      // fixed grid; a tenth or less: the list (an empty dense launch would cost 0.18 ms of
      // idle workgroups on smooth terrain).  Either is correct for any count -- the counts only
      // pick the faster one; they come from a pinned mirror the previous call filled (no
      // synchronisation: a value one call late is as good).
      bool rej_dense = false;
      if (bin_z && cap0 <= 2048 && !sparse && c->host_tile_stats && prev_ntiles > 0) {
        const volatile unsigned* hs = c->host_tile_stats;
        const double rejected = (double)hs[4] + (double)hs[5] + (double)hs[6] + (double)hs[7];
        rej_dense = rejected * 10.0 > (double)prev_ntiles;
      }
      // The capacity-class launches behind the main one (denser tiles: classes 1, 2, the wave-per-
      // block kernel) are SKIPPED when the previous call of this kind left all three lists empty
      // (pinned counters, never waited for): the main launch then takes every class -- a stray
      // denser tile walks the global bins (gather_tile: use_lds false), slower, same result -- and
      // the counters this call leaves bring the class launches back for the next one.
      unsigned long long tsig = 1469598103934665603ull;
      for (const int v : {p.rows, p.cols, p.i_off, p.j_off, p.tile_j, p.lds_cap, p.B, p.M, p.w[0], p.pcl_mode, cap1, cap2})
        tsig = (tsig ^ (unsigned long long)(unsigned)v) * 1099511628211ull;
      const volatile unsigned* hst = c->host_tile_stats;
      const bool skip_classes = !f32 && !sparse && hst && c->tile_stats_sig == tsig && hst[1] == 0u && hst[2] == 0u &&
                                hst[3] == 0u && !no_launch_skips();
      c->tile_stats_sig = tsig;
      const int main_class = skip_classes ? -1 : 0;
      ProloguePart pp;
      pp.zpart = c->range_parts ? c->zpart : nullptr;
      pp.nparts = c->range_parts;
      pp.range = c->range_running;
      pp.call_range = c->dev_zrange + 2;
      pp.nocc = (ntiles + 255) / 256;
      pp.nrange = c->range_parts ? 32u : 0u;
      pp.ticket = c->dev_tickets + 1;
      // (single-precision mode: its gather kernels still append to lists 4 .. 6 -- copied at the end)
      pp.host_stats = f32 ? nullptr : c->host_tile_stats;
      c->range_parts = 0;
      hipLaunchKernelGGL(k_dsm_tile_occupancy, dim3(pp.nocc + pp.nrange), dim3(256), 0, c->stream,
                         p, p.tile_j, c->bin_start, c->tile_occ, lists, sparse ? 1 : 0, ccap0, ccap1, ccap2,
                         bin_z, rej_own ? 1 : 0, cap2, rej_dense ? 1 : 0, pts_view.zref, pp);
      if (p.knn_k > 0) {
        // the capped mode: one dense launch over the same LDS image (no capacity classes: a
        // denser tile takes the global bins cell by cell)
#define AMHIP_LAUNCH_KNN_(TJ_, CAP_, SET_)                                                       \
  do {                                                                                        \
    AMHIP_TRY(hipFuncSetAttribute(                                                            \
        reinterpret_cast<const void*>(k_dsm_gather_tiled_knn<512, TJ_, CAP_, SET_>),          \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes));                       \
    hipLaunchKernelGGL((k_dsm_gather_tiled_knn<512, TJ_, CAP_, SET_>), dim3(ntiles), dim3(512), \
                       p.lds_bytes, c->stream, p, c->bin_start, pts_view, c->tile_occ,       \
                       cell_out);                                                             \
  } while (0)
  // (a set of 4 registers for k <= 4 measured the same as the set of 8: the shift's cost follows k)
#define AMHIP_LAUNCH_KNN(TJ_, CAP_) AMHIP_LAUNCH_KNN_(TJ_, CAP_, kMaxKnn)
        if (p.tile_j == 16 && cap0 == 1024) AMHIP_LAUNCH_KNN(16, 1024);
        else if (p.tile_j == 16 && cap0 == 2048) AMHIP_LAUNCH_KNN(16, 2048);
        else if (p.tile_j == 32 && cap0 == 2048) AMHIP_LAUNCH_KNN(32, 2048);
        else return arg_failure("internal: no capped-mode gather for this tile shape");
#undef AMHIP_LAUNCH_KNN
#undef AMHIP_LAUNCH_KNN_
        AMHIP_TRY(hipGetLastError());
        return AMHIP_OK;
      }
      // tuning knob gather_nt: threads per gather workgroup (tuning knob; 512 measured best)
#ifdef AMHIP_TIMING_PROBES
      const int nt = (int)tuning("gather_nt", 512.0);
#else
      constexpr int nt = 512;
      (void)nt;
#endif
      auto with_cap = [&](int cap) {
        DsmParams q = p;
        q.lds_cap = cap;
        q.lds_bytes = (unsigned)((int)p.lds_bytes + (cap - cap0) * 24);
        return q;
      };
#define AMHIP_LAUNCH_DENSE(NT_, TJ_, CAP_)                                                    \
  do {                                                                                        \
    AMHIP_TRY(hipFuncSetAttribute(                                                            \
        reinterpret_cast<const void*>(k_dsm_gather_tiled<NT_, TJ_, CAP_>),                    \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes));                       \
    hipLaunchKernelGGL((k_dsm_gather_tiled<NT_, TJ_, CAP_>), dim3(ntiles), dim3(NT_),         \
                       p.lds_bytes, c->stream, p, c->bin_start, pts_view, c->tile_occ,       \
                       cell_out, main_class);                                                 \
  } while (0)
      // the FP64 kernel, one workgroup per tile, over the tiles the pre-pass classified for it
      // (occ = 1 + 4) and nothing else: hardware dispatch and the XCD-aware order instead of a
      // fixed grid walking list 4 (28 ns per tile against 37, and no serial list)
#define AMHIP_LAUNCH_REJECTED_DENSE(TJ_, CAP_)                                                \
  do {                                                                                        \
    const DsmParams q = with_cap(cap0);                                                       \
    AMHIP_TRY(hipFuncSetAttribute(                                                            \
        reinterpret_cast<const void*>(k_dsm_gather_tiled<512, TJ_, CAP_>),                    \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds_bytes));                       \
    hipLaunchKernelGGL((k_dsm_gather_tiled<512, TJ_, CAP_>), dim3(ntiles), dim3(512),         \
                       q.lds_bytes, c->stream, q, c->bin_start, pts_view, c->tile_occ,       \
                       cell_out, 4);                                                          \
  } while (0)
      // FP64 list launch: list LIST_ with an LDS image of CAPV_ points (CAP_ sizes the
      // instance's registers; the LDS image holds what fits)
#define AMHIP_LAUNCH_LIST_EX(NT_, TJ_, CAP_, CAPV_, LIST_, GRID_)                             \
  do {                                                                                        \
    const DsmParams q = with_cap(CAPV_);                                                      \
    AMHIP_TRY(hipFuncSetAttribute(                                                            \
        reinterpret_cast<const void*>(k_dsm_gather_tiled_sparse<NT_, TJ_, CAP_>),             \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds_bytes));                       \
    hipLaunchKernelGGL((k_dsm_gather_tiled_sparse<NT_, TJ_, CAP_>), dim3(GRID_), dim3(NT_),   \
                       q.lds_bytes, c->stream, q, c->bin_start, pts_view, c->tile_occ,       \
                       lists + kListHdr + (size_t)(LIST_) * ntiles, tile_count + (LIST_),     \
                       cell_out);                                                             \
  } while (0)
#define AMHIP_LAUNCH_LIST(NT_, TJ_, CAP_, CLS_, GRID_)                                        \
  AMHIP_LAUNCH_LIST_EX(NT_, TJ_, CAP_,                                                        \
                       ((CLS_) == 0 ? cap0 : std::min((int)(CAP_), (CLS_) == 1 ? cap1 : cap2)), \
                       CLS_, GRID_)
      // class-0 tiles in single precision (dense or list 0), then the tiles it handed back
      // (list 4) through the FP64 kernel
#define AMHIP_LAUNCH_F32(TJ_, CAP_)                                                           \
  do {                                                                                        \
    int* const xl = lists + kListHdr + (size_t)4 * ntiles;                                    \
    if (sparse) {                                                                             \
      AMHIP_TRY(hipFuncSetAttribute(                                                          \
          reinterpret_cast<const void*>(k_dsm_gather_f32_list<512, TJ_, CAP_>),               \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes_f32));                 \
      hipLaunchKernelGGL((k_dsm_gather_f32_list<512, TJ_, CAP_>), dim3(8192), dim3(512),      \
                         p.lds_bytes_f32, c->stream, p, c->bin_start, pts_view, c->tile_occ, \
                         lists + kListHdr, tile_count, cell_out, xl, tile_count + 4,          \
                         (int*)nullptr, (unsigned*)nullptr, 0x7FFFFFFF);                      \
    } else if (AMHIP_PROBE_SELECTED(TJ_, CAP_)) {                                             \
      AMHIP_F32_DENSE(16, 1024, f32_variant);                                                 \
    } else {                                                                                  \
      AMHIP_TRY(hipFuncSetAttribute(                                                          \
          reinterpret_cast<const void*>(k_dsm_gather_f32<512, TJ_, CAP_>),                    \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes_f32));                 \
      hipLaunchKernelGGL((k_dsm_gather_f32<512, TJ_, CAP_>), dim3(ntiles), dim3(512),         \
                         p.lds_bytes_f32, c->stream, p, c->bin_start, pts_view, c->tile_occ, \
                         cell_out, xl, tile_count + 4);                                       \
    }                                                                                         \
    if (rej_dense) AMHIP_LAUNCH_REJECTED_DENSE(TJ_, CAP_);                                    \
    AMHIP_LAUNCH_LIST_EX(512, TJ_, CAP_, cap0, 4, 4096);                                      \
  } while (0)
      // (rejected tiles: the FP64 kernel with an image of cap0 points while that fits a CU --
      // list 4 --, else its largest image -- list 5 -- or the wave-per-block kernel -- list 6)
#define AMHIP_LAUNCH_F32_WIDE(TJ_, CAP_)                                                      \
  do {                                                                                        \
    const bool own = ((long)cap0 + 2) * 24 + fixed_bytes <= 150 * 1024;                       \
    int* const xl = lists + kListHdr + (size_t)(own ? 4 : 5) * ntiles;                        \
    unsigned* const xc = tile_count + (own ? 4 : 5);                                          \
    int* const bl = lists + kListHdr + (size_t)6 * ntiles;                                    \
    const int bnp = own ? 0x7FFFFFFF : cap2;                                                  \
    if (sparse) {                                                                             \
      AMHIP_TRY(hipFuncSetAttribute(                                                          \
          reinterpret_cast<const void*>(k_dsm_gather_f32_list<512, TJ_, CAP_>),               \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes_f32));                 \
      hipLaunchKernelGGL((k_dsm_gather_f32_list<512, TJ_, CAP_>), dim3(8192), dim3(512),      \
                         p.lds_bytes_f32, c->stream, p, c->bin_start, pts_view, c->tile_occ, \
                         lists + kListHdr, tile_count, cell_out, xl, xc, bl, tile_count + 6,  \
                         bnp);                                                                \
    } else {                                                                                  \
      AMHIP_TRY(hipFuncSetAttribute(                                                          \
          reinterpret_cast<const void*>(k_dsm_gather_f32_wide<512, TJ_, CAP_>),               \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes_f32));                 \
      hipLaunchKernelGGL((k_dsm_gather_f32_wide<512, TJ_, CAP_>), dim3(ntiles), dim3(512),    \
                         p.lds_bytes_f32, c->stream, p, c->bin_start, pts_view, c->tile_occ, \
                         cell_out, xl, xc, bl, tile_count + 6, bnp);                          \
    }                                                                                         \
    if (own) AMHIP_LAUNCH_LIST_EX(512, TJ_, 4096, cap0, 4, 4096);                             \
  } while (0)
#ifdef AMHIP_TIMING_PROBES
      const int f32_variant = (int)tuning("f32_variant", 0.0);
#define AMHIP_PROBE_SELECTED(TJ_, CAP_) (f32_variant && (TJ_) == 16 && (CAP_) == 1024)
#else
      constexpr int f32_variant = 0;
      (void)f32_variant;
#define AMHIP_PROBE_SELECTED(TJ_, CAP_) false
#endif
      // (timing probes of the 64 x 16 / 1024-point instance: -DAMHIP_TIMING_PROBES builds only)
#define AMHIP_F32_DENSE_V(V_)                                                                 \
  do {                                                                                        \
    AMHIP_TRY(hipFuncSetAttribute(                                                            \
        reinterpret_cast<const void*>(k_dsm_gather_f32<512, 16, 1024, V_>),                   \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes_f32));                   \
    hipLaunchKernelGGL((k_dsm_gather_f32<512, 16, 1024, V_>), dim3(ntiles), dim3(512),        \
                       p.lds_bytes_f32, c->stream, p, c->bin_start, pts_view, c->tile_occ,   \
                       cell_out, lists + kListHdr + (size_t)4 * ntiles, tile_count + 4);      \
  } while (0)
#ifdef AMHIP_TIMING_PROBES
#define AMHIP_F32_DENSE(TJ_, CAP_, VAR_)                                                      \
  do {                                                                                        \
    if ((VAR_) == 1) AMHIP_F32_DENSE_V(1);                                                    \
    else if ((VAR_) == 2) AMHIP_F32_DENSE_V(2);                                               \
    else if ((VAR_) == 5) AMHIP_F32_DENSE_V(5);                                               \
    else AMHIP_F32_DENSE_V(6);                                                                \
  } while (0)
#else
#define AMHIP_F32_DENSE(TJ_, CAP_, VAR_) do { } while (0)
#endif
      // (tile height, LDS point capacity) picked by make_dsm_params from the
      // cloud's mean density: 64x16 / 1024 points runs 4 workgroups per CU
      // single-precision list launch of class CLS_ (1, 2): image of CAPV_ points, register
      // instance CAP_; rejected tiles -> list 5 (FP64, image cap2) / list 6 (wave per block)
#define AMHIP_LAUNCH_F32_CLASS(TJ_, CAP_, CAPV_, CLS_, GRID_)                                 \
  do {                                                                                        \
    DsmParams q = p;                                                                          \
    q.lds_cap = (CAPV_);                                                                      \
    q.lds_bytes_f32 = (unsigned)((int)p.lds_bytes_f32 + ((CAPV_) - cap0) * 16);               \
    AMHIP_TRY(hipFuncSetAttribute(                                                            \
        reinterpret_cast<const void*>(k_dsm_gather_f32_list<512, TJ_, CAP_>),                 \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds_bytes_f32));                   \
    hipLaunchKernelGGL((k_dsm_gather_f32_list<512, TJ_, CAP_>), dim3(GRID_), dim3(512),       \
                       q.lds_bytes_f32, c->stream, q, c->bin_start, pts_view, c->tile_occ,   \
                       lists + kListHdr + (size_t)(CLS_) * ntiles, tile_count + (CLS_),       \
                       cell_out, lists + kListHdr + (size_t)5 * ntiles, tile_count + 5,       \
                       lists + kListHdr + (size_t)6 * ntiles, tile_count + 6, cap2);          \
  } while (0)
      if (p.tile_j == 16 && cap0 == 1024) {
        if (f32) AMHIP_LAUNCH_F32(16, 1024);
        else if (sparse) AMHIP_LAUNCH_LIST(512, 16, 1024, 0, 8192);
#ifdef AMHIP_TIMING_PROBES
        else if (nt == 256) AMHIP_LAUNCH_DENSE(256, 16, 1024);
#endif
        else AMHIP_LAUNCH_DENSE(512, 16, 1024);
      } else if (p.tile_j == 16 && cap0 > 2048) {
        // (make_dsm_params picks 4096 / 7680 only for the single-precision mode)
        if (!f32) return arg_failure("internal: 4096-point tiles outside the single-precision mode");
        if (cap0 == 4096) AMHIP_LAUNCH_F32_WIDE(16, 4096);
        else AMHIP_LAUNCH_F32_WIDE(16, 7680);
      } else if (p.tile_j == 16) {
        if (f32) AMHIP_LAUNCH_F32(16, 2048);
        else if (sparse) AMHIP_LAUNCH_LIST(512, 16, 2048, 0, 8192);
        else AMHIP_LAUNCH_DENSE(512, 16, 2048);
      } else {
        if (f32) AMHIP_LAUNCH_F32(32, 2048);
        else if (sparse) AMHIP_LAUNCH_LIST(512, 32, 2048, 0, 8192);
#ifdef AMHIP_TIMING_PROBES
        else if (nt == 256) AMHIP_LAUNCH_DENSE(256, 32, 2048);
        else if (nt == 1024) AMHIP_LAUNCH_DENSE(1024, 32, 2048);
#endif
        else AMHIP_LAUNCH_DENSE(512, 32, 2048);
      }
      if (skip_classes) {
        // (nothing: the main launch took every class)
      } else if (p.tile_j == 16) {
        if (f32) {
          AMHIP_LAUNCH_F32_CLASS(16, 4096, capf1, 1, 2048);
          AMHIP_LAUNCH_F32_CLASS(16, 7680, capf2, 2, 1024);
          AMHIP_LAUNCH_LIST_EX(512, 16, 5600, cap2, 5, 1024);
        } else {
          AMHIP_LAUNCH_LIST(512, 16, 2752, 1, 2048);
          AMHIP_LAUNCH_LIST(512, 16, 5600, 2, 1024);
        }
      } else {
        if (f32) {
          AMHIP_LAUNCH_F32_CLASS(32, 4096, capf1, 1, 2048);
          AMHIP_LAUNCH_F32_CLASS(32, 7680, capf2, 2, 1024);
          AMHIP_LAUNCH_LIST_EX(512, 32, 5200, cap2, 5, 1024);
        } else {
          AMHIP_LAUNCH_LIST(512, 32, 2432, 1, 2048);
          AMHIP_LAUNCH_LIST(512, 32, 5200, 2, 1024);
        }
      }
#undef AMHIP_LAUNCH_F32_CLASS
#undef AMHIP_LAUNCH_DENSE
#undef AMHIP_LAUNCH_LIST
#undef AMHIP_LAUNCH_LIST_EX
#undef AMHIP_LAUNCH_F32
#undef AMHIP_LAUNCH_F32_WIDE
#undef AMHIP_F32_DENSE
#undef AMHIP_F32_DENSE_V
#undef AMHIP_PROBE_SELECTED
      if (f32)
        hipLaunchKernelGGL(k_dsm_gather_dense<true>, dim3(4096), dim3(256), 0, c->stream, p, p.tile_j,
                           c->bin_start, pts_view, lists + kListHdr + (size_t)3 * ntiles,
                           tile_count + 3, cell_out);
      else if (!skip_classes)
        hipLaunchKernelGGL(k_dsm_gather_dense<false>, dim3(4096), dim3(256), 0, c->stream, p, p.tile_j,
                           c->bin_start, pts_view, lists + kListHdr + (size_t)3 * ntiles,
                           tile_count + 3, cell_out);
      // (list 6: tiles the single-precision list launches handed back for their height spread)
      if (f32)
        hipLaunchKernelGGL(k_dsm_gather_dense<false>, dim3(1024), dim3(256), 0, c->stream, p, p.tile_j,
                           c->bin_start, pts_view, lists + kListHdr + (size_t)6 * ntiles,
                           tile_count + 6, cell_out);
      // (what the next call's choice between list and dense launch reads; never waited for)
      if (c->host_tile_stats && f32)   // (FP64 pipeline: the prologue's last workgroup wrote them)
        AMHIP_TRY(hipMemcpyAsync(c->host_tile_stats, tile_count, kListHdr * sizeof(unsigned),
                                 hipMemcpyDeviceToHost, c->stream));
    } else {
      if (c->range_parts) {
        hipLaunchKernelGGL(k_range_reduce, dim3(32), dim3(256), 0, c->stream, (const double*)c->zpart,
                           c->range_parts, c->range_running, c->dev_zrange + 2);
        c->range_parts = 0;
      }
      dim3 grid((unsigned)((p.rows + 63) / 64), (unsigned)((p.cols + 3) / 4));
      if (p.knn_k > 0)
        hipLaunchKernelGGL(k_dsm_gather_knn, grid, dim3(256), 0, c->stream, p, c->bin_start,
                           pts_view, cell_out);
      else
        hipLaunchKernelGGL(k_dsm_gather, grid, dim3(256), 0, c->stream, p, c->bin_start,
                           pts_view, cell_out);
    }
    AMHIP_TRY(hipGetLastError());
  }
  return AMHIP_OK;
}

}  // namespace amhip
