// amhip_ortho.hip -- grid-based backward-projection orthomosaic on MI355X.
//
// Replaces ortho::OrthoBackwardGrid::updateOrthomosaicLayerMultiThreaded
// (aerial_mapper_ortho/src/ortho-backward-grid.cc:128-221): for every cell the
// centre (x, y, elevation) is projected into every frame in ascending frame
// order, the most-nadir view wins (strict `>` against the FLOAT-rounded
// running maximum, which makes the fold order dependent) and the nearest
// pixel of the winning frame is sampled.
//
// One workgroup owns a tile of 64 x 64 cells.  It reduces the tile's elevation
// range, builds the tile's frame list ONCE -- cull against the bounding sphere
// of the tile's landmarks with the four side planes of the view pyramid
// (conservative: a culled frame is invisible from every cell of the tile, so
// skipping it does not change the fold), then dominance pruning of the
// survivors (amhip_ortho_fold.h) -- and folds the list, in ascending order,
// into the tile's cells, one slab of 64 x 16 cells (4 per lane) at a time.  The
// reference brute-forces all F frames per cell; with ~4x overlap one or two
// frames are left per tile.
//
// Two builds of the fold:
//   k_ortho_backward        every pair in the reference's arithmetic
//                           (oracle/amo_compat.h), operation for operation, in
//                           double without fused multiply-add
//                           (-ffp-contract=off): minkindr transform of the
//                           landmark, aslam pinhole project3 (+ radtan /
//                           equidistant distortion), asin(|z| / ||p||);
//   k_ortho_backward_fast   undistorted pinhole + unit quaternions: a
//                           bounded-error evaluation whose every decision
//                           carries a margin, the reference's arithmetic only
//                           for the cells a margin cannot settle
//                           (amhip_ortho_fold.h).
// Both give the reference's layers bit for bit.
#include <cstdlib>

#include "amhip_common.h"
#include "amhip_atan_cr.h"
#include "amhip_device.h"

namespace amhip {

constexpr int kTileI = 64;
constexpr int kTileJ = 64;   // a workgroup's tile ...
// ... is folded in slabs of kSlab cell columns (template parameter of the kernels): 16 = four
// cells per lane for the kernel that folds every pair in the reference's arithmetic, 8 = two for
// the margin-guarded one (it then fits 128 VGPRs = 4 waves per SIMD almost without spills:
// 1.36 -> 1.27 ms; with four cells per lane the 128-VGPR build spills 26 registers: 1.69 ms)
constexpr int kOrthoThreads = 256;
constexpr int kChunk = 1024;  // frames culled per pass

// V3, cross3, transform_point, exact_view_inline, fold_init / fold_pair / fold_finish: amhip_ortho_fold.h

__device__ __forceinline__ void distort_point(const OrthoParams& p, double* px,
                                              double* py) {
  double x = *px, y = *py;
  if (p.distortion == AMHIP_DIST_RADTAN) {
    const double k1 = p.dist[0], k2 = p.dist[1], p1 = p.dist[2], p2 = p.dist[3];
    const double mx2 = x * x;
    const double my2 = y * y;
    const double mxy = x * y;
    const double rho2 = mx2 + my2;
    const double rad = k1 * rho2 + k2 * rho2 * rho2;
    const double nx = x + (x * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2));
    const double ny = y + (y * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2));
    x = nx;
    y = ny;
  } else if (p.distortion == AMHIP_DIST_EQUIDISTANT) {
    const double r = sqrt(x * x + y * y);
    // atan to 76 bits, rounded once (amhip_atan_cr.h: the correctly rounded value in all but one
    // call in ~8 million) -- what a host libm returns in 99.9 % of its calls (glibc 2.35),
    // whichever libm that is; the device library's own differs from the host's by an ulp far
    // more often, and an ulp here can move a keypoint across a pixel boundary
    const double theta = atan_device(r);
    const double th2 = theta * theta;
    const double th4 = th2 * th2;
    const double th6 = th4 * th2;
    const double th8 = th4 * th4;
    const double thetad =
        theta * (1.0 + p.dist[0] * th2 + p.dist[1] * th4 + p.dist[2] * th6 +
                 p.dist[3] * th8);
    const double scaling = (r > 1e-8) ? thetad / r : 1.0;
    x = x * scaling;
    y = y * scaling;
  }
  *px = x;
  *py = y;
}

// aslam::PinholeCamera::project3 + the visibility test of
// ortho-backward-grid.cc:164-171.
__device__ __forceinline__ bool project_visible(const OrthoParams& p,
                                                const V3& c, double* u,
                                                double* v) {
  const double rz = 1.0 / c.z;
  double kx = c.x * rz;
  double ky = c.y * rz;
  if (p.distortion != AMHIP_DIST_NONE) distort_point(p, &kx, &ky);
  *u = p.fu * kx + p.cu;
  *v = p.fv * ky + p.cv;
  const bool in_box = (*u >= 0.0) && (*v >= 0.0) && (*u < (double)p.width) &&
                      (*v < (double)p.height);
  // status not in {POINT_BEHIND_CAMERA, PROJECTION_INVALID} <=> z > 1e-10
  return in_box && (c.z > 1e-10);
}

// asin(|z| / ||p||) exactly as the reference evaluates it
// (ortho-backward-grid.cc:173-176).  Deliberately NOT inlined: it is needed
// once per cell and in rare near ties, and inlining libm's asin three times per
// unrolled cell costs ~25 VGPRs (one wave per SIMD of occupancy).
__device__ __noinline__ double view_angle(double abs_z, double n2) {
  const double norm = sqrt(n2);
  return asin(abs_z / norm);
}

// The reference's arithmetic for the cells the margin-guarded fold could not
// settle (amhip_ortho_fold.h).  Out of line, called after the hot loop with
// almost nothing live: their registers do not add to the loop's.
__device__ __noinline__ FoldResult slow_finish(const double* __restrict__ cam,
                                               const FramePose* __restrict__ pose, double lx,
                                               double ly, double lz, int best_f, int accepted) {
  return exact_finish(cam, *pose, lx, ly, lz, best_f, accepted);
}

__device__ __noinline__ FoldResult slow_refold(const double* __restrict__ cam,
                                               const FramePose* __restrict__ poses,
                                               const int* cand, int n, double lx, double ly,
                                               double lz, float layer_angle) {
  return exact_refold(cam, poses, cand, n, lx, ly, lz, layer_angle);
}

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}

// Can the frame with pose T (= T_C_G) see any point of the sphere?  Conservative:
// false only if the sphere lies entirely behind the camera or outside one of
// the four side planes of the view pyramid.
__device__ __forceinline__ bool frame_may_see(const OrthoParams& p, const FramePose& T,
                                              const V3& centre, double radius) {
  const V3 cc = transform_point(T, centre);
  bool keep = !(cc.z < -radius);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double d = p.pl[k][0] * cc.x + p.pl[k][1] * cc.y + p.pl[k][2] * cc.z;
    if (d < -radius) keep = false;
  }
  return keep;
}

// aerial-mapper-grid-map.cc:40-48: elevation_angle 0, observation_index NaN,
// ortho 255 / colored_ortho NaN
__device__ __forceinline__ void write_initial(const OrthoParams& p, float* __restrict__ angle,
                                              float* __restrict__ index, float* __restrict__ out,
                                              int i, int j) {
  const size_t at = (size_t)i + (size_t)j * (size_t)p.rows;
  angle[at] = 0.0f;
  index[at] = __builtin_nanf("");
  out[at] = p.colored ? __builtin_nanf("") : 255.0f;
}

// a wave-uniform double, moved to scalar registers
__device__ __forceinline__ double uniform_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmin(v, __shfl_xor(v, d, 64));
  return v;
}

// The same test with the pose as a matrix around the camera centre (FrameFast):
// 24 FP64 operations instead of 45, and within ~1e-15 (|centre| + distance) of
// the exact transform (amhip_ortho_fold.h); the radius grows by 2^-40 of that.
__device__ __forceinline__ bool frame_may_see_fast(const OrthoParams& p, const FrameFast& Q,
                                                   const V3& centre, double cmag, double radius) {
  const double dx = centre.x - Q.p[0], dy = centre.y - Q.p[1], dz = centre.z - Q.p[2];
  const double r = fma(0x1p-40, cmag + (fabs(dx) + fabs(dy) + fabs(dz)), radius);
  const double cx = fma(Q.m[2], dz, fma(Q.m[1], dy, Q.m[0] * dx));
  const double cy = fma(Q.m[5], dz, fma(Q.m[4], dy, Q.m[3] * dx));
  const double cz = fma(Q.m[8], dz, fma(Q.m[7], dy, Q.m[6] * dx));
  bool keep = !(cz < -r);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double d = fma(p.pl[k][2], cz, fma(p.pl[k][1], cy, p.pl[k][0] * cx));
    if (d < -r) keep = false;
  }
  return keep;
}

// Phase B: the tile's frame list for one chunk of frames, ascending, in s_cand.
//   stage 1  one thread per frame: bounding sphere of the tile's landmarks
//            against the view pyramid (conservative: a dropped frame is
//            invisible from every cell of the tile), ballot compaction;
//   stage 2  (p.prune) one thread per SURVIVOR: frame_bounds() -> drop the
//            frames that a frame fully visible over the tile beats at every
//            landmark (amhip_ortho_fold.h: dominated), second compaction.
// Returns the number of frames left (the same value in every thread of the
// block); ends with a barrier.
template <bool kFast>
__device__ __forceinline__ int cull_chunk(const OrthoParams& p, const FramePose* __restrict__ poses,
                                          const FrameFast* __restrict__ fast_tab,
                                          const V3& centre, double radius, double slack,
                                          int chunk0, int chunk_n, int* s_cand, int* s_wave_cnt,
                                          double* s_best) {
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const double cmag = fabs(centre.x) + fabs(centre.y) + fabs(centre.z);
  int ncand = 0;
  for (int r = 0; r < chunk_n; r += kOrthoThreads) {
    const int f = chunk0 + r + (int)threadIdx.x;
    bool keep = false;
    if (f < chunk0 + chunk_n) {
      keep = true;
      if (p.cull) {
        if constexpr (kFast)
          keep = frame_may_see_fast(p, fast_tab[f], centre, cmag, radius);
        else
          keep = frame_may_see(p, poses[f], centre, radius);
      }
    }
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();  // previous round's readers of s_wave_cnt are done
    if (lane == 0) s_wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int base = ncand, tot = 0;
#pragma unroll
    for (int w = 0; w < kOrthoThreads / 64; ++w) {
      const int cw = s_wave_cnt[w];
      if (w < wid) base += cw;
      tot += cw;
    }
    if (keep && base + before < kChunk) s_cand[base + before] = f;
    ncand += tot;
  }
  __syncthreads();  // s_cand complete
  // more survivors than the list holds: the caller splits the frames into chunks
  if (ncand > kChunk) return -1;
  if (!p.prune || ncand <= 1) return ncand;

  // ---- stage 2: every thread takes up to kChunk / kOrthoThreads survivors -------
  constexpr int kPer = kChunk / kOrthoThreads;
  int f[kPer];
  double tmin[kPer];
  double tmax_full = __builtin_huge_val();
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int k = q * kOrthoThreads + (int)threadIdx.x;
    f[q] = -1;
    tmin[q] = 0.0;
    if (k < ncand) {
      f[q] = s_cand[k];
      const FrameBounds b = frame_bounds(p.pl, poses[f[q]], centre, radius, slack, p.r_in);
      tmin[q] = b.tmin;
      if (b.full) tmax_full = fmin(tmax_full, b.tmax);
    }
  }
  // smallest tmax of a fully visible frame
  {
    const double wmin = wave_min_d(tmax_full);
    if (lane == 0) s_best[wid] = wmin;
  }
  __syncthreads();  // s_best written, everyone has read its s_cand entries
  double best = s_best[0];
#pragma unroll
  for (int w = 1; w < kOrthoThreads / 64; ++w) best = fmin(best, s_best[w]);
  int kept = 0;
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    if (q * kOrthoThreads >= ncand) break;  // (uniform)
    const bool keep = f[q] >= 0 && !dominated(tmin[q], best);
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();  // previous round's readers of s_wave_cnt are done
    if (lane == 0) s_wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int base = kept, tot = 0;
#pragma unroll
    for (int w = 0; w < kOrthoThreads / 64; ++w) {
      const int cw = s_wave_cnt[w];
      if (w < wid) base += cw;
      tot += cw;
    }
    // (in place: entry `base + before` <= the entry read, and every thread has
    // read all of its entries before the first barrier above)
    if (keep) s_cand[base + before] = f[q];
    kept += tot;
  }
  __syncthreads();  // s_cand complete
  return kept;
}

// The sampled pixel of a cell's winning view as the output layer stores it
// (ortho-backward-grid.cc:186-208).
__device__ __forceinline__ float read_pixel(const OrthoParams& p, const uint8_t* __restrict__ frames,
                                            int frame, int kp_x, int kp_y) {
  const uint8_t* px = frames + (size_t)frame * p.frame_stride + (size_t)kp_y * p.row_step;
  if (p.colored) {
    // cv::Vec3b = (B, G, R); colorVectorToValue packs R<<16 | G<<8 | B
    px += (size_t)kp_x * 3u;
    const unsigned bits = ((unsigned)px[2] << 16) | ((unsigned)px[1] << 8) | (unsigned)px[0];
    return __uint_as_float(bits);
  }
  return (float)px[kp_x];
}

// One cell's results (ortho-backward-grid.cc:181-208): angle, frame index, the
// `num_observations += itself` updates and the sampled pixel.
__device__ __forceinline__ void store_cell(const OrthoParams& p, float* __restrict__ elevation_angle,
                                           float* __restrict__ observation_index,
                                           float* __restrict__ num_observations,
                                           float* __restrict__ out_layer, int i, int j, float angle,
                                           int frame, int accepted, float pixel) {
  const size_t at = (size_t)i + (size_t)j * (size_t)p.rows;
  elevation_angle[at] = angle;
  observation_index[at] = (float)frame;
  // layer_num_observations(x, y) += layer_num_observations(x, y), once per
  // accepted update (ortho-backward-grid.cc:183): doubles the stored value.
  if (!p.virt_nobs) {
    float nobs = num_observations[at];
    if (nobs != 0.0f) {
      for (int n = 0; n < accepted; ++n) nobs += nobs;
      num_observations[at] = nobs;
    }
  }
  out_layer[at] = pixel;
}

__device__ __forceinline__ void write_cell(const OrthoParams& p, const uint8_t* __restrict__ frames,
                                           float* __restrict__ elevation_angle,
                                           float* __restrict__ observation_index,
                                           float* __restrict__ num_observations,
                                           float* __restrict__ out_layer, int i, int j, float angle,
                                           int frame, int accepted, int kp_x, int kp_y) {
  store_cell(p, elevation_angle, observation_index, num_observations, out_layer, i, j, angle, frame,
             accepted, read_pixel(p, frames, frame, kp_x, kp_y));
}

// Deferred write-back of the margin-guarded fold (round 4).  A slab's pixel reads -- single bytes
// gathered from 2 MB frames, the longest latency of the kernel -- used to be issued and awaited
// at the end of the slab, with four waves per SIMD to cover them.  Now they are ISSUED at the end
// of slab s, the rest of what the stores need is parked in LDS (angle, frame, count: 8 bytes per
// cell), and the stores happen after slab s + 1's arithmetic: by then the bytes have arrived.
// Two registers per lane stay live (six with colour) instead of a memory round trip per slab.
template <int kCells>
struct SlabCarry {
  unsigned raw[kCells][3];  // the pixel bytes as loaded (gray: [0]; colour: B, G, R)
  int js;                   // first column of the parked slab; -1: nothing parked
};
// meta word: what (2 bits) << 30 | accepted (14) << 16 | frame (16)
constexpr unsigned kParkFrameMax = 0xFFFFu, kParkAcceptMax = 0x3FFFu;

template <int kCells>
__device__ __forceinline__ void slab_commit(const OrthoParams& p, const SlabCarry<kCells>& carry,
                                            const float* s_park_angle, const unsigned* s_park_meta,
                                            float* __restrict__ elevation_angle,
                                            float* __restrict__ observation_index,
                                            float* __restrict__ num_observations,
                                            float* __restrict__ out_layer, int i) {
  if (carry.js < 0) return;  // (block-uniform)
  const int wid = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < kCells; ++c) {
    const int j = carry.js + wid + c * (kOrthoThreads / 64);
    const unsigned meta = s_park_meta[c * kOrthoThreads + threadIdx.x];
    const unsigned what = meta >> 30;
    if (what == (unsigned)kFoldNone) {
      if (p.virt_out) write_initial(p, elevation_angle, observation_index, out_layer, i, j);
    } else if (what == (unsigned)kFoldDone) {
      const float pix = p.colored ? __uint_as_float((carry.raw[c][2] << 16) | (carry.raw[c][1] << 8) |
                                                    carry.raw[c][0])
                                  : (float)carry.raw[c][0];
      store_cell(p, elevation_angle, observation_index, num_observations, out_layer, i, j,
                 s_park_angle[c * kOrthoThreads + threadIdx.x], (int)(meta & kParkFrameMax),
                 (int)((meta >> 16) & kParkAcceptMax), pix);
    }  // (3: no such cell, or a cell the slow path has already written)
  }
}

// One slab (64 x kSlab cells, kSlab / 4 per lane) of the block's tile: fold
// the tile's frame list into the slab's cells and write them back.
//
// kFast: the margin-guarded fold of amhip_ortho_fold.h (undistorted pinhole,
// unit quaternions) -- ~30 FP64 operations per pair instead of ~85, the
// reference's own arithmetic once per cell for the winner and for the pairs a
// margin cannot decide.  !kFast: every pair in the reference's arithmetic.
//
// single: the whole frame list fits one cull chunk and is already in s_cand
// (ncand0 entries); otherwise the chunks are culled here, per slab.
template <bool kFast, int kSlab>
__device__ __forceinline__ void ortho_slab(
    const OrthoParams& p, const FramePose* __restrict__ poses,
    const FrameFast* __restrict__ fast_tab, const uint8_t* __restrict__ frames,
    const float* __restrict__ elevation, float* __restrict__ elevation_angle,
    float* __restrict__ observation_index, float* __restrict__ num_observations,
    float* __restrict__ out_layer, unsigned* __restrict__ dev_err, int* s_cand, int* s_wave_cnt,
    double* s_best, const double* s_atan, const float* s_elev, float* s_park_angle,
    unsigned* s_park_meta, SlabCarry<kSlab / (kOrthoThreads / 64)>* carry, const V3& centre,
    double radius, double slack, int i, bool i_ok, int js, int j0, bool single, int ncand0) {
  constexpr int kCellsPerLane = kSlab / (kOrthoThreads / 64);
  const int wid = threadIdx.x >> 6;
  const double lx = p.base_x + p.res * (-(double)(i + p.i_off));
  // the tile's elevations were parked in LDS by the range pass (phase A): an LDS round trip per
  // slab instead of an L2 one (the same wave wrote what it reads: no barrier)
  float elev[kCellsPerLane];
#pragma unroll
  for (int c = 0; c < kCellsPerLane; ++c) {
    const int j = js + wid + c * (kOrthoThreads / 64);
    float e = __builtin_nanf("");
    if (i_ok && j < p.cols) e = s_elev[(j - j0) * kTileI + (threadIdx.x & 63)];
    elev[c] = e;
  }

  if constexpr (kFast) {
    // ---- margin-guarded fold (amhip_ortho_fold.h) -----------------------------
    const double* cam_tab = reinterpret_cast<const double*>(fast_tab + p.num_frames);
    CellFold st[kCellsPerLane];
    double lz[kCellsPerLane];
    bool valid[kCellsPerLane];
    const double ly0 = p.base_y + p.res * (-(double)(js + wid + p.j_off));
    const double dly = p.res * (-(double)(kOrthoThreads / 64));
    double magL = 0.0;  // >= |lx| + |ly| + |lz| of every cell of the lane
#pragma unroll
    for (int c = 0; c < kCellsPerLane; ++c) {
      const int j = js + wid + c * (kOrthoThreads / 64);
      float a0 = 0.0f;
      if (i_ok && j < p.cols && !p.virt_out)
        a0 = elevation_angle[(size_t)i + (size_t)j * (size_t)p.rows];
      fold_init(&st[c], a0);
      lz[c] = (double)elev[c];
      valid[c] = elev[c] == elev[c];  // NaN elevation (or no such cell) is never visible
      // an infinite elevation makes mag infinite: every pair of the lane is then
      // inside the margins and the cells are replayed by slow_refold()
      if (valid[c]) magL = fmax(magL, fabs(lz[c]));
    }
    // (the fma chain steps ly through the slab as ly0 + c * dly: within 1.5 ulp of
    // the grid's value, a term of the error budget; the slow routines use the grid's)
    magL += fabs(lx) + fmax(fabs(ly0), fabs(ly0 + dly * (double)(kCellsPerLane - 1)));
    const double epsL = 0x1p-48 * magL;

    for (int chunk0 = 0; chunk0 < p.num_frames; chunk0 += kChunk) {
      int ncand = ncand0;
      if (!single) {
        const int chunk_n = min(kChunk, p.num_frames - chunk0);
        ncand = cull_chunk<kFast>(p, poses, fast_tab, centre, radius, slack, chunk0, chunk_n, s_cand,
                                  s_wave_cnt, s_best);
      }
      if (i_ok) {
        for (int k = 0; k < ncand; ++k) {
          const int f = __builtin_amdgcn_readfirstlane(s_cand[k]);
          const FrameFast& Q = fast_tab[f];
          // c = M (L - p): the large coordinates cancel in the differences
          const double dx = lx - Q.p[0], dy0 = ly0 - Q.p[1];
          const double bx = fma(Q.m[1], dy0, Q.m[0] * dx);
          const double by = fma(Q.m[4], dy0, Q.m[3] * dx);
          const double bz = fma(Q.m[7], dy0, Q.m[6] * dx);
          const double sx = Q.m[1] * dly, sy = Q.m[4] * dly, sz = Q.m[7] * dly;
#pragma unroll
          for (int c = 0; c < kCellsPerLane; ++c) {
            const double dz = lz[c] - Q.p[2];
            const double cx = fma(Q.m[2], dz, fma(sx, (double)c, bx));
            const double cy = fma(Q.m[5], dz, fma(sy, (double)c, by));
            const double cz = fma(Q.m[8], dz, fma(sz, (double)c, bz));
            fold_pair(&st[c], f, p.fold, valid[c], cx, cy, cz, epsL);
          }
        }
      }
      if (single) break;
      __syncthreads();  // everyone is done with s_cand before the next chunk
    }

    // ---- write back, pass 1: the winner's keypoint and stored angle from the
    // approximate point wherever that is provably the reference's result --------
    // The four cells of the lane side by side: first all the arithmetic (no branch between
    // the cells, so their dependent FP64 chains overlap), then the four pixel reads, then
    // the stores.
    int pending = 0;  // 2 bits per cell: kFoldFinish / kFoldRedo
    int what[kCellsPerLane], ku[kCellsPerLane], kv[kCellsPerLane];
    float angle[kCellsPerLane];
#pragma unroll
    for (int c = 0; c < kCellsPerLane; ++c) {
      const int j = js + wid + c * (kOrthoThreads / 64);
      ku[c] = kv[c] = 0;
      angle[c] = 0.0f;
      what[c] = fold_finish(&st[c], p.fold, s_atan, epsL, p.width, p.height, &ku[c], &kv[c],
                            &angle[c]);
      if (!(i_ok && j < p.cols)) what[c] = -1;  // no such cell
    }
    // (unconditional, so that the lane's reads leave together instead of one per branch: a cell
    // without a settled winner reads pixel (0, 0) of frame 0 and drops it)
    // (and the colour / gray branch OUTSIDE the loop over the cells: inside, the compiler waits
    // for one cell's pixel before it asks for the next)
    SlabCarry<kCellsPerLane> mine;
    mine.js = js;
    {
      const uint8_t* px[kCellsPerLane];
#pragma unroll
      for (int c = 0; c < kCellsPerLane; ++c) {
        const bool done = what[c] == kFoldDone;
        px[c] = frames + (size_t)(done ? st[c].best_f : 0) * p.frame_stride +
                (size_t)(done ? kv[c] : 0) * p.row_step + (size_t)(done ? ku[c] : 0) * (p.colored ? 3u : 1u);
      }
      if (p.colored) {
        // cv::Vec3b = (B, G, R); colorVectorToValue packs R<<16 | G<<8 | B (slab_commit)
#pragma unroll
        for (int c = 0; c < kCellsPerLane; ++c) {
          mine.raw[c][0] = px[c][0];
          mine.raw[c][1] = px[c][1];
          mine.raw[c][2] = px[c][2];
        }
      } else {
#pragma unroll
        for (int c = 0; c < kCellsPerLane; ++c) {
          mine.raw[c][0] = px[c][0];
          mine.raw[c][1] = mine.raw[c][2] = 0u;
        }
      }
    }
    // the PREVIOUS slab's stores, while this slab's bytes are on their way (slab_commit reads
    // the parked words before they are overwritten below; same thread: no barrier)
    slab_commit<kCellsPerLane>(p, *carry, s_park_angle, s_park_meta, elevation_angle, observation_index,
                               num_observations, out_layer, i);
    const bool parkable = p.num_frames <= (int)kParkFrameMax;
#pragma unroll
    for (int c = 0; c < kCellsPerLane; ++c) {
      unsigned code = 3u;  // nothing to do at commit time
      if (what[c] == kFoldNone) {
        code = (unsigned)kFoldNone;
      } else if (what[c] == kFoldDone) {
        if (parkable && st[c].accepted <= (int)kParkAcceptMax) {
          code = (unsigned)kFoldDone;
        } else {  // (more frames than the packed word holds: finish this cell the slow way)
          what[c] = kFoldFinish;
        }
      }
      if (what[c] > 0 && what[c] != kFoldDone) pending |= what[c] << (2 * c);
      s_park_angle[c * kOrthoThreads + threadIdx.x] = angle[c];
      s_park_meta[c * kOrthoThreads + threadIdx.x] =
          (code << 30) | (((unsigned)st[c].accepted & kParkAcceptMax) << 16) |
          ((unsigned)max(st[c].best_f, 0) & kParkFrameMax);
    }
    *carry = mine;
    // ---- pass 2 (rare): the reference's arithmetic -------------------------------
    if (pending) {
      const bool whole_list = single;  // else: every frame, like the reference itself
      bool bad_alpha = false;
#pragma unroll
      for (int c = 0; c < kCellsPerLane; ++c) {
        const int what = (pending >> (2 * c)) & 3;
        if (what == 0) continue;
        const int j = js + wid + c * (kOrthoThreads / 64);
        const size_t at = (size_t)i + (size_t)j * (size_t)p.rows;
        const double ly = p.base_y + p.res * (-(double)(j + p.j_off));
        const double lzc = (double)elevation[at];
        FoldResult r;
        if (what == kFoldFinish) {
          r = slow_finish(cam_tab, poses + st[c].best_f, lx, ly, lzc, st[c].best_f, st[c].accepted);
        } else {
          const float a0 = p.virt_out ? 0.0f : elevation_angle[at];
          r = slow_refold(cam_tab, poses, whole_list ? s_cand : nullptr,
                          whole_list ? ncand0 : p.num_frames, lx, ly, lzc, a0);
        }
        if (r.bad_alpha) bad_alpha = true;
        if (r.best_f < 0) {
          if (p.virt_out) write_initial(p, elevation_angle, observation_index, out_layer, i, j);
        } else {
          write_cell(p, frames, elevation_angle, observation_index, num_observations, out_layer, i,
                     j, r.best, r.best_f, r.accepted, r.kp_x, r.kp_y);
        }
      }
      if (bad_alpha) atomicOr(dev_err, kDevErrAlphaNonPos);
    }
  } else {
    // ---- per-lane fold state -------------------------------------------------
    // Running best view per cell.  The reference keeps (float)asin(|z|/||p||) and
    // accepts a view iff its asin exceeds that float (widened to double).  asin is
    // monotonic, so unless the two sines are within 2.5e-6 (relative, squared) of
    // each other the outcome is decided by comparing |z|^2/||p||^2 -- no sqrt, no
    // division, no asin.  Only near ties (where the float rounding of the stored
    // angle matters) take the exact route; the winning angle itself is evaluated
    // once per cell at the end.  Margin: d(asin)/ds >= 1 and asin(s) <= (pi/2) s,
    // so a relative gap of 1e-6 in s is > 10x the 6e-8 float rounding of the angle.
    float best[kCellsPerLane];    // stored angle (valid iff have_f)
    bool have_f[kCellsPerLane];
    double zb[kCellsPerLane];     // |z| and ||p||^2 of the current best view
    double n2b[kCellsPerLane];
    int best_f[kCellsPerLane];
    int best_u[kCellsPerLane];
    int best_v[kCellsPerLane];
    int accepted[kCellsPerLane];
    bool bad_alpha = false;
#pragma unroll
    for (int c = 0; c < kCellsPerLane; ++c) {
      const int j = js + wid + c * (kOrthoThreads / 64);
      best[c] = 0.0f;
      if (i_ok && j < p.cols && !p.virt_out)
        best[c] = elevation_angle[(size_t)i + (size_t)j * (size_t)p.rows];
      have_f[c] = true;
      n2b[c] = 1.0;
      if (best[c] >= 1.5707964f)
        zb[c] = __builtin_huge_val();  // no asin exceeds (float)(pi/2)
      else if (best[c] > 0.0f)
        zb[c] = sin((double)best[c]);  // incremental mode: angle left by earlier batches
      else if (best[c] == best[c])
        zb[c] = 0.0;                   // fresh layer: every visible view wins (alpha > 0)
      else
        zb[c] = __builtin_huge_val();  // NaN in the layer: `alpha > NaN` never holds
      best_f[c] = -1;
      best_u[c] = 0;
      best_v[c] = 0;
      accepted[c] = 0;
    }

    for (int chunk0 = 0; chunk0 < p.num_frames; chunk0 += kChunk) {
      int ncand = ncand0;
      if (!single) {
        const int chunk_n = min(kChunk, p.num_frames - chunk0);
        ncand = cull_chunk<kFast>(p, poses, fast_tab, centre, radius, slack, chunk0, chunk_n, s_cand,
                                  s_wave_cnt, s_best);
      }
      // ---- fold the candidates, ascending --------------------------------------
      if (i_ok) {
        for (int k = 0; k < ncand; ++k) {
          const int f = __builtin_amdgcn_readfirstlane(s_cand[k]);
          const FramePose T = poses[f];
#pragma unroll
          for (int c = 0; c < kCellsPerLane; ++c) {
            const int j = js + wid + c * (kOrthoThreads / 64);
            const double ly = p.base_y + p.res * (-(double)(j + p.j_off));
            const V3 landmark = {lx, ly, (double)elev[c]};
            const V3 cp = transform_point(T, landmark);
            double u, v;
            if (!project_visible(p, cp, &u, &v)) continue;
            const double zz = cp.z * cp.z;
            const double n2 = cp.x * cp.x + cp.y * cp.y + zz;
            const double lhs = zz * n2b[c];
            const double rhs = (zb[c] * zb[c]) * n2;
            bool accept = lhs > rhs * (1.0 + 2.5e-6);
            const bool reject = lhs < rhs * (1.0 - 2.5e-6);
            bool exact = false;
            if (!accept && !reject) {
              // near tie: the reference's own arithmetic decides
              asm volatile("" ::: "memory");
              if (!have_f[c]) {
                best[c] = (float)view_angle(zb[c], n2b[c]);
                have_f[c] = true;
              }
              const double alpha = view_angle(fabs(cp.z), n2);
              if (!(alpha > 0.0)) bad_alpha = true;  // CHECK(alpha > 0.0)
              if (alpha > (double)best[c]) {
                best[c] = (float)alpha;
                accept = true;
                exact = true;
              }
            }
            if (accept) {
              have_f[c] = exact;
              zb[c] = fabs(cp.z);
              n2b[c] = n2;
              best_f[c] = f;
              accepted[c]++;
              best_v[c] = min((int)round(v), p.height - 1);
              best_u[c] = min((int)round(u), p.width - 1);
            }
          }
        }
      }
      if (single) break;
      __syncthreads();  // everyone is done with s_cand before the next chunk
    }

    if (bad_alpha) atomicOr(dev_err, kDevErrAlphaNonPos);

    // ---- write back ------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < kCellsPerLane; ++c) {
      const int j = js + wid + c * (kOrthoThreads / 64);
      if (!(i_ok && j < p.cols)) continue;
      if (accepted[c] == 0) {
        if (p.virt_out) write_initial(p, elevation_angle, observation_index, out_layer, i, j);
        continue;
      }
      if (!have_f[c]) {
        // the winning view's angle, evaluated exactly like the reference does
        const double alpha = view_angle(zb[c], n2b[c]);
        if (!(alpha > 0.0)) atomicOr(dev_err, kDevErrAlphaNonPos);  // CHECK(alpha > 0.0)
        best[c] = (float)alpha;
      }
      write_cell(p, frames, elevation_angle, observation_index, num_observations, out_layer, i, j,
                 best[c], best_f[c], accepted[c], best_u[c], best_v[c]);
    }
  }
}

// One workgroup owns a tile of 64 x kTileJ cells = kTileJ / kSlab slabs.  The
// frame list is built ONCE per tile (elevation range -> bounding sphere -> cull
// + dominance pruning, one thread per frame: that is 250 transforms, a sqrt and
// two divisions per tile, as much work as folding a slab) and every slab folds it.
template <bool kFast, int kSlab>
__device__ __forceinline__ void ortho_backward_tile(
    const OrthoParams& p, const FramePose* __restrict__ poses,
    const FrameFast* __restrict__ fast_tab, const uint8_t* __restrict__ frames,
    const float* __restrict__ elevation, float* __restrict__ elevation_angle,
    float* __restrict__ observation_index, float* __restrict__ num_observations,
    float* __restrict__ out_layer, unsigned* __restrict__ dev_err,
    const unsigned long long* __restrict__ zrange, float* s_red, int* s_cand, int* s_wave_cnt,
    double* s_best, double* s_atan, float* s_elev, float* s_park_angle, unsigned* s_park_meta,
    const int tile_x, const int tile_y) {
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int i = tile_x * kTileI + lane;
  const int j0 = tile_y * kTileJ;
  const bool i_ok = i < p.rows;
  if constexpr (kFast) {
    // fold_finish()'s atan table (doubles 8 .. 24 behind the frame table) next to the lanes: a
    // global read in the middle of the write-back's dependent chain costs a cache round trip
    // per cell (the barriers of the frame cull below order it before its first use)
    const double* cam_tab = reinterpret_cast<const double*>(fast_tab + p.num_frames);
    if (threadIdx.x < kAtanTabSize) s_atan[threadIdx.x] = cam_tab[8 + threadIdx.x];
  }

  // tile extents (cell centres)
  const int i_hi = min(tile_x * kTileI + kTileI, p.rows) - 1;
  const int j_hi = min(j0 + kTileJ, p.cols) - 1;
  const double xa = p.base_x + p.res * (-(double)(tile_x * kTileI + p.i_off));
  const double xb = p.base_x + p.res * (-(double)(i_hi + p.i_off));
  const double ya = p.base_y + p.res * (-(double)(j0 + p.j_off));
  const double yb = p.base_y + p.res * (-(double)(j_hi + p.j_off));
  const double hx = 0.5 * fabs(xa - xb), hy = 0.5 * fabs(ya - yb);

  // ---- phase 0 (small batches): can ANY frame see this tile at all, given the
  // range of heights the DSM has ever written?  If not, leave before touching
  // the tile's elevation.  (Incremental mapping: a few frames, a huge map.)
  bool nothing = false;
  if (p.coarse) {
    const double glo = from_ordered_key(zrange[0]), ghi = from_ordered_key(zrange[1]);
    bool any = false;
    if (glo <= ghi) {
      const double ghz = 0.5 * (ghi - glo) * (1.0 + 1e-6) + 1e-3;  // float-rounded heights
      const V3 gc = {0.5 * (xa + xb), 0.5 * (ya + yb), 0.5 * (glo + ghi)};
      const double gr =
          (sqrt(hx * hx + hy * hy + ghz * ghz) * (1.0 + 1e-9) + 1e-6) * p.radius_scale;
      for (int f = threadIdx.x; f < p.num_frames; f += kOrthoThreads)
        any = any || frame_may_see(p, poses[f], gc, gr);
    }
    nothing = !__syncthreads_or(any);
  }

  // ---- phase A: the tile's elevation range ----------------------------------
  float zmin = __builtin_huge_valf(), zmax = -__builtin_huge_valf();
  if (!nothing) {
    if (i_ok) {
      for (int j = j0 + wid; j <= j_hi; j += kOrthoThreads / 64) {
        const float e = elevation[(size_t)i + (size_t)j * (size_t)p.rows];
        s_elev[(j - j0) * kTileI + lane] = e;  // (the slab that folds row j is this wave's: j = js + wid + 4 c)
        if (e == e) {  // NaN elevation can never be visible
          zmin = fminf(zmin, e);
          zmax = fmaxf(zmax, e);
        }
      }
    }
    zmin = wave_min_f(zmin);
    zmax = wave_max_f(zmax);
    if (lane == 0) {
      s_red[wid] = zmin;
      s_red[kOrthoThreads / 64 + wid] = zmax;
    }
    __syncthreads();
    zmin = s_red[0];
    zmax = s_red[kOrthoThreads / 64];
#pragma unroll
    for (int w = 1; w < kOrthoThreads / 64; ++w) {
      zmin = fminf(zmin, s_red[w]);
      zmax = fmaxf(zmax, s_red[kOrthoThreads / 64 + w]);
    }
  }
  if (!(zmin <= zmax)) {
    // nothing can see the tile / no finite elevation in it: every cell keeps its values
    if (p.virt_out && i_ok)
      for (int j = j0 + wid; j <= j_hi; j += kOrthoThreads / 64)
        write_initial(p, elevation_angle, observation_index, out_layer, i, j);
    return;
  }

  // bounding sphere of the tile's landmarks (cell centres x elevation range)
  const V3 centre = {0.5 * (xa + xb), 0.5 * (ya + yb),
                     0.5 * ((double)zmin + (double)zmax)};
  const double hz = 0.5 * ((double)zmax - (double)zmin);
  // generous slack: the cull only has to be conservative
  const double radius =
      (sqrt(hx * hx + hy * hy + hz * hz) * (1.0 + 1e-9) + 1e-6) * p.radius_scale;
  // "fully visible" margin of frame_bounds(): 1e-6 m + 2^-40 of the coordinate
  // magnitudes (the pose translations are of the same order as the map's)
  const double slack =
      1e-6 + 0x1p-40 * (fabs(centre.x) + fabs(centre.y) + fabs(centre.z) + radius) * 2.0;

  // (block-uniform values the slabs need: keep them in scalar registers)
  const V3 ucentre = {uniform_d(centre.x), uniform_d(centre.y), uniform_d(centre.z)};
  const double uradius = uniform_d(radius), uslack = uniform_d(slack);
  // The whole frame list in one go: the survivors of the sphere cull are what
  // the list holds, not the frames; only if more than kChunk frames can see the
  // tile the slabs split the frames into chunks (and cull per slab and chunk).
  int ncand0 = cull_chunk<kFast>(p, poses, fast_tab, ucentre, uradius, uslack, 0, p.num_frames,
                                 s_cand, s_wave_cnt, s_best);
  const bool single = ncand0 >= 0;
  if (!single) ncand0 = 0;
  SlabCarry<kSlab / (kOrthoThreads / 64)> carry;  // (fast path: the previous slab's parked write-back)
  carry.js = -1;
  if constexpr (kTileJ == kSlab) {
    ortho_slab<kFast, kSlab>(p, poses, fast_tab, frames, elevation, elevation_angle, observation_index,
                      num_observations, out_layer, dev_err, s_cand, s_wave_cnt, s_best, s_atan, s_elev,
                      s_park_angle, s_park_meta, &carry, ucentre, uradius, uslack, i, i_ok, j0, j0, single,
                      ncand0);
  } else {
#pragma unroll 1
    for (int js = j0; js <= j_hi; js += kSlab)
      ortho_slab<kFast, kSlab>(p, poses, fast_tab, frames, elevation, elevation_angle, observation_index,
                        num_observations, out_layer, dev_err, s_cand, s_wave_cnt, s_best, s_atan, s_elev,
                        s_park_angle, s_park_meta, &carry, ucentre, uradius, uslack, i, i_ok, js, j0, single,
                        ncand0);
  }
  if constexpr (kFast)  // the last slab's stores
    slab_commit<kSlab / (kOrthoThreads / 64)>(p, carry, s_park_angle, s_park_meta, elevation_angle,
                                              observation_index, num_observations, out_layer, i);
}

#define AMHIP_ORTHO_KERNEL_ARGS                                                              \
  OrthoParams p, const FramePose *__restrict__ poses, const FrameFast *__restrict__ fast_tab, \
      const uint8_t *__restrict__ frames, const float *__restrict__ elevation,               \
      float *__restrict__ elevation_angle, float *__restrict__ observation_index,            \
      float *__restrict__ num_observations, float *__restrict__ out_layer,                   \
      unsigned *__restrict__ dev_err, const unsigned long long *__restrict__ zrange
#define AMHIP_ORTHO_KERNEL_LDS(SLAB)                                                           \
  __shared__ float s_red[2 * (kOrthoThreads / 64)];                                          \
  __shared__ int s_cand[kChunk];                                                             \
  __shared__ int s_wave_cnt[kOrthoThreads / 64];                                             \
  __shared__ double s_best[kOrthoThreads / 64];                                              \
  __shared__ double s_atan[kAtanTabSize];                                                    \
  __shared__ float s_elev[kTileI * kTileJ];                                                  \
  __shared__ float s_park_angle[kOrthoThreads * ((SLAB) / (kOrthoThreads / 64))];            \
  __shared__ unsigned s_park_meta[kOrthoThreads * ((SLAB) / (kOrthoThreads / 64))];
// one workgroup per tile of the map (blockIdx = tile)
#define AMHIP_ORTHO_KERNEL_BODY(FAST, SLAB)                                                      \
  AMHIP_ORTHO_KERNEL_LDS(SLAB)                                                               \
  ortho_backward_tile<FAST, SLAB>(p, poses, fast_tab, frames, elevation, elevation_angle,    \
                                  observation_index, num_observations, out_layer, dev_err,   \
                                  zrange, s_red, s_cand, s_wave_cnt, s_best, s_atan, s_elev, \
                                  s_park_angle, s_park_meta, (int)blockIdx.x, (int)blockIdx.y);
// a fixed grid walks the list k_ortho_tile_list made of the tiles some frame of the batch can
// see (kernels of their own: the loop's extra live values would cost the dense kernels registers)
#define AMHIP_ORTHO_KERNEL_BODY_LIST(FAST, SLAB)                                                 \
  AMHIP_ORTHO_KERNEL_LDS(SLAB)                                                               \
  const unsigned count = *tile_count;                                                        \
  const int ntx = (p.rows + kTileI - 1) / kTileI;                                            \
  for (unsigned t = blockIdx.x; t < count; t += gridDim.x) {                                 \
    const int tile = tile_list[t];                                                           \
    ortho_backward_tile<FAST, SLAB>(p, poses, fast_tab, frames, elevation, elevation_angle,  \
                                    observation_index, num_observations, out_layer, dev_err, \
                                    zrange, s_red, s_cand, s_wave_cnt, s_best, s_atan,       \
                                    s_elev, s_park_angle, s_park_meta, tile % ntx,           \
                                    tile / ntx);                                             \
    __syncthreads(); /* the next tile reuses every LDS array */                              \
  }

// every pair in the reference's arithmetic (distorted cameras, non-unit quaternions)
__global__ void __launch_bounds__(kOrthoThreads) __attribute__((amdgpu_waves_per_eu(3)))
k_ortho_backward(AMHIP_ORTHO_KERNEL_ARGS) {
  AMHIP_ORTHO_KERNEL_BODY(false, 16)
}
// margin-guarded fold, two cells per lane, held to 128 VGPRs (4 waves per SIMD): the default
__global__ void __launch_bounds__(kOrthoThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_ortho_backward_fast4(AMHIP_ORTHO_KERNEL_ARGS) {
  AMHIP_ORTHO_KERNEL_BODY(true, 8)
}
// the same two over a tile list (small batches onto a large map, below)
__global__ void __launch_bounds__(kOrthoThreads) __attribute__((amdgpu_waves_per_eu(3)))
k_ortho_backward_list(AMHIP_ORTHO_KERNEL_ARGS, const int* __restrict__ tile_list,
                      const unsigned* __restrict__ tile_count) {
  AMHIP_ORTHO_KERNEL_BODY_LIST(false, 16)
}
__global__ void __launch_bounds__(kOrthoThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_ortho_backward_fast4_list(AMHIP_ORTHO_KERNEL_ARGS, const int* __restrict__ tile_list,
                            const unsigned* __restrict__ tile_count) {
  AMHIP_ORTHO_KERNEL_BODY_LIST(true, 8)
}

// Small batches onto a large map (incremental mapping: one frame or a 64-frame batch sees a few
// per cent of a 40 000 x 40 000 map): a dense launch spends its time DISPATCHING workgroups that
// leave at once (390 K tiles: 0.5 ms of the 1.5 ms of a 64-frame batch, all of the 0.56 ms of a
// single frame).  One lane per tile asks the dense launch's own first question -- can ANY frame of
// the batch see the tile's bounding sphere over the range of heights the DSM ever wrote (phase 0
// of ortho_backward_tile, the same arithmetic) -- and the tiles that pass go onto a list a fixed
// grid walks.  Only while every layer is materialized: a lazily reset layer has to be written
// everywhere, which is the dense launch's job.
__global__ void __launch_bounds__(256)
k_ortho_tile_list(OrthoParams p, const FramePose* __restrict__ poses,
                  const unsigned long long* __restrict__ zrange, int* __restrict__ list,
                  unsigned* __restrict__ count) {
  const int ntx = (p.rows + kTileI - 1) / kTileI, nty = (p.cols + kTileJ - 1) / kTileJ;
  const int tile = blockIdx.x * 256 + threadIdx.x;
  bool any = false;
  if (tile < ntx * nty) {
    const int tx = tile % ntx, ty = tile / ntx;
    const int i_hi = min(tx * kTileI + kTileI, p.rows) - 1;
    const int j0 = ty * kTileJ, j_hi = min(j0 + kTileJ, p.cols) - 1;
    const double xa = p.base_x + p.res * (-(double)(tx * kTileI + p.i_off));
    const double xb = p.base_x + p.res * (-(double)(i_hi + p.i_off));
    const double ya = p.base_y + p.res * (-(double)(j0 + p.j_off));
    const double yb = p.base_y + p.res * (-(double)(j_hi + p.j_off));
    const double hx = 0.5 * fabs(xa - xb), hy = 0.5 * fabs(ya - yb);
    const double glo = from_ordered_key(zrange[0]), ghi = from_ordered_key(zrange[1]);
    if (glo <= ghi) {
      const double ghz = 0.5 * (ghi - glo) * (1.0 + 1e-6) + 1e-3;  // float-rounded heights
      const V3 gc = {0.5 * (xa + xb), 0.5 * (ya + yb), 0.5 * (glo + ghi)};
      const double gr = (sqrt(hx * hx + hy * hy + ghz * ghz) * (1.0 + 1e-9) + 1e-6) * p.radius_scale;
      for (int f = 0; f < p.num_frames && !any; ++f) any = frame_may_see(p, poses[f], gc, gr);
    }
  }
  // (one returning atomic per wave)
  const unsigned long long m = __ballot(any);
  if (m) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(m));
    base = __shfl(base, leader, 64);
    if (any) list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = tile;
    // the listed tiles' bounding box (what the call can write: the session downloads only that):
    // ints 4 .. 7 behind the count = min tx, max tx, min ty, max ty
    int tx_lo = any ? tile % ntx : 0x7FFFFFFF, tx_hi = any ? tile % ntx : -1;
    int ty_lo = any ? tile / ntx : 0x7FFFFFFF, ty_hi = any ? tile / ntx : -1;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      tx_lo = min(tx_lo, __shfl_xor(tx_lo, d, 64));
      tx_hi = max(tx_hi, __shfl_xor(tx_hi, d, 64));
      ty_lo = min(ty_lo, __shfl_xor(ty_lo, d, 64));
      ty_hi = max(ty_hi, __shfl_xor(ty_hi, d, 64));
    }
    if (lane == leader) {
      int* box = reinterpret_cast<int*>(count) + 4;
      atomicMin(&box[0], tx_lo);
      atomicMax(&box[1], tx_hi);
      atomicMin(&box[2], ty_lo);
      atomicMax(&box[3], ty_hi);
    }
  }
}
__global__ void k_ortho_tile_list_reset(int* __restrict__ hdr) {
  hdr[0] = 0;
  hdr[4] = 0x7FFFFFFF;
  hdr[5] = -1;
  hdr[6] = 0x7FFFFFFF;
  hdr[7] = -1;
}

int ortho_run(Ctx* c, const OrthoParams& p, const FramePose* dev_poses, const FrameFast* dev_fast,
              const uint8_t* dev_frames) {
  ScopedTimer t(c, AMHIP_K_ORTHO);
  dim3 grid((unsigned)((p.rows + kTileI - 1) / kTileI),
            (unsigned)((p.cols + kTileJ - 1) / kTileJ));
  float* out = p.colored ? c->layers[AMHIP_LAYER_COLORED_ORTHO]
                         : c->layers[AMHIP_LAYER_ORTHO];
  auto kernel = !p.fast ? k_ortho_backward : k_ortho_backward_fast4;
  c->dirty_on_device = false;  // (a dense launch can write anywhere in the window)
  c->dirty[0] = c->dirty[1] = 0;
  c->dirty[2] = c->win_rows;
  c->dirty[3] = c->win_cols;
  // small batch, big map, the output layers materialized: only the tiles some frame can see
  // (tuning knob ortho_no_tile_list: the dense launch -- A-B and tests).  num_observations may stay
  // lazily initial: the kernels neither read nor write it then.
  const size_t ntiles = (size_t)grid.x * (size_t)grid.y;
  // (from 16 K tiles: below, dispatching every tile costs less than the list's extra launch)
  if (p.coarse && !p.virt_out && ntiles >= 16384 && !tuning_on("ortho_no_tile_list")) {
    int rc;
    if ((rc = ensure_capacity(&c->ortho_list, &c->ortho_list_cap, ntiles + 16))) return rc;
    unsigned* cnt = reinterpret_cast<unsigned*>(c->ortho_list);
    hipLaunchKernelGGL(k_ortho_tile_list_reset, dim3(1), dim3(1), 0, c->stream, c->ortho_list);
    hipLaunchKernelGGL(k_ortho_tile_list, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, c->stream,
                       p, dev_poses, c->dev_zrange, c->ortho_list + 8, cnt);
    c->dirty_on_device = true;  // (ctx_last_dirty reads the listed tiles' box back on demand)
    OrthoParams q = p;
    q.coarse = 0;  // (the list kernel asked the question already)
    const dim3 lgrid((unsigned)std::min<size_t>(ntiles, 256 * 16));
    hipLaunchKernelGGL(p.fast ? k_ortho_backward_fast4_list : k_ortho_backward_list, lgrid,
                       dim3(kOrthoThreads), 0, c->stream, q, dev_poses, dev_fast, dev_frames,
                       c->layers[AMHIP_LAYER_ELEVATION], c->layers[AMHIP_LAYER_ELEVATION_ANGLE],
                       c->layers[AMHIP_LAYER_OBSERVATION_INDEX],
                       c->layers[AMHIP_LAYER_NUM_OBSERVATIONS], out, c->dev_err,
                       (const unsigned long long*)nullptr, (const int*)(c->ortho_list + 8),
                       (const unsigned*)cnt);
    AMHIP_TRY(hipGetLastError());
    return AMHIP_OK;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(kOrthoThreads), 0, c->stream,
                     p, dev_poses, dev_fast, dev_frames,
                     c->layers[AMHIP_LAYER_ELEVATION],
                     c->layers[AMHIP_LAYER_ELEVATION_ANGLE],
                     c->layers[AMHIP_LAYER_OBSERVATION_INDEX],
                     c->layers[AMHIP_LAYER_NUM_OBSERVATIONS], out, c->dev_err,
                     p.coarse ? c->dev_zrange : nullptr);
  AMHIP_TRY(hipGetLastError());
  return AMHIP_OK;
}

}  // namespace amhip
