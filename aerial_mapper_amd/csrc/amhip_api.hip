// amhip_api.hip -- C ABI of libaerial_mapper_hip.so (see
// include/aerial_mapper_hip.h): context, layers, parameter set-up, host
// staging, timing.  Kernels live in amhip_dsm.hip / amhip_ortho.hip.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>

#include "amhip_common.h"

namespace amhip {

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  std::snprintf(buf, sizeof(buf), "HIP error %d (%s) in `%s` at %s:%d", (int)e,
                hipGetErrorString(e), what, file, line);
  set_last_error(buf);
  return e == hipErrorOutOfMemory ? AMHIP_ERR_NOMEM : AMHIP_ERR_HIP;
}

static int arg_fail(const char* msg) {
  set_last_error(msg);
  return AMHIP_ERR_ARG;
}

// ---------------------------------------------------------------------------
// memory
// ---------------------------------------------------------------------------
int ensure_bytes(void** ptr, size_t* cap_bytes, size_t need_bytes) {
  if (*cap_bytes >= need_bytes && *ptr) return AMHIP_OK;
  if (*ptr) {
    AMHIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *cap_bytes = 0;
  }
  // grow by 1/8 so that slightly larger follow-up clouds do not reallocate
  size_t want = need_bytes + need_bytes / 8 + 256;
  hipError_t e = hipMalloc(ptr, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    want = need_bytes;
    AMHIP_TRY(hipMalloc(ptr, want));
  }
  *cap_bytes = want;
  return AMHIP_OK;
}

template <typename T>
int ensure_capacity(T** ptr, size_t* cap, size_t need) {
  size_t cap_bytes = *cap * sizeof(T);
  void* p = *ptr;
  const int rc = ensure_bytes(&p, &cap_bytes, need * sizeof(T));
  *ptr = static_cast<T*>(p);
  *cap = cap_bytes / sizeof(T);
  return rc;
}
template int ensure_capacity<double>(double**, size_t*, size_t);
template int ensure_capacity<uint32_t>(uint32_t**, size_t*, size_t);
template int ensure_capacity<uint8_t>(uint8_t**, size_t*, size_t);
template int ensure_capacity<int32_t>(int32_t**, size_t*, size_t);
template int ensure_capacity<FramePose>(FramePose**, size_t*, size_t);

// ---------------------------------------------------------------------------
// timing
// ---------------------------------------------------------------------------
ScopedTimer::ScopedTimer(Ctx* ctx, int slot) : c(ctx), on(ctx->timing) {
  if (!on) return;
  if (!c->free_regions.empty()) {
    r = c->free_regions.back();
    c->free_regions.pop_back();
  } else {
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
      on = false;
      return;
    }
  }
  r.slot = slot;
  (void)hipEventRecord(r.a, c->stream);
}

ScopedTimer::~ScopedTimer() {
  if (!on) return;
  (void)hipEventRecord(r.b, c->stream);
  c->regions.push_back(r);
}

static void drain_timers(Ctx* c) {
  for (TimedRegion& r : c->regions) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess &&
        hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      c->slot_ms[r.slot] += ms;
      c->slot_launches[r.slot] += 1;
    }
    c->free_regions.push_back(r);
  }
  c->regions.clear();
}

// ---------------------------------------------------------------------------
// host-side pose math (minkindr; mirrors oracle/amo_compat.h)
// ---------------------------------------------------------------------------
struct H3 {
  double x, y, z;
};
static H3 hcross(const H3& a, const H3& b) {
  H3 r;
  r.x = a.y * b.z - a.z * b.y;
  r.y = a.z * b.x - a.x * b.z;
  r.z = a.x * b.y - a.y * b.x;
  return r;
}
static H3 hrotate(double qw, double qx, double qy, double qz, const H3& v) {
  const H3 qv = {qx, qy, qz};
  H3 uv = hcross(qv, v);
  uv.x = uv.x + uv.x;
  uv.y = uv.y + uv.y;
  uv.z = uv.z + uv.z;
  const H3 c2 = hcross(qv, uv);
  H3 r;
  r.x = (v.x + qw * uv.x) + c2.x;
  r.y = (v.y + qw * uv.y) + c2.y;
  r.z = (v.z + qw * uv.z) + c2.z;
  return r;
}

HPose hpose_from7(const double* p) {
  HPose r;
  r.tx = p[0];
  r.ty = p[1];
  r.tz = p[2];
  r.qw = p[3];
  r.qx = p[4];
  r.qy = p[5];
  r.qz = p[6];
  return r;
}

// inverse() = (q*, -(q* (x) t))
HPose hpose_inverse(const HPose& T) {
  HPose r;
  r.qw = T.qw;
  r.qx = -T.qx;
  r.qy = -T.qy;
  r.qz = -T.qz;
  const H3 t = {T.tx, T.ty, T.tz};
  const H3 rt = hrotate(r.qw, r.qx, r.qy, r.qz, t);
  r.tx = -rt.x;
  r.ty = -rt.y;
  r.tz = -rt.z;
  return r;
}

// A * B = (qA qB, tA + qA (x) tB)
HPose hpose_compose(const HPose& A, const HPose& B) {
  HPose r;
  r.qw = A.qw * B.qw - A.qx * B.qx - A.qy * B.qy - A.qz * B.qz;
  r.qx = A.qw * B.qx + A.qx * B.qw + A.qy * B.qz - A.qz * B.qy;
  r.qy = A.qw * B.qy + A.qy * B.qw + A.qz * B.qx - A.qx * B.qz;
  r.qz = A.qw * B.qz + A.qz * B.qw + A.qx * B.qy - A.qy * B.qx;
  const H3 tb = {B.tx, B.ty, B.tz};
  const H3 rt = hrotate(A.qw, A.qx, A.qy, A.qz, tb);
  r.tx = A.tx + rt.x;
  r.ty = A.ty + rt.y;
  r.tz = A.tz + rt.z;
  return r;
}

// ---------------------------------------------------------------------------
// parameter set-up
// ---------------------------------------------------------------------------
static void grid_bases(const amhip_grid_desc& g, double* bx, double* by) {
  // getPosition: (mapPosition + (0.5*length - 0.5*resolution)) + resolution*(-i)
  const double off_x = 0.5 * g.length_x - 0.5 * g.resolution;
  const double off_y = 0.5 * g.length_y - 0.5 * g.resolution;
  *bx = g.pos_x + off_x;
  *by = g.pos_y + off_y;
}

// The opt-in single-precision mode protects itself on rough terrain.  A
// tile whose height range leaves no room under the error bound is done by the FP64 kernel, which
// in that mode stages its points from the caller's UNSORTED cloud through the records' row
// indices (24 of every 64 bytes it touches); beyond about half the tiles that costs more than the
// records save in the sort.  The context knows the previous call's share (the pinned mirror of
// its tile counters, never waited for -- a value one or two calls late is as good): above one
// half, this call and the next 15 run the FP64 pipeline outright (sorted doubles), then the
// single-precision pipeline is tried once more.  Results: FP64 is the stricter arithmetic, every
// bar of the single-precision mode holds.  tuning knob dsm_no_rough_switch disables the switch.
static void dsm_rough_policy(Ctx* c) {
  const bool off = tuning_on("dsm_no_rough_switch");  // (looked up per call: tests toggle it)
  c->dsm_exact_now = 0;
  if (c->dsm_exact || c->dsm_knn || off) return;
  if (c->rough_hold > 0) {
    --c->rough_hold;
    c->dsm_exact_now = 1;
    return;
  }
  if (c->last_call_f32 && c->last_ntiles > 0 && c->host_tile_stats) {
    const volatile unsigned* hs = c->host_tile_stats;
    const double rejected = (double)hs[4] + (double)hs[5] + (double)hs[6] + (double)hs[7];
    if (rejected * 2.0 > (double)c->last_ntiles) {
      c->rough_hold = 15;
      c->dsm_exact_now = 1;
    }
  }
}

// A SMALL cloud onto a LARGE resident map (the incremental demo: one stereo pair's 360 K points
// per call onto a map of 1e8 .. 1.6e9 cells, main-ortho-backward-grid-incremental.cc:143-157).
// Binned over the context's whole window the call costs O(map): a 400 MB memset and scan of the
// bin table of a 40 000 x 40 000 map, an occupancy pre-pass over its 1.5 M gather tiles -- 0.75 ms
// for 0.05 ms of work.  Instead: the bounding box of the points the binning would accept (one
// pass over the cloud, five integers back to the host: the one synchronisation of such a call),
// grown by the last fallback radius -- no cell outside it can receive a value -- becomes the
// window of THIS call; the kernels run unchanged on it and write into the full layer
// (DsmParams::out_*).  Only while the layer is materialized (a lazily reset layer has to be
// written everywhere).  tuning knob dsm_no_subwindow: never.
// Returns 1: *p_sub is the call's parameter set; 0: run on the whole window; 2: no point near the
// window, nothing to do; < 0: error.
static int make_dsm_params(const Ctx& c, int radius_sq, double center_easting, double center_northing,
                           DsmParams* out, int mode, int pcl_lambda, size_t num_points);
static int dsm_subwindow(Ctx* c, const double* dev_xyz, size_t n, int radius_sq, double center_easting,
                         double center_northing, const DsmParams& p_full, DsmParams* p_sub) {
  if (n > (1u << 20) || c->cells < (size_t)(4u << 20) || !c->dev_bbox || !c->host_bbox ||
      tuning_on("dsm_no_subwindow"))
    return 0;
  int rc;
  if ((rc = dsm_bbox_run(c, dev_xyz, n, p_full, c->dev_bbox))) return -rc;
  if (hipMemcpyAsync(c->host_bbox, c->dev_bbox, 5 * sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess)
    return -hip_fail(hipGetLastError(), "bounding box of the cloud", __FILE__, __LINE__);
  if (c->host_bbox[4] == 0) return 2;
  // (k_dsm_bbox's biased words: an all-zero buffer is the empty box)
  const int b[4] = {kBboxBias - c->host_bbox[0], c->host_bbox[1] - kBboxBias,
                    kBboxBias - c->host_bbox[2], c->host_bbox[3] - kBboxBias};
  const int grow = p_full.w[p_full.nlevels - 1] + 2;   // last radius of the ladder, in cells (+ slack)
  const int i_lo = std::max(0, b[0] - grow), i_hi = std::min(p_full.rows - 1, b[1] + grow);
  const int j_lo = std::max(0, b[2] - grow), j_hi = std::min(p_full.cols - 1, b[3] + grow);
  if (i_lo > i_hi || j_lo > j_hi) return 2;             // (every binned point lies beyond the border by more than the radius)
  const size_t sub = (size_t)(i_hi - i_lo + 1) * (size_t)(j_hi - j_lo + 1);
  if (sub * 4 > c->cells) return 0;                     // not small against the map: nothing to win
  // the same context, seen through the sub-window
  const int wi0 = c->win_i0, wj0 = c->win_j0, wr = c->win_rows, wc = c->win_cols;
  c->win_i0 = wi0 + i_lo;
  c->win_j0 = wj0 + j_lo;
  c->win_rows = i_hi - i_lo + 1;
  c->win_cols = j_hi - j_lo + 1;
  rc = make_dsm_params(*c, radius_sq, center_easting, center_northing, p_sub, 0, 1, n);
  c->win_i0 = wi0;
  c->win_j0 = wj0;
  c->win_rows = wr;
  c->win_cols = wc;
  if (rc) return -rc;
  p_sub->out_i0 = i_lo;
  p_sub->out_j0 = j_lo;
  p_sub->out_pitch = wr;
  return 1;
}

// mode 0: dsm::Dsm ladder.  mode 1: ortho::OrthoFromPcl, one search with the
// squared radius `radius_sq * pcl_lambda` (pcl_lambda = 1 for the first search,
// 10, 100, ... for the adaptive retries, ortho-from-pcl.cc:63-71).
static int make_dsm_params(const Ctx& c, int radius_sq,
                           double center_easting, double center_northing,
                           DsmParams* out, int mode, int pcl_lambda,
                           size_t num_points) {
  const amhip_grid_desc& g = c.grid;
  DsmParams p;
  std::memset(&p, 0, sizeof(p));
  grid_bases(g, &p.base_x, &p.base_y);
  p.res = g.resolution;
  p.inv_res = 1.0 / g.resolution;
  p.rows = c.win_rows;
  p.cols = c.win_cols;
  p.i_off = c.win_i0;
  p.j_off = c.win_j0;
  p.sub_x = center_northing;  // dsm.cc:42
  p.sub_y = center_easting;   // dsm.cc:43
  {
    p.canon_all = tuning_on("dsm_canon_all") ? 1 : 0;
  }

  // Squared search radii in the order dsm.cc:127-144 tries them: the initial
  // search with T = R, then lambda*R for lambda = 1, 1.1, 1.1^2, ... where
  // lambda is updated by `lambda *= 1.1` and the loop stops once lambda*R > 7.
  int n = 0;
  p.pcl_mode = mode;
  if (mode == 0) {
    p.T[n++] = static_cast<double>(radius_sq);
    double lambda = 1.0;
    for (;;) {
      if (n >= kMaxLevels) return arg_fail("radius ladder too long");
      p.T[n++] = lambda * radius_sq;
      lambda *= 1.1;
      if (lambda * radius_sq > 7.0) break;
    }
  } else {
    p.T[n++] = static_cast<double>(pcl_lambda * radius_sq);  // int product, like the reference
  }
  p.nlevels = n;
  double tmax = 0.0;
  for (int k = 0; k < n; ++k) {
    // a point within sqrt(T) of the centre of cell i lies in a cell whose index
    // differs from i by at most floor(sqrt(T)/res + 0.5) (+ slack for rounding)
    p.w[k] = static_cast<int>(
        std::floor(std::sqrt(p.T[k]) / g.resolution + 0.5 + 1e-6));
    if (p.T[k] > tmax) tmax = p.T[k];
  }
  int wmax = 0;
  for (int k = 0; k < n; ++k)
    if (p.w[k] > wmax) wmax = p.w[k];

  // Bin edge = the first search radius in cells: the LDS gather then needs
  // exactly one ring of bins around a tile.
  int B = p.w[0];
  if (B < 1) B = 1;
  if (B > 8) B = 8;
  if (mode == 1 && pcl_lambda > 1) {
    // adaptive retries: huge radii -> coarse bins, global gather only
    B = p.w[0] / 4;
    if (B < 1) B = 1;
  }
  p.B = B;
  p.M = ((wmax + B - 1) / B) * B;
  const long long ex = (long long)p.rows + 2LL * p.M;
  const long long ey = (long long)p.cols + 2LL * p.M;
  p.nbx = static_cast<int>((ex + B - 1) / B);
  p.nby = static_cast<int>((ey + B - 1) / B);
  const unsigned long long nbins =
      (unsigned long long)p.nbx * (unsigned long long)p.nby;
  if (nbins + 1 >= 0xFFFFFFFFull) return arg_fail("grid too large for 32-bit bin ids");

  // ---- three-pass partition sort plan (amhip_dsm.hip) ---------------------------
  // Worth it once the cloud is large enough that the sort is bandwidth bound;
  // sub-partitions are sized for ~1.5 K points (a pass-3 workgroup sorts up to
  // p3_cap of them in LDS, more through a direct-placement fallback).
  p.p3_n1 = 0;
  {
    const size_t min_pts = (size_t)tuning("p3_min_points", (double)(1u << 20));
    const int r1 = (p.nby + 127) / 128;
    const int n1 = (p.nby + r1 - 1) / r1;
    int cmax = 256 / r1;
    if (r1 <= 256 && cmax >= 1 && num_points >= min_pts) {
      const double target = tuning("p3_target", 1536.0);
      int cc = static_cast<int>((double)num_points / ((double)p.nby * target) + 0.5);
      if (cc < 1) cc = 1;
      if (cc > cmax) cc = cmax;
      if (cc > p.nbx) cc = p.nbx;
      const int w = (p.nbx + cc - 1) / cc;
      if (w <= 4096 && (long long)n1 * r1 * cc <= 32768) {
        p.p3_r1 = r1;
        p.p3_c = cc;
        p.p3_w = w;
        p.p3_n1 = n1;
        p.p3_n2 = r1 * cc;
        p.p3_cap = (int)tuning("p3_cap", 2048.0);
        if (p.p3_cap > 2048) p.p3_cap = 2048;  // kP3PlaceMaxCap
        if (p.p3_cap < 64) p.p3_cap = 64;
      }
    }
  }

  // ---- division-free keys for the sort passes (DsmParams::mul_*) --------------------------
  {
    auto magic = [](unsigned long long n_max, int d) -> unsigned {
      if (d <= 1) return 0u;
      if (n_max * (unsigned long long)d >= (1ull << 32)) return 0xFFFFFFFFu;
      return (unsigned)((1ull << 32) / (unsigned long long)d) + 1u;
    };
    const unsigned long long cells_max =
        (unsigned long long)std::max(p.rows, p.cols) + 2ull * (unsigned long long)p.M + 1ull;
    p.mul_B = magic(cells_max, p.B);
    p.mul_r1 = p.p3_n1 ? magic((unsigned long long)p.nby + 1ull, p.p3_r1) : 0xFFFFFFFFu;
    p.mul_w = p.p3_n1 ? magic((unsigned long long)p.nbx + 1ull, p.p3_w) : 0xFFFFFFFFu;
  }

  // ---- LDS-tiled gather set-up (amhip_dsm.hip: k_dsm_gather_tiled) ----------
  const int kTileI = 64;
  // (the single-precision mode's sort records hold a cell in 16 + 16 bits and a row of the cloud
  // in 32: larger maps / clouds stay in FP64)
  const bool rec_fits = (long long)p.rows + 2LL * p.M <= 65535 && (long long)p.cols + 2LL * p.M <= 65535 &&
                        num_points < 0xFFFFFFFFull;
  // Tile height and LDS point capacity from the cloud's MEAN density (points
  // per cell): the tile's region must hold E + 5 sqrt(E) points.  64x16 tiles
  // with 1024 slots need ~37 KB of LDS -> 4 workgroups per CU; denser clouds
  // take 64x32 / 2048 (2 per CU), then 64x16 / 2048.  A tile that still
  // overflows (clustered cloud) takes the global-memory path on its own.
  int kTileJ = 32;
  int cap = 2048;
  {
    const int w0h = p.w[0];
    const double rho = (double)num_points / ((double)p.rows * (double)p.cols);
    auto need = [&](int tj) {
      // bins an interior tile's region spans (same integer arithmetic as the kernel)
      const int bi = (kTileI + kTileI - 1 + w0h + p.M) / B - (kTileI - w0h + p.M) / B + 1;
      const int bj = (tj + tj - 1 + w0h + p.M) / B - (tj - w0h + p.M) / B + 1;
      const double e = rho * (double)(bi * B) * (double)(bj * B);
      return e + 5.0 * std::sqrt(e);
    };
    // (single-precision mode: 16-byte records, so 4096 points still leave two workgroups per
    // CU -- clouds of ~1.2 .. 2.2 points per cell keep the one-workgroup-per-tile launch)
    const bool want_f32 = mode == 0 && !c.dsm_exact && !c.dsm_exact_now && !c.dsm_knn && rec_fits;
    if (need(16) <= 1024.0) {
      kTileJ = 16;
      cap = 1024;
    } else if (need(32) <= 2048.0) {
      kTileJ = 32;
      cap = 2048;
    } else if (need(16) <= 2048.0 || !want_f32 || need(16) > 7680.0) {
      kTileJ = 16;
      cap = 2048;
    } else {
      kTileJ = 16;
      cap = need(16) <= 4096.0 ? 4096 : 7680;  // (7680: one workgroup per CU)
    }
#ifdef AMHIP_TIMING_PROBES
    if (tuning("gather_tj", 0.0) != 0.0) {  // tuning knob
      kTileJ = (int)tuning("gather_tj", 0.0) == 16 ? 16 : 32;
      cap = (kTileJ == 16 && need(16) <= 1024.0) ? 1024 : 2048;  // (the density still picks the capacity)
    }
#endif
  }
  p.tile_j = kTileJ;
  p.tiles_i = (p.rows + kTileI - 1) / kTileI;
  p.tiles_j = (p.cols + kTileJ - 1) / kTileJ;
  const int w0 = p.w[0];
  p.lds_ok = (w0 >= 1 && w0 <= kMaxW0 && !(mode == 1 && pcl_lambda > 1)) ? 1 : 0;
  if (p.lds_ok) {
    // disc-shaped window: a point whose cell row differs by dj from the query's
    // is at least (|dj| - 0.5) * res away in y; what is left of the radius
    // bounds its column distance.
    const double r2 = p.T[0] / (g.resolution * g.resolution);  // in cells^2
    for (int dj = -w0; dj <= w0; ++dj) {
      const double dy = std::fabs((double)dj) - 0.5 - 1e-6;
      const double rem = dy > 0.0 ? r2 - dy * dy : r2;
      int wr = rem > 0.0 ? static_cast<int>(std::floor(std::sqrt(rem) + 0.5 + 1e-6)) : 0;
      if (wr > w0) wr = w0;
      p.wr[dj + w0] = wr;
    }
    for (int r = 0; r <= 2 * w0 + 1; ++r) {
      const int a = r <= 2 * w0 ? p.wr[r] : 0;
      const int b = r >= 1 ? p.wr[r - 1] : 0;
      p.wr2[r] = a > b ? a : b;
    }
    const int rw = kTileI + 2 * w0 + 2 * (B - 1);
    const int rh = kTileJ + 2 * w0 + 2 * (B - 1);
    for (int k = 0; k <= w0; ++k)
      p.wrp[k] = p.wr2[2 * k] > p.wr2[2 * k + 1] ? p.wr2[2 * k] : p.wr2[2 * k + 1];
    // row pairs, +1 pair for the parity shift, +1 spill; (lds_cells + 1) entries, a multiple of
    // four with three to spare: the single-precision gather clears / scans the table in quads
    p.lds_cells = ((rw * (rh + 4) + 4 + 3) & ~3) - 1;
    p.lds_cap = cap;  // == kCap of the kernel instance
    const size_t bytes = ((size_t)p.lds_cap + 2) * 24 + ((size_t)p.lds_cells + 1) * 4 +
                         (96 + 97 + 24 + 4 + 4 * kMaxW0 + 4) * 4 + (size_t)kTileI * kTileJ * 2 + 64;
    p.lds_bytes = static_cast<unsigned>((bytes + 15) & ~size_t(15));
    // (a 4096- / 7680-point main launch exists in single precision only; the FP64 kernels then
    // run on lists with images of their own size, amhip_dsm.hip: dsm_run)
    if (p.lds_bytes > 150 * 1024 && cap <= 2048) p.lds_ok = 0;
  }
  // ---- single-precision gather with exact guards (amhip_dsm.hip: k_dsm_gather_f32) ----
  // Only for dsm::Dsm (heights): OrthoFromPcl interpolates 8-bit intensities whose spread
  // (up to 255) leaves no room under the error bound.
  p.knn_k = mode == 0 ? c.dsm_knn : 0;
  // (capped mode: the LDS-tiled gather's own build, k_dsm_gather_tiled_knn; tuning knob knn_global_bins:
  // one lane per cell on the global bins, the round-1 kernel)
  if (p.knn_k && tuning_on("knn_global_bins")) p.lds_ok = 0;
  p.fx_ok = 0;
  if (p.lds_ok && mode == 0 && !c.dsm_exact && !c.dsm_exact_now && rec_fits) {
    int S = 28;
    while (((long long)(w0 + 2) << S) >= (1LL << 31)) --S;
    const double scale2 = std::ldexp(1.0, 2 * S);
    const double tc = p.T[0] / (g.resolution * g.resolution) * scale2;
    // |d2_f32 - d2_reference| <= 3e-7 relative at the search radius; anything
    // inside +-2e-6 is decided by the FP64 routine
    const double margin = 2e-6;
    // (cells within theta of a point go to the FP64 routine: 0.02 -> 0.005 takes 0.07 ms of
    // redo tails off the 50 M-point gather and costs the error budget a factor 2.2 in eps_w --
    // a tile's height half-range limit goes from 7.6 to 6.2 m; measured)
    double theta = 0.005;  // cells
#ifdef AMHIP_TIMING_PROBES
    theta = tuning("fx_theta", theta);
#endif
    const double q = std::ldexp(1.0, -(S + 1));
    p.fx_S = S;
    p.fx_thi = static_cast<float>(tc * (1.0 + margin));
    p.fx_tlo = static_cast<float>(tc * (1.0 - margin));
    p.fx_denmax = static_cast<float>(1.0 / (theta * theta * scale2));
    p.fx_epsw = static_cast<float>(2.0 * std::sqrt(2.0) * q / theta + 4e-7);
    const size_t bytes = ((size_t)p.lds_cap + 2) * 16 + ((size_t)p.lds_cells + 1) * 4 +
                         (96 + 97 + 24 + 8 + 4 * kMaxW0 + 4) * 4 + 64 + (size_t)kTileI * kTileJ * 2 + 64;
    p.lds_bytes_f32 = static_cast<unsigned>((bytes + 15) & ~size_t(15));
    p.fx_ok = 1;
  }
  p.out_i0 = p.out_j0 = 0;
  p.out_pitch = p.rows;
  *out = p;
  return AMHIP_OK;
}

static void normalize3(double* v) {
  const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= n;
  v[1] /= n;
  v[2] /= n;
}

// Conservative view cone of a camera WITH distortion: the largest normalised
// radius rho = sqrt(x^2 + y^2) / z a landmark can have and still project into
// the image (plus margins).  The reference tests nothing but the image box
// (ortho-backward-grid.cc:164-171), so where the distortion polynomial folds
// back, far off-axis landmarks DO count as visible -- the bound covers that.
// A landmark is visible only if |distort(p)| <= B, B = farthest image corner
// (+1 pixel) in distorted normalised coordinates; |distort(p)| is bounded from
// below by g(|p|), and the supremum of {r : g(r) <= B} is found on a grid with
// a Lipschitz margin per step (so no dip between samples is missed) plus an
// analytic tail.  false = no usable bound (the kernel then tests every frame).
static bool distorted_view_cone(const amhip_camera& cam, double* cone) {
  double B = 0.0;
  const double us[2] = {-1.0, (double)cam.width}, vs[2] = {-1.0, (double)cam.height};
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const double x = (us[a] - cam.cu) / cam.fu, y = (vs[b] - cam.cv) / cam.fv;
      B = std::max(B, std::sqrt(x * x + y * y));
    }
  B = B * (1.0 + 1e-9) + 1e-9;
  if (cam.distortion == AMHIP_DIST_RADTAN) {
    const double k1 = cam.dist[0], k2 = cam.dist[1];
    const double c = 4.5 * (std::fabs(cam.dist[2]) + std::fabs(cam.dist[3]));  // tangential <= c r^2
    auto g = [&](double r) {
      const double r2 = r * r;
      return r * std::fabs(1.0 + k1 * r2 + k2 * r2 * r2) - c * r2;
    };
    auto lip = [&](double r) {
      const double r2 = r * r;
      return 1.0 + 3.0 * std::fabs(k1) * r2 + 5.0 * std::fabs(k2) * r2 * r2 + 2.0 * c * r;
    };
    const double Rmax = 1e4;
    // tail r >= Rmax: the leading term must dominate for good
    if (k2 != 0.0) {
      const double lead = std::fabs(k2) * Rmax * Rmax * Rmax * Rmax;
      if (!(lead > 2.0 * (std::fabs(k1) * Rmax * Rmax + 1.0 + c * Rmax)) ||
          !(0.5 * lead * Rmax > 2.0 * B))
        return false;
    } else if (k1 != 0.0) {
      const double lead = std::fabs(k1) * Rmax * Rmax;
      if (!(lead > 2.0 * (1.0 + c * Rmax)) || !(0.5 * lead * Rmax > 2.0 * B)) return false;
    } else if (c != 0.0) {
      return false;  // r - c r^2: no bound
    }
    double rstar = 0.0;
    double r = 0.0;
    while (r < Rmax) {
      const double h = r < 20.0 ? 5e-4 : r * 2.5e-5;
      const double rn = r + h;
      if (g(r) <= B + h * lip(rn)) rstar = rn;
      r = rn;
    }
    if (!(rstar < 0.5 * Rmax)) return false;
    *cone = rstar * (1.0 + 1e-6) + 1e-6;
    return true;
  }
  if (cam.distortion == AMHIP_DIST_EQUIDISTANT) {
    const double* k = cam.dist;
    auto td = [&](double th) {
      const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      return std::fabs(th * (1.0 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8));
    };
    const double half_pi = 1.5707963267948966;
    const double t2 = half_pi * half_pi, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double L = 1.0 + 3.0 * std::fabs(k[0]) * t2 + 5.0 * std::fabs(k[1]) * t4 +
                     7.0 * std::fabs(k[2]) * t6 + 9.0 * std::fabs(k[3]) * t8;
    const double h = 2e-5;
    double tstar = 0.0;
    for (double th = 0.0; th < half_pi; th += h)
      if (td(th) <= B + h * L) tstar = th + h;
    if (!(tstar < half_pi - 2e-3)) return false;  // sees (numerically) the whole half space
    *cone = std::tan(tstar) * (1.0 + 1e-6) + 1e-6;
    return true;
  }
  return false;
}

// Tighter bounds for cameras with a distortion model, given the view cone
// `cone` of distorted_view_cone() (every visible landmark has |p| <= cone * z):
//   ax, ay  outer rectangle: visible => |x| <= ax z and |y| <= ay z.  From
//           x_d = x s(r) + t_x with |t_x| <= c r^2 (radtan; s = the radial
//           factor, t = the tangential terms) resp. x_d = x s(r) (equidistant)
//           and |x_d| <= Bx for a visible landmark:  |x| <= (Bx + c cone^2) / min s.
//   rin     inner cone: |p| <= rin z  =>  visible by a clear margin.  The
//           distorted point of such a landmark has |x_d|, |y_d| <= r |s(r)| + c r^2
//           =: g(r); rin = the largest r with g <= (smallest distance of the
//           principal point to an image edge, normalised) on [0, r].
// Both scans carry a Lipschitz margin per step.  Values of 0 mean "no bound".
static void distorted_rectangle_and_inner_cone(const amhip_camera& cam, double cone, double* ax,
                                               double* ay, double* rin) {
  *ax = *ay = cone;
  *rin = 0.0;
  const double W = cam.width, H = cam.height;
  // (the same one-pixel allowance as distorted_view_cone)
  const double Bx = std::max(std::fabs(-1.0 - cam.cu), std::fabs(W - cam.cu)) / cam.fu;
  const double By = std::max(std::fabs(-1.0 - cam.cv), std::fabs(H - cam.cv)) / cam.fv;
  const double bmin = std::min(std::min(cam.cu / cam.fu, (W - cam.cu) / cam.fu),
                               std::min(cam.cv / cam.fv, (H - cam.cv) / cam.fv));
  double c = 0.0;
  // s(r): radial factor; ds: an upper bound of |s'| that grows with r
  auto s_of = [&](double r) {
    if (cam.distortion == AMHIP_DIST_RADTAN) {
      const double r2 = r * r;
      return 1.0 + cam.dist[0] * r2 + cam.dist[1] * r2 * r2;
    }
    if (r < 1e-8) return 1.0;
    const double th = std::atan(r);
    const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    return th * (1.0 + cam.dist[0] * t2 + cam.dist[1] * t4 + cam.dist[2] * t6 + cam.dist[3] * t8) / r;
  };
  auto ds_of = [&](double r) {
    if (cam.distortion == AMHIP_DIST_RADTAN)
      return 2.0 * std::fabs(cam.dist[0]) * r + 4.0 * std::fabs(cam.dist[1]) * r * r * r;
    // s = theta_d(theta) / r, theta = atan r, theta_d = theta P(theta^2):
    //   s' = [P (r / (1 + r^2) - theta) + 2 theta^2 P' r / (1 + r^2)] / r^2
    // |theta_d'| <= L, |theta_d| <= L theta <= L r            =>  |s'| <= 2 L / r
    // |r / (1 + r^2) - atan r| <= r^3, theta <= r, |P| <= L   =>  |s'| <= (L + 2 Lp) r
    const double hp = 1.5707963267948966, t2 = hp * hp, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double L = 1.0 + 3.0 * std::fabs(cam.dist[0]) * t2 + 5.0 * std::fabs(cam.dist[1]) * t4 +
                     7.0 * std::fabs(cam.dist[2]) * t6 + 9.0 * std::fabs(cam.dist[3]) * t8;
    const double Lp = std::fabs(cam.dist[0]) + 2.0 * std::fabs(cam.dist[1]) * t2 +
                      3.0 * std::fabs(cam.dist[2]) * t4 + 4.0 * std::fabs(cam.dist[3]) * t6;
    return std::min(2.0 * L / std::max(r, 1e-12), (L + 2.0 * Lp) * r);
  };
  if (cam.distortion == AMHIP_DIST_RADTAN)
    c = 4.5 * (std::fabs(cam.dist[2]) + std::fabs(cam.dist[3]));
  if (!(cone > 0.0) || !(cone < 1e3)) return;
  // ---- outer rectangle: min |s| on [0, cone]
  {
    const int n = 20000;
    const double h = cone / n;
    double smin = 1e300;
    for (int k = 0; k <= n; ++k) {
      const double r = k * h;
      smin = std::min(smin, std::fabs(s_of(r)) - h * ds_of(r + h));
    }
    if (smin > 0.05) {
      const double tx = c * cone * cone;
      *ax = std::min(cone, ((Bx + tx) / smin) * (1.0 + 1e-6) + 1e-6);
      *ay = std::min(cone, ((By + tx) / smin) * (1.0 + 1e-6) + 1e-6);
    }
  }
  // ---- inner cone
  if (bmin > 0.0) {
    const double bound = bmin * (1.0 - 1e-6) - 1e-9;
    const double h = 2e-5;
    double r = 0.0, best = 0.0;
    while (r < cone) {
      const double rn = r + h;
      // g(r) = r |s(r)| + c r^2;  |g'| <= |s| + r |s'| + 2 c r
      const double g = r * std::fabs(s_of(r)) + c * r * r;
      const double lip = std::fabs(s_of(rn)) + h * ds_of(rn) + rn * ds_of(rn) + 2.0 * c * rn;
      if (!(g + h * lip <= bound)) break;
      best = rn;
      r = rn;
    }
    *rin = best;
  }
}

static void make_ortho_params(Ctx& c, const amhip_camera& cam,
                              size_t F, size_t frame_stride, size_t row_step,
                              int channels, int colored, OrthoParams* out) {
  const amhip_grid_desc& g = c.grid;
  OrthoParams p;
  std::memset(&p, 0, sizeof(p));
  grid_bases(g, &p.base_x, &p.base_y);
  p.res = g.resolution;
  p.rows = c.win_rows;
  p.cols = c.win_cols;
  p.i_off = c.win_i0;
  p.j_off = c.win_j0;
  p.fu = cam.fu;
  p.fv = cam.fv;
  p.cu = cam.cu;
  p.cv = cam.cv;
  for (int k = 0; k < 4; ++k) p.dist[k] = cam.dist[k];
  p.width = cam.width;
  p.height = cam.height;
  p.distortion = cam.distortion;
  p.num_frames = static_cast<int>(F);
  p.channels = channels;
  p.colored = colored;
  p.frame_stride = frame_stride;
  p.row_step = row_step;
  // Side planes of the undistorted pinhole frustum, camera frame, through the
  // optical centre; a landmark with z > 0 projects into [0,W) x [0,H) only if
  // it is on the inner side of all four.
  p.cull = (cam.distortion == AMHIP_DIST_NONE && cam.fu > 0.0 && cam.fv > 0.0) ? 1 : 0;
  double cone = 0.0;
  bool have_cone = false;
  if (cam.distortion != AMHIP_DIST_NONE && cam.fu > 0.0 && cam.fv > 0.0 &&
      !tuning_on("no_distorted_cull")) {
    // (a few hundred thousand evaluations: cached per camera)
    if (c.cone_state == 0 || std::memcmp(&c.cone_cam, &cam, sizeof(cam)) != 0) {
      c.cone_cam = cam;
      c.cone_state = distorted_view_cone(cam, &c.cone) ? 1 : 2;
      c.cone_state_rect = 0;
    }
    have_cone = c.cone_state == 1;
    cone = c.cone;
  }
  double ax = cone, ay = cone;
  p.r_in = 0.0;
  if (have_cone) {
    if (c.cone_state_rect != 1 || c.cone_rect_cone != cone) {
      distorted_rectangle_and_inner_cone(cam, cone, &c.cone_ax, &c.cone_ay, &c.cone_rin);
      c.cone_state_rect = 1;
      c.cone_rect_cone = cone;
    }
    ax = c.cone_ax;
    ay = c.cone_ay;
    if (!tuning_on("no_distorted_prune")) p.r_in = c.cone_rin;
    if (tuning_on("distorted_square_cull")) ax = ay = cone;  // (A/B: the circumscribed square)
  }
  if (have_cone) {
    // every visible landmark has |x| <= ax * z and |y| <= ay * z
    p.cull = 1;
    double l[3] = {1.0, 0.0, ax}, r[3] = {-1.0, 0.0, ax};
    double t[3] = {0.0, 1.0, ay}, b[3] = {0.0, -1.0, ay};
    normalize3(l);
    normalize3(r);
    normalize3(t);
    normalize3(b);
    for (int k = 0; k < 3; ++k) {
      p.pl[0][k] = l[k];
      p.pl[1][k] = r[k];
      p.pl[2][k] = t[k];
      p.pl[3][k] = b[k];
    }
  } else if (p.cull) {
    const double W = cam.width, H = cam.height;
    double l[3] = {cam.fu, 0.0, cam.cu};          // u >= 0
    double r[3] = {-cam.fu, 0.0, W - cam.cu};     // u <  W
    double t[3] = {0.0, cam.fv, cam.cv};          // v >= 0
    double b[3] = {0.0, -cam.fv, H - cam.cv};     // v <  H
    normalize3(l);
    normalize3(r);
    normalize3(t);
    normalize3(b);
    for (int k = 0; k < 3; ++k) {
      p.pl[0][k] = l[k];
      p.pl[1][k] = r[k];
      p.pl[2][k] = t[k];
      p.pl[3][k] = b[k];
    }
  }
  *out = p;
}

static float layer_init_value(int layer) {
  // aerial-mapper-grid-map.cc:40-48
  switch (layer) {
    case AMHIP_LAYER_ORTHO:
      return 255.0f;
    case AMHIP_LAYER_ELEVATION_ANGLE:
    case AMHIP_LAYER_NUM_OBSERVATIONS:
      return 0.0f;
    default:
      return std::numeric_limits<float>::quiet_NaN();
  }
}

static int use_device(Ctx* c) {
  AMHIP_TRY(hipSetDevice(c->device));
  return AMHIP_OK;
}

static int fetch_status(Ctx* c) {
  // dev_err -> pinned mirror, then clear
  AMHIP_TRY(hipMemcpyAsync(c->host_err, c->dev_err, sizeof(unsigned),
                           hipMemcpyDeviceToHost, c->stream));
  AMHIP_TRY(hipMemsetAsync(c->dev_err, 0, sizeof(unsigned), c->stream));
  AMHIP_TRY(hipStreamSynchronize(c->stream));
  const unsigned e = *c->host_err;
  if (e & kDevErrExactHit) {
    set_last_error(
        "a point coincides with a cell centre (reference: dsm.cc:165 "
        "CHECK(distances[i] > 0.0))");
    return AMHIP_ERR_EXACT_HIT;
  }
  if (e & kDevErrAlphaNonPos) {
    set_last_error(
        "observation angle alpha <= 0 (reference: ortho-backward-grid.cc:178 "
        "CHECK(alpha > 0.0))");
    return AMHIP_ERR_ALPHA_NONPOS;
  }
  if (e & kDevErrRectifyZeroW) {
    set_last_error("rectification: w == 0 (reference: rectifier.cpp:93,99 CHECK_NE(xyw(2), 0.0))");
    return AMHIP_ERR_ARG;
  }
  if (e & kDevErrHaloOverflow) {
    set_last_error(
        "tiled DSM: more halo points for a neighbouring window than send rows were reserved "
        "(cap_per_dest): the elevation near that edge is incomplete");
    return AMHIP_ERR_HALO_OVERFLOW;
  }
  return AMHIP_OK;
}

// tiled DSM: a selection that did not fit its send rows must fail the step (the points beyond
// the capacity were not shipped), without a host round trip inside the step
__global__ void k_halo_overflow_check(const unsigned long long* __restrict__ counts, int nd,
                                      unsigned long long cap, unsigned* __restrict__ dev_err) {
  if ((int)threadIdx.x < nd && counts[threadIdx.x] > cap) atomicOr(dev_err, kDevErrHaloOverflow);
}

static bool valid_layer(int l) { return l >= 0 && l < AMHIP_NUM_LAYERS; }

}  // namespace amhip

using namespace amhip;

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int amhip_abi_version(void) { return AMHIP_ABI_VERSION; }

const char* amhip_last_error(void) { return g_last_error.c_str(); }

void amhip_make_grid(double length_x, double length_y, double resolution,
                     double pos_x, double pos_y, amhip_grid_desc* out) {
  if (!out) return;
  out->rows = static_cast<int>(std::round(length_x / resolution));
  out->cols = static_cast<int>(std::round(length_y / resolution));
  out->resolution = resolution;
  out->length_x = static_cast<double>(out->rows) * resolution;
  out->length_y = static_cast<double>(out->cols) * resolution;
  out->pos_x = pos_x;
  out->pos_y = pos_y;
}

void amhip_cell_position(const amhip_grid_desc* grid, int i, int j, double* x,
                         double* y) {
  double bx, by;
  grid_bases(*grid, &bx, &by);
  if (x) *x = bx + grid->resolution * (-static_cast<double>(i));
  if (y) *y = by + grid->resolution * (-static_cast<double>(j));
}

int amhip_ctx_create(const amhip_grid_desc* grid, int device, amhip_ctx** out) {
  if (!grid) return arg_fail("amhip_ctx_create: null argument");
  return amhip_ctx_create_window(grid, 0, 0, grid->rows, grid->cols, device, out);
}

int amhip_ctx_create_window(const amhip_grid_desc* grid, int i0, int j0, int rows, int cols,
                            int device, amhip_ctx** out) {
  if (!grid || !out) return arg_fail("amhip_ctx_create: null argument");
  if (grid->rows <= 0 || grid->cols <= 0 || !(grid->resolution > 0.0))
    return arg_fail("amhip_ctx_create: empty grid");
  if (i0 < 0 || j0 < 0 || rows <= 0 || cols <= 0 || i0 + rows > grid->rows ||
      j0 + cols > grid->cols)
    return arg_fail("amhip_ctx_create_window: window outside the map");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    set_last_error("no HIP device visible (libaerial_mapper_hip needs a gfx950 GPU)");
    return AMHIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) return arg_fail("amhip_ctx_create: bad device index");
  amhip_ctx* h = new (std::nothrow) amhip_ctx();
  if (!h) return AMHIP_ERR_NOMEM;
  Ctx* c = &h->impl;
  c->grid = *grid;
  c->device = device;
  c->win_i0 = i0;
  c->win_j0 = j0;
  c->win_rows = rows;
  c->win_cols = cols;
  c->cells = static_cast<size_t>(rows) * static_cast<size_t>(cols);
  // reference-identical by default; FAST is opt-in (setter, or AMHIP_DSM_FAST=1 for hosts that
  // cannot be recompiled; AMHIP_DSM_EXACT=1 wins over it)
  {
    c->dsm_exact = amhip_default_dsm_precision() == AMHIP_DSM_EXACT ? 1 : 0;
  }
  int rc = AMHIP_OK;
  do {
    if ((rc = use_device(c))) break;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
      rc = hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
      break;
    }
    c->stream = c->own_stream;
    hipError_t e = hipSuccess;
    for (int l = 0; l < AMHIP_NUM_LAYERS && e == hipSuccess; ++l)
      e = hipMalloc(reinterpret_cast<void**>(&c->layers[l]), c->cells * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->dev_err), 4 * sizeof(unsigned));
    if (e == hipSuccess)
      e = hipMalloc(reinterpret_cast<void**>(&c->dev_zrange), 4 * sizeof(unsigned long long));  // [2], [3]: the last DSM call's own range
    if (e == hipSuccess)
      e = hipHostMalloc(reinterpret_cast<void**>(&c->host_err), sizeof(unsigned), 0);
    if (e == hipSuccess) {
      e = hipHostMalloc(reinterpret_cast<void**>(&c->host_tile_stats), 8 * sizeof(unsigned), 0);
      if (e == hipSuccess) std::memset(c->host_tile_stats, 0, 8 * sizeof(unsigned));
    }
    if (e == hipSuccess) {
      e = hipHostMalloc(reinterpret_cast<void**>(&c->host_sort_stats), 4 * sizeof(unsigned), 0);
      if (e == hipSuccess) std::memset(c->host_sort_stats, 0xFF, 4 * sizeof(unsigned));  // ("unknown")
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->dev_tickets), 16 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemsetAsync(c->dev_tickets, 0, 16 * sizeof(unsigned), c->stream);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->dev_bbox), 8 * sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->host_bbox), 8 * sizeof(int), 0);
    if (e == hipSuccess) e = hipMemsetAsync(c->dev_err, 0, sizeof(unsigned), c->stream);
    if (e != hipSuccess) {
      rc = hip_fail(e, "context allocation", __FILE__, __LINE__);
      break;
    }
    rc = amhip_layers_reset(h);
  } while (0);
  if (rc != AMHIP_OK) {
    const std::string keep = g_last_error;
    amhip_ctx_destroy(h);
    set_last_error(keep);
    return rc;
  }
  *out = h;
  return AMHIP_OK;
}

void amhip_ctx_destroy(amhip_ctx* h) {
  if (!h) return;
  Ctx* c = &h->impl;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  drain_timers(c);
  for (TimedRegion& r : c->free_regions) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (int l = 0; l < AMHIP_NUM_LAYERS; ++l)
    if (c->layers[l]) (void)hipFree(c->layers[l]);
  void* bufs[] = {c->dev_bbox, c->ortho_list, c->zpart, c->dev_zrange, c->tile_list, c->tile_occ, c->fill_mask, c->stage_values, c->dev_err, c->sorted,       c->rank,        c->bin_start, c->bin_z, c->rec_a, c->rec_b, c->rec16, c->sidx, c->zref, c->zall, c->tmp_points, c->stripe_ws,
                  c->scan_partials, c->stage_points, c->frame_poses, c->stage_frames};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (c->host_err) (void)hipHostFree(c->host_err);
  if (c->host_tile_stats) (void)hipHostFree(c->host_tile_stats);
  if (c->host_sort_stats) (void)hipHostFree(c->host_sort_stats);
  if (c->dev_tickets) (void)hipFree(c->dev_tickets);
  if (c->host_bbox) (void)hipHostFree(c->host_bbox);
  if (c->order_event) (void)hipEventDestroy(c->order_event);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete h;
}

int amhip_ctx_set_dsm_precision(amhip_ctx* h, int mode) {
  if (!h) return arg_fail("null context");
  if (mode != AMHIP_DSM_FAST && mode != AMHIP_DSM_EXACT) return arg_fail("unknown DSM precision mode");
  h->impl.dsm_exact = mode == AMHIP_DSM_EXACT ? 1 : 0;
  return AMHIP_OK;
}

int amhip_ctx_set_dsm_knn(amhip_ctx* h, int k) {
  if (!h) return arg_fail("null context");
  if (k < 0 || k > 8) return arg_fail("amhip_ctx_set_dsm_knn: k must be 0 (off) .. 8");
  h->impl.dsm_knn = k;
  return AMHIP_OK;
}

int amhip_ctx_set_stream(amhip_ctx* h, void* hip_stream) {
  if (!h) return arg_fail("null context");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  AMHIP_TRY(hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  return AMHIP_OK;
}

int amhip_ctx_synchronize(amhip_ctx* h) {
  if (!h) return arg_fail("null context");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  return fetch_status(c);
}

int amhip_layers_reset(amhip_ctx* h) {
  if (!h) return arg_fail("null context");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  // Lazy: nothing is written here.  A layer that is already in its initial
  // state (physically, 0, or logically, 3) stays as it is; a written one becomes
  // "logically initial": the next kernel that produces it writes the initial
  // value into the cells it leaves alone (fused fill), anything else that
  // needs the memory (download, device pointer, a partial writer) fills it
  // first (materialize()).  Layers whose device pointer was handed out are
  // refilled eagerly.  tuning knob eager_reset restores the plain fills.
  const bool eager = tuning_on("eager_reset");
  {  // every elevation is NaN again: empty height range
    static const unsigned long long empty[2] = {0xFFF0000000000000ull, 0x000FFFFFFFFFFFFFull};
    AMHIP_TRY(hipMemcpyAsync(c->dev_zrange, empty, sizeof(empty), hipMemcpyHostToDevice, c->stream));
    c->zrange_valid = true;
  }
  for (int l = 0; l < AMHIP_NUM_LAYERS; ++l) {
    unsigned char& st = c->layer_state[l];
    if (st == 0 || st == 3) continue;
    if (st == 1 && !eager) {
      st = 3;
      continue;
    }
    if ((rc = launch_fill(c, c->layers[l], c->cells, layer_init_value(l)))) return rc;
    if (st == 1) st = 0;
  }
  return AMHIP_OK;
}

// the layer's memory is about to be read or partially written by someone who
// does not know about the lazy state
static int materialize(Ctx* c, int layer) {
  if (c->layer_state[layer] != 3) return AMHIP_OK;
  const int rc = launch_fill(c, c->layers[layer], c->cells, layer_init_value(layer));
  if (rc) return rc;
  c->layer_state[layer] = 0;
  return AMHIP_OK;
}

// What the last writing call can have written: the whole window, unless the call narrows it
// afterwards (a small cloud's sub-window, a small batch's tile list).  Every writer passes here --
// touch() / overwrite() / the fused-fill branches -- so a rectangle never outlives its call
// (the tiled / OrthoFromPcl / empty-cloud entry points used to leave the previous one).
static void dirty_full(Ctx* c) {
  c->dirty_on_device = false;
  c->dirty[0] = c->dirty[1] = 0;
  c->dirty[2] = c->win_rows;
  c->dirty[3] = c->win_cols;
}

// partial writer: materialize, then dirty
static int touch(Ctx* c, int layer) {
  dirty_full(c);
  const int rc = materialize(c, layer);
  if (rc) return rc;
  if (c->layer_state[layer] == 0) c->layer_state[layer] = 1;
  return AMHIP_OK;
}

// full overwrite (upload): no need to fill first
static void overwrite(Ctx* c, int layer) {
  dirty_full(c);
  if (c->layer_state[layer] != 2) c->layer_state[layer] = 1;
  if (layer == AMHIP_LAYER_ELEVATION) c->zrange_valid = false;  // heights from outside
}

}  // extern "C"

namespace amhip {
int ctx_use_device(Ctx* c) { return use_device(c); }
int ctx_last_dirty(Ctx* c, int rect[4]) {
  if (c->dirty_on_device) {
    // (k_ortho_tile_list left [min tx, max tx, min ty, max ty] of the listed tiles behind the count)
    if (!c->ortho_list || !c->host_bbox) return arg_fail("no tile list");
    AMHIP_TRY(hipMemcpyAsync(c->host_bbox, c->ortho_list, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    AMHIP_TRY(hipStreamSynchronize(c->stream));
    const int* b = c->host_bbox;
    if (b[0] == 0 || b[4] > b[5] || b[6] > b[7]) {
      c->dirty[0] = c->dirty[1] = c->dirty[2] = c->dirty[3] = 0;
    } else {
      const int i0 = b[4] * 64, j0 = b[6] * 64;   // (the mosaic's tiles: 64 x 64 cells)
      c->dirty[0] = i0;
      c->dirty[1] = j0;
      c->dirty[2] = std::min(c->win_rows, (b[5] + 1) * 64) - i0;
      c->dirty[3] = std::min(c->win_cols, (b[7] + 1) * 64) - j0;
    }
    c->dirty_on_device = false;
  }
  for (int k = 0; k < 4; ++k) rect[k] = c->dirty[k];
  return AMHIP_OK;
}
int ctx_materialize(Ctx* c, int layer) { return materialize(c, layer); }
void ctx_overwrite(Ctx* c, int layer) { overwrite(c, layer); }
int ctx_fetch_status(Ctx* c) { return fetch_status(c); }
float ctx_layer_init_value(int layer) { return layer_init_value(layer); }
bool ctx_layer_is_initial(const Ctx* c, int layer) {
  return c->layer_state[layer] == 0 || c->layer_state[layer] == 3;
}
int arg_failure(const char* msg) { return arg_fail(msg); }
int ctx_layer_set_initial(Ctx* c, int layer) {
  unsigned char& st = c->layer_state[layer];
  if (layer == AMHIP_LAYER_ELEVATION) {  // every elevation is NaN again: empty height range
    static const unsigned long long empty[2] = {0xFFF0000000000000ull, 0x000FFFFFFFFFFFFFull};
    AMHIP_TRY(hipMemcpyAsync(c->dev_zrange, empty, sizeof(empty), hipMemcpyHostToDevice, c->stream));
    c->zrange_valid = true;
  }
  if (st == 0 || st == 3) return AMHIP_OK;
  if (st == 1) {
    st = 3;
    return AMHIP_OK;
  }
  return launch_fill(c, c->layers[layer], c->cells, layer_init_value(layer));  // (handed out: eager)
}
}  // namespace amhip

extern "C" {

int amhip_layer_upload(amhip_ctx* h, int layer, const float* host) {
  if (!h || !host || !valid_layer(layer)) return arg_fail("amhip_layer_upload: bad argument");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  overwrite(c, layer);
  AMHIP_TRY(hipMemcpyAsync(c->layers[layer], host, c->cells * sizeof(float),
                           hipMemcpyHostToDevice, c->stream));
  AMHIP_TRY(hipStreamSynchronize(c->stream));
  return AMHIP_OK;
}

int amhip_layer_download(amhip_ctx* h, int layer, float* host) {
  if (!h || !host || !valid_layer(layer)) return arg_fail("amhip_layer_download: bad argument");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = materialize(c, layer))) return rc;
  AMHIP_TRY(hipMemcpyAsync(host, c->layers[layer], c->cells * sizeof(float),
                           hipMemcpyDeviceToHost, c->stream));
  AMHIP_TRY(hipStreamSynchronize(c->stream));
  return AMHIP_OK;
}

void* amhip_layer_device_ptr(amhip_ctx* h, int layer) {
  if (!h || !valid_layer(layer)) return nullptr;
  Ctx* c = &h->impl;
  if (use_device(c) != AMHIP_OK || materialize(c, layer) != AMHIP_OK) return nullptr;
  c->layer_state[layer] = 2;  // the caller may write through the pointer at any time
  if (layer == AMHIP_LAYER_ELEVATION) c->zrange_valid = false;
  return c->layers[layer];
}

// ---- DSM ------------------------------------------------------------------

int amhip_dsm_process_dev(amhip_ctx* h, const double* dev_xyz, size_t n,
                          int radius_sq, double center_easting,
                          double center_northing) {
  if (!h) return arg_fail("null context");  // CHECK(map), dsm.cc:194
  if (n == 0) {                               // empty cloud: warning + return -- nothing written
    h->impl.dirty_on_device = false;
    h->impl.dirty[0] = h->impl.dirty[1] = h->impl.dirty[2] = h->impl.dirty[3] = 0;
    return AMHIP_OK;
  }
  if (!dev_xyz) return arg_fail("amhip_dsm_process_dev: null point pointer");
  if (radius_sq <= 0)
    return arg_fail("interpolation_radius must be >= 1 (the reference loops forever on 0)");
  if (n >= 0x7FFFFFFFull)
    return arg_fail("more than 2^31-1 points (the reference indexes results with int)");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  DsmParams p;
  dsm_rough_policy(c);
  if ((rc = make_dsm_params(*c, radius_sq, center_easting, center_northing, &p, 0, 1, n)))
    return rc;
  // a logically-initial elevation layer is filled by the gather itself
  const bool fused_fill = c->layer_state[AMHIP_LAYER_ELEVATION] == 3;
  if (fused_fill)
    c->layer_state[AMHIP_LAYER_ELEVATION] = 1;
  else if ((rc = touch(c, AMHIP_LAYER_ELEVATION)))
    return rc;
  dirty_full(c);
  if (!fused_fill) {  // a small cloud onto a large materialized map: its bounding box is the window
    DsmParams ps;
    const int sw = dsm_subwindow(c, dev_xyz, n, radius_sq, center_easting, center_northing, p, &ps);
    if (sw < 0) return -sw;
    if (sw == 2) {  // (no point within the last radius of any cell: every cell stays)
      c->dirty[2] = c->dirty[3] = 0;
      return AMHIP_OK;
    }
    if (sw == 1) {
      p = ps;
      c->dirty[0] = ps.out_i0;
      c->dirty[1] = ps.out_j0;
      c->dirty[2] = ps.rows;
      c->dirty[3] = ps.cols;
    }
  }
  return dsm_run(c, dev_xyz, nullptr, n, p, c->layers[AMHIP_LAYER_ELEVATION], nullptr, nullptr,
                 fused_fill, layer_init_value(AMHIP_LAYER_ELEVATION),
                 c->zrange_valid ? c->dev_zrange : nullptr);
}

int amhip_dsm_process(amhip_ctx* h, const double* host_xyz, size_t n,
                      int radius_sq, double center_easting,
                      double center_northing, float* elevation) {
  if (!h) return arg_fail("null context");
  if (n == 0) return AMHIP_OK;
  if (!host_xyz || !elevation) return arg_fail("amhip_dsm_process: null buffer");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = ensure_capacity(&c->stage_points, &c->stage_points_cap, 3 * n))) return rc;
  overwrite(c, AMHIP_LAYER_ELEVATION);
  AMHIP_TRY(hipMemcpyAsync(c->layers[AMHIP_LAYER_ELEVATION], elevation,
                           c->cells * sizeof(float), hipMemcpyHostToDevice, c->stream));
  AMHIP_TRY(hipMemcpyAsync(c->stage_points, host_xyz, 3 * n * sizeof(double),
                           hipMemcpyHostToDevice, c->stream));
  if ((rc = amhip_dsm_process_dev(h, c->stage_points, n, radius_sq, center_easting,
                                  center_northing)))
    return rc;
  AMHIP_TRY(hipMemcpyAsync(elevation, c->layers[AMHIP_LAYER_ELEVATION],
                           c->cells * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  return fetch_status(c);
}

// ---- OrthoFromPcl -------------------------------------------------------------

int amhip_ortho_from_pcl_process_dev(amhip_ctx* h, const double* dev_xyz,
                                     const int32_t* dev_intensities, size_t n, int radius_sq,
                                     int adaptive) {
  if (!h) return arg_fail("null context");                    // CHECK(map)
  if (n == 0 || !dev_xyz || !dev_intensities)
    return arg_fail("empty point cloud (CHECK(!pointcloud.empty()))");
  if (radius_sq <= 0) return arg_fail("interpolation_radius must be >= 1");
  if (n >= 0x7FFFFFFFull) return arg_fail("more than 2^31-1 points");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  float* out = c->layers[AMHIP_LAYER_ORTHO];
  DsmParams p;
  if ((rc = make_dsm_params(*c, radius_sq, 0.0, 0.0, &p, 1, 1, n))) return rc;
  if (!adaptive) {
    const bool fused_fill = c->layer_state[AMHIP_LAYER_ORTHO] == 3;
    dirty_full(c);
    if (fused_fill)
      c->layer_state[AMHIP_LAYER_ORTHO] = 1;
    else if ((rc = touch(c, AMHIP_LAYER_ORTHO)))
      return rc;
    return dsm_run(c, dev_xyz, dev_intensities, n, p, out, nullptr, nullptr, fused_fill,
                   layer_init_value(AMHIP_LAYER_ORTHO));
  }
  if ((rc = touch(c, AMHIP_LAYER_ORTHO))) return rc;

  // use_adaptive_interpolation: cells whose search is empty retry with the
  // squared radius x10, x100, ... (int lambda, ortho-from-pcl.cc:63-71).  Every
  // retry is a full pass restricted to the cells no earlier pass has filled.
  if ((rc = ensure_capacity(&c->fill_mask, &c->fill_mask_cap, c->cells))) return rc;
  unsigned* unfilled = c->dev_err + 1;  // second word of the device status block
  AMHIP_TRY(hipMemsetAsync(c->fill_mask, 0, c->cells, c->stream));
  AMHIP_TRY(hipMemsetAsync(unfilled, 0, sizeof(unsigned), c->stream));
  if ((rc = dsm_run(c, dev_xyz, dev_intensities, n, p, out, c->fill_mask, unfilled))) return rc;
  long long lambda = 10;
  for (;;) {
    unsigned left = 0;
    AMHIP_TRY(hipMemcpyAsync(&left, unfilled, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    AMHIP_TRY(hipStreamSynchronize(c->stream));
    if (left == 0) break;
    if (lambda * radius_sq > 0x7FFFFFFFll) break;  // the reference's int product overflows here
    if ((rc = make_dsm_params(*c, radius_sq, 0.0, 0.0, &p, 1, static_cast<int>(lambda), n))) return rc;
    p.only_unfilled = 1;
    AMHIP_TRY(hipMemsetAsync(unfilled, 0, sizeof(unsigned), c->stream));
    if ((rc = dsm_run(c, dev_xyz, dev_intensities, n, p, out, c->fill_mask, unfilled))) return rc;
    lambda *= 10;
  }
  return AMHIP_OK;
}

int amhip_ortho_from_pcl_process(amhip_ctx* h, const double* host_xyz,
                                 const int32_t* host_intensities, size_t n, int radius_sq,
                                 int adaptive, float* ortho) {
  if (!h) return arg_fail("null context");
  if (n == 0 || !host_xyz || !host_intensities || !ortho)
    return arg_fail("empty point cloud / null buffer (CHECK(!pointcloud.empty()))");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  if ((rc = ensure_capacity(&c->stage_points, &c->stage_points_cap, 3 * n))) return rc;
  if ((rc = ensure_capacity(&c->stage_values, &c->stage_values_cap, n))) return rc;
  overwrite(c, AMHIP_LAYER_ORTHO);
  AMHIP_TRY(hipMemcpyAsync(c->layers[AMHIP_LAYER_ORTHO], ortho, c->cells * sizeof(float),
                           hipMemcpyHostToDevice, c->stream));
  AMHIP_TRY(hipMemcpyAsync(c->stage_points, host_xyz, 3 * n * sizeof(double),
                           hipMemcpyHostToDevice, c->stream));
  AMHIP_TRY(hipMemcpyAsync(c->stage_values, host_intensities, n * sizeof(int32_t),
                           hipMemcpyHostToDevice, c->stream));
  if ((rc = amhip_ortho_from_pcl_process_dev(h, c->stage_points, c->stage_values, n, radius_sq,
                                             adaptive)))
    return rc;
  AMHIP_TRY(hipMemcpyAsync(ortho, c->layers[AMHIP_LAYER_ORTHO], c->cells * sizeof(float),
                           hipMemcpyDeviceToHost, c->stream));
  return fetch_status(c);
}

// ---- densifier reprojection -----------------------------------------------------

int amhip_densify_dev(amhip_ctx* h, const float* dev_disparity, size_t disp_step,
                      const uint8_t* dev_image_left, size_t img_step, int width, int height,
                      const double* K, double baseline, const double* R_G_C,
                      const double* t_G_C1, double* dev_xyz_out, int32_t* dev_intensity_out,
                      size_t capacity, int64_t* dev_count) {
  if (!h) return arg_fail("null context");
  if (!dev_disparity || !dev_image_left || !K || !R_G_C || !t_G_C1 || !dev_xyz_out ||
      !dev_intensity_out || !dev_count)
    return arg_fail("amhip_densify_dev: null argument");
  if (width <= 0 || height <= 0 || disp_step < (size_t)width * sizeof(float) ||
      img_step < (size_t)width)
    return arg_fail("amhip_densify_dev: bad image geometry");
  if (baseline == 0.0) return arg_fail("CHECK_NE(baseline, 0.0)");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  DensifyParams p;
  std::memset(&p, 0, sizeof(p));
  p.width = width;
  p.height = height;
  p.disp_step = disp_step;
  p.img_step = img_step;
  // stereo projection matrix Q (densifier.cpp:39-46), K row-major
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  p.Q03 = -cx;
  p.Q11 = fx / fy;
  p.Q13 = -cy * (fx / fy);
  p.Q23 = fx;
  p.Q32 = 1.0 / baseline;
  for (int k = 0; k < 9; ++k) p.R[k] = R_G_C[k];
  for (int k = 0; k < 3; ++k) p.t[k] = t_G_C1[k];
  return densify_run(c, p, dev_disparity, dev_image_left, dev_xyz_out, dev_intensity_out,
                     capacity, reinterpret_cast<long long*>(dev_count));
}

// ---- multi-GPU halo ----------------------------------------------------------

}  // extern "C"
namespace amhip {
int make_halo_params(const Ctx& c, double center_easting, double center_northing,
                     const int32_t* dest_windows, int nd, double margin_m,
                     size_t cap_per_dest, HaloParams* out) {
  HaloParams hp;
  std::memset(&hp, 0, sizeof(hp));
  grid_bases(c.grid, &hp.base_x, &hp.base_y);
  hp.inv_res = 1.0 / c.grid.resolution;
  hp.sub_x = center_northing;  // dsm.cc:42
  hp.sub_y = center_easting;   // dsm.cc:43
  hp.nd = nd;
  hp.cap = cap_per_dest;
  const double mc = margin_m / c.grid.resolution;  // margin in cells
  for (int d = 0; d < nd; ++d) {
    const int32_t* w = dest_windows + 4 * d;
    if (w[2] <= 0 || w[3] <= 0) return arg_fail("halo selection: empty window");
    hp.lo_i[d] = (double)w[0] - 0.5 - mc;
    hp.hi_i[d] = (double)(w[0] + w[2]) - 0.5 + mc;
    hp.lo_j[d] = (double)w[1] - 0.5 - mc;
    hp.hi_j[d] = (double)(w[1] + w[3]) - 0.5 + mc;
    hp.off[d] = (unsigned long long)d * (unsigned long long)cap_per_dest;
    hp.cap_d[d] = cap_per_dest;
  }
  // the context's own window shrunk by the margin, if no destination reaches into it
  // (windows of one tiling never do)
  hp.in_lo_i = (double)c.win_i0 - 0.5 + mc + 1.0;
  hp.in_hi_i = (double)(c.win_i0 + c.win_rows) - 0.5 - mc - 1.0;
  hp.in_lo_j = (double)c.win_j0 - 0.5 + mc + 1.0;
  hp.in_hi_j = (double)(c.win_j0 + c.win_cols) - 0.5 - mc - 1.0;
  bool clear = hp.in_lo_i < hp.in_hi_i && hp.in_lo_j < hp.in_hi_j;
  for (int d = 0; d < nd && clear; ++d)
    clear = hp.hi_i[d] < hp.in_lo_i || hp.lo_i[d] > hp.in_hi_i || hp.hi_j[d] < hp.in_lo_j ||
            hp.lo_j[d] > hp.in_hi_j;
  if (!clear) {
    hp.in_lo_i = hp.in_lo_j = 1.0;
    hp.in_hi_i = hp.in_hi_j = 0.0;
  }
  *out = hp;
  return AMHIP_OK;
}
}  // namespace amhip
extern "C" {

int amhip_halo_select_dev(amhip_ctx* h, const double* dev_xyz, size_t n,
                          double center_easting, double center_northing,
                          const int32_t* dest_windows, int nd, double margin_m,
                          double* dev_out, size_t cap_per_dest, int64_t* dev_counts) {
  if (!h) return arg_fail("null context");
  if (nd < 0 || nd > kMaxHaloDests) return arg_fail("amhip_halo_select_dev: 0..8 destinations");
  if (nd == 0) return AMHIP_OK;
  if ((n && !dev_xyz) || !dest_windows || !dev_out || !dev_counts || !(margin_m >= 0.0))
    return arg_fail("amhip_halo_select_dev: bad argument");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  HaloParams hp;
  if ((rc = make_halo_params(*c, center_easting, center_northing, dest_windows, nd, margin_m,
                             cap_per_dest, &hp)))
    return rc;
  return halo_select_run(c, dev_xyz, n, hp, dev_out,
                         reinterpret_cast<unsigned long long*>(dev_counts));
}

int amhip_dsm_tiled_begin_dev(amhip_ctx* h, const double* dev_xyz, size_t n_owned,
                              size_t n_total, int radius_sq, double center_easting,
                              double center_northing, const int32_t* dest_windows, int nd,
                              double margin_m, double* dev_out, size_t cap_per_dest,
                              int64_t* dev_counts) {
  if (!h) return arg_fail("null context");
  Ctx* c = &h->impl;
  c->tiled_pending = false;
  if (!dev_xyz || n_total == 0 || n_owned > n_total)
    return arg_fail("amhip_dsm_tiled_begin_dev: bad point arguments");
  if (radius_sq <= 0) return arg_fail("interpolation_radius must be >= 1");
  if (n_total >= 0x7FFFFFFFull) return arg_fail("more than 2^31-1 points");
  if (nd < 1 || nd > kMaxHaloDests || !dest_windows || !dev_out || !dev_counts ||
      !(margin_m >= 0.0) || cap_per_dest == 0)
    return arg_fail("amhip_dsm_tiled_begin_dev: bad halo arguments (1..8 destinations)");
  int rc = use_device(c);
  if (rc) return rc;
  SortSplit sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.phase = 1;
  sp.n_prefix = n_owned;
  if ((rc = make_halo_params(*c, center_easting, center_northing, dest_windows, nd, margin_m,
                             cap_per_dest, &sp.hp)))
    return rc;
  sp.halo_out = dev_out;
  sp.halo_counts = reinterpret_cast<unsigned long long*>(dev_counts);
  DsmParams p;
  dsm_rough_policy(c);
  if ((rc = make_dsm_params(*c, radius_sq, center_easting, center_northing, &p, 0, 1, n_total)))
    return rc;
  if ((rc = dsm_run(c, dev_xyz, nullptr, n_total, p, nullptr, nullptr, nullptr, false, 0.0f,
                    nullptr, &sp)))
    return rc;
  c->tiled_pending = true;
  c->tiled_xyz = dev_xyz;
  c->tiled_n = n_total;
  c->tiled_radius_sq = radius_sq;
  c->tiled_ce = center_easting;
  c->tiled_cn = center_northing;
  c->tiled_split = sp;
  c->tiled_params = p;
  return AMHIP_OK;
}

int amhip_dsm_tiled_finish_dev(amhip_ctx* h) {
  if (!h) return arg_fail("null context");
  Ctx* c = &h->impl;
  if (!c->tiled_pending)
    return arg_fail("amhip_dsm_tiled_finish_dev without amhip_dsm_tiled_begin_dev");
  c->tiled_pending = false;
  int rc = use_device(c);
  if (rc) return rc;
  // (the plan of the begin call, not a fresh one: amhip_ctx_set_dsm_precision / _knn between the
  // two calls must not make the gather disagree with the binning of phase 1)
  const DsmParams p = c->tiled_params;
  SortSplit sp = c->tiled_split;
  sp.phase = 2;
  hipLaunchKernelGGL(k_halo_overflow_check, dim3(1), dim3(64), 0, c->stream, sp.halo_counts,
                     sp.hp.nd, sp.hp.cap, c->dev_err);
  const bool fused_fill = c->layer_state[AMHIP_LAYER_ELEVATION] == 3;
  dirty_full(c);
  if (fused_fill)
    c->layer_state[AMHIP_LAYER_ELEVATION] = 1;
  else if ((rc = touch(c, AMHIP_LAYER_ELEVATION)))
    return rc;
  return dsm_run(c, c->tiled_xyz, nullptr, c->tiled_n, p, c->layers[AMHIP_LAYER_ELEVATION],
                 nullptr, nullptr, fused_fill, layer_init_value(AMHIP_LAYER_ELEVATION),
                 c->zrange_valid ? c->dev_zrange : nullptr, &sp);
}

// ---- ortho ----------------------------------------------------------------

void amhip_compose_T_G_C(const double* T_G_B, const double* T_C_B, size_t F,
                         double* T_G_C) {
  const HPose T_B_C = hpose_inverse(hpose_from7(T_C_B));
  for (size_t f = 0; f < F; ++f) {
    const HPose p = hpose_compose(hpose_from7(T_G_B + 7 * f), T_B_C);
    double* o = T_G_C + 7 * f;
    o[0] = p.tx;
    o[1] = p.ty;
    o[2] = p.tz;
    o[3] = p.qw;
    o[4] = p.qx;
    o[5] = p.qy;
    o[6] = p.qz;
  }
}

int amhip_camera_view_bounds(const amhip_camera* cam, double* out4) {
  if (!cam || !out4) return arg_fail("amhip_camera_view_bounds: null argument");
  if (!(cam->fu > 0.0) || !(cam->fv > 0.0) || cam->width <= 0 || cam->height <= 0)
    return arg_fail("amhip_camera_view_bounds: bad camera");
  out4[0] = out4[1] = out4[2] = out4[3] = 0.0;
  if (cam->distortion == AMHIP_DIST_NONE) {
    out4[1] = std::max(cam->cu, (double)cam->width - cam->cu) / cam->fu;
    out4[2] = std::max(cam->cv, (double)cam->height - cam->cv) / cam->fv;
    return AMHIP_OK;
  }
  double cone = 0.0;
  if (!distorted_view_cone(*cam, &cone)) return AMHIP_OK;
  out4[0] = cone;
  distorted_rectangle_and_inner_cone(*cam, cone, &out4[1], &out4[2], &out4[3]);
  return AMHIP_OK;
}

int amhip_ortho_backward_process_dev(amhip_ctx* h, const amhip_camera* cam,
                                     const double* host_T_G_C, size_t F,
                                     const uint8_t* dev_frames,
                                     size_t frame_stride, size_t row_step,
                                     int channels, int colored) {
  if (!h || !cam) return arg_fail("null context / camera");
  if (F == 0 || !host_T_G_C) return arg_fail("empty pose list (CHECK(!T_G_Bs.empty()))");
  if (!dev_frames) return arg_fail("null frame pointer");
  if (colored ? channels != 3 : channels != 1)
    return arg_fail("colored_ortho needs 8UC3 frames, gray needs 8UC1");
  if (cam->width <= 0 || cam->height <= 0) return arg_fail("bad image size");
  if (row_step < (size_t)cam->width * (size_t)channels ||
      frame_stride < row_step * (size_t)cam->height)
    return arg_fail("frame strides smaller than the image");
  if (F >= (1u << 24)) return arg_fail("too many frames for a float observation_index");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  // one buffer, one upload: FramePose[F], then FrameFast[F + 2] (two FramePose
  // slots each; the two extra entries carry the camera for exact_view(),
  // doubles 0..5, and the atan table of fold_finish(), doubles 8..24)
  const size_t slots = F + 2 * (F + 2);
  if ((rc = ensure_capacity(&c->frame_poses, &c->frame_pose_cap, slots))) return rc;

  // T_C_G[f] = T_G_C[f].inverse()  (ortho-backward-grid.cc:157-158; the
  // reference recomputes it per cell and frame, the value is the same)
  static_assert(sizeof(FrameFast) == 2 * sizeof(FramePose), "frame table layout");
  std::vector<FramePose> inv(slots);
  FrameFast* fast = reinterpret_cast<FrameFast*>(inv.data() + F);
  double qdev = 0.0;  // max | |q|^2 - 1 |
  bool fast_ok = cam->distortion == AMHIP_DIST_NONE && cam->fu > 0.0 && cam->fv > 0.0 &&
                 !tuning_on("ortho_exact_fold");
  for (size_t f = 0; f < F; ++f) {
    const HPose T = hpose_inverse(hpose_from7(host_T_G_C + 7 * f));
    inv[f].qw = T.qw;
    inv[f].qx = T.qx;
    inv[f].qy = T.qy;
    inv[f].qz = T.qz;
    inv[f].tx = T.tx;
    inv[f].ty = T.ty;
    inv[f].tz = T.tz;
    inv[f]._pad = 0.0;
    // (non-unit quaternion / non-finite pose: the whole call takes the exact kernel)
    if (!make_frame_fast(inv[f], &fast[f])) fast_ok = false;
    const double dev = std::fabs(T.qw * T.qw + T.qx * T.qx + T.qy * T.qy + T.qz * T.qz - 1.0);
    if (!(dev <= qdev)) qdev = dev;  // (NaN sticks)
  }
  {
    double* camd = reinterpret_cast<double*>(&fast[F]);
    std::memset(camd, 0, 2 * sizeof(FrameFast));
    make_atan_table(camd + 8);
    camd[0] = cam->fu;
    camd[1] = cam->fv;
    camd[2] = cam->cu;
    camd[3] = cam->cv;
    camd[4] = (double)cam->width;
    camd[5] = (double)cam->height;
  }
  // pageable source: hipMemcpyAsync returns once it has been staged
  AMHIP_TRY(hipMemcpyAsync(c->frame_poses, inv.data(), slots * sizeof(FramePose),
                           hipMemcpyHostToDevice, c->stream));

  OrthoParams p;
  make_ortho_params(*c, *cam, F, frame_stride, row_step, channels, colored, &p);
  // layer states (lazy reset): the kernel reads elevation; elevation_angle,
  // observation_index and the output layer are either all produced with a
  // fused fill or all materialized; num_observations stays logically 0
  const int out_layer = colored ? AMHIP_LAYER_COLORED_ORTHO : AMHIP_LAYER_ORTHO;
  if ((rc = materialize(c, AMHIP_LAYER_ELEVATION))) return rc;
  const int outs[3] = {AMHIP_LAYER_ELEVATION_ANGLE, AMHIP_LAYER_OBSERVATION_INDEX, out_layer};
  p.virt_out = 1;
  for (int k = 0; k < 3; ++k)
    if (c->layer_state[outs[k]] != 3) p.virt_out = 0;
  dirty_full(c);  // (ortho_run narrows it to the tile list's box where one is walked)
  for (int k = 0; k < 3; ++k) {
    if (p.virt_out)
      c->layer_state[outs[k]] = 1;
    else if ((rc = touch(c, outs[k])))
      return rc;
  }
  p.radius_scale = 1.0 + 2.0 * qdev;
  if (!(qdev < 0.25)) p.cull = 0;  // not a rotation at all (or NaN): every frame is tested
  // small batches on a big map: most tiles are out of every frame's sight
  p.coarse = (p.cull && c->zrange_valid && F <= 64 && !tuning_on("no_coarse_cull")) ? 1 : 0;
  // num_observations is zero everywhere while it holds its initial value,
  // filled (0) or not (3): `+= itself` keeps it zero, the kernel neither reads
  // nor writes it and the layer stays in that state
  {
    const unsigned char st = c->layer_state[AMHIP_LAYER_NUM_OBSERVATIONS];
    p.virt_nobs = (st == 3 || st == 0) ? 1 : 0;
  }
  if (!p.virt_nobs && (rc = touch(c, AMHIP_LAYER_NUM_OBSERVATIONS))) return rc;
  p.prune = (p.cull && (cam->distortion == AMHIP_DIST_NONE || p.r_in > 0.0) && p.virt_nobs &&
             !tuning_on("ortho_no_prune")) ? 1 : 0;
  p.fast = fast_ok ? 1 : 0;
  p.fold = make_fold_cam(cam->fu, cam->fv, cam->cu, cam->cv, cam->width, cam->height);
  return ortho_run(c, p, c->frame_poses, reinterpret_cast<const FrameFast*>(c->frame_poses + F),
                   dev_frames);
}

int amhip_ortho_backward_process(
    amhip_ctx* h, const amhip_camera* cam, const double* host_T_G_C, size_t F,
    const uint8_t* const* images, const size_t* steps, int channels,
    int colored, const float* elevation, float* elevation_angle,
    float* observation_index, float* num_observations, float* ortho,
    float* colored_ortho) {
  if (!h || !cam) return arg_fail("null context / camera");
  if (F == 0 || !host_T_G_C || !images || !steps)
    return arg_fail("empty pose / image list (CHECK(!T_G_Bs.empty()))");
  if (cam->width <= 0 || cam->height <= 0) return arg_fail("bad image size");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  const size_t bytes = c->cells * sizeof(float);
  const float* ups[AMHIP_NUM_LAYERS] = {ortho,           elevation,
                                        elevation_angle, num_observations,
                                        observation_index, colored_ortho};
  for (int l = 0; l < AMHIP_NUM_LAYERS; ++l)
    if (ups[l]) {
      overwrite(c, l);
      AMHIP_TRY(hipMemcpyAsync(c->layers[l], ups[l], bytes, hipMemcpyHostToDevice,
                               c->stream));
    }
  // stage the frames densely: [F][H][W*channels]
  const size_t row = (size_t)cam->width * (size_t)channels;
  const size_t frame = row * (size_t)cam->height;
  if ((rc = ensure_capacity(&c->stage_frames, &c->stage_frames_cap, frame * F))) return rc;
  for (size_t f = 0; f < F; ++f) {
    if (!images[f]) return arg_fail("null image");
    if (steps[f] < row) return arg_fail("image step smaller than a row");
    if (steps[f] == row)  // dense rows (the usual cv::Mat): one linear copy
      AMHIP_TRY(hipMemcpyAsync(c->stage_frames + f * frame, images[f], frame,
                               hipMemcpyHostToDevice, c->stream));
    else
      AMHIP_TRY(hipMemcpy2DAsync(c->stage_frames + f * frame, row, images[f], steps[f],
                                 row, (size_t)cam->height, hipMemcpyHostToDevice,
                                 c->stream));
  }
  if ((rc = amhip_ortho_backward_process_dev(h, cam, host_T_G_C, F, c->stage_frames,
                                             frame, row, channels, colored)))
    return rc;
  float* downs[AMHIP_NUM_LAYERS] = {ortho,           nullptr,
                                    elevation_angle, num_observations,
                                    observation_index, colored_ortho};
  for (int l = 0; l < AMHIP_NUM_LAYERS; ++l)
    if (downs[l]) {
      if ((rc = materialize(c, l))) return rc;
      AMHIP_TRY(hipMemcpyAsync(downs[l], c->layers[l], bytes, hipMemcpyDeviceToHost,
                               c->stream));
    }
  return fetch_status(c);
}

// ---- measurement ------------------------------------------------------------

int amhip_ctx_enable_timing(amhip_ctx* h, int on) {
  if (!h) return arg_fail("null context");
  h->impl.timing = on != 0;
  return AMHIP_OK;
}

int amhip_ctx_timing_reset(amhip_ctx* h) {
  if (!h) return arg_fail("null context");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  drain_timers(c);
  for (int k = 0; k < AMHIP_NUM_KERNELS; ++k) {
    c->slot_ms[k] = 0.0;
    c->slot_launches[k] = 0;
  }
  return AMHIP_OK;
}

int amhip_ctx_kernel_time(amhip_ctx* h, int kernel, double* total_ms,
                          int64_t* launches) {
  if (!h || kernel < 0 || kernel >= AMHIP_NUM_KERNELS) return arg_fail("bad kernel slot");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  drain_timers(c);
  if (total_ms) *total_ms = c->slot_ms[kernel];
  if (launches) *launches = c->slot_launches[kernel];
  return AMHIP_OK;
}

const char* amhip_kernel_name(int kernel) {
  switch (kernel) {
    // sort slots: named after the kernels of the default path for large clouds
    // (three-pass partition sort); the one-level sort of small clouds reports its
    // count / scan / scatter launches in the same slots
    case AMHIP_K_DSM_BIN_COUNT:
      return "k_dsm_p3_count";    // + k_dsm_p3_reduce, k_dsm_p3_scan
    case AMHIP_K_DSM_SCATTER:
      return "k_dsm_p3_scatter";  // both scatter passes
    case AMHIP_K_DSM_SCAN:
      return "k_dsm_p3_place";
    case AMHIP_K_DSM_GATHER:
      return "k_dsm_gather";
    case AMHIP_K_ORTHO:
      return "k_ortho_backward";
    case AMHIP_K_MISC:
      return "memset/fill";
    case AMHIP_K_HALO_SELECT:
      return "k_halo_select";
    default:
      return "?";
  }
}

int amhip_set_tuning(const char* key, double value) {
  if (!tuning_set(key, value)) return arg_fail("amhip_set_tuning: unknown key (include/aerial_mapper_hip.h lists them)");
  return AMHIP_OK;
}

double amhip_get_tuning(const char* key, double dflt) { return key ? tuning(key, dflt) : dflt; }

int amhip_default_dsm_precision(void) {
  // AMHIP_DSM_FAST=1 (and not AMHIP_DSM_EXACT): hosts that cannot be recompiled opt in to the
  // single-precision gather; everything else starts in the reference's arithmetic
  const char* fast = std::getenv("AMHIP_DSM_FAST");
  return (fast && fast[0] && fast[0] != '0' && !std::getenv("AMHIP_DSM_EXACT")) ? AMHIP_DSM_FAST : AMHIP_DSM_EXACT;
}

int amhip_ctx_order_after(amhip_ctx* h, void* other_stream) {
  if (!h) return arg_fail("null context");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  hipStream_t other = static_cast<hipStream_t>(other_stream);
  if (other == c->stream) return AMHIP_OK;  // (already ordered)
  if (!c->order_event) AMHIP_TRY(hipEventCreateWithFlags(&c->order_event, hipEventDisableTiming));
  AMHIP_TRY(hipEventRecord(c->order_event, c->stream));
  AMHIP_TRY(hipStreamWaitEvent(other, c->order_event, 0));
  return AMHIP_OK;
}

int amhip_ctx_dsm_gather_stats(amhip_ctx* h, int64_t* out8) {
  if (!h || !out8) return arg_fail("amhip_ctx_dsm_gather_stats: null argument");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  if (!c->tile_list || c->last_ntiles == 0) return AMHIP_OK;
  unsigned hdr[8];
  AMHIP_TRY(hipMemcpyAsync(hdr, c->tile_list, sizeof(hdr), hipMemcpyDeviceToHost, c->stream));
  AMHIP_TRY(hipStreamSynchronize(c->stream));
  for (int k = 0; k < 7; ++k) out8[k] = hdr[k];
  out8[4] += hdr[7];  // (pre-classified tiles the dense FP64 launch took instead of list 4)
  out8[7] = c->last_ntiles;
  return AMHIP_OK;
}

int amhip_ctx_dsm_stats(amhip_ctx* h, int64_t* points_binned, int64_t* num_bins,
                        int32_t* bin_cells) {
  if (!h) return arg_fail("null context");
  Ctx* c = &h->impl;
  int rc = use_device(c);
  if (rc) return rc;
  if (points_binned) {
    *points_binned = 0;
    if (c->bin_start && c->last_num_bins > 0) {
      uint32_t total = 0;
      AMHIP_TRY(hipMemcpyAsync(&total, c->bin_start + c->last_num_bins,
                               sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      AMHIP_TRY(hipStreamSynchronize(c->stream));
      *points_binned = total;
    }
  }
  if (num_bins) *num_bins = c->last_num_bins;
  if (bin_cells) *bin_cells = c->last_bin_cells;
  return AMHIP_OK;
}

}  // extern "C"
