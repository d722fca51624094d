// tests/cpp/ortho_fold_emul.cc -- CPU emulation of the margin-guarded fold the
// kernel k_ortho_backward<true> runs (aerial_mapper_amd/csrc/amhip_ortho_fold.h
// is compiled here unchanged, with g++ -ffp-contract=off; fma() is libm's
// correctly rounded one, as on the device).  TEST INFRASTRUCTURE: lets the CPU
// suite check the fast path's decisions against the oracle on millions of
// (cell, frame) pairs, including engineered ties and image-border hits, before
// anything runs on a GPU.  No culling here (culling only removes frames that
// are invisible from the whole tile).
#include <cmath>
#include <cstdint>
#include <vector>

#include "amhip_ortho_fold.h"

using namespace amhip;

namespace {

struct Provider {
  const double* cam;
  const FramePose* poses;
  double lx, ly, lz;
  long long* stats;
  ExactView view(int f) const {
    stats[3]++;
    return exact_view_inline(cam, poses[f], lx, ly, lz);
  }
  double angle(double absz, double n2) const {
    stats[4]++;
    const double norm = std::sqrt(n2);
    return std::asin(absz / norm);
  }
};

// minkindr inverse(): (q*, -(q* (x) t)), through the header's transform
FramePose inverse_pose(const double* T7) {
  FramePose r;
  r.qw = T7[3];
  r.qx = -T7[4];
  r.qy = -T7[5];
  r.qz = -T7[6];
  r.tx = r.ty = r.tz = 0.0;
  r._pad = 0.0;
  const V3 t = {T7[0], T7[1], T7[2]};
  const V3 rt = transform_point(r, t);  // + 0.0 leaves the rotation's doubles as they are
  r.tx = -rt.x;
  r.ty = -rt.y;
  r.tz = -rt.z;
  return r;
}

}  // namespace

extern "C" {

// Layers are column-major rows x cols (i + j * rows), like the map's.
//   stats[0] pairs folded, [1] visibility decided by exact_view, [2] near ties,
//   [3] exact_view calls, [4] asin calls, [5] poses the fast path refused
// Returns 0, or 3 (AMO_ERR_ALPHA_NONPOS) when CHECK(alpha > 0) would fire.
int emul_ortho_fold(int rows, int cols, double base_x, double base_y, double res,
                    const double* camera /* fu fv cu cv W H */, const double* T_G_C, int F,
                    const float* elevation, float* elevation_angle, float* observation_index,
                    int32_t* kp_x, int32_t* kp_y, int32_t* accepted, long long* stats) {
  std::vector<FramePose> poses(F);
  std::vector<FrameFast> fast(F);
  bool all_ok = true;
  for (int f = 0; f < F; ++f) {
    poses[f] = inverse_pose(T_G_C + 7 * f);
    if (!make_frame_fast(poses[f], &fast[f])) {
      all_ok = false;
      stats[5]++;
    }
  }
  if (!all_ok) return -1;  // the library then runs its exact kernel
  const FoldCam k = make_fold_cam(camera[0], camera[1], camera[2], camera[3], (int)camera[4],
                                  (int)camera[5]);
  bool bad = false;
  for (int j = 0; j < cols; ++j) {
    for (int i = 0; i < rows; ++i) {
      const size_t at = (size_t)i + (size_t)j * (size_t)rows;
      const float e = elevation[at];
      kp_x[at] = kp_y[at] = -1;
      accepted[at] = 0;
      if (!(e == e)) continue;  // NaN elevation is never visible
      const double lx = base_x + res * (-(double)i);
      const double ly = base_y + res * (-(double)j);
      const double lz = (double)e;
      const double magL = std::fabs(lx) + std::fabs(ly) + std::fabs(lz);
      CellFold s;
      fold_init(&s, elevation_angle[at]);
      Provider ex = {camera, poses.data(), lx, ly, lz, stats};
      for (int f = 0; f < F; ++f) {
        const FrameFast& Q = fast[f];
        const double mag = magL + Q.tmag;
        const double zthr = fma(0x1p-22, mag, 1e-10);
        const double muv = k.kuv * mag;
        const double bx = fma(Q.m[0], lx, Q.t[0]);
        const double by = fma(Q.m[3], lx, Q.t[1]);
        const double bz = fma(Q.m[6], lx, Q.t[2]);
        const double cx = fma(Q.m[2], lz, fma(Q.m[1], ly, bx));
        const double cy = fma(Q.m[5], lz, fma(Q.m[4], ly, by));
        const double cz = fma(Q.m[8], lz, fma(Q.m[7], ly, bz));
        stats[0]++;
        const long long views_before = stats[3];
        const long long asin_before = stats[4];
        fold_pair(&s, f, k, true, cx, cy, cz, zthr, muv, ex, &bad);
        if (stats[4] == asin_before && stats[3] != views_before) stats[1]++;
        if (stats[4] != asin_before) stats[2]++;
      }
      int ku = 0, kv = 0;
      if (fold_finish(&s, ex, (int)camera[4], (int)camera[5], &ku, &kv, &bad)) {
        elevation_angle[at] = s.best;
        observation_index[at] = (float)s.best_f;
        kp_x[at] = ku;
        kp_y[at] = kv;
        accepted[at] = s.accepted;
      }
    }
  }
  return bad ? 3 : 0;
}

}  // extern "C"
