// tests/cpp/ortho_fold_emul.cc -- CPU emulation of the margin-guarded fold the
// kernel k_ortho_backward<true> runs (aerial_mapper_amd/csrc/amhip_ortho_fold.h
// is compiled here unchanged, with g++ -ffp-contract=off; fma() is libm's
// correctly rounded one, as on the device).  TEST INFRASTRUCTURE: lets the CPU
// suite check the fast path's decisions against the oracle on millions of
// (cell, frame) pairs, including engineered ties and image-border hits, before
// anything runs on a GPU.  No culling here (culling only removes frames that
// are invisible from the whole tile).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "amhip_ortho_fold.h"

using namespace amhip;

namespace {

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
//   stats[0] pairs folded, [1] cells replayed by exact_refold (a decision inside a margin),
//   [5] poses the fast path refused,
//   [6] (tile, frame) pairs dropped as dominated, [7] frames kept over all tiles,
//   [8] cells whose keypoint / angle came from exact_finish
// prune: build every 64 x 16 tile's frame list like cull_chunk() does (sphere cull +
// dominance pruning); 0: every frame for every cell.
// Returns 0, or 3 (AMO_ERR_ALPHA_NONPOS) when CHECK(alpha > 0) would fire.
int emul_ortho_fold(int rows, int cols, double base_x, double base_y, double res,
                    const double* camera /* fu fv cu cv W H */, const double* T_G_C, int F,
                    const float* elevation, float* elevation_angle, float* observation_index,
                    int32_t* kp_x, int32_t* kp_y, int32_t* accepted, int prune, long long* stats) {
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
  double atan_tab[kAtanTabSize];
  make_atan_table(atan_tab);
  bool bad = false;
  // side planes of the undistorted view pyramid, unit inward normals
  // (amhip_api.hip: make_ortho_params)
  double pl[4][3] = {{camera[0], 0.0, camera[2]},
                     {-camera[0], 0.0, camera[4] - camera[2]},
                     {0.0, camera[1], camera[3]},
                     {0.0, -camera[1], camera[5] - camera[3]}};
  for (auto& n : pl) {
    const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (double& v : n) v /= len;
  }
  const int kTileI = 64, kTileJ = 64;  // amhip_ortho.hip
  std::vector<int> cand;
  for (int j0 = 0; j0 < cols; j0 += kTileJ) {
    for (int i0 = 0; i0 < rows; i0 += kTileI) {
      const int i_hi = std::min(i0 + kTileI, rows) - 1;
      const int j_hi = std::min(j0 + kTileJ, cols) - 1;
      // ---- the tile's frame list: cull + dominance pruning, like cull_chunk() ----
      cand.clear();
      if (!prune) {
        for (int f = 0; f < F; ++f) cand.push_back(f);
      } else {
        float zmin = HUGE_VALF, zmax = -HUGE_VALF;
        for (int j = j0; j <= j_hi; ++j)
          for (int i = i0; i <= i_hi; ++i) {
            const float e = elevation[(size_t)i + (size_t)j * (size_t)rows];
            if (e == e) {
              zmin = std::fmin(zmin, e);
              zmax = std::fmax(zmax, e);
            }
          }
        if (zmin <= zmax) {
          const double xa = base_x + res * (-(double)i0), xb = base_x + res * (-(double)i_hi);
          const double ya = base_y + res * (-(double)j0), yb = base_y + res * (-(double)j_hi);
          const double hx = 0.5 * std::fabs(xa - xb), hy = 0.5 * std::fabs(ya - yb);
          const V3 centre = {0.5 * (xa + xb), 0.5 * (ya + yb), 0.5 * ((double)zmin + (double)zmax)};
          const double hz = 0.5 * ((double)zmax - (double)zmin);
          const double radius = std::sqrt(hx * hx + hy * hy + hz * hz) * (1.0 + 1e-9) + 1e-6;
          const double slack =
              1e-6 + 0x1p-40 * (std::fabs(centre.x) + std::fabs(centre.y) + std::fabs(centre.z) + radius) * 2.0;
          std::vector<FrameBounds> fb(F);
          double best = HUGE_VAL;
          for (int f = 0; f < F; ++f) {
            fb[f] = frame_bounds(pl, poses[f], centre, radius, slack);
            if (fb[f].full) best = std::fmin(best, fb[f].tmax);
          }
          for (int f = 0; f < F; ++f) {
            if (!fb[f].keep) continue;
            if (dominated(fb[f].tmin, best)) {
              stats[6]++;
              continue;
            }
            cand.push_back(f);
          }
          stats[7] += (long long)cand.size();
        }
      }
      for (int j = j0; j <= j_hi; ++j) {
        for (int i = i0; i <= i_hi; ++i) {
          const size_t at = (size_t)i + (size_t)j * (size_t)rows;
          const float e = elevation[at];
          kp_x[at] = kp_y[at] = -1;
          accepted[at] = 0;
          if (!(e == e)) continue;  // NaN elevation is never visible
          const double lx = base_x + res * (-(double)i);
          const double ly = base_y + res * (-(double)j);
          const double lz = (double)e;
          const double magL = std::fabs(lx) + std::fabs(ly) + std::fabs(lz);
          const double epsL = 0x1p-48 * magL;
          const float a0 = elevation_angle[at];
          CellFold s;
          fold_init(&s, a0);
          for (int f : cand) {
            const FrameFast& Q = fast[f];
            // (the kernel steps ly through a slab as ly0 + c * dly; the error
            // budget carries that term, the emulation takes the grid's value)
            const double dx = lx - Q.p[0], dy = ly - Q.p[1], dz = lz - Q.p[2];
            const double cx = fma(Q.m[2], dz, fma(Q.m[1], dy, Q.m[0] * dx));
            const double cy = fma(Q.m[5], dz, fma(Q.m[4], dy, Q.m[3] * dx));
            const double cz = fma(Q.m[8], dz, fma(Q.m[7], dy, Q.m[6] * dx));
            stats[0]++;
            fold_pair(&s, f, k, true, cx, cy, cz, epsL);
          }
          int ku = 0, kv = 0;
          float angle = 0.0f;
          const int what = fold_finish(&s, k, atan_tab, epsL, (int)camera[4], (int)camera[5], &ku,
                                       &kv, &angle);
          FoldResult r;
          r.best_f = -1;
          if (what == kFoldDone) {
            r.best = angle;
            r.best_f = s.best_f;
            r.accepted = s.accepted;
            r.kp_x = ku;
            r.kp_y = kv;
            r.bad_alpha = 0;
          } else if (what == kFoldFinish) {
            stats[8]++;
            r = exact_finish(camera, poses[s.best_f], lx, ly, lz, s.best_f, s.accepted);
          } else if (what == kFoldRedo) {
            stats[1]++;
            r = exact_refold(camera, poses.data(), cand.data(), (int)cand.size(), lx, ly, lz, a0);
          }
          if (r.best_f >= 0) {
            if (r.bad_alpha) bad = true;
            elevation_angle[at] = r.best;
            observation_index[at] = (float)r.best_f;
            kp_x[at] = r.kp_x;
            kp_y[at] = r.kp_y;
            accepted[at] = r.accepted;
          } else if (what == kFoldRedo && r.bad_alpha) {
            bad = true;
          }
        }
      }
    }
  }
  return bad ? 3 : 0;
}

}  // extern "C"
