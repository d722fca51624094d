/*
 * oracle/amo_rectify.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * stereo::Rectifier::rectifyStereoPair restated
 * (aerial_mapper_dense_pcl/src/rectifier.cpp:34-114, computeMask :116-128): the rectified
 * rotation and baseline, the two rectification maps (float, per pixel), the remapped pair and
 * the mask.  The adopted arithmetic of the Eigen / OpenCV calls is in amo_rectify.h; the flow
 * is pinned by oracle/ref_loops_rectify.cc (rectifier.cpp compiled unchanged over refkit).
 */
#include "amo_rectify.h"

#include <cstring>

#include "amo_types.h"

extern "C" {

/* K, R1, R2 row-major 3x3 (R_G_C of the left / right camera), t1, t2 their positions.  Images
 * 8UC1, rows `*_step` bytes apart.  Outputs (any may be NULL): R_G_C_out row-major 3x3,
 * baseline, maps = 4 planes of width*height floats (x1, y1, x2, y2), the two rectified images
 * and the mask, dense width*height bytes.  Returns AMO_OK, AMO_ERR_ARG, or AMO_ERR_EXACT_HIT when
 * CHECK_NE(xyw(2), 0.0) would have fired. */
int amo_rectify_stereo_pair(const double* K, const double* R1, const double* R2, const double* t1,
                            const double* t2, int width, int height, const uint8_t* left,
                            size_t left_step, const uint8_t* right, size_t right_step,
                            double* R_G_C_out, double* baseline_out, float* maps,
                            uint8_t* rect_left, uint8_t* rect_right, uint8_t* mask) {
  using namespace amo::rect;
  if (!K || !R1 || !R2 || !t1 || !t2 || width <= 0 || height <= 0) return AMO_ERR_ARG;
  M3 k, r1, r2;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      k.m[i][j] = K[3 * i + j];
      r1.m[i][j] = R1[3 * i + j];
      r2.m[i][j] = R2[3 * i + j];
    }
  const V3d a = {{t1[0], t1[1], t1[2]}}, b = {{t2[0], t2[1], t2[2]}};
  const Plan p = make_plan(k, r1, r2, a, b);
  if (baseline_out) *baseline_out = p.baseline;
  if (R_G_C_out)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R_G_C_out[3 * i + j] = p.R_G_C.m[i][j];
  const size_t n = static_cast<size_t>(width) * height;
  int rc = AMO_OK;
  for (int v = 0; v < height; ++v)
    for (int u = 0; u < width; ++u) {
      float x1, y1, x2, y2;
      if (!map_pixel(p.T1_inv, u, v, &x1, &y1)) rc = AMO_ERR_EXACT_HIT;
      if (!map_pixel(p.T2_inv, u, v, &x2, &y2)) rc = AMO_ERR_EXACT_HIT;
      const size_t o = static_cast<size_t>(v) * width + u;
      if (maps) {
        maps[o] = x1;
        maps[n + o] = y1;
        maps[2 * n + o] = x2;
        maps[3 * n + o] = y2;
      }
      if (rect_left && left) rect_left[o] = remap_bilinear(left, left_step, width, height, x1, y1);
      if (rect_right && right) rect_right[o] = remap_bilinear(right, right_step, width, height, x2, y2);
    }
  if (mask) {
    int cx[4], cy[4];
    mask_corners(p.T1, width, height, cx, cy);
    for (int v = 0; v < height; ++v)
      for (int u = 0; u < width; ++u)
        mask[static_cast<size_t>(v) * width + u] = in_quad(cx, cy, u, v) ? 255 : 0;
  }
  return rc;
}

}  // extern "C"
