/*
 * oracle/amo_types.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C structs shared by the oracle's C entry points.  Nothing under
 * aerial_mapper_amd/ (the product) may include, link or call anything under
 * oracle/; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do, and only as the checker.
 *
 * Layout mirrors include/aerial_mapper_hip.h (amhip_grid_desc / amhip_camera)
 * on purpose so a test can fill one ctypes.Structure and hand it to both.
 */
#ifndef AMO_TYPES_H_
#define AMO_TYPES_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Geometry of one grid_map::GridMap as aerial_mapper_grid_map creates it
 * (reference: aerial_mapper_grid_map/src/aerial-mapper-grid-map.cc:30-33).
 * rows <-> index(0) <-> x/easting, cols <-> index(1) <-> y/northing.
 * Layers are Eigen::MatrixXf: float32, column-major, (i,j) at i + j*rows. */
typedef struct amo_grid {
  int32_t rows;
  int32_t cols;
  double resolution;
  double length_x; /* = rows * resolution (grid_map_core setGeometry) */
  double length_y; /* = cols * resolution */
  double pos_x;    /* map centre (center_easting)  */
  double pos_y;    /* map centre (center_northing) */
} amo_grid;

enum { AMO_DIST_NONE = 0, AMO_DIST_RADTAN = 1, AMO_DIST_EQUIDISTANT = 2 };

/* aslam::PinholeCamera intrinsics + distortion (external dependency,
 * un-vendored; formulas adopted in SURVEY.md section 8c). */
typedef struct amo_camera {
  double fu, fv, cu, cv;
  int32_t width;
  int32_t height;
  int32_t distortion; /* AMO_DIST_* */
  int32_t _pad;
  double dist[4]; /* radtan: k1,k2,p1,p2; equidistant: k1..k4 */
} amo_camera;

/* Return codes (the reference aborts through glog CHECK in these cases). */
enum {
  AMO_OK = 0,
  AMO_ERR_ARG = 1,
  AMO_ERR_EXACT_HIT = 2,    /* dsm.cc:165  CHECK(distances[i] > 0.0)  */
  AMO_ERR_ALPHA_NONPOS = 3  /* ortho-backward-grid.cc:178 CHECK(alpha > 0.0) */
};

#ifdef __cplusplus
}
#endif
#endif /* AMO_TYPES_H_ */
