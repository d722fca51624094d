// aerial-mapper-deps.h -- one place that decides where the external types in
// the hot path's signatures come from: the real grid_map_core / Eigen /
// minkindr / aslam_cv2 / OpenCV headers when a catkin workspace provides them,
// the minimal stand-ins under include/aerial-mapper-compat/ otherwise (this
// image has none of them).
#ifndef AERIAL_MAPPER_DEPS_H_
#define AERIAL_MAPPER_DEPS_H_

#if defined(__has_include) && __has_include(<grid_map_core/GridMap.hpp>) && \
    __has_include(<aslam/cameras/ncamera.h>) && __has_include(<opencv2/core/core.hpp>)
#define AERIAL_MAPPER_REAL_DEPS 1
#include <Eigen/Dense>
#include <aslam/cameras/camera-pinhole.h>
#include <aslam/cameras/camera.h>
#include <aslam/cameras/distortion.h>
#include <aslam/cameras/ncamera.h>
#include <grid_map_core/GridMap.hpp>
#include <opencv2/core/core.hpp>
#else
#define AERIAL_MAPPER_REAL_DEPS 0
#include "aerial-mapper-compat/eigen_lite.h"
#include "aerial-mapper-compat/grid_map_lite.h"
#include "aerial-mapper-compat/kindr_aslam_cv_lite.h"
#endif

#endif  // AERIAL_MAPPER_DEPS_H_
