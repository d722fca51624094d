/*
 * oracle/ref_loops_forward.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C entry points over the reference's OWN ortho::OrthoForwardHomography
 * (aerial_mapper_ortho/src/ortho-forward-homography.cc, compiled unchanged from
 * /root/reference; see refkit/refkit.h).  What that pins is the reference's own part of the
 * path: the four border rays and their ground points (incl. the width-for-height offset of
 * batch()), which operations run on what in which order (undistort, homography, warp,
 * GRAY2RGB / 16SC3 / mask, feed, blend, the re-fed running mosaic of the incremental mode,
 * the final unobserved-pixel pass of batch()).  The operations themselves are OpenCV's /
 * aslam's and forward to the oracle's restatements (amo_cvlike.h) -- NOT pinned.
 * The class keeps its mosaic private; what it hands to cv::imwrite is what is returned here.
 */
#include <aerial-mapper-ortho/ortho-forward-homography.h>

#include "amo_types.h"
#include "refkit/refkit.h"

namespace {

struct Handle {
  amo_camera cam;
  int width, height;
  std::shared_ptr<aslam::NCamera> ncameras;
  std::unique_ptr<ortho::OrthoForwardHomography> mosaic;
};

int copy_result(const Handle& h, int16_t* result) {
  const cv::Mat& m = cv::last_written();
  if (m.rows != h.height || m.cols != h.width || m.type() != CV_16SC3) return AMO_ERR_ARG;
  for (int r = 0; r < m.rows; ++r)
    std::memcpy(result + 3 * static_cast<size_t>(r) * m.cols, m.ptr<int16_t>(r), 6 * static_cast<size_t>(m.cols));
  return refkit::check_state().failed ? AMO_ERR_ARG : AMO_OK;
}

cv::Mat view(const Handle& h, const void* image, size_t step, int channels) {
  return cv::Mat(h.cam.height, h.cam.width, static_cast<const uint8_t*>(image), step,
                 channels == 3 ? CV_8UC3 : CV_8UC1);
}

}  // namespace

extern "C" {

typedef struct amr_mosaic_desc {  // = amo_mosaic_desc
  int32_t width_mosaic_pixels;
  int32_t height_mosaic_pixels;
  double ground_plane_elevation_m;
  double origin[3];
} amr_mosaic_desc;

void* amr_fwd_create(const amo_camera* cam, const amr_mosaic_desc* desc, const double* T_C_B) {
  if (!cam || !desc || !T_C_B) return nullptr;
  refkit::check_reset();
  Handle* h = new Handle();
  h->cam = *cam;
  h->width = desc->width_mosaic_pixels;
  h->height = desc->height_mosaic_pixels;
  h->ncameras.reset(new aslam::NCamera(*cam, aslam::Transformation(amo::pose_from7(T_C_B))));
  ortho::Settings settings;
  settings.ground_plane_elevation_m = desc->ground_plane_elevation_m;
  settings.width_mosaic_pixels = static_cast<size_t>(desc->width_mosaic_pixels);
  settings.height_mosaic_pixels = static_cast<size_t>(desc->height_mosaic_pixels);
  settings.origin = Eigen::Vector3d(desc->origin[0], desc->origin[1], desc->origin[2]);
  h->mosaic.reset(new ortho::OrthoForwardHomography(h->ncameras, settings));
  return h;
}

void amr_fwd_destroy(void* handle) { delete static_cast<Handle*>(handle); }

/* OrthoForwardHomography::batch (:137-189).  result: height x width x 3 int16. */
int amr_fwd_batch(void* handle, const double* T_G_B, const void* const* images, const size_t* steps,
                  int channels, size_t F, int16_t* result) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !T_G_B || !images || !steps || !result || (channels != 1 && channels != 3))
    return AMO_ERR_ARG;
  refkit::check_reset();
  Poses T_G_Bs;
  Images frames;
  for (size_t f = 0; f < F; ++f) {
    T_G_Bs.push_back(Pose(amo::pose_from7(T_G_B + 7 * f)));
    frames.push_back(view(*h, images[f], steps[f], channels));
  }
  h->mosaic->batch(T_G_Bs, frames);
  return copy_result(*h, result);
}

/* OrthoForwardHomography::updateOrthomosaic (:74-135). */
int amr_fwd_update(void* handle, const double* T_G_B7, const void* image, size_t step, int channels,
                   int16_t* result) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !T_G_B7 || !image || !result || (channels != 1 && channels != 3)) return AMO_ERR_ARG;
  refkit::check_reset();
  h->mosaic->updateOrthomosaic(Pose(amo::pose_from7(T_G_B7)), view(*h, image, step, channels));
  return copy_result(*h, result);
}

}  // extern "C"
