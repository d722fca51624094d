/*
 * oracle/ref_loops_ortho_backward.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * C entry point over the reference's OWN ortho::OrthoBackwardGrid
 * (aerial_mapper_ortho/src/ortho-backward-grid.cc, compiled unchanged from /root/reference;
 * see refkit/refkit.h).  Same arguments as the restated oracle's
 * amo_ortho_backward_process (amo_ortho.cc).
 */
#include <aerial-mapper-ortho/ortho-backward-grid.h>

#include "ref_loops_common.h"

extern "C" {

int amr_ortho_backward_process(const amo_grid* grid, const amo_camera* cam, const double* T_G_B,
                               const double* T_C_B, const uint8_t* const* images,
                               const size_t* steps, int channels, size_t F, int colored,
                               int multi_thread, const float* elevation, float* elevation_angle,
                               float* observation_index, float* num_observations, float* ortho,
                               float* colored_ortho, double* timing) {
  if (!grid || !cam || !T_G_B || !T_C_B || !images || !steps || !elevation) return AMO_ERR_ARG;
  if ((colored && channels != 3) || (!colored && channels != 1)) return AMO_ERR_ARG;
  refkit::check_reset();
  grid_map::GridMap map({"ortho", "elevation", "elevation_angle", "num_observations",
                         "observation_index", "colored_ortho"});
  ref_loops::set_geometry(*grid, &map);
  if (!ref_loops::same_geometry(map.geometry(), *grid)) return AMO_ERR_ARG;
  ref_loops::layer_in(elevation, &map["elevation"]);
  ref_loops::layer_in(elevation_angle, &map["elevation_angle"]);
  ref_loops::layer_in(observation_index, &map["observation_index"]);
  ref_loops::layer_in(num_observations, &map["num_observations"]);
  ref_loops::layer_in(ortho, &map["ortho"]);
  ref_loops::layer_in(colored_ortho, &map["colored_ortho"]);

  std::shared_ptr<aslam::NCamera> ncameras(
      new aslam::NCamera(*cam, aslam::Transformation(amo::pose_from7(T_C_B))));
  Poses T_G_Bs;
  Images frames;
  for (size_t f = 0; f < F; ++f) {
    T_G_Bs.push_back(Pose(amo::pose_from7(T_G_B + 7 * f)));
    frames.push_back(cv::Mat(cam->height, cam->width, images[f], steps[f]));
  }
  ortho::Settings settings;
  settings.colored_ortho = colored != 0;
  settings.use_multi_threads = multi_thread != 0;
  const double t0 = ref_loops::now_s();
  ortho::OrthoBackwardGrid mosaic(ncameras, settings, &map);
  const double t1 = ref_loops::now_s();
  mosaic.process(T_G_Bs, frames, &map);
  if (timing) {  // [0] constructor (ortho-backward-grid.cc:22-40), [1] process()
    timing[0] = t1 - t0;
    timing[1] = ref_loops::now_s() - t1;
  }

  ref_loops::layer_out(map["elevation_angle"], elevation_angle);
  ref_loops::layer_out(map["observation_index"], observation_index);
  ref_loops::layer_out(map["num_observations"], num_observations);
  ref_loops::layer_out(map["ortho"], ortho);
  ref_loops::layer_out(map["colored_ortho"], colored_ortho);
  return ref_loops::check_result();
}

}  // extern "C"
