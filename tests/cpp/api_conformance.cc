// One source, compiled (syntax only) twice by tests/test_api_conformance.py: against the
// REFERENCE's public headers (from /root/reference, externals from oracle/refkit/) and against
// this repository's drop-in headers (include/).  Every static_assert below is a statement about
// the class API the reference's demos call: if both compilations pass, the
// public surface the drop-in offers is the reference's -- same namespaces, names, Settings
// fields and their types, constructor and member signatures, constness.
//   -DAPI_DSM / -DAPI_BACKWARD / -DAPI_FROM_PCL / -DAPI_FORWARD pick the header (the three ortho
//   headers each define ortho::Settings, as in the reference).
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#if defined(API_DSM)
#include <aerial-mapper-dsm/dsm.h>
#elif defined(API_BACKWARD)
#include <aerial-mapper-ortho/ortho-backward-grid.h>
#elif defined(API_FROM_PCL)
#include <aerial-mapper-ortho/ortho-from-pcl.h>
#elif defined(API_FORWARD)
#include <aerial-mapper-ortho/ortho-forward-homography.h>
#elif defined(API_IO)
#include <aerial-mapper-io/aerial-mapper-io.h>
#endif

#define FIELD(S, name, T) \
  static_assert(std::is_same<decltype(S::name), T>::value, #S "::" #name " is not " #T)

typedef AlignedType<std::vector, Eigen::Vector3d>::type Cloud;

#if defined(API_DSM)
FIELD(dsm::Settings, interpolation_radius, int);
FIELD(dsm::Settings, adaptive_interpolation, bool);
FIELD(dsm::Settings, center_easting, double);
FIELD(dsm::Settings, center_northing, double);
FIELD(dsm::Settings, use_multi_threads, bool);
static_assert(std::is_constructible<dsm::Dsm, const dsm::Settings&, grid_map::GridMap*>::value,
              "Dsm(const Settings&, GridMap*)");
static_assert(std::is_same<decltype(&dsm::Dsm::process),
                           void (dsm::Dsm::*)(const Cloud&, grid_map::GridMap*)>::value,
              "void Dsm::process(const cloud&, GridMap*)");
// main-dsm.cc:103-107
inline void caller(grid_map::GridMap* map, const Cloud& point_cloud) {
  dsm::Settings settings_dsm;
  settings_dsm.center_easting = 0.0;
  settings_dsm.center_northing = 0.0;
  dsm::Dsm digital_surface_map(settings_dsm, map);
  digital_surface_map.process(point_cloud, map);
}
#endif

#if defined(API_BACKWARD)
FIELD(ortho::Settings, show_orthomosaic_opencv, bool);
FIELD(ortho::Settings, save_orthomosaic_jpg, bool);
FIELD(ortho::Settings, orthomosaic_jpg_filename, std::string);
FIELD(ortho::Settings, orthomosaic_elevation_m, double);
FIELD(ortho::Settings, use_digital_elevation_map, bool);
FIELD(ortho::Settings, colored_ortho, bool);
FIELD(ortho::Settings, use_multi_threads, bool);
static_assert(std::is_constructible<ortho::OrthoBackwardGrid, const std::shared_ptr<aslam::NCamera>,
                                    const ortho::Settings&, grid_map::GridMap*>::value,
              "OrthoBackwardGrid(shared_ptr<NCamera>, const Settings&, GridMap*)");
static_assert(std::is_constructible<ortho::OrthoBackwardGrid, const std::shared_ptr<aslam::NCamera>,
                                    const ortho::Settings&>::value,
              "the map argument has a default");
static_assert(std::is_same<decltype(&ortho::OrthoBackwardGrid::process),
                           void (ortho::OrthoBackwardGrid::*)(const Poses&, const Images&,
                                                              grid_map::GridMap*) const>::value,
              "void OrthoBackwardGrid::process(const Poses&, const Images&, GridMap*) const");
static_assert(std::is_same<Poses, std::vector<Pose> >::value &&
                  std::is_same<Pose, kindr::minimal::QuatTransformation>::value &&
                  std::is_same<Images, std::vector<Image> >::value &&
                  std::is_same<Image, cv::Mat>::value,
              "the typedefs of aerial-mapper-io.h:17-20");
// main-ortho-backward-grid.cc:134-141
inline void caller(std::shared_ptr<aslam::NCamera> ncameras, const Poses& T_G_Bs, const Images& images,
                   grid_map::GridMap* map) {
  ortho::Settings settings_ortho;
  settings_ortho.colored_ortho = false;
  ortho::OrthoBackwardGrid mosaic(ncameras, settings_ortho, map);
  mosaic.process(T_G_Bs, images, map);
}
#endif

#if defined(API_FROM_PCL)
FIELD(ortho::Settings, show_orthomosaic_opencv, bool);
FIELD(ortho::Settings, interpolation_radius, int);
FIELD(ortho::Settings, use_adaptive_interpolation, bool);
FIELD(ortho::Settings, save_orthomosaic_jpg, bool);
FIELD(ortho::Settings, orthomosaic_jpg_filename, std::string);
static_assert(std::is_constructible<ortho::OrthoFromPcl, const ortho::Settings&>::value,
              "OrthoFromPcl(const Settings&)");
static_assert(std::is_same<decltype(&ortho::OrthoFromPcl::process),
                           void (ortho::OrthoFromPcl::*)(const Cloud&, const std::vector<int>&,
                                                         grid_map::GridMap*) const>::value,
              "void OrthoFromPcl::process(const cloud&, const vector<int>&, GridMap*) const");
#endif

#if defined(API_FORWARD)
FIELD(ortho::Settings, batch, bool);
FIELD(ortho::Settings, ground_plane_elevation_m, double);
FIELD(ortho::Settings, width_mosaic_pixels, size_t);
FIELD(ortho::Settings, height_mosaic_pixels, size_t);
FIELD(ortho::Settings, origin, Eigen::Vector3d);
FIELD(ortho::Settings, nframe_id, std::string);
FIELD(ortho::Settings, filename_mosaic_output, std::string);
static_assert(std::is_constructible<ortho::OrthoForwardHomography,
                                    const std::shared_ptr<aslam::NCamera>&, const ortho::Settings&>::value,
              "OrthoForwardHomography(const shared_ptr<NCamera>&, const Settings&)");
static_assert(std::is_same<decltype(&ortho::OrthoForwardHomography::updateOrthomosaic),
                           void (ortho::OrthoForwardHomography::*)(const Pose&, const Image&)>::value,
              "void updateOrthomosaic(const Pose&, const Image&)");
static_assert(std::is_same<decltype(&ortho::OrthoForwardHomography::batch),
                           void (ortho::OrthoForwardHomography::*)(const Poses&, const Images&)>::value,
              "void batch(const Poses&, const Images&)");
#endif

int main() { return 0; }

#if defined(API_IO)
// io::AerialMapperIO: the loaders in front of the path and the writers behind it
// (aerial-mapper-io.h:32-64; callers main-dsm.cc:84-90, main-ortho-backward-grid.cc:86-101)
static_assert(std::is_default_constructible<io::AerialMapperIO>::value, "AerialMapperIO()");
static_assert(std::is_same<decltype(&io::AerialMapperIO::loadPosesFromFileStandard),
                           void (io::AerialMapperIO::*)(const std::string&, Poses*)>::value,
              "void loadPosesFromFileStandard(const std::string&, Poses*)");
static_assert(std::is_same<decltype(static_cast<void (io::AerialMapperIO::*)(
                               const std::string&, Cloud*, std::vector<int>*)>(
                               &io::AerialMapperIO::loadPointCloudFromFile)),
                           void (io::AerialMapperIO::*)(const std::string&, Cloud*,
                                                        std::vector<int>*)>::value,
              "void loadPointCloudFromFile(const std::string&, cloud*, std::vector<int>*)");
static_assert(std::is_same<decltype(static_cast<void (io::AerialMapperIO::*)(
                               const std::string&, Cloud*)>(
                               &io::AerialMapperIO::loadPointCloudFromFile)),
                           void (io::AerialMapperIO::*)(const std::string&, Cloud*)>::value,
              "void loadPointCloudFromFile(const std::string&, cloud*)");
static_assert(std::is_same<decltype(&io::AerialMapperIO::subtractOriginFromPoses),
                           void (io::AerialMapperIO::*)(const Eigen::Vector3d&, Poses*)>::value,
              "void subtractOriginFromPoses(const Eigen::Vector3d&, Poses*)");
static_assert(std::is_same<decltype(&io::AerialMapperIO::toGeoTiff),
                           void (io::AerialMapperIO::*)(const cv::Mat&, const Eigen::Vector2d&,
                                                        const std::string&)>::value,
              "void toGeoTiff(const cv::Mat&, const Eigen::Vector2d&, const std::string&)");
static_assert(std::is_same<decltype(&io::AerialMapperIO::writeDataToDEMGeoTiffColor),
                           void (io::AerialMapperIO::*)(const cv::Mat&, const Eigen::Vector2d&,
                                                        const std::string&)>::value,
              "void writeDataToDEMGeoTiffColor(const cv::Mat&, const Eigen::Vector2d&, "
              "const std::string&)");
static_assert(std::is_same<Image, cv::Mat>::value && std::is_same<Images, std::vector<cv::Mat> >::value,
              "typedef cv::Mat Image; typedef std::vector<Image> Images");
#endif
