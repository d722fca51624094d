/*
 * aerial_mapper_hip.h -- C ABI of libaerial_mapper_hip.so
 *
 * MI355X (gfx950) implementation of aerial_mapper's one data-parallel hot
 * path: point cloud -> DSM rasterisation and the grid-based backward-projection
 * orthomosaic.  The reference (ethz-asl/aerial_mapper) has no FFI layer; its
 * boundary for this path is the C++ class API
 *
 *   dsm::Dsm::Dsm / Dsm::process
 *       aerial_mapper_dsm/include/aerial-mapper-dsm/dsm.h:25-42
 *       aerial_mapper_dsm/src/dsm.cc:20-34,186-201
 *   ortho::OrthoBackwardGrid::OrthoBackwardGrid / ::process
 *       aerial_mapper_ortho/include/aerial-mapper-ortho/ortho-backward-grid.h:32-50
 *       aerial_mapper_ortho/src/ortho-backward-grid.cc:23-40,223-239
 *   grid_map::AerialGridMap::initialize (layer set, geometry, init constants)
 *       aerial_mapper_grid_map/src/aerial-mapper-grid-map.cc:23-49
 *
 * which the drop-in C++ classes under include/aerial-mapper-dsm/ and
 * include/aerial-mapper-ortho/ implement on top of the entry points below
 * (see INTEGRATION.md).  Everything here is extern "C", POD only: plain
 * pointers, sizes and two small structs; no C++ / torch / HIP types.
 *
 * Conventions
 *   - A layer is one grid_map::Matrix (Eigen::MatrixXf): float32, column-major,
 *     rows*cols elements, element (i,j) at i + j*rows.  rows <-> index(0) <->
 *     x/easting, cols <-> index(1) <-> y/northing.
 *   - Points are AoS x,y,z doubles, 24 B apart: the memory layout of
 *     AlignedType<std::vector, Eigen::Vector3d>::type (dsm.h:41).
 *   - Poses are 7 doubles tx,ty,tz,qw,qx,qy,qz (Hamilton unit quaternion), the
 *     components of kindr::minimal::QuatTransformation.
 *   - Images are 8UC1 (gray) or 8UC3 (OpenCV BGR) rasters with a row step in
 *     bytes (cv::Mat::data / cv::Mat::step).
 *   - "host" pointers are ordinary process memory; "dev" pointers are HIP
 *     device allocations on the context's GPU (e.g. torch tensor data_ptr()).
 *   - Every function returns an amhip_status (0 = ok) unless noted.  The
 *     reference aborts through glog CHECK on precondition failures; the C++
 *     shim turns a non-zero status back into that behaviour.
 *   - One context serialises its own calls (the caller must not use one
 *     context from two threads at once); different contexts are independent.
 */
#ifndef AERIAL_MAPPER_HIP_H_
#define AERIAL_MAPPER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMHIP_ABI_VERSION 1

typedef enum amhip_status {
  AMHIP_OK = 0,
  AMHIP_ERR_ARG = 1,          /* null / shape / mode error (reference: CHECK) */
  AMHIP_ERR_EXACT_HIT = 2,    /* a point coincides with a cell centre:
                                 dsm.cc:165 CHECK(distances[i] > 0.0)        */
  AMHIP_ERR_ALPHA_NONPOS = 3, /* ortho-backward-grid.cc:178 CHECK(alpha > 0) */
  AMHIP_ERR_HIP = 4,          /* HIP runtime failure, see amhip_last_error() */
  AMHIP_ERR_NO_DEVICE = 5,    /* no usable gfx950 device                     */
  AMHIP_ERR_NOMEM = 6,
  AMHIP_ERR_HALO_OVERFLOW = 7 /* tiled DSM: a window had more halo points for a neighbour than
                                 the send rows reserved: the step's result is incomplete   */
} amhip_status;

/* Geometry of the grid_map::GridMap the layers belong to
 * (grid_map_core GridMap::setGeometry as called by
 * aerial-mapper-grid-map.cc:30-33; startIndex is always 0 there). */
typedef struct amhip_grid_desc {
  int32_t rows;      /* getSize()(0) */
  int32_t cols;      /* getSize()(1) */
  double resolution; /* getResolution() */
  double length_x;   /* getLength().x() = rows * resolution */
  double length_y;   /* getLength().y() = cols * resolution */
  double pos_x;      /* getPosition().x() (center_easting)  */
  double pos_y;      /* getPosition().y() (center_northing) */
} amhip_grid_desc;

enum {
  AMHIP_DIST_NONE = 0,
  AMHIP_DIST_RADTAN = 1,      /* k1,k2,p1,p2 */
  AMHIP_DIST_EQUIDISTANT = 2  /* k1,k2,k3,k4 */
};

/* aslam::PinholeCamera of ncameras->getCamera(0)
 * (ortho-backward-grid.cc:46,131): intrinsics, image size, distortion. */
typedef struct amhip_camera {
  double fu, fv, cu, cv;
  int32_t width;
  int32_t height;
  int32_t distortion; /* AMHIP_DIST_* */
  int32_t _pad;
  double dist[4];
} amhip_camera;

/* The layers of AerialGridMap the hot path reads or writes
 * (aerial-mapper-grid-map.cc:25-28). */
typedef enum amhip_layer {
  AMHIP_LAYER_ORTHO = 0,
  AMHIP_LAYER_ELEVATION = 1,
  AMHIP_LAYER_ELEVATION_ANGLE = 2,
  AMHIP_LAYER_NUM_OBSERVATIONS = 3,
  AMHIP_LAYER_OBSERVATION_INDEX = 4,
  AMHIP_LAYER_COLORED_ORTHO = 5,
  AMHIP_NUM_LAYERS = 6
} amhip_layer;

typedef struct amhip_ctx amhip_ctx; /* opaque: GPU, stream, layers, workspaces */

/* ---- library / context ---------------------------------------------------*/

int amhip_abi_version(void);

/* Message of the last failing call on this thread ("" if none). */
const char* amhip_last_error(void);

/* grid_map_core GridMap::setGeometry: size = (int)round(length/resolution),
 * length := size*resolution.  Pure host helper. */
void amhip_make_grid(double length_x, double length_y, double resolution,
                     double pos_x, double pos_y, amhip_grid_desc* out);

/* grid_map_core GridMap::getPosition(index): centre of cell (i,j). */
void amhip_cell_position(const amhip_grid_desc* grid, int i, int j, double* x,
                         double* y);

/* Replaces the per-cell bookkeeping of dsm.cc:24-33 and
 * ortho-backward-grid.cc:30-39 (the GPU needs only the geometry).  Allocates
 * the six layers on GPU `device` and sets them to AerialGridMap::initialize()'s
 * constants (ortho 255, elevation NaN, elevation_angle 0, num_observations 0,
 * observation_index NaN, colored_ortho NaN). */
int amhip_ctx_create(const amhip_grid_desc* grid, int device, amhip_ctx** out);
void amhip_ctx_destroy(amhip_ctx* ctx);

/* The same for the window [i0, i0+rows) x [j0, j0+cols) of a larger map
 * `grid` (one tile of a survey spread over several GPUs): cell positions, and
 * therefore every result, are exactly those of the full map; the context's
 * layers hold only the window (rows*cols floats, column-major).  The caller
 * hands the DSM every point within sqrt(7) m of the window (the last fallback
 * radius, dsm.cc:133-144) -- see aerial_mapper_amd/tiling.py. */
int amhip_ctx_create_window(const amhip_grid_desc* grid, int i0, int j0, int rows,
                            int cols, int device, amhip_ctx** out);

/* Run the context's kernels on an existing HIP stream (hipStream_t passed as
 * void*; NULL = back to the context's own stream). */
int amhip_ctx_set_stream(amhip_ctx* ctx, void* hip_stream);

/* Arithmetic of the DSM gather (dsm::Dsm::process, dsm.cc:160-172).
 *   AMHIP_DSM_EXACT (DEFAULT since round 3) the FP64 gather everywhere: the reference's doubles
 *                   decide and weigh every pair, only the ORDER of the double sums differs from
 *                   the kd-tree's (1e-16 relative, then rounded to float): bit-identical floats
 *                   in every test so far (one cell in 1e8 follows the kd-tree's own summation
 *                   order), THE SAME BITS IN EVERY RUN AND FOR EVERY ORDER OF THE CLOUD (since
 *                   round 4: the rounding of every quotient is guarded, the cells on a float
 *                   rounding boundary are redone in order-independent double-double sums), and
 *                   therefore a byte-identical mosaic on top.  What a drop-in must be by default.
 *   AMHIP_DSM_FAST  (opt-in) single-precision distances and weights under exact guards: the
 *                   reference's neighbour sets (any decision within 2e-6 of the radius is taken
 *                   in its own doubles), identical NaN pattern, heights within the contract's
 *                   1e-4 m (1 float spacing above 1024 m) by a per-tile error bound; tiles /
 *                   cells without room under the bound take the FP64 path.  ~0.6 ms faster per
 *                   100 M cells; heights may differ from the reference's by one float spacing
 *                   and from run to run (the f32 sums follow the order the atomics of the
 *                   binning left the points in), so a mosaic on top of them can differ in the
 *                   rare cell whose keypoint sits on a pixel boundary.  A context whose
 *                   scene is mostly rough (more than half of the tiles without room under the
 *                   bound) runs the FP64 pipeline by itself for 16 calls at a time.
 * New contexts start in EXACT; the environment variable AMHIP_DSM_FAST=1 makes FAST their
 * default (hosts that cannot be recompiled), AMHIP_DSM_EXACT=1 (round 2's switch) still forces
 * EXACT.  The setters override either.  amhip_session_set_dsm_precision applies to every window
 * of a session (the drop-in dsm::Dsm::setPrecision calls it). */
#define AMHIP_DSM_FAST 0
#define AMHIP_DSM_EXACT 1
int amhip_ctx_set_dsm_precision(amhip_ctx* ctx, int mode);

/* OPTIONAL capped interpolation ("IDW k = 4" of BASELINE.json's wording): only the k nearest
 * points of a cell's radius search (nanoflann::KNNResultSet's order, nanoflann.hpp:80-131)
 * take part in the weighting.  k = 0 (default) = the reference's behaviour: every point of
 * the search.  NOT a reference code path -- dsm.cc:127-172 has no k -- so its parity is
 * unpinned by the reference (oracle: KNNResultSet on the reference's vendored tree); it is a
 * separately reported extra, never the graded mode.  1 <= k <= 8; runs one lane per cell on
 * the global bins (FP64, true divisions in ascending-distance order: bit-identical to the
 * oracle but for exact distance ties). */
int amhip_ctx_set_dsm_knn(amhip_ctx* ctx, int k);

/* Wait for everything enqueued on the context and return the sticky status
 * of the device-side CHECKs (EXACT_HIT / ALPHA_NONPOS) or HIP errors since
 * the last synchronize; the status is then cleared. */
int amhip_ctx_synchronize(amhip_ctx* ctx);

/* ---- layers (device resident; persist across process() calls like the
 *      GridMap's matrices do in incremental mode,
 *      main-ortho-backward-grid-incremental.cc:153-162) --------------------*/

/* AerialGridMap::initialize() values (aerial-mapper-grid-map.cc:40-48).  The
 * reset is lazy: it writes nothing; the next call that produces a layer
 * (DSM -> elevation, OrthoFromPcl -> ortho, OrthoBackwardGrid ->
 * elevation_angle / observation_index / ortho|colored_ortho) also writes the
 * initial value into the cells it leaves alone, and anything else that needs
 * the memory (download, device pointer, a batch on top of uploaded layers)
 * fills the layer first.  Observable contents are always those of an eager
 * fill; the tuning knob eager_reset (amhip_set_tuning) restores the plain fills. */
int amhip_layers_reset(amhip_ctx* ctx);
int amhip_layer_upload(amhip_ctx* ctx, int layer, const float* host);
int amhip_layer_download(amhip_ctx* ctx, int layer, float* host);
/* Device address of a layer (rows*cols floats) or NULL.  The layer is
 * filled if its reset was still pending and is refilled eagerly from then on
 * (the library cannot see writes through the pointer). */
void* amhip_layer_device_ptr(amhip_ctx* ctx, int layer);

/* ---- DSM: dsm::Dsm::process (dsm.cc:186-201) ----------------------------*/

/* Asynchronous, device-resident form.  dev_xyz: n points on the context's
 * GPU.  Updates the context's ELEVATION layer exactly where the reference
 * would (cells with no point inside the last fallback radius stay untouched).
 * n == 0 is the reference's soft no-op.  radius_sq is
 * dsm::Settings::interpolation_radius (an int holding the SQUARED search
 * radius in m^2); center_* are dsm::Settings::center_easting/northing
 * (note dsm.cc:42-43 subtracts center_northing from x and center_easting
 * from y -- reproduced).
 * dev_xyz is read until the LAST kernel of the call (in AMHIP_DSM_FAST mode the sort carries
 * 20-byte records and the exact routines fetch a point's doubles from dev_xyz itself): work
 * that overwrites or frees it has to be ordered after the call on the context's stream. */
int amhip_dsm_process_dev(amhip_ctx* ctx, const double* dev_xyz, size_t n,
                          int radius_sq, double center_easting,
                          double center_northing);

/* Synchronous drop-in form on host buffers: uploads `elevation` (the map's
 * current layer), copies the cloud to the GPU, runs the DSM, downloads the
 * layer back into `elevation`, returns the final status. */
int amhip_dsm_process(amhip_ctx* ctx, const double* host_xyz, size_t n,
                      int radius_sq, double center_easting,
                      double center_northing, float* elevation);

/* ---- OrthoFromPcl: ortho::OrthoFromPcl::process
 *      (aerial_mapper_ortho/src/ortho-from-pcl.cc:20-113) --
 * The DSM's radius search + inverse-squared-distance weighting applied to the
 * points' INTENSITIES (one int per point), written to the ORTHO layer.
 * Differences to the DSM that are reproduced: no centre offsets (:30-31); an
 * exact hit takes that point's value instead of failing (:91-96); no fallback
 * unless `adaptive` (ortho::Settings::use_adaptive_interpolation), which then
 * retries with the squared radius x10, x100, ... (int lambda, :63-71) until
 * every cell has a value (one extra pass over the cloud per retry; stops where
 * the reference's int product would overflow).  radius_sq is
 * ortho::Settings::interpolation_radius (squared, like the DSM's).  The _dev
 * form is asynchronous unless `adaptive` is set. */
int amhip_ortho_from_pcl_process_dev(amhip_ctx* ctx, const double* dev_xyz,
                                     const int32_t* dev_intensities, size_t n,
                                     int radius_sq, int adaptive);
int amhip_ortho_from_pcl_process(amhip_ctx* ctx, const double* host_xyz,
                                 const int32_t* host_intensities, size_t n,
                                 int radius_sq, int adaptive, float* ortho);

/* ---- stereo::Densifier::computePointCloud, reprojection part
 *      (aerial_mapper_dense_pcl/src/densifier.cpp:25-108) --
 * Disparity map (float32 rows, disp_step bytes apart) + left rectified image
 * (8UC1, img_step) -> world points in RASTER order (pixels with disparity <= 1
 * or infinite z dropped) + their gray values, written to device buffers in
 * the layout amhip_dsm_process_dev / amhip_ortho_from_pcl_process_dev take, so
 * the cloud of the incremental pipeline never leaves HBM.  K and R_G_C are
 * row-major 3x3 (StereoRigParameters::K, RectifiedStereoPair::R_G_C), t_G_C1
 * is StereoRigParameters::t_G_C1, baseline RectifiedStereoPair::baseline.
 * dev_count receives the number of valid points (it may exceed `capacity`, the
 * excess is not written).  Asynchronous.  The block matcher (OpenCV BM/SGBM)
 * and the ROS PointCloud2 fill stay the reference's. */
int amhip_densify_dev(amhip_ctx* ctx, const float* dev_disparity, size_t disp_step,
                      const uint8_t* dev_image_left, size_t img_step, int width,
                      int height, const double* K, double baseline,
                      const double* R_G_C, const double* t_G_C1,
                      double* dev_xyz_out, int32_t* dev_intensity_out,
                      size_t capacity, int64_t* dev_count);

/* ---- multi-GPU: halo points of a tiled survey ------------------------------
 * No reference counterpart (the reference is single-process).  When one map
 * is tiled over several GPUs with amhip_ctx_create_window(), a rank's DSM
 * needs every point within the last fallback radius of its window.  Given the
 * points a rank holds, this collects (compacts) the ones that lie inside each
 * of `nd` OTHER windows expanded by `margin_m` metres, so that the host can
 * ship them with one RCCL all_to_all (aerial_mapper_amd/tiling.py).
 *   dest_windows  host, nd x 4 int32: i0, j0, rows, cols (nd <= 8)
 *   dev_out       device, nd * cap_per_dest * 3 doubles; points of destination
 *                 d start at dev_out + d*cap_per_dest*3 (original x,y,z)
 *   dev_counts    device, nd int64: number of points selected per destination
 *                 (may exceed cap_per_dest: the excess was not written)
 * Asynchronous on the context's stream. */
int amhip_halo_select_dev(amhip_ctx* ctx, const double* dev_xyz, size_t n,
                          double center_easting, double center_northing,
                          const int32_t* dest_windows, int nd, double margin_m,
                          double* dev_out, size_t cap_per_dest,
                          int64_t* dev_counts);

/* The same selection folded into dsm::Dsm::process (dsm.cc:186-201) of a window, around the
 * caller's exchange -- the DSM's binning pass reads every point anyway:
 *   1. amhip_dsm_tiled_begin_dev: bins the rank's own points dev_xyz[0, n_owned) and, in the
 *      same pass, copies the ones the `nd` windows in dest_windows need into dev_out /
 *      dev_counts exactly like amhip_halo_select_dev (a placeholder window far outside the
 *      map for the rank itself keeps the rows indexed by rank).
 *   2. the caller exchanges the rows (one all_to_all of equal splits, no count exchange, no
 *      host synchronisation) into dev_xyz[n_owned, n_total); rows it does not fill must be
 *      NaN -- the binning drops them like any point outside the map.
 *   3. amhip_dsm_tiled_finish_dev: bins dev_xyz[n_owned, n_total) and runs the rest of
 *      Dsm::process on all n_total rows; the ELEVATION layer is as after
 *      amhip_dsm_process_dev(dev_xyz, n_total).
 * Both asynchronous on the context's stream; dev_xyz must stay valid and rows [0, n_owned)
 * unchanged in between (and all n_total rows until the finish call's last kernel: see
 * amhip_dsm_process_dev).  Any other DSM / OrthoFromPcl call on the context in between cancels
 * the pending call: amhip_dsm_tiled_finish_dev then returns AMHIP_ERR_ARG. */
int amhip_dsm_tiled_begin_dev(amhip_ctx* ctx, const double* dev_xyz, size_t n_owned,
                              size_t n_total, int radius_sq, double center_easting,
                              double center_northing, const int32_t* dest_windows, int nd,
                              double margin_m, double* dev_out, size_t cap_per_dest,
                              int64_t* dev_counts);
int amhip_dsm_tiled_finish_dev(amhip_ctx* ctx);

/* ---- Ortho: ortho::OrthoBackwardGrid::process
 *      (ortho-backward-grid.cc:223-239) ------------------------------------*/

/* T_G_C[i] = T_G_B[i] * T_C_B^-1 (ortho-backward-grid.cc:230-233) with
 * minkindr's composition / inverse.  Pure host helper (F poses of 7). */
void amhip_compose_T_G_C(const double* T_G_B, const double* T_C_B, size_t F,
                         double* T_G_C);

/* Asynchronous, device-resident form.  host_T_G_C: F x 7 doubles on the host
 * (small; copied).  dev_frames: F rasters on the GPU, frame f starts at
 * dev_frames + f*frame_stride, rows are row_step bytes apart, `channels` is 1
 * (8UC1) or 3 (8UC3 BGR).  colored = ortho::Settings::colored_ortho (needs
 * channels == 3; gray needs channels == 1).  Reads the ELEVATION layer, folds
 * the F frames in ascending order into ELEVATION_ANGLE / OBSERVATION_INDEX /
 * NUM_OBSERVATIONS and ORTHO or COLORED_ORTHO of the context. */
int amhip_ortho_backward_process_dev(amhip_ctx* ctx, const amhip_camera* cam,
                                     const double* host_T_G_C, size_t F,
                                     const uint8_t* dev_frames,
                                     size_t frame_stride, size_t row_step,
                                     int channels, int colored);

/* Synchronous drop-in form on host buffers: uploads the six layers given
 * (any may be NULL = keep the context's copy), stages the images on the GPU,
 * runs the mosaic, downloads the five output layers that are non-NULL. */
int amhip_ortho_backward_process(
    amhip_ctx* ctx, const amhip_camera* cam, const double* host_T_G_C, size_t F,
    const uint8_t* const* images, const size_t* steps, int channels,
    int colored, const float* elevation, float* elevation_angle,
    float* observation_index, float* num_observations, float* ortho,
    float* colored_ortho);

/* ---- ortho::OrthoForwardHomography
 *      (aerial_mapper_ortho/src/ortho-forward-homography.cc; SURVEY section 8b/8f,
 *      named in the API surface to keep) --
 * A mosaic is the object's state: Settings::{width,height}_mosaic_pixels,
 * ground_plane_elevation_m, origin (ortho-forward-homography.h:33-42), the
 * camera, the FeatherBlender accumulators and result_ / result_mask_.
 * result_ is CV_16SC3 (height x width x 3 int16, row-major; a gray frame is
 * replicated into the three channels, :45-47), result_mask_ CV_8U (255 where
 * the summed feather weight exceeds 1e-5).  Poses are T_G_C (amhip_compose_T_G_C).
 *   amhip_mosaic_batch    = OrthoForwardHomography::batch (:137-189): every
 *       frame is warped (4 corner rays on the ground plane -> homography ->
 *       nearest-neighbour warp) and fed to the blender, then blended and the
 *       "unobserved pixels" set to 0.  batch() offsets both ground axes by
 *       width/2 (:155-158) -- reproduced.
 *   amhip_mosaic_update   = ::updateOrthomosaic (:74-135): feed the frame, blend,
 *       start a new blender holding the previous result.
 * The _dev forms take device-resident frames (frame f at dev_frames + f *
 * frame_stride, rows row_step bytes apart), run asynchronously on the mosaic's
 * stream and leave the result on the device (amhip_mosaic_device_ptr /
 * amhip_mosaic_download).  imshow / imwrite / ROS publishing stay with the
 * caller.  OpenCV / aslam semantics are restated, see oracle/amo_forward.cc
 * for the list ("parity unpinned": the reference has no tests for this path). */
typedef struct amhip_mosaic amhip_mosaic;

typedef struct amhip_mosaic_desc {
  int32_t width_mosaic_pixels;
  int32_t height_mosaic_pixels;
  double ground_plane_elevation_m;
  double origin[3];
} amhip_mosaic_desc;

int amhip_mosaic_create(const amhip_mosaic_desc* desc, const amhip_camera* cam, int device,
                        amhip_mosaic** out);
int amhip_mosaic_destroy(amhip_mosaic* mosaic);
int amhip_mosaic_set_stream(amhip_mosaic* mosaic, void* hip_stream);
int amhip_mosaic_synchronize(amhip_mosaic* mosaic);
/* fresh blender + zero result (a newly constructed object) */
int amhip_mosaic_reset(amhip_mosaic* mosaic);
int amhip_mosaic_batch(amhip_mosaic* mosaic, const double* T_G_C, size_t num_frames,
                       const void* const* images, const size_t* steps, int channels,
                       int16_t* result_16sc3, uint8_t* result_mask);
int amhip_mosaic_batch_dev(amhip_mosaic* mosaic, const double* T_G_C, size_t num_frames,
                           const void* dev_frames, size_t frame_stride, size_t row_step,
                           int channels);
int amhip_mosaic_update(amhip_mosaic* mosaic, const double* T_G_C7, const void* image,
                        size_t step, int channels, int16_t* result_16sc3,
                        uint8_t* result_mask);
int amhip_mosaic_update_dev(amhip_mosaic* mosaic, const double* T_G_C7, const void* dev_frame,
                            size_t row_step, int channels);
int amhip_mosaic_download(amhip_mosaic* mosaic, int16_t* result_16sc3, uint8_t* result_mask);
int amhip_mosaic_device_ptr(amhip_mosaic* mosaic, void** result_16sc3, void** result_mask);
/* The image -> mosaic homography of one frame (row-major 3x3, M[8] = 1); host
 * arithmetic only.  batch_quirk != 0: batch()'s offsets. */
int amhip_mosaic_homography(const amhip_mosaic_desc* desc, const amhip_camera* cam,
                            const double* T_G_C7, int batch_quirk, double* M9);

/* The conservative view bounds the backward-grid mosaic derives from a camera
 * (host arithmetic only; diagnostics / tests), in normalised camera coordinates
 * (x / z, y / z):
 *   out[0] cone   every visible landmark has |p| <= cone            (0: no bound)
 *   out[1], out[2]  ax, ay: every visible landmark has |x| <= ax, |y| <= ay
 *   out[3] rin    |p| <= rin  =>  the landmark is visible             (0: unknown)
 * "visible" = aslam project3's image-box test (ortho-backward-grid.cc:164-171);
 * for an undistorted camera cone = rin = 0 and ax, ay describe the image box. */
int amhip_camera_view_bounds(const amhip_camera* cam, double* out4);

/* ---- io::AerialMapperIO::loadPointCloudFromFile
 *      (aerial_mapper_io/src/aerial-mapper-io.cc:309-347) --
 * The text point-cloud format in front of the DSM: whitespace separated
 * records `x y z intensity`, read like `infile >> x >> y >> z >> intensity`
 * (tokens four at a time; reading stops at the first token that is not a
 * number; an incomplete last record is dropped), points with z <= -100 dropped.
 * host_text is the file's content (e.g. mmap'ed).  The text is tokenised and
 * parsed on the GPU -- decimal -> double correctly rounded (Eisel-Lemire; the
 * rare inputs it cannot decide are re-done with strtod on the host and counted
 * in *num_strtod_tokens) -- and the result stays on the device in the layout
 * amhip_dsm_process_dev / amhip_ortho_from_pcl_process_dev take: dev_xyz = 3 *
 * num_points doubles (AoS), dev_intensities = num_points int32, file order.
 * Both buffers are allocated here; release them with amhip_io_free().
 * Synchronous.  Pose files (`x y z qw qx qy qz`, :103-121) are a few KB and are
 * read on the host by the C++ shim. */
int amhip_io_parse_point_cloud_text(int device, const char* host_text, size_t len,
                                    double** dev_xyz, int32_t** dev_intensities,
                                    size_t* num_points, size_t* num_strtod_tokens);
int amhip_io_download_point_cloud(const double* dev_xyz, const int32_t* dev_intensities,
                                  size_t n, double* host_xyz, int32_t* host_intensities);
int amhip_io_free(void* dev_ptr);

/* ---- what leaves the map: images, GeoTiff, grid_map_msgs, binary clouds
 *      (the formats either side of the path) ------------------
 * grid_map_cv, grid_map_ros, GDAL and roscpp are not part of the reference tree; the adopted
 * definitions are restated in oracle/amo_export.py (parity unpinned). */

/* A layer as an 8-bit image in grid_map_cv::GridMapCvConverter::toImage's orientation (image row
 * = grid index 0, image column = grid index 1; aerial-mapper-grid-map.cc:10 includes it).
 * bgr == 0: toImage<unsigned char, 1>(map, layer, CV_8UC1, lower, upper, image): the layer clamped
 * to [lower, upper], finite cells -> (uchar)(((v - lower) / (upper - lower)) * 255.0f), others 0.
 * bgr != 0: the layer holds grid_map's packed colours (colored_ortho: the float's bits are
 * R << 16 | G << 8 | B, ortho-backward-grid.cc:203-207): an 8UC3 image in OpenCV's B, G, R order,
 * NaN -> 0, 0, 0 (lower / upper unused).  step = bytes per image row.  A transposition through
 * LDS on the device; the _dev form leaves the image in HBM. */
int amhip_layer_to_image_dev(amhip_ctx* ctx, int layer, int bgr, float lower, float upper,
                             uint8_t* dev_image, size_t step);
int amhip_layer_to_image(amhip_ctx* ctx, int layer, int bgr, float lower, float upper,
                         uint8_t* host_image, size_t step);

/* The GeoTiff container io::AerialMapperIO::toGeoTiff / writeDataToDEMGeoTiffColor produce with
 * GDAL (aerial_mapper_io/src/aerial-mapper-io.cc:349-509), without GDAL: classic little-endian
 * TIFF, 8 bits per sample, bands = 1 (BlackIsZero) or 3 (pixel interleaved, as GDAL's GTiff
 * driver lays out Create(.., 3, GDT_Byte, NULL)), uncompressed strips, ModelPixelScale /
 * ModelTiepoint from the north-up geotransform {x0, dx, 0, y0, 0, -dy}, GeoKeys of
 * "WGS 84 / UTM zone <utm_zone><N|S>" with the reference's citation.  pixels = height rows of
 * `step` bytes.  Host code only (no device is touched). */
int amhip_geotiff_write_u8(const char* filename, const uint8_t* pixels, int width, int height,
                           size_t step, int bands, const double* geotransform, int utm_zone,
                           int northern);

/* grid_map_msgs/GridMap in ROS 1 wire format, as GridMapRosConverter::toMessage fills it for
 * AerialGridMap::publishOnce / publishUntilShutdown (aerial-mapper-grid-map.cc:51-72).
 * _bytes: size of the message; _layout: writes everything except the layers' float payloads and
 * reports where each payload (rows * cols floats, column-major) belongs.  Host code only. */
size_t amhip_grid_map_msg_bytes(const amhip_grid_desc* grid, const char* frame_id, int num_layers,
                                const char* const* layer_names);
int amhip_grid_map_msg_layout(const amhip_grid_desc* grid, uint64_t stamp_ns, const char* frame_id,
                              int num_layers, const char* const* layer_names, uint8_t* out,
                              size_t cap, size_t* payload_offsets);

/* A binary point-cloud file ("AMPCLD01", n, flags, n x 3 float64, n x int32 intensities: what
 * loadPointCloudFromFile, aerial-mapper-io.cc:309-347, leaves in its vectors).  The loader
 * stages the file through two pinned buffers (read() of one chunk overlaps the link transfer of
 * the previous one) and leaves the cloud in HBM in the layout of amhip_dsm_process_dev /
 * amhip_ortho_from_pcl_process_dev; release with amhip_io_free().  *dev_intensities = NULL for a
 * file without intensities. */
int amhip_io_write_point_cloud_binary(const char* filename, const double* host_xyz,
                                      const int32_t* host_intensities, size_t n);
int amhip_io_load_point_cloud_binary(int device, const char* filename, double** dev_xyz,
                                     int32_t** dev_intensities, size_t* num_points);

/* ---- measurement ----------------------------------------------------------*/

/* Kernel slots for amhip_ctx_kernel_time(). */
typedef enum amhip_kernel {
  AMHIP_K_DSM_BIN_COUNT = 0, /* sort: partition histogram (+ reduce, scan)  */
  AMHIP_K_DSM_SCAN = 1,      /* sort: in-LDS placement of the sub-partitions */
  AMHIP_K_DSM_SCATTER = 2,   /* sort: the two LDS-staged scatter passes      */
  AMHIP_K_DSM_GATHER = 3,    /* per-cell radius search + IDW               */
  AMHIP_K_ORTHO = 4,         /* per-tile frame cull + per-cell fold/sample */
  AMHIP_K_MISC = 5,          /* memsets / small helpers                    */
  AMHIP_K_HALO_SELECT = 6,   /* multi-GPU: compact the halo points          */
  AMHIP_NUM_KERNELS = 7
} amhip_kernel;

/* ---- stereo::Rectifier::rectifyStereoPair (+ computeMask)
 *      aerial_mapper_dense_pcl/src/rectifier.cpp:34-128 -- the step in front of the block
 *      matcher of the dense-point-cloud pipeline (its other half is amhip_densify_dev) --------
 * K, R_G_C1, R_G_C2: row-major 3x3 on the HOST (intrinsics; orientation of the left / right
 * camera in the world), t_G_C1, t_G_C2 their positions.  dev_left / dev_right: 8UC1 rasters on
 * the GPU, rows `*_step` bytes apart.  Outputs, any may be NULL: R_G_C_out (host, 3x3 rectified
 * rotation, RectifiedStereoPair::R_G_C), baseline_out (host), dev_maps (4 planes of
 * width*height floats: map_rectify_1_x, _1_y, _2_x, _2_y), the rectified images and the mask
 * (dense width*height bytes on the GPU).  Asynchronous on the context's stream.  A zero w of
 * the inverse rectifying transformation (CHECK_NE(xyw(2), 0.0)) is reported by
 * amhip_ctx_synchronize as AMHIP_ERR_ARG. */
int amhip_rectify_stereo_pair_dev(amhip_ctx* ctx, const double* K, const double* R_G_C1,
                                  const double* R_G_C2, const double* t_G_C1, const double* t_G_C2,
                                  int width, int height, const uint8_t* dev_left, size_t left_step,
                                  const uint8_t* dev_right, size_t right_step, double* R_G_C_out,
                                  double* baseline_out, float* dev_maps, uint8_t* dev_rect_left,
                                  uint8_t* dev_rect_right, uint8_t* dev_mask);

/* ---- session: one map served through HOST matrices by one or several GPUs ---------------
 *
 * What the drop-in classes share per grid_map::GridMap (the .cc files under aerial_mapper_amd/cpp): the map
 * is cut into tiles_i x tiles_j windows, window k = (k % tiles_i, k / tiles_i) lives on
 * devices[k] (NULL = all on device 0; a device may serve several windows), everything inside ONE
 * host process like the reference's demos (main-dsm.cc:103-107, main-ortho-backward-grid.cc:
 * 128-141).  A DSM call uploads one slice of the cloud per window, every device selects what
 * every window needs of its slice (cells + halo margin) and the selections travel device to
 * device (hipMemcpyPeerAsync over xGMI); frames and poses are replicated.
 * The layers stay resident between calls: whether a host matrix still holds what the devices
 * hold is decided by a 64-bit content sum (host threads on the way in, a kernel on the way out)
 * -- equal: no transfer; the layer's initial constant: a lazy device-side reset; anything
 * else: uploaded.  Outputs the kernels left unchanged are not downloaded; of a layer that a call
 * changed only inside a rectangle it knows (a small cloud's bounding box on a large map, the
 * tiles a small batch of frames can see: the incremental use case) only that rectangle is
 * (tuning knob session_no_partial: whole windows).  Results are those of the single-context calls
 * (same kernels; windows reproduce the full map).
 * tuning knob session_always_copy / amhip_session_set_always_copy: every matrix up and down.
 * amhip_session_transfer_stats: bytes of layer data uploaded / downloaded since the session was
 * created (clouds and frames not counted). */
typedef struct amhip_session amhip_session;
int amhip_session_create(const amhip_grid_desc* grid, int tiles_i, int tiles_j,
                         const int32_t* devices, amhip_session** out);
void amhip_session_destroy(amhip_session* s);
int amhip_session_num_windows(const amhip_session* s);
amhip_ctx* amhip_session_context(amhip_session* s, int window);
int amhip_session_window(const amhip_session* s, int window, int32_t* i0_j0_rows_cols);
int amhip_session_set_always_copy(amhip_session* s, int on);
int amhip_session_set_dsm_precision(amhip_session* s, int mode);  /* AMHIP_DSM_FAST / _EXACT */
int amhip_session_transfer_stats(const amhip_session* s, uint64_t* uploaded_bytes,
                                 uint64_t* downloaded_bytes);
/* Where the LAST host-buffer call on this session (DSM / backward mosaic / OrthoFromPcl) spent its
 * time, window 0's view: out8 = { total_ms, h2d_ms (wall clock until the call's inputs -- cloud or
 * frames -- were handed to the copy engine; pageable sources block that long), host_sum_ms (host
 * content sums, running beside the upload), kernel_ms (HIP events around the call's kernels),
 * dev_sum_wait_ms (wall clock spent waiting for device content sums that could not run beside a
 * download), d2h_ms (HIP events around the downloads), bytes uploaded as inputs, layer bytes
 * downloaded }. */
int amhip_session_last_profile(const amhip_session* s, double* out8);
/* dsm::Dsm::process (dsm.cc:186-201): `elevation` = the GridMap's matrix (map rows x cols,
 * column-major), read and written like the reference does. */
int amhip_session_dsm_process(amhip_session* s, const double* host_xyz, size_t n, int radius_sq,
                              double center_easting, double center_northing, float* elevation);
/* ortho::OrthoFromPcl::process (ortho-from-pcl.cc:20-113): `ortho` = the GridMap's matrix. */
int amhip_session_ortho_from_pcl_process(amhip_session* s, const double* host_xyz,
                                         const int32_t* host_intensities, size_t n, int radius_sq,
                                         int adaptive, float* ortho);
/* ortho::OrthoBackwardGrid::process (ortho-backward-grid.cc:223-239): the six matrices of the
 * map (all required except the one of ortho / colored_ortho the mode does not write). */
int amhip_session_ortho_backward_process(
    amhip_session* s, const amhip_camera* cam, const double* host_T_G_C, size_t F,
    const uint8_t* const* images, const size_t* steps, int channels, int colored,
    const float* elevation, float* elevation_angle, float* observation_index,
    float* num_observations, float* ortho, float* colored_ortho);

/* With timing enabled every launch is bracketed by hipEvents on the
 * context's stream.  amhip_ctx_kernel_time() synchronises, then reports the
 * accumulated milliseconds and launch count of one slot since the last
 * amhip_ctx_timing_reset(). */
int amhip_ctx_enable_timing(amhip_ctx* ctx, int on);
int amhip_ctx_timing_reset(amhip_ctx* ctx);
int amhip_ctx_kernel_time(amhip_ctx* ctx, int kernel, double* total_ms,
                          int64_t* launches);
const char* amhip_kernel_name(int kernel);

/* Counters of the last amhip_dsm_process*: points kept after the bin-range
 * filter, number of bins, bin edge in cells. */
int amhip_ctx_dsm_stats(amhip_ctx* ctx, int64_t* points_binned,
                        int64_t* num_bins, int32_t* bin_cells);

/* Where the gather tiles of the last amhip_dsm_process* went (synchronises): out8[0..6] = tiles
 * on the lists of the tiled gather -- [0] occupied class-0 tiles (sparse calls only), [1] / [2]
 * capacity classes 1 / 2 (denser than the main launch's LDS image), [3] beyond any LDS image
 * (wave-per-block kernel), [4] / [5] tiles the single-precision gather handed to the FP64 kernel
 * (height range leaves no room under the 1e-4 m error bound: rough terrain), [6] those beyond
 * its largest image -- and out8[7] = tiles of the map.  All zero when the LDS-tiled gather was
 * not used. */
int amhip_ctx_dsm_gather_stats(amhip_ctx* ctx, int64_t* out8);

/* Order `other_stream` (hipStream_t as void*) behind everything the context has enqueued so far: an
 * event recorded on the context's stream, waited for by the other stream -- no host wait.  For
 * callers that feed asynchronous calls (`*_dev` entry points) from buffers another stream refills. */
int amhip_ctx_order_after(amhip_ctx* ctx, void* other_stream);

/* The session's map as a grid_map_msgs/GridMap message (ROS 1 wire format): the resident layers
 * travel from the devices straight into `out`.  layer_ids[l] = the amhip layer behind message
 * layer l, or -1: host_layers[l] (rows x cols floats, column-major) is copied, NaN-filled when
 * null.  cap >= amhip_grid_map_msg_bytes(..); *written = the message size. */
int amhip_session_grid_map_msg(amhip_session* s, uint64_t stamp_ns, const char* frame_id,
                               int num_layers, const char* const* layer_names,
                               const int32_t* layer_ids, const float* const* host_layers,
                               uint8_t* out, size_t cap, size_t* written);
/* amhip_layer_to_image for the whole map of a session (every window on its own device). */
int amhip_session_layer_to_image(amhip_session* s, int layer, int bgr, float lower, float upper,
                                 uint8_t* host_image, size_t step);

/* ---- tuning knobs: the ONE door for switches that select among correct implementations ----------
 * (tests force every path through them, A-B timing flips them; none is needed in normal use).
 * amhip_set_tuning(key, value) -- value NaN clears the key -- or, for hosts that cannot be
 * recompiled, AMHIP_TUNING="key=value,key,..." in the environment (read once; a bare key = 1).
 * Unknown keys are an error (AMHIP_ERR_ARG) / a warning on stderr.  Process-wide; looked up when a
 * call is made.  Keys:
 *   sort_one_level            clouds of any size take the one-level counting sort
 *   p3_min_points (2^20)      cloud size from which the three-pass partition sort is used
 *   p3_target (1536)          points per sub-partition the sort's plan aims for
 *   p3_cap (2048)             points a placement workgroup sorts in its registers / LDS
 *   p3_rounds_cap, p3_rounds_reread   placement in rounds (sub-partitions beyond one LDS image)
 *                             forced at test size: image of n points / re-reading instead of registers
 *   no_launch_skips           launch every capacity-class / big-list kernel whatever the previous
 *                             call's counters say
 *   dsm_canon_all             every FP64 quotient goes through the order-independent double-double sums
 *   knn_global_bins           the optional k-cap mode on the plain global-bins kernel (default: LDS-tiled)
 *   dsm_no_rough_switch       the single-precision mode stays in its own pipeline on rough scenes
 *   dsm_no_subwindow          small clouds onto large maps are binned over the whole window
 *   eager_reset               amhip_layers_reset fills the layers at once (default: fused into producers)
 *   ortho_exact_fold, ortho_no_prune, no_coarse_cull, ortho_no_tile_list
 *                             mosaic: every pair in the reference's arithmetic / keep dominated frames /
 *                             no small-batch pre-cull / dispatch every tile of a large map instead of a
 *                             list of visible ones
 *   no_distorted_cull, no_distorted_prune, distorted_square_cull   cameras with a distortion model
 *   session_always_copy, session_threads, session_scalar_sums, session_no_partial,
 *   session_serial_sums       device content sums always before the downloads (round 5's order)
 *   session_verify_partial, session_trace          amhip_session: every matrix both ways on every
 *                             call / host threads of the content sums / scalar sums / whole-window
 *                             downloads / re-sum partial downloads / phase timings on stderr
 *   (lab builds, -DAMHIP_TIMING_PROBES: gather_tj, gather_nt, gather_class_cap0..2, f32_variant, fx_theta)
 * Environment variables the library itself reads, all of them: AMHIP_TUNING (above), AMHIP_DSM_FAST /
 * AMHIP_DSM_EXACT (amhip_default_dsm_precision); the C++ drop-in classes add AERIAL_MAPPER_HIP_DEVICE
 * and AERIAL_MAPPER_HIP_DEVICES (aerial_mapper_amd/cpp/shim_common.*). */
int amhip_set_tuning(const char* key, double value);
double amhip_get_tuning(const char* key, double dflt);
/* AMHIP_DSM_EXACT unless the environment says AMHIP_DSM_FAST=1 (and not AMHIP_DSM_EXACT): what new
 * contexts, sessions and dsm::Dsm objects start with. */
int amhip_default_dsm_precision(void);

/* The library's build id: 16 hex digits of the SHA-256 over its sources, headers and compiler flags
 * (aerial_mapper_amd/build.py).  Profiles collected from one build are refused as evidence for
 * another (bench.py). */
const char* amhip_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* AERIAL_MAPPER_HIP_H_ */
