// amhip_common.h -- internal declarations shared by the HIP translation units
// of libaerial_mapper_hip.so (gfx950 only; not part of the public C ABI).
#ifndef AMHIP_COMMON_H_
#define AMHIP_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "aerial_mapper_hip.h"
#include "amhip_tuning.h"
#include "amhip_ortho_fold.h"

namespace amhip {

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
void set_last_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define AMHIP_TRY(expr)                                                  \
  do {                                                                   \
    hipError_t _e = (expr);                                              \
    if (_e != hipSuccess) return ::amhip::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

// ---------------------------------------------------------------------------
// device-side constants
// ---------------------------------------------------------------------------
constexpr int kMaxLevels = 24;  // radius ladder: R, then lambda_k * R (<= 22 for R = 1)
constexpr int kMaxW0 = 16;      // largest first-level window (cells) the LDS gather handles

// Geometry + search parameters of one DSM call, passed by value to kernels.
struct DsmParams {
  // grid_map_core getPosition: x_i = base_x + res * (-(double)i)
  double base_x, base_y, res;
  int rows, cols;    // size of the window this context owns
  int i_off, j_off;  // its position inside the (global) map
  // dsm.cc:42-43: px = p.x - center_northing, py = p.y - center_easting
  double sub_x, sub_y;
  // 0: dsm::Dsm (exact hit = CHECK failure); 1: ortho::OrthoFromPcl (exact hit
  // = that point's value, ortho-from-pcl.cc:91-96)
  int pcl_mode;
  // adaptive OrthoFromPcl passes: touch only cells no earlier pass has filled
  int only_unfilled;
  // binning: bins of B x B cells, grid extended by M cells on every side
  double inv_res;
  int B, M, nbx, nby;
  // radius ladder (squared radii, exactly the doubles dsm.cc:127-144 uses)
  int nlevels;
  double T[kMaxLevels];
  int w[kMaxLevels];  // conservative window half-width in cells for T[k]
  // LDS-tiled gather (first level only): per window row dj in [-w0, w0] the
  // half-width in cells of the disc of squared radius T[0]
  int lds_ok;                 // 0 -> every tile takes the global-memory path
  // three-pass partition sort plan (p3_n1 == 0: not used)
  int p3_r1;                  // bin rows per first-pass partition
  int p3_c, p3_w;             // column blocks per bin row, bins per column block
  int p3_n1, p3_n2;           // partitions of pass 1; sub-partitions of each (= p3_r1 * p3_c)
  int p3_cap;                 // points a pass-3 workgroup can sort in LDS
  // Every sort pass turns every point into its keys again: four integer divisions by
  // B / p3_r1 / p3_w per point and pass were ~half of the passes' VALU time.  Multipliers
  // m = floor(2^32 / d) + 1: n / d == umulhi(n, m) for n * d < 2^32 (0: d == 1; ~0: n * d may
  // reach 2^32 on this grid -> the real division); div_by() in amhip_sort.hip
  unsigned mul_B, mul_r1, mul_w;
  int wr[2 * kMaxW0 + 1];
  int wr2[2 * kMaxW0 + 2];    // the same for a pair of cells (j, j+1): max of both
  int wrp[kMaxW0 + 1];        // per trip (window rows 2k, 2k+1 of the pair): max of wr2
  int lds_cap;                // points the tile's LDS image can hold
  int lds_cells;              // cell-offset table entries reserved (+1 sentinel)
  int tile_j;                 // tile height in cells: 32, or 16 for dense clouds
  int tiles_i, tiles_j;
  unsigned lds_bytes;
  // single-precision gather with exact guards (k_dsm_gather_f32): point
  // positions as 32-bit fixed point in units of 2^-fx_S cells (wrapping: only differences of
  // at most w0 + 2 cells are ever formed), squared distances in f32 in (2^-fx_S cells)^2
  // OPTIONAL capped mode (not a reference code path): only the knn_k nearest points of a
  // cell's search result take part (0 = off = the reference's behaviour)
  int knn_k;
  int fx_ok;                  // 0 -> the FP64 gather only; 1 -> the sort leaves 16-byte records
                              // (amhip_sort.hip, "the record pipeline") and the single-precision
                              // gather runs on them
  int fx_S;
  float fx_thi, fx_tlo;       // T[0] in those units, widened / narrowed by the decision margin
  float fx_denmax;            // a hit nearer than fx_theta cells makes the weight sum exceed this
  float fx_epsw;              // bound on the relative error of one weight
  unsigned lds_bytes_f32;
  int canon_all;              // tests: every FP64 quotient goes through canonical_search (amhip_dsm.hip)
  // Where cell (i, j) of THIS call's window lives in the layer it writes: out_i0 + i + (out_j0 + j)
  // * out_pitch.  A call on the context's whole window: 0, 0, rows.  A small cloud onto a large map
  // runs on a SUB-window around the cloud's bounding box (amhip_api.hip: dsm_subwindow) and
  // writes into the full layer.
  int out_i0, out_j0, out_pitch;
};

// The binned cloud as the gather sees it (amhip_dsm.hip: pts_x / pts_y / pts_z; filled by
// dsm_sort): the doubles pipeline (sorted), the record pipeline of the single-precision
// gather (rec + sidx + cloud), or both.
struct PtsView {
  const double* sorted;   // 3 doubles per sorted point (px, py, z), or null
  const uint4* rec;       // 16-byte record per sorted point, or null: x = cell ix | iy << 16
                          // (map cells + margin M), y / z = offsets from the cell centre in
                          // units of 2^-fx_S cells (int32), w = f32 bits of z - zref[0]
  const uint32_t* sidx;   // row of the caller's cloud behind sorted point g (with rec)
  const double* cloud;    // the caller's cloud (AoS x, y, z), untouched
  const double* zref;     // device: [0] the reference height of the records' offsets
  double sub_x, sub_y;    // dsm.cc:42-43's centre offsets (applied when reading `cloud`)
};

// FramePose / FrameFast (per-frame inverse pose T_C_G = T_G_C^-1): amhip_ortho_fold.h

struct OrthoParams {
  double base_x, base_y, res;
  int rows, cols;
  int i_off, j_off;
  // camera
  double fu, fv, cu, cv;
  double dist[4];
  int width, height, distortion;
  // frames
  int num_frames;
  int channels, colored;
  size_t frame_stride, row_step;
  // frustum side planes in the camera frame (unit normals, inside = n.p >= 0),
  // only valid when distortion == NONE
  double pl[4][3];
  int cull;
  // lazy reset: elevation_angle / observation_index / the output layer are
  // logically in their initial state but their memory is not filled -- do not
  // read them, and write the initial values into every cell no view is
  // accepted for; same for num_observations (initial 0: `+= itself` keeps it)
  int virt_out, virt_nobs;
  // the global elevation range is available: pre-cull the frames with it and
  // leave the tile before touching its elevation when nothing can see it
  int coarse;
  // margin-guarded fold (amhip_ortho_fold.h): undistorted pinhole, unit
  // quaternions; the frame table follows the poses in the same buffer
  int fast;
  FoldCam fold;
  // dominance pruning of a tile's frame list (frame_bounds / dominated):
  // undistorted pinhole, num_observations logically zero
  int prune;
  // the bounding spheres assume rigid poses; |q|^2 = 1 + dev scales distances
  // by that much: radii are multiplied by 1 + 2 max|dev|
  double radius_scale;
  // cameras with a distortion model: |p| <= r_in z  =>  visible by a clear
  // margin (0: unknown); lets frame_bounds() call a frame fully visible
  double r_in;
};

// Device error word bits (sticky until amhip_ctx_synchronize).
enum : unsigned { kDevErrExactHit = 1u, kDevErrAlphaNonPos = 2u, kDevErrHaloOverflow = 4u,
                  kDevErrRectifyZeroW = 8u };

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
constexpr int kMaxHaloDests = 8;
struct HaloParams {
  double base_x, base_y, inv_res, sub_x, sub_y;
  int nd;
  double lo_i[kMaxHaloDests], hi_i[kMaxHaloDests], lo_j[kMaxHaloDests], hi_j[kMaxHaloDests];
  unsigned long long cap;
  // where destination d's rows start in the output (in points) and how many it may take:
  // d * cap / cap for the equal-split layout of the RCCL exchange; prefix sums of the counts /
  // the counts themselves for the session's count-then-compact routing (cap_d = 0: count only)
  unsigned long long off[kMaxHaloDests], cap_d[kMaxHaloDests];
  // a box (open) no destination reaches into -- the inside of the context's own window --
  // or an empty one: a point in it is done after four comparisons
  double in_lo_i, in_hi_i, in_lo_j, in_hi_j;
};
// Multi-GPU tiling (amhip_dsm_tiled_begin_dev / _finish_dev): the sort's count pass in two
// parts around the caller's halo exchange.  phase 1: count rows [0, n_prefix) of the cloud
// and copy the points other windows need into their send rows, then stop; phase 2: count
// rows [n_prefix, n) (what the exchange delivered) and carry on.
struct SortSplit {
  int phase;
  size_t n_prefix;
  HaloParams hp;
  double* halo_out;
  unsigned long long* halo_counts;
};

struct TimedRegion {
  hipEvent_t a, b;
  int slot;
};

struct Ctx {
  amhip_grid_desc grid;       // the (global) map
  int win_i0 = 0, win_j0 = 0; // window of it this context owns
  int win_rows = 0, win_cols = 0;
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t order_event = nullptr;  // amhip_ctx_order_after
  size_t cells = 0;
  int dsm_exact = 0;          // amhip_ctx_set_dsm_precision
  // single-precision mode on rough terrain (amhip_api.hip: dsm_rough_policy): when a call filed
  // more than half of its tiles for the FP64 kernel, the next calls run the FP64 pipeline
  // outright (sorted doubles instead of records + the caller's unsorted cloud)
  int dsm_exact_now = 0;      // this call
  int rough_hold = 0;         // calls left before the single-precision pipeline is tried again
  bool last_call_f32 = false; // the last DSM call ran the record pipeline with bin ranges
  int dsm_knn = 0;            // amhip_ctx_set_dsm_knn

  float* layers[AMHIP_NUM_LAYERS] = {nullptr, nullptr, nullptr,
                                     nullptr, nullptr, nullptr};
  // 0: holds its initial value since the last reset (a reset need not touch
  // it); 1: possibly written; 2: its device pointer was handed out (the caller
  // may write at any time: always refilled); 3: logically initial, memory not
  // filled (lazy reset: the next producer kernel fuses the fill)
  unsigned char layer_state[AMHIP_NUM_LAYERS] = {1, 1, 1, 1, 1, 1};
  // [min, max] of the heights every DSM call since the last reset could have
  // written (ordered 64-bit keys, amhip_device.h); lets the mosaic kernel drop
  // tiles no frame of a small batch can see before it reads their elevation.
  // Not valid once the elevation layer was written from outside.
  unsigned long long* dev_zrange = nullptr;
  bool zrange_valid = false;
  double* zpart = nullptr;       // per-workgroup partials of one call
  size_t zpart_cap = 0;
  unsigned* dev_err = nullptr;   // device error word
  unsigned* host_err = nullptr;  // pinned mirror

  // DSM workspaces (grow on demand)
  PtsView pts = {};              // what the last dsm_sort left for the gather
  double* sorted = nullptr;      // 3 doubles per binned point (px, py, z)
  size_t sorted_cap = 0;         // in points
  uint32_t* rank = nullptr;      // per input point: rank inside its bin
  size_t rank_cap = 0;
  uint32_t* bin_start = nullptr; // nbins + 1 (+ scan partials behind it)
  size_t bin_cap = 0;
  // record pipeline (single-precision gather): 20-byte records between the sort passes, the
  // 16-byte records + rows the gather reads, the records' reference height
  uint32_t* rec_a = nullptr;
  size_t rec_a_cap = 0;
  uint32_t* rec_b = nullptr;
  size_t rec_b_cap = 0;
  uint32_t* rec16 = nullptr;     // (uint4 per point)
  size_t rec16_cap = 0;
  uint32_t* sidx = nullptr;
  size_t sidx_cap = 0;
  double* zref = nullptr;        // [0] reference height, [1] / [2] range of the cloud's heights
  size_t zref_cap = 0;
  double* zall = nullptr;        // per-wave [min, max] partials of the count pass
  size_t zall_cap = 0;
  uint32_t* bin_z = nullptr;     // per bin: ordered keys of its lowest / highest height (uint2)
  size_t bin_z_cap = 0;
  bool bin_z_valid = false;      // written by the last sort (three-pass, single-precision mode)
  uint32_t* scan_partials = nullptr;
  size_t partial_cap = 0;
  double* tmp_points = nullptr;    // stripe-ordered points (level 1 of the sort)
  size_t tmp_points_cap = 0;
  uint32_t* stripe_ws = nullptr;   // stripe counts / starts / cursors
  size_t stripe_ws_cap = 0;
  uint8_t* tile_occ = nullptr;         // per gather tile: any point within the last radius
  size_t tile_occ_cap = 0;
  int* tile_list = nullptr;            // sparse gather: [count (4 ints)] [occupied tile ids]
  size_t tile_list_cap = 0;
  unsigned char* fill_mask = nullptr;  // OrthoFromPcl adaptive passes
  size_t fill_mask_cap = 0;
  int32_t* stage_values = nullptr;     // H2D staging of host intensities
  size_t stage_values_cap = 0;
  double* stage_points = nullptr;  // H2D staging of host clouds
  size_t stage_points_cap = 0;

  // ortho workspaces
  amhip_camera cone_cam = {};    // camera the cached view cone belongs to
  double cone = 0.0;             // distorted_view_cone() result
  int cone_state = 0;            // 0: none cached, 1: usable bound, 2: no bound
  int cone_state_rect = 0;       // distorted_rectangle_and_inner_cone() cached for `cone_rect_cone`
  double cone_rect_cone = 0.0, cone_ax = 0.0, cone_ay = 0.0, cone_rin = 0.0;
  FramePose* frame_poses = nullptr;
  size_t frame_pose_cap = 0;
  uint8_t* stage_frames = nullptr;
  size_t stage_frames_cap = 0;
  // The cells of the window the LAST DSM / mosaic call can have written (window coordinates):
  // the whole window, the sub-window of a small cloud, or the bounding box of the mosaic's tile
  // list (then still on the device: dirty_on_device, read back by ctx_last_dirty()).  What the
  // session downloads after the call (amhip_session.hip) instead of whole layers.
  int dirty[4] = {0, 0, 0, 0};   // i0, j0, rows, cols
  bool dirty_on_device = false;
  int* dev_bbox = nullptr;       // k_dsm_bbox: [min i, max i, min j, max j, count] of a small cloud
  int* host_bbox = nullptr;      // pinned mirror
  int* ortho_list = nullptr;     // [count (4 ints)] [tiles some frame of a small batch can see]
  size_t ortho_list_cap = 0;

  // amhip_dsm_tiled_begin_dev .. amhip_dsm_tiled_finish_dev
  bool tiled_pending = false;
  const double* tiled_xyz = nullptr;
  size_t tiled_n = 0;
  int tiled_radius_sq = 0;
  double tiled_ce = 0.0, tiled_cn = 0.0;
  SortSplit tiled_split = {};
  DsmParams tiled_params = {};  // the begin call's plan: the finish call must bin with the same

  // stats of the last DSM call
  int64_t last_points_binned = 0;
  int64_t last_num_bins = 0;
  int32_t last_bin_cells = 0;
  int64_t last_ntiles = 0;        // gather tiles of the last call (0: not the LDS-tiled gather)
  // Launch bookkeeping (round 5): what the sort's single-workgroup kernel resets for the gather
  // (SortAux), where the scatter waves' height partials end, the ticket of the count pass's
  // reduce-then-scan kernel, and the pinned words the device leaves for the NEXT call's launch
  // policy (never waited for): [0] sub-partitions beyond the placement workgroup's registers.
  uint32_t* aux_zero_words = nullptr;
  int aux_nzero = 0;
  bool aux_done = false;
  size_t range_parts = 0;
  unsigned long long* range_running = nullptr;
  unsigned* dev_tickets = nullptr;      // 16 words, zero between launches
  unsigned* host_sort_stats = nullptr;  // pinned, 4 words
  unsigned long long sort_stats_sig = 0, tile_stats_sig = 0;  // the geometry the pinned counters describe
  unsigned* host_tile_stats = nullptr;  // pinned mirror of the last call's tile-list counters

  // timing
  bool timing = false;
  std::vector<TimedRegion> regions;
  std::vector<TimedRegion> free_regions;
  double slot_ms[AMHIP_NUM_KERNELS] = {};
  int64_t slot_launches[AMHIP_NUM_KERNELS] = {};
};

// RAII-less helper: bracket a launch sequence with events when timing is on.
struct ScopedTimer {
  Ctx* c;
  TimedRegion r;
  bool on;
  ScopedTimer(Ctx* ctx, int slot);
  ~ScopedTimer();
};

template <typename T>
int ensure_capacity(T** ptr, size_t* cap, size_t need);

int ensure_bytes(void** ptr, size_t* cap_bytes, size_t need_bytes);

// ---------------------------------------------------------------------------
// launchers (defined in amhip_dsm.hip / amhip_ortho.hip)
// ---------------------------------------------------------------------------
int launch_fill(Ctx* c, float* dst, size_t n, float value);

// stereo densifier reprojection (amhip_densify.hip)
struct DensifyParams {
  int width, height;
  size_t disp_step, img_step;  // bytes per row
  double Q03, Q11, Q13, Q23, Q32;
  double R[9], t[3];
};
int densify_run(Ctx* c, const DensifyParams& p, const float* dev_disparity,
                const uint8_t* dev_image_left, double* dev_xyz_out, int32_t* dev_intensity_out,
                size_t capacity, long long* dev_count);

// multi-GPU halo selection
int halo_select_run(Ctx* c, const double* dev_xyz, size_t n, const HaloParams& hp,
                    double* dev_out, unsigned long long* dev_counts);
// cells (of p's window, point_bin's arithmetic) the points of a cloud fall into: dev_bbox5 =
// [kBboxBias - min i, max i + kBboxBias, kBboxBias - min j, max j + kBboxBias, points inside the
// binned area] (amhip_sort.hip; all zero = the empty box)
constexpr int kBboxBias = 1 << 30;
int dsm_bbox_run(Ctx* c, const double* dev_xyz, size_t n, const DsmParams& p, int* dev_bbox5);
// geometry of a selection for `nd` destination windows (amhip_api.hip); off / cap_d are set to
// the equal-split layout (d * cap_per_dest, cap_per_dest)
int make_halo_params(const Ctx& c, double center_easting, double center_northing,
                     const int32_t* dest_windows, int nd, double margin_m, size_t cap_per_dest,
                     HaloParams* out);
// values: nullptr -> interpolate the points' z; else one int per point
// (OrthoFromPcl intensities).  out: the layer to write.  mask (may be null): one
// byte per cell, set where this call wrote a value; unfilled (may be null):
// device counter of cells left without a value.
// amhip_sort.hip: bin-sort the cloud into c->sorted / c->bin_start
bool no_launch_skips();            // amhip_sort.hip: tuning knob no_launch_skips
int dsm_sort(Ctx* c, const double* dev_xyz, const int32_t* dev_values, size_t n,
             const DsmParams& p, unsigned long long* zrange, const SortSplit* split = nullptr);
int dsm_run(Ctx* c, const double* dev_xyz, const int32_t* dev_values, size_t n,
            const DsmParams& p, float* out, unsigned char* mask, unsigned* unfilled,
            bool fill_untouched = false, float init_value = 0.0f,
            unsigned long long* zrange = nullptr, const SortSplit* split = nullptr);
// dev_fast: FrameFast[num_frames] followed by one entry holding the camera
// (fu fv cu cv W H) for exact_view(); only read when p.fast
int ortho_run(Ctx* c, const OrthoParams& p, const FramePose* dev_poses, const FrameFast* dev_fast,
              const uint8_t* dev_frames);

// context internals shared with amhip_session.hip (defined in amhip_api.hip)
// i0, j0, rows, cols (window coordinates) of what the last DSM / mosaic call can have written;
// synchronises the stream when the box is still on the device.  rows == 0: nothing.
int ctx_last_dirty(Ctx* c, int rect[4]);
int ctx_use_device(Ctx* c);
int ctx_materialize(Ctx* c, int layer);      // lazy-initial -> filled
void ctx_overwrite(Ctx* c, int layer);       // about to be fully overwritten from outside
int ctx_layer_set_initial(Ctx* c, int layer);  // one layer back to its (lazy) initial state
int ctx_fetch_status(Ctx* c);                // synchronize + sticky device status
float ctx_layer_init_value(int layer);
bool ctx_layer_is_initial(const Ctx* c, int layer);
int arg_failure(const char* msg);

// host-side restatements of the external pose math (minkindr), used to build
// T_G_C and T_C_G; kept in one place so the composition and the per-cell
// transform agree operation for operation.
struct HPose {
  double qw, qx, qy, qz, tx, ty, tz;
};
HPose hpose_from7(const double* p);
HPose hpose_inverse(const HPose& T);
HPose hpose_compose(const HPose& A, const HPose& B);

}  // namespace amhip

struct amhip_ctx {
  amhip::Ctx impl;
};

#endif  // AMHIP_COMMON_H_
