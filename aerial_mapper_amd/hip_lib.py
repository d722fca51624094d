"""ctypes binding of libaerial_mapper_hip.so (include/aerial_mapper_hip.h).

There is NO CPU fallback: if the library is missing this module raises, and if
no gfx950 device is visible `Context()` raises.  (The CPU oracle under oracle/
is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# (AMHIP_LIB_PATH: another build of the same library -- A-B timing of two builds on one box,
# tools/gpu_ab.sh; never a different implementation)
LIB_PATH = os.environ.get("AMHIP_LIB_PATH") or os.path.join(PKG, "lib", "libaerial_mapper_hip.so")

ABI_VERSION = 1

# amhip_status
OK, ERR_ARG, ERR_EXACT_HIT, ERR_ALPHA_NONPOS, ERR_HIP, ERR_NO_DEVICE, ERR_NOMEM, ERR_HALO_OVERFLOW = range(8)
STATUS_NAMES = {OK: "AMHIP_OK", ERR_ARG: "AMHIP_ERR_ARG", ERR_EXACT_HIT: "AMHIP_ERR_EXACT_HIT",
                ERR_ALPHA_NONPOS: "AMHIP_ERR_ALPHA_NONPOS", ERR_HIP: "AMHIP_ERR_HIP",
                ERR_NO_DEVICE: "AMHIP_ERR_NO_DEVICE", ERR_NOMEM: "AMHIP_ERR_NOMEM",
                ERR_HALO_OVERFLOW: "AMHIP_ERR_HALO_OVERFLOW"}

# amhip_layer
(LAYER_ORTHO, LAYER_ELEVATION, LAYER_ELEVATION_ANGLE, LAYER_NUM_OBSERVATIONS,
 LAYER_OBSERVATION_INDEX, LAYER_COLORED_ORTHO) = range(6)
NUM_LAYERS = 6
LAYER_NAMES = ["ortho", "elevation", "elevation_angle", "num_observations",
               "observation_index", "colored_ortho"]

# amhip_kernel
(K_DSM_BIN_COUNT, K_DSM_SCAN, K_DSM_SCATTER, K_DSM_GATHER, K_ORTHO, K_MISC,
 K_HALO_SELECT) = range(7)
NUM_KERNELS = 7

DIST_NONE, DIST_RADTAN, DIST_EQUIDISTANT = 0, 1, 2

# every symbol include/aerial_mapper_hip.h declares
EXPORTS = [
    "amhip_abi_version", "amhip_last_error", "amhip_make_grid", "amhip_cell_position",
    "amhip_ctx_create", "amhip_ctx_create_window", "amhip_ctx_destroy", "amhip_ctx_set_dsm_precision", "amhip_ctx_set_dsm_knn", "amhip_ctx_set_stream", "amhip_ctx_synchronize",
    "amhip_layers_reset", "amhip_layer_upload", "amhip_layer_download",
    "amhip_layer_device_ptr", "amhip_dsm_process_dev", "amhip_dsm_process",
    "amhip_ortho_from_pcl_process_dev", "amhip_ortho_from_pcl_process",
    "amhip_densify_dev", "amhip_rectify_stereo_pair_dev", "amhip_halo_select_dev", "amhip_dsm_tiled_begin_dev",
    "amhip_dsm_tiled_finish_dev", "amhip_compose_T_G_C", "amhip_ortho_backward_process_dev",
    "amhip_ortho_backward_process", "amhip_ctx_enable_timing", "amhip_ctx_timing_reset",
    "amhip_ctx_kernel_time", "amhip_kernel_name", "amhip_ctx_dsm_stats", "amhip_ctx_dsm_gather_stats", "amhip_ctx_order_after", "amhip_session_last_profile", "amhip_build_id", "amhip_set_tuning", "amhip_get_tuning", "amhip_default_dsm_precision",
    "amhip_mosaic_create", "amhip_mosaic_destroy", "amhip_mosaic_set_stream",
    "amhip_mosaic_synchronize", "amhip_mosaic_reset", "amhip_mosaic_batch",
    "amhip_mosaic_batch_dev", "amhip_mosaic_update", "amhip_mosaic_update_dev",
    "amhip_mosaic_download", "amhip_mosaic_device_ptr", "amhip_mosaic_homography",
    "amhip_camera_view_bounds",
    "amhip_session_create", "amhip_session_destroy", "amhip_session_num_windows",
    "amhip_session_context", "amhip_session_window", "amhip_session_set_always_copy",
    "amhip_session_set_dsm_precision", "amhip_session_transfer_stats",
    "amhip_session_dsm_process", "amhip_session_ortho_backward_process",
    "amhip_session_ortho_from_pcl_process",
    "amhip_io_parse_point_cloud_text", "amhip_io_download_point_cloud", "amhip_io_free",
    "amhip_layer_to_image_dev", "amhip_layer_to_image", "amhip_geotiff_write_u8",
    "amhip_grid_map_msg_bytes", "amhip_grid_map_msg_layout", "amhip_io_write_point_cloud_binary",
    "amhip_io_load_point_cloud_binary", "amhip_session_grid_map_msg", "amhip_session_layer_to_image",
]


class GridDesc(C.Structure):
    """amhip_grid_desc"""
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32),
                ("resolution", C.c_double),
                ("length_x", C.c_double), ("length_y", C.c_double),
                ("pos_x", C.c_double), ("pos_y", C.c_double)]


class Camera(C.Structure):
    """amhip_camera"""
    _fields_ = [("fu", C.c_double), ("fv", C.c_double),
                ("cu", C.c_double), ("cv", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32),
                ("distortion", C.c_int32), ("_pad", C.c_int32),
                ("dist", C.c_double * 4)]


class MosaicDesc(C.Structure):
    """amhip_mosaic_desc"""
    _fields_ = [("width_mosaic_pixels", C.c_int32), ("height_mosaic_pixels", C.c_int32),
                ("ground_plane_elevation_m", C.c_double), ("origin", C.c_double * 3)]


class AmhipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: run `python -m aerial_mapper_amd.build` "
            "(there is no CPU fallback for the hot path)" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 /
    # libhsa-runtime64.  If torch is importable, load it FIRST so that this
    # library binds (by SONAME) to the runtime torch uses -- device pointers and
    # streams are then shareable, and the second runtime that would otherwise
    # come up blind ("No HIP GPUs are available") never exists.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, f64p, f32p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
    gp, cp = C.POINTER(GridDesc), C.POINTER(Camera)
    lib.amhip_abi_version.restype = C.c_int
    lib.amhip_last_error.restype = C.c_char_p
    lib.amhip_make_grid.restype = None
    lib.amhip_make_grid.argtypes = [C.c_double] * 5 + [gp]
    lib.amhip_cell_position.restype = None
    lib.amhip_cell_position.argtypes = [gp, C.c_int, C.c_int, f64p, f64p]
    lib.amhip_ctx_create.argtypes = [gp, C.c_int, C.POINTER(vp)]
    lib.amhip_ctx_create_window.argtypes = [gp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(vp)]
    lib.amhip_ctx_destroy.restype = None
    lib.amhip_ctx_destroy.argtypes = [vp]
    lib.amhip_ctx_set_stream.argtypes = [vp, vp]
    lib.amhip_ctx_set_dsm_precision.argtypes = [vp, C.c_int]
    lib.amhip_ctx_set_dsm_knn.argtypes = [vp, C.c_int]
    lib.amhip_ctx_synchronize.argtypes = [vp]
    lib.amhip_layers_reset.argtypes = [vp]
    lib.amhip_layer_upload.argtypes = [vp, C.c_int, vp]
    lib.amhip_layer_download.argtypes = [vp, C.c_int, vp]
    lib.amhip_layer_device_ptr.restype = vp
    lib.amhip_layer_device_ptr.argtypes = [vp, C.c_int]
    lib.amhip_dsm_process_dev.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_double, C.c_double]
    lib.amhip_dsm_process.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_double, C.c_double, vp]
    lib.amhip_ortho_from_pcl_process_dev.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.c_int]
    lib.amhip_ortho_from_pcl_process.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.c_int, vp]
    lib.amhip_densify_dev.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, f64p,
                                      C.c_double, f64p, f64p, vp, vp, C.c_size_t, vp]
    lib.amhip_halo_select_dev.argtypes = [vp, vp, C.c_size_t, C.c_double, C.c_double,
                                          C.POINTER(C.c_int32), C.c_int, C.c_double, vp,
                                          C.c_size_t, vp]
    lib.amhip_dsm_tiled_begin_dev.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_double,
                                              C.c_double, C.POINTER(C.c_int32), C.c_int,
                                              C.c_double, vp, C.c_size_t, vp]
    lib.amhip_dsm_tiled_finish_dev.argtypes = [vp]
    lib.amhip_compose_T_G_C.restype = None
    lib.amhip_compose_T_G_C.argtypes = [f64p, f64p, C.c_size_t, f64p]
    lib.amhip_ortho_backward_process_dev.argtypes = [
        vp, cp, f64p, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
    lib.amhip_ortho_backward_process.argtypes = [
        vp, cp, f64p, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_int, C.c_int,
        vp, vp, vp, vp, vp, vp]
    lib.amhip_session_create.argtypes = [gp, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(vp)]
    lib.amhip_session_destroy.restype = None
    lib.amhip_session_destroy.argtypes = [vp]
    lib.amhip_session_num_windows.argtypes = [vp]
    lib.amhip_session_context.restype = vp
    lib.amhip_session_context.argtypes = [vp, C.c_int]
    lib.amhip_session_window.argtypes = [vp, C.c_int, C.POINTER(C.c_int32)]
    lib.amhip_session_set_always_copy.argtypes = [vp, C.c_int]
    lib.amhip_session_set_dsm_precision.argtypes = [vp, C.c_int]
    lib.amhip_ctx_order_after.argtypes = [vp, vp]
    lib.amhip_session_transfer_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.amhip_session_last_profile.argtypes = [vp, C.POINTER(C.c_double)]
    lib.amhip_session_dsm_process.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_double, C.c_double, vp]
    lib.amhip_session_ortho_from_pcl_process.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.c_int, vp]
    lib.amhip_session_ortho_backward_process.argtypes = [
        vp, cp, f64p, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_int, C.c_int,
        vp, vp, vp, vp, vp, vp]
    lib.amhip_rectify_stereo_pair_dev.argtypes = [vp, f64p, f64p, f64p, f64p, f64p, C.c_int, C.c_int,
                                                  vp, C.c_size_t, vp, C.c_size_t, f64p, f64p, vp, vp, vp, vp]
    lib.amhip_ctx_enable_timing.argtypes = [vp, C.c_int]
    lib.amhip_ctx_timing_reset.argtypes = [vp]
    lib.amhip_ctx_kernel_time.argtypes = [vp, C.c_int, f64p, C.POINTER(C.c_int64)]
    lib.amhip_kernel_name.restype = C.c_char_p
    lib.amhip_set_tuning.argtypes = [C.c_char_p, C.c_double]
    lib.amhip_get_tuning.restype = C.c_double
    lib.amhip_get_tuning.argtypes = [C.c_char_p, C.c_double]
    lib.amhip_default_dsm_precision.argtypes = []
    lib.amhip_build_id.restype = C.c_char_p
    lib.amhip_build_id.argtypes = []
    lib.amhip_kernel_name.argtypes = [C.c_int]
    lib.amhip_ctx_dsm_gather_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.amhip_ctx_dsm_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int32)]
    mp = C.POINTER(MosaicDesc)
    lib.amhip_mosaic_create.argtypes = [mp, cp, C.c_int, C.POINTER(vp)]
    lib.amhip_mosaic_destroy.argtypes = [vp]
    lib.amhip_mosaic_set_stream.argtypes = [vp, vp]
    lib.amhip_mosaic_synchronize.argtypes = [vp]
    lib.amhip_mosaic_reset.argtypes = [vp]
    lib.amhip_mosaic_batch.argtypes = [vp, f64p, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t),
                                       C.c_int, vp, vp]
    lib.amhip_mosaic_batch_dev.argtypes = [vp, f64p, C.c_size_t, vp, C.c_size_t, C.c_size_t,
                                           C.c_int]
    lib.amhip_mosaic_update.argtypes = [vp, f64p, vp, C.c_size_t, C.c_int, vp, vp]
    lib.amhip_mosaic_update_dev.argtypes = [vp, f64p, vp, C.c_size_t, C.c_int]
    lib.amhip_mosaic_download.argtypes = [vp, vp, vp]
    lib.amhip_mosaic_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.amhip_mosaic_homography.argtypes = [mp, cp, f64p, C.c_int, f64p]
    lib.amhip_camera_view_bounds.argtypes = [cp, f64p]
    lib.amhip_io_parse_point_cloud_text.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(vp),
                                                    C.POINTER(vp), C.POINTER(C.c_size_t),
                                                    C.POINTER(C.c_size_t)]
    lib.amhip_io_download_point_cloud.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.amhip_io_free.argtypes = [vp]
    u8p = C.POINTER(C.c_uint8)
    lib.amhip_layer_to_image_dev.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float, vp, C.c_size_t]
    lib.amhip_layer_to_image.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float, vp, C.c_size_t]
    lib.amhip_geotiff_write_u8.argtypes = [C.c_char_p, vp, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                           f64p, C.c_int, C.c_int]
    lib.amhip_grid_map_msg_bytes.restype = C.c_size_t
    lib.amhip_grid_map_msg_bytes.argtypes = [gp, C.c_char_p, C.c_int, C.POINTER(C.c_char_p)]
    lib.amhip_grid_map_msg_layout.argtypes = [gp, C.c_uint64, C.c_char_p, C.c_int,
                                              C.POINTER(C.c_char_p), vp, C.c_size_t,
                                              C.POINTER(C.c_size_t)]
    lib.amhip_io_write_point_cloud_binary.argtypes = [C.c_char_p, vp, vp, C.c_size_t]
    lib.amhip_io_load_point_cloud_binary.argtypes = [C.c_int, C.c_char_p, C.POINTER(vp),
                                                     C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.amhip_session_grid_map_msg.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_int,
                                               C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                                               C.POINTER(vp), vp, C.c_size_t,
                                               C.POINTER(C.c_size_t)]
    lib.amhip_session_layer_to_image.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float, vp,
                                                 C.c_size_t]
    del u8p
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise ImportError("libaerial_mapper_hip.so lacks %s (stale build?)" % missing)
    if lib.amhip_abi_version() != ABI_VERSION:
        raise ImportError("libaerial_mapper_hip.so ABI %d != expected %d"
                          % (lib.amhip_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def set_tuning(key, value=1.0):
    """amhip_set_tuning: a process-wide knob that selects among correct implementations (None clears)."""
    check(load().amhip_set_tuning(key.encode(), float("nan") if value is None else float(value)))


def tuning_env(**knobs):
    """{"AMHIP_TUNING": "key=value,..."} for a child process's environment"""
    return {"AMHIP_TUNING": ",".join("%s=%s" % (k, v) for k, v in knobs.items())}


def build_id():
    """amhip_build_id(): SHA-256 prefix over the loaded library's sources, headers and flags."""
    return load().amhip_build_id().decode()


def last_error():
    return load().amhip_last_error().decode("utf-8", "replace")


def check(status):
    if status != OK:
        raise AmhipError(status, last_error())
    return status


def make_grid(length_x, length_y, resolution, pos_x=0.0, pos_y=0.0):
    """grid_map_core setGeometry as called by AerialGridMap::initialize
    (Length(delta_easting, delta_northing), resolution, Position(center_easting,
    center_northing))."""
    g = GridDesc()
    load().amhip_make_grid(length_x, length_y, resolution, pos_x, pos_y, C.byref(g))
    return g


def cell_position(g, i, j):
    x, y = C.c_double(), C.c_double()
    load().amhip_cell_position(C.byref(g), i, j, C.byref(x), C.byref(y))
    return x.value, y.value
