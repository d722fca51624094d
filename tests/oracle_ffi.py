"""ctypes binding of the CPU oracle (oracle/liboracle.so, oracle/_ref/liboracle_ref.so,
oracle/_ref/libref_loops_*.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package aerial_mapper_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
PORT_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "liboracle_ref.so")


class Grid(C.Structure):
    """amo_grid / amhip_grid_desc (identical layout)."""
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32),
                ("resolution", C.c_double),
                ("length_x", C.c_double), ("length_y", C.c_double),
                ("pos_x", C.c_double), ("pos_y", C.c_double)]


class Camera(C.Structure):
    """amo_camera / amhip_camera (identical layout)."""
    _fields_ = [("fu", C.c_double), ("fv", C.c_double),
                ("cu", C.c_double), ("cv", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32),
                ("distortion", C.c_int32), ("_pad", C.c_int32),
                ("dist", C.c_double * 4)]


class MosaicDesc(C.Structure):
    """amo_mosaic_desc / amhip_mosaic_desc (identical layout)."""
    _fields_ = [("width_mosaic_pixels", C.c_int32), ("height_mosaic_pixels", C.c_int32),
                ("ground_plane_elevation_m", C.c_double), ("origin", C.c_double * 3)]


DIST_NONE, DIST_RADTAN, DIST_EQUIDISTANT = 0, 1, 2
OK, ERR_ARG, ERR_EXACT_HIT, ERR_ALPHA_NONPOS = 0, 1, 2, 3


def build(force=False):
    """(Re)build the oracle libraries with oracle/Makefile (gcc only)."""
    if force or not os.path.exists(PORT_SO):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "all"],
                              stdout=subprocess.DEVNULL)


def _bind(lib):
    f64p = C.POINTER(C.c_double)
    f32p = C.POINTER(C.c_float)
    lib.amo_uses_vendored_nanoflann.restype = C.c_int
    lib.amo_dsm_process.restype = C.c_int
    lib.amo_dsm_process.argtypes = [
        f64p, C.c_size_t, C.POINTER(Grid), C.c_int, C.c_double, C.c_double,
        C.c_int, C.c_int, f32p, f64p]
    lib.amo_dsm_process_knn.restype = C.c_int
    lib.amo_dsm_process_knn.argtypes = [
        f64p, C.c_size_t, C.POINTER(Grid), C.c_int, C.c_double, C.c_double, C.c_int,
        C.c_int, C.c_int, f32p]
    lib.amo_dsm_radius_probe.restype = C.c_int
    lib.amo_dsm_radius_probe.argtypes = [
        f64p, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_int,
        C.POINTER(C.c_int), f64p]
    lib.amo_make_grid.restype = None
    lib.amo_make_grid.argtypes = [C.c_double] * 5 + [C.POINTER(Grid)]
    lib.amo_cell_position.restype = None
    lib.amo_cell_position.argtypes = [C.POINTER(Grid), C.c_int, C.c_int, f64p, f64p]
    lib.amo_ortho_backward_process.restype = C.c_int
    lib.amo_ortho_backward_process.argtypes = [
        C.POINTER(Grid), C.POINTER(Camera), f64p, f64p,
        C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_size_t,
        C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, f32p]
    lib.amo_ortho_from_pcl_process.restype = C.c_int
    lib.amo_ortho_from_pcl_process.argtypes = [f64p, C.POINTER(C.c_int32), C.c_size_t,
                                               C.POINTER(Grid), C.c_int, C.c_int, f32p]
    lib.amo_densify.restype = C.c_long
    lib.amo_densify.argtypes = [f32p, C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.c_int,
                                f64p, C.c_double, f64p, f64p, f64p, C.POINTER(C.c_int32)]
    lib.amo_fwd_homography.restype = C.c_int
    lib.amo_fwd_homography.argtypes = [C.POINTER(Camera), C.POINTER(MosaicDesc), f64p, C.c_int, f64p]
    lib.amo_fwd_distance_l1.restype = None
    lib.amo_fwd_distance_l1.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.amo_fwd_create.restype = C.c_void_p
    lib.amo_fwd_create.argtypes = [C.POINTER(Camera), C.POINTER(MosaicDesc)]
    lib.amo_fwd_destroy.restype = None
    lib.amo_fwd_destroy.argtypes = [C.c_void_p]
    lib.amo_fwd_batch.restype = C.c_int
    lib.amo_fwd_batch.argtypes = [C.c_void_p, f64p, f64p, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_size_t), C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.amo_fwd_update.restype = C.c_int
    lib.amo_fwd_update.argtypes = [C.c_void_p, f64p, f64p, C.c_void_p, C.c_size_t, C.c_int,
                                   C.c_void_p, C.c_void_p]
    lib.amo_io_load_point_cloud.restype = C.c_size_t
    lib.amo_io_load_point_cloud.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.amo_io_load_poses.restype = C.c_size_t
    lib.amo_io_load_poses.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.amo_compose_T_G_C.restype = None
    lib.amo_compose_T_G_C.argtypes = [f64p, f64p, C.c_size_t, f64p]
    lib.amo_project_probe.restype = None
    lib.amo_project_probe.argtypes = [C.POINTER(Camera), f64p, f64p, f64p]
    lib.amo_color_value_bgr.restype = C.c_float
    lib.amo_color_value_bgr.argtypes = [C.c_uint8] * 3
    return lib


_libs = {}


def lib(which="port"):
    """which: 'port' (own kd-tree) or 'ref' (vendored nanoflann build)."""
    if which not in _libs:
        path = PORT_SO if which == "port" else REF_SO
        if which == "port":
            build()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        _libs[which] = _bind(C.CDLL(path))
    return _libs[which]


def have_ref():
    return os.path.exists(REF_SO)


# ---- the reference's OWN loops (oracle/_ref/libref_loops_*.so: dsm.cc, ortho-backward-grid.cc,
# ortho-from-pcl.cc compiled unchanged against oracle/refkit/; `which="loops"` below) ----------
LOOPS_SO = {name: os.path.join(ORACLE_DIR, "_ref", "libref_loops_%s.so" % name)
            for name in ("dsm", "ortho_backward", "ortho_from_pcl", "grid_map", "densify", "forward", "rectify")}
_loops_libs = {}


def have_loops():
    return all(os.path.exists(p) for p in LOOPS_SO.values())


def _loops(name):
    if name not in _loops_libs:
        if not os.path.exists(LOOPS_SO[name]):
            raise FileNotFoundError(LOOPS_SO[name])
        f64p, f32p = C.POINTER(C.c_double), C.POINTER(C.c_float)
        so = C.CDLL(LOOPS_SO[name])
        if name == "dsm":
            so.amr_dsm_process.restype = C.c_int
            so.amr_dsm_process.argtypes = [f64p, C.c_size_t, C.POINTER(Grid), C.c_int, C.c_double,
                                           C.c_double, C.c_int, f32p, f64p]
        elif name == "ortho_backward":
            so.amr_ortho_backward_process.restype = C.c_int
            so.amr_ortho_backward_process.argtypes = [
                C.POINTER(Grid), C.POINTER(Camera), f64p, f64p, C.POINTER(C.c_void_p),
                C.POINTER(C.c_size_t), C.c_int, C.c_size_t, C.c_int, C.c_int,
                f32p, f32p, f32p, f32p, f32p, f32p, f64p]
        elif name == "forward":
            so.amr_fwd_create.restype = C.c_void_p
            so.amr_fwd_create.argtypes = [C.POINTER(Camera), C.POINTER(MosaicDesc), f64p]
            so.amr_fwd_destroy.restype = None
            so.amr_fwd_destroy.argtypes = [C.c_void_p]
            so.amr_fwd_batch.restype = C.c_int
            so.amr_fwd_batch.argtypes = [C.c_void_p, f64p, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_size_t), C.c_int, C.c_size_t, C.c_void_p]
            so.amr_fwd_update.restype = C.c_int
            so.amr_fwd_update.argtypes = [C.c_void_p, f64p, C.c_void_p, C.c_size_t, C.c_int,
                                          C.c_void_p]
        elif name == "densify":
            so.amr_densify.restype = C.c_long
            so.amr_densify.argtypes = [f32p, C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t, C.c_int,
                                       C.c_int, f64p, C.c_double, f64p, f64p, f64p,
                                       C.POINTER(C.c_int32)]
        elif name == "grid_map":
            so.amr_grid_map_initialize.restype = C.c_int
            so.amr_grid_map_initialize.argtypes = [C.c_double] * 5 + [C.POINTER(Grid),
                                                                      C.POINTER(C.c_void_p)]
        elif name == "rectify":
            pass  # (bound by rectify_stereo_pair)
        else:
            so.amr_ortho_from_pcl_process.restype = C.c_int
            so.amr_ortho_from_pcl_process.argtypes = [f64p, C.POINTER(C.c_int32), C.c_size_t,
                                                      C.POINTER(Grid), C.c_int, C.c_int, f32p]
        _loops_libs[name] = so
    return _loops_libs[name]


LAYER_ORDER = ["ortho", "elevation", "elevation_angle", "num_observations", "observation_index",
               "colored_ortho"]


def reference_grid_map(center_easting, center_northing, delta_easting, delta_northing, resolution):
    """The reference's own grid_map::AerialGridMap(settings) (aerial-mapper-grid-map.cc,
    compiled unchanged): returns (Grid, {layer: initial values})."""
    so = _loops("grid_map")
    g = Grid()
    args = (center_easting, center_northing, delta_easting, delta_northing, resolution)
    assert so.amr_grid_map_initialize(*args, C.byref(g), None) == OK
    layers = {n: np.empty((g.cols, g.rows), np.float32) for n in LAYER_ORDER}
    ptrs = (C.c_void_p * 6)(*[layers[n].ctypes.data for n in LAYER_ORDER])
    assert so.amr_grid_map_initialize(*args, C.byref(g), ptrs) == OK
    return g, layers


def _f64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_grid(length_x, length_y, resolution, pos_x=0.0, pos_y=0.0, which="port"):
    g = Grid()
    lib(which).amo_make_grid(length_x, length_y, resolution, pos_x, pos_y, C.byref(g))
    return g


def cell_position(g, i, j, which="port"):
    x, y = C.c_double(), C.c_double()
    lib(which).amo_cell_position(C.byref(g), i, j, C.byref(x), C.byref(y))
    return x.value, y.value


def new_layers(g):
    """The 6 layers the hot path touches, initialised like
    aerial-mapper-grid-map.cc:40-48 (column-major float32, shape (cols, rows)
    in numpy C-order == Eigen (rows, cols) column-major)."""
    shape = (g.cols, g.rows)
    return {
        "ortho": np.full(shape, 255.0, np.float32),
        "elevation": np.full(shape, np.nan, np.float32),
        "elevation_angle": np.zeros(shape, np.float32),
        "num_observations": np.zeros(shape, np.float32),
        "observation_index": np.full(shape, np.nan, np.float32),
        "colored_ortho": np.full(shape, np.nan, np.float32),
    }


def dsm_process(xyz, g, radius_sq=1, center_easting=0.0, center_northing=0.0,
                elevation=None, multi_thread=True, num_threads=0, which="port"):
    """Returns (rc, elevation, (t_build, t_cells))."""
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    if elevation is None:
        elevation = np.full((g.cols, g.rows), np.nan, np.float32)
    assert elevation.dtype == np.float32 and elevation.flags.c_contiguous
    t = np.zeros(2)
    if which == "loops":
        rc = _loops("dsm").amr_dsm_process(_f64(xyz), xyz.shape[0], C.byref(g), int(radius_sq),
                                           center_easting, center_northing,
                                           int(bool(multi_thread)), _f32(elevation), _f64(t))
        # (t[0] = the constructor's one-sample-per-cell table, t[1] = Dsm::process: kd-tree
        # build + cell loop together)
        return rc, elevation, (t[0], t[1])
    rc = lib(which).amo_dsm_process(
        _f64(xyz), xyz.shape[0], C.byref(g), int(radius_sq),
        center_easting, center_northing, int(bool(multi_thread)),
        int(num_threads), _f32(elevation), _f64(t))
    return rc, elevation, (t[0], t[1])


def dsm_process_knn(xyz, g, k, radius_sq=1, center_easting=0.0, center_northing=0.0,
                    elevation=None, multi_thread=True, which="port"):
    """The OPTIONAL capped mode (only the k nearest points of a cell's search result take part;
    not a reference code path).  Returns (rc, elevation)."""
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    if elevation is None:
        elevation = np.full((g.cols, g.rows), np.nan, np.float32)
    rc = lib(which).amo_dsm_process_knn(_f64(xyz), xyz.shape[0], C.byref(g), int(radius_sq),
                                        center_easting, center_northing, int(k),
                                        int(bool(multi_thread)), 0, _f32(elevation))
    return rc, elevation


def rectify_stereo_pair(K, R1, R2, t1, t2, left, right, which="port"):
    """stereo::Rectifier::rectifyStereoPair.  left / right: (H, W) uint8.  Returns (rc, dict with
    R_G_C (3,3), baseline, maps (4,H,W) float32, left, right, mask (H,W) uint8)."""
    f64 = lambda a, n: np.ascontiguousarray(a, np.float64).reshape(n)
    K, R1, R2, t1, t2 = f64(K, 9), f64(R1, 9), f64(R2, 9), f64(t1, 3), f64(t2, 3)
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    H, W = left.shape
    R = np.zeros(9)
    b = C.c_double()
    maps = np.zeros((4, H, W), np.float32)
    ol, orr, mask = (np.zeros((H, W), np.uint8) for _ in range(3))
    u8 = lambda a: a.ctypes.data_as(C.c_void_p)
    if which == "loops":
        fn = _loops("rectify").amr_rectify_stereo_pair
    else:
        fn = lib(which).amo_rectify_stereo_pair
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_double)] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                                 C.c_size_t, C.POINTER(C.c_double),
                                                 C.POINTER(C.c_double), C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
    rc = fn(_f64(K), _f64(R1), _f64(R2), _f64(t1), _f64(t2), W, H, u8(left), left.strides[0],
            u8(right), right.strides[0], _f64(R), C.byref(b), u8(maps), u8(ol), u8(orr), u8(mask))
    return rc, {"R_G_C": R.reshape(3, 3), "baseline": b.value, "maps": maps, "left": ol,
                "right": orr, "mask": mask}


def ortho_from_pcl(xyz, intensities, g, radius_sq=2, adaptive=False, ortho=None, which="port"):
    """ortho::OrthoFromPcl::process.  Returns (rc, ortho layer)."""
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    inten = np.ascontiguousarray(intensities, np.int32).reshape(-1)
    assert inten.shape[0] == xyz.shape[0]
    if ortho is None:
        ortho = np.full((g.cols, g.rows), 255.0, np.float32)
    if which == "loops":
        rc = _loops("ortho_from_pcl").amr_ortho_from_pcl_process(
            _f64(xyz), inten.ctypes.data_as(C.POINTER(C.c_int32)), xyz.shape[0], C.byref(g),
            int(radius_sq), int(bool(adaptive)), _f32(ortho))
        return rc, ortho
    rc = lib(which).amo_ortho_from_pcl_process(
        _f64(xyz), inten.ctypes.data_as(C.POINTER(C.c_int32)), xyz.shape[0], C.byref(g),
        int(radius_sq), int(bool(adaptive)), _f32(ortho))
    return rc, ortho


def densify(disparity, image_left, K, baseline, R_G_C, t_G_C1, which="port"):
    """stereo::Densifier::computePointCloud -> (points (n,3) f64, intensities (n,) i32)."""
    disp = np.ascontiguousarray(disparity, np.float32)
    img = np.ascontiguousarray(image_left, np.uint8)
    h, w = disp.shape
    assert img.shape == (h, w)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    R = np.ascontiguousarray(R_G_C, np.float64).reshape(9)
    t = np.ascontiguousarray(t_G_C1, np.float64).reshape(3)
    xyz = np.empty((h * w, 3), np.float64)
    inten = np.empty(h * w, np.int32)
    fn = _loops("densify").amr_densify if which == "loops" else lib(which).amo_densify
    n = fn(_f32(disp), disp.strides[0], img.ctypes.data_as(C.POINTER(C.c_uint8)),
                               img.strides[0], w, h, _f64(K), float(baseline), _f64(R), _f64(t),
                               _f64(xyz), inten.ctypes.data_as(C.POINTER(C.c_int32)))
    return xyz[:n].copy(), inten[:n].copy()


def radius_probe(xyz, qx, qy, radius_sq, cap=4096, which="port"):
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    idx = np.zeros(cap, np.int32)
    d2 = np.zeros(cap, np.float64)
    n = lib(which).amo_dsm_radius_probe(
        _f64(xyz), xyz.shape[0], qx, qy, radius_sq, cap,
        idx.ctypes.data_as(C.POINTER(C.c_int)), _f64(d2))
    k = min(n, cap)
    return n, idx[:k].copy(), d2[:k].copy()


def ortho_process(g, cam, T_G_B, T_C_B, images, layers, colored=False,
                  multi_thread=True, num_threads=0, which="port", timing=None):
    """images: list of uint8 arrays (H,W) or (H,W,3 BGR). layers updated in place."""
    T_G_B = np.ascontiguousarray(T_G_B, np.float64).reshape(-1, 7)
    T_C_B = np.ascontiguousarray(T_C_B, np.float64).reshape(7)
    F = T_G_B.shape[0]
    assert len(images) == F
    ch = 3 if colored else 1
    ptrs = (C.c_void_p * F)()
    steps = (C.c_size_t * F)()
    keep = []
    for k, im in enumerate(images):
        assert im.dtype == np.uint8
        assert im.strides[-1] == 1 and (im.ndim == 2 or im.strides[1] == 3)
        keep.append(im)
        ptrs[k] = im.ctypes.data
        steps[k] = im.strides[0]
    if which == "loops":
        return _loops("ortho_backward").amr_ortho_backward_process(
            C.byref(g), C.byref(cam), _f64(T_G_B), _f64(T_C_B), ptrs, steps, ch, F,
            int(bool(colored)), int(bool(multi_thread)),
            _f32(layers["elevation"]), _f32(layers["elevation_angle"]),
            _f32(layers["observation_index"]), _f32(layers["num_observations"]),
            _f32(layers["ortho"]), _f32(layers["colored_ortho"]),
            _f64(timing) if timing is not None else None)
    rc = lib(which).amo_ortho_backward_process(
        C.byref(g), C.byref(cam), _f64(T_G_B), _f64(T_C_B), ptrs, steps, ch, F,
        int(bool(colored)), int(bool(multi_thread)), int(num_threads),
        _f32(layers["elevation"]), _f32(layers["elevation_angle"]),
        _f32(layers["observation_index"]), _f32(layers["num_observations"]),
        _f32(layers["ortho"]), _f32(layers["colored_ortho"]))
    return rc


def compose_T_G_C(T_G_B, T_C_B, which="port"):
    T_G_B = np.ascontiguousarray(T_G_B, np.float64).reshape(-1, 7)
    T_C_B = np.ascontiguousarray(T_C_B, np.float64).reshape(7)
    out = np.zeros_like(T_G_B)
    lib(which).amo_compose_T_G_C(_f64(T_G_B), _f64(T_C_B), T_G_B.shape[0], _f64(out))
    return out


def project_probe(cam, T_G_C7, landmark, which="port"):
    T = np.ascontiguousarray(T_G_C7, np.float64).reshape(7)
    L = np.ascontiguousarray(landmark, np.float64).reshape(3)
    out = np.zeros(7)
    lib(which).amo_project_probe(C.byref(cam), _f64(T), _f64(L), _f64(out))
    return dict(u=out[0], v=out[1], alpha=out[2], status=int(out[3]), C=out[4:7].copy())


def color_value_bgr(b, g, r, which="port"):
    return np.float32(lib(which).amo_color_value_bgr(b, g, r))


def mosaic_desc(width, height, ground, origin=(0.0, 0.0, 0.0)):
    d = MosaicDesc()
    d.width_mosaic_pixels, d.height_mosaic_pixels = int(width), int(height)
    d.ground_plane_elevation_m = float(ground)
    for k in range(3):
        d.origin[k] = float(origin[k])
    return d


def fwd_homography(cam, desc, T_G_C7, batch_quirk=True, which="port"):
    T = np.ascontiguousarray(T_G_C7, np.float64).reshape(7)
    M = np.zeros(9)
    rc = lib(which).amo_fwd_homography(C.byref(cam), C.byref(desc), _f64(T), int(bool(batch_quirk)),
                                       _f64(M))
    return rc, M.reshape(3, 3)


def fwd_distance_l1(mask, which="port"):
    mask = np.ascontiguousarray(mask, np.uint8)
    out = np.empty(mask.shape, np.float32)
    lib(which).amo_fwd_distance_l1(C.c_void_p(mask.ctypes.data), mask.shape[1], mask.shape[0],
                                   C.c_void_p(out.ctypes.data))
    return out


class ForwardMosaic(object):
    """ortho::OrthoForwardHomography on the CPU oracle (stateful, like the class)."""

    def __init__(self, cam, desc, T_C_B=(0, 0, 0, 1, 0, 0, 0), which="port"):
        self.lib = lib(which)
        self.cam, self.desc = cam, desc
        self.T_C_B = np.ascontiguousarray(T_C_B, np.float64).reshape(7)
        self.h = self.lib.amo_fwd_create(C.byref(cam), C.byref(desc))
        assert self.h
        hh, ww = desc.height_mosaic_pixels, desc.width_mosaic_pixels
        self.result = np.zeros((hh, ww, 3), np.int16)
        self.mask = np.zeros((hh, ww), np.uint8)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.amo_fwd_destroy(self.h)
            self.h = None

    def batch(self, T_G_B, images):
        T_G_B = np.ascontiguousarray(T_G_B, np.float64).reshape(-1, 7)
        F = T_G_B.shape[0]
        assert len(images) == F
        ch = 3 if (F and images[0].ndim == 3) else 1
        ptrs = (C.c_void_p * max(F, 1))()
        steps = (C.c_size_t * max(F, 1))()
        for k, im in enumerate(images):
            assert im.dtype == np.uint8 and im.strides[-1] == 1
            ptrs[k] = im.ctypes.data
            steps[k] = im.strides[0]
        rc = self.lib.amo_fwd_batch(self.h, _f64(T_G_B), _f64(self.T_C_B), ptrs, steps, ch, F,
                                    C.c_void_p(self.result.ctypes.data),
                                    C.c_void_p(self.mask.ctypes.data))
        return rc

    def update(self, T_G_B7, image):
        T = np.ascontiguousarray(T_G_B7, np.float64).reshape(7)
        ch = 3 if image.ndim == 3 else 1
        assert image.dtype == np.uint8 and image.strides[-1] == 1
        return self.lib.amo_fwd_update(self.h, _f64(T), _f64(self.T_C_B),
                                       C.c_void_p(image.ctypes.data), image.strides[0], ch,
                                       C.c_void_p(self.result.ctypes.data),
                                       C.c_void_p(self.mask.ctypes.data))


class ReferenceForwardMosaic(object):
    """The reference's own ortho::OrthoForwardHomography (ortho-forward-homography.cc compiled
    unchanged against oracle/refkit/).  The class keeps its mosaic private: `result` is what
    it hands to cv::imwrite (no mask)."""

    def __init__(self, cam, desc, T_C_B=(0, 0, 0, 1, 0, 0, 0)):
        self.lib = _loops("forward")
        T = np.ascontiguousarray(T_C_B, np.float64).reshape(7)
        self.h = self.lib.amr_fwd_create(C.byref(cam), C.byref(desc), _f64(T))
        assert self.h
        self.result = np.zeros((desc.height_mosaic_pixels, desc.width_mosaic_pixels, 3), np.int16)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.amr_fwd_destroy(self.h)
            self.h = None

    def batch(self, T_G_B, images):
        T_G_B = np.ascontiguousarray(T_G_B, np.float64).reshape(-1, 7)
        F = T_G_B.shape[0]
        assert len(images) == F
        ch = 3 if (F and images[0].ndim == 3) else 1
        ptrs = (C.c_void_p * max(F, 1))()
        steps = (C.c_size_t * max(F, 1))()
        for k, im in enumerate(images):
            assert im.dtype == np.uint8 and im.strides[-1] == 1
            ptrs[k] = im.ctypes.data
            steps[k] = im.strides[0]
        return self.lib.amr_fwd_batch(self.h, _f64(T_G_B), ptrs, steps, ch, F,
                                      C.c_void_p(self.result.ctypes.data))

    def update(self, T_G_B7, image):
        T = np.ascontiguousarray(T_G_B7, np.float64).reshape(7)
        ch = 3 if image.ndim == 3 else 1
        assert image.dtype == np.uint8 and image.strides[-1] == 1
        return self.lib.amr_fwd_update(self.h, _f64(T), C.c_void_p(image.ctypes.data),
                                       image.strides[0], ch, C.c_void_p(self.result.ctypes.data))


def io_load_point_cloud(text, with_intensities=True, which="port"):
    """io::AerialMapperIO::loadPointCloudFromFile on an in-memory file (bytes)."""
    cap = text.count(b"\n") + text.count(b" ") // 3 + 8
    for _ in range(2):
        xyz = np.empty((cap, 3), np.float64)
        inten = np.empty(cap, np.int32)
        n = lib(which).amo_io_load_point_cloud(text, len(text), C.c_void_p(xyz.ctypes.data),
                                               C.c_void_p(inten.ctypes.data) if with_intensities else None,
                                               cap)
        if n <= cap:
            break
        cap = n            # (tokens that hold several numbers: "1-2-3-4" is a whole record)
    assert n <= cap
    return xyz[:n].copy(), inten[:n].copy()


def io_load_poses(text, which="port"):
    cap = len(text.split()) // 7 + 4
    out = np.empty((cap, 7), np.float64)
    n = lib(which).amo_io_load_poses(text, len(text), C.c_void_p(out.ctypes.data), cap)
    return out[:n].copy()
