"""Host-side mirror of the reference's class API for the hot path, over the C ABI.

  grid_map::AerialGridMap   aerial_mapper_grid_map/include/aerial-mapper-grid-map/
                            aerial-mapper-grid-map.h:23-54
  dsm::Settings / dsm::Dsm  aerial_mapper_dsm/include/aerial-mapper-dsm/dsm.h:25-42
  ortho::Settings / ortho::OrthoBackwardGrid
                            aerial_mapper_ortho/include/aerial-mapper-ortho/
                            ortho-backward-grid.h:32-50

Same names, argument meaning and error behaviour (a failed reference CHECK
surfaces as AmhipError instead of abort()).  The C++ drop-in classes live under
include/aerial-mapper-dsm/ and include/aerial-mapper-ortho/; this Python layer
exists so the parity tests and bench.py can drive the very same C ABI.

The layers stay resident in HBM between calls (what the reference keeps in the
GridMap's matrices); `AerialGridMap.get()` downloads one for publishing.
"""
import ctypes as C

import numpy as np

from . import hip_lib as L


class GridMapSettings(object):
    """grid_map::Settings (aerial-mapper-grid-map.h:23-29)."""

    def __init__(self, center_easting=0.0, center_northing=0.0, delta_easting=0.0,
                 delta_northing=0.0, resolution=1.0):
        self.center_easting = center_easting
        self.center_northing = center_northing
        self.delta_easting = delta_easting
        self.delta_northing = delta_northing
        self.resolution = resolution


class AerialGridMap(object):
    """grid_map::AerialGridMap: geometry + the layers, device resident."""

    def __init__(self, settings, device=0, window=None):
        """window = (i0, j0, rows, cols): own only that part of the map (one tile
        of a survey spread over several GPUs); positions and results are those of
        the full map, the layers hold just the window."""
        self.settings = settings
        lib = L.load()
        self.grid = L.make_grid(settings.delta_easting, settings.delta_northing,
                                settings.resolution, settings.center_easting,
                                settings.center_northing)
        if window is None:
            window = (0, 0, self.grid.rows, self.grid.cols)
        self.window = tuple(int(v) for v in window)
        h = C.c_void_p()
        L.check(lib.amhip_ctx_create_window(C.byref(self.grid), self.window[0], self.window[1],
                                            self.window[2], self.window[3], int(device),
                                            C.byref(h)))
        self._h = h
        self._lib = lib
        self.device = int(device)

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.amhip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- geometry ---------------------------------------------------------
    @property
    def rows(self):
        return self.window[2]

    @property
    def cols(self):
        return self.window[3]

    @property
    def num_cells(self):
        return self.window[2] * self.window[3]

    @property
    def handle(self):
        return self._h

    # -- layers -----------------------------------------------------------
    @staticmethod
    def _layer_id(layer):
        return L.LAYER_NAMES.index(layer) if isinstance(layer, str) else int(layer)

    def get(self, layer, out=None):
        """Download a layer: float32 array of shape (cols, rows) -- numpy C order
        of an Eigen column-major (rows, cols) matrix, so a[j, i] == layer(i, j).
        `out`: an existing array of that shape to fill (like the GridMap's own
        matrices in the C++ shim) instead of a fresh allocation."""
        if out is None:
            out = np.empty((self.cols, self.rows), np.float32)
        assert out.dtype == np.float32 and out.shape == (self.cols, self.rows) and out.flags.c_contiguous
        L.check(self._lib.amhip_layer_download(self._h, self._layer_id(layer),
                                               out.ctypes.data))
        return out

    def host_mirror(self, layer):
        """The layer's persistent host-side matrix (the GridMap's own matrix in the
        C++ shim), brought up to date with the device copy if something may have
        changed that since the last synchronisation."""
        if not hasattr(self, "_mirrors"):
            self._mirrors, self._mirror_fresh = {}, set()
        buf = self._mirrors.get(layer)
        if buf is None:
            buf = self._mirrors[layer] = np.zeros((self.cols, self.rows), np.float32)
            buf.fill(0.0)  # touch the pages once
        if layer not in self._mirror_fresh:
            self.get(layer, out=buf)
            self._mirror_fresh.add(layer)
        return buf

    def _touched(self, *layers):
        """The device copy of these layers (all, if none is named) may have changed."""
        if hasattr(self, "_mirror_fresh"):
            if layers:
                self._mirror_fresh.difference_update(layers)
            else:
                self._mirror_fresh.clear()

    def set(self, layer, values):
        self._touched(layer)
        a = np.ascontiguousarray(values, np.float32)
        assert a.shape == (self.cols, self.rows), a.shape
        L.check(self._lib.amhip_layer_upload(self._h, self._layer_id(layer), a.ctypes.data))

    def device_ptr(self, layer):
        self._touched(layer)
        self._external = getattr(self, "_external", set()) | {layer}
        return self._lib.amhip_layer_device_ptr(self._h, self._layer_id(layer))

    def as_torch(self, layer):
        """Zero-copy torch view (cols, rows) of a device-resident layer.  The view lives on the
        map's memory: torch work on it (a clone, a reduction) runs on TORCH's stream and must have
        finished -- torch.cuda.synchronize() / set_stream(torch's stream) -- before the next call
        on the map writes the layer (reset(), process()); the map's stream does not wait for
        torch's readers."""
        import torch
        ptr = self.device_ptr(layer)
        n = self.num_cells

        class _Holder(object):
            pass

        holder = _Holder()
        holder.__cuda_array_interface__ = {
            "shape": (self.cols, self.rows), "typestr": "<f4",
            "data": (int(ptr), False), "version": 2}
        holder._keepalive = self
        assert n > 0
        return torch.as_tensor(holder, device="cuda:%d" % self.device)

    def reset(self):
        """AerialGridMap::initialize() constants."""
        self._touched()
        L.check(self._lib.amhip_layers_reset(self._h))

    # -- stream / sync / timing ------------------------------------------
    def set_stream(self, stream_handle):
        """Run this context's kernels on an existing HIP stream (e.g.
        torch.cuda.Stream().cuda_stream); None/0 = the context's own stream."""
        L.check(self._lib.amhip_ctx_set_stream(self._h, C.c_void_p(stream_handle or 0)))
        self._stream_handle = int(stream_handle or 0)

    def wait_for_torch(self, tensor):
        """Device inputs made by torch on ANOTHER stream than this context's must be
        complete before the context's kernels read them."""
        import torch
        cur = torch.cuda.current_stream(tensor.device)
        if int(cur.cuda_stream) == 0 or int(cur.cuda_stream) != getattr(self, "_stream_handle", 0):
            cur.synchronize()

    def torch_waits(self):
        """The reverse of wait_for_torch: what this context's kernels wrote must be
        complete before torch work on ANOTHER stream reads it.

        Only torch's CURRENT stream (on this map's device) is ordered behind the call, and the host
        does not wait: a device-side CHECK (exact hit, alpha <= 0) of a sync=False call is raised by the
        next synchronize(), not here; an input tensor that is freed or refilled from a THIRD stream
        before the call has run still races with it -- keep it alive (or record_stream() it) until
        synchronize()."""
        import torch
        cur = torch.cuda.current_stream(torch.device("cuda", self.device))
        # (handle 0 = the map runs on its OWN stream, which is never torch's: always order)
        if int(cur.cuda_stream) == 0 or int(cur.cuda_stream) != getattr(self, "_stream_handle", 0):
            # an event on the map's stream that torch's current stream waits for: the host does
            # not wait (sync=False stays asynchronous whatever the stream arrangement)
            L.check(self._lib.amhip_ctx_order_after(self._h, C.c_void_p(int(cur.cuda_stream))))

    def synchronize(self):
        """Waits for the GPU and raises if a device-side CHECK fired."""
        L.check(self._lib.amhip_ctx_synchronize(self._h))

    def set_dsm_precision(self, exact):
        """AMHIP_DSM_EXACT (True, the default of new maps): the FP64 gather everywhere --
        reference-identical floats, deterministic; AMHIP_DSM_FAST (False, opt-in): single
        precision under exact guards, heights within 1e-4 m (include/aerial_mapper_hip.h)."""
        L.check(self._lib.amhip_ctx_set_dsm_precision(self._h, 1 if exact else 0))

    def set_dsm_knn(self, k):
        """OPTIONAL capped mode: only the k nearest points of a cell's search take part (0 = off,
        the reference's behaviour; include/aerial_mapper_hip.h)."""
        L.check(self._lib.amhip_ctx_set_dsm_knn(self._h, int(k)))

    def enable_timing(self, on=True):
        L.check(self._lib.amhip_ctx_enable_timing(self._h, int(bool(on))))

    def timing_reset(self):
        L.check(self._lib.amhip_ctx_timing_reset(self._h))

    def kernel_times(self):
        """{kernel name: (total_ms, launches)} since the last timing_reset()."""
        out = {}
        for k in range(L.NUM_KERNELS):
            ms, n = C.c_double(), C.c_int64()
            L.check(self._lib.amhip_ctx_kernel_time(self._h, k, C.byref(ms), C.byref(n)))
            out[self._lib.amhip_kernel_name(k).decode()] = (ms.value, n.value)
        return out

    def dsm_stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int32()
        L.check(self._lib.amhip_ctx_dsm_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"points_binned": a.value, "num_bins": b.value, "bin_cells": c.value}

    def dsm_gather_stats(self):
        """amhip_ctx_dsm_gather_stats: where the gather tiles of the last DSM call went."""
        out = (C.c_int64 * 8)()
        L.check(self._lib.amhip_ctx_dsm_gather_stats(self._h, out))
        v = [int(x) for x in out]
        return {"tiles": v[7], "sparse_list": v[0], "class1": v[1], "class2": v[2],
                "beyond_lds": v[3], "f32_to_fp64": v[4] + v[5], "f32_to_fp64_beyond": v[6]}


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class DsmSettings(object):
    """dsm::Settings (dsm.h:25-32).  interpolation_radius is an int holding the
    SQUARED search radius (it is handed to nanoflann's RadiusResultSet as is)."""

    def __init__(self, interpolation_radius=1, adaptive_interpolation=False,
                 center_easting=0.0, center_northing=0.0, use_multi_threads=True):
        self.interpolation_radius = int(interpolation_radius)
        self.adaptive_interpolation = adaptive_interpolation  # unused by the reference too
        self.center_easting = center_easting
        self.center_northing = center_northing
        self.use_multi_threads = use_multi_threads  # both reference variants give one result


class Dsm(object):
    """dsm::Dsm."""

    def __init__(self, settings, map):
        if map is None:
            raise L.AmhipError(L.ERR_ARG, "CHECK(map) (dsm.cc:22)")
        self.settings = settings

    def process(self, point_cloud, map, sync=True):
        """point_cloud: (N,3) float64 numpy array (host path, like the reference)
        or a CUDA torch tensor (device-resident path).  Updates map's
        'elevation' layer.

        sync=False (device path): returns with the kernels enqueued on the map's stream.  The
        tensor must stay unchanged until they have run: work enqueued on the SAME stream
        afterwards is ordered automatically (map.set_stream(torch's stream)); with any other
        stream arrangement torch's CURRENT stream is made to wait for the call on the device
        (amhip_ctx_order_after: an event, no host wait) -- work on a third stream is the
        caller's to order."""
        if map is None:
            raise L.AmhipError(L.ERR_ARG, "CHECK(map) (dsm.cc:194)")
        s = self.settings
        lib = L.load()
        if _is_torch(point_cloud):
            assert point_cloud.is_cuda and point_cloud.dtype.is_floating_point
            assert point_cloud.element_size() == 8 and point_cloud.is_contiguous()
            n = point_cloud.numel() // 3
            map.wait_for_torch(point_cloud)
            map._touched("elevation")
            L.check(lib.amhip_dsm_process_dev(
                map.handle, C.c_void_p(point_cloud.data_ptr()), n,
                s.interpolation_radius, s.center_easting, s.center_northing))
            if sync:
                map.synchronize()
            else:
                # LIFETIME: until the call's last gather kernel has run, the library reads
                # `point_cloud` itself (single-precision mode: the FP64 redo routines fetch the
                # doubles from the caller's cloud through the records' row indices).  The call is
                # ordered on the context's stream; when that is torch's current stream, later
                # torch work on the tensor is ordered behind it.  On any OTHER stream torch
                # knows nothing of these kernels: its current stream is ordered behind them.
                map.torch_waits()
            return
        pts = np.ascontiguousarray(point_cloud, np.float64).reshape(-1, 3)
        if pts.shape[0] == 0:
            return  # "Passed empty point cloud to DSM module" (dsm.cc:189-192)
        # The host-buffer entry point works on the caller's elevation matrix like
        # the C++ drop-in does on the GridMap's: uploaded, updated, downloaded.
        # The matrix is the map's persistent host mirror (allocated and touched
        # once -- a fresh 400 MB array per call would cost more in page faults
        # than the transfers).
        if "elevation" in getattr(map, "_external", ()):
            map._touched("elevation")  # someone holds the device pointer: always refresh
        elev = map.host_mirror("elevation")
        L.check(lib.amhip_dsm_process(map.handle, pts.ctypes.data, pts.shape[0],
                                      s.interpolation_radius, s.center_easting,
                                      s.center_northing, elev.ctypes.data))
        # `elev` is what the reference would have left in the host GridMap; the
        # device copy is identical.


class OrthoSettings(object):
    """ortho::Settings (ortho-backward-grid.h:32-41); only colored_ortho and
    use_multi_threads influence the result."""

    def __init__(self, show_orthomosaic_opencv=True, save_orthomosaic_jpg=True,
                 orthomosaic_jpg_filename="", orthomosaic_elevation_m=0.0,
                 use_digital_elevation_map=True, colored_ortho=False,
                 use_multi_threads=True):
        self.show_orthomosaic_opencv = show_orthomosaic_opencv
        self.save_orthomosaic_jpg = save_orthomosaic_jpg
        self.orthomosaic_jpg_filename = orthomosaic_jpg_filename
        self.orthomosaic_elevation_m = orthomosaic_elevation_m
        self.use_digital_elevation_map = use_digital_elevation_map
        self.colored_ortho = colored_ortho
        self.use_multi_threads = use_multi_threads


class NCamera(object):
    """What the hot path reads from aslam::NCamera: camera 0's pinhole model
    and T_C_B(0) (7 doubles tx,ty,tz,qw,qx,qy,qz)."""

    def __init__(self, fu, fv, cu, cv, width, height, distortion=L.DIST_NONE,
                 dist=(0.0, 0.0, 0.0, 0.0), T_C_B=(0, 0, 0, 1, 0, 0, 0)):
        self.camera = L.Camera()
        self.camera.fu, self.camera.fv, self.camera.cu, self.camera.cv = fu, fv, cu, cv
        self.camera.width, self.camera.height = int(width), int(height)
        self.camera.distortion = int(distortion)
        for k in range(4):
            self.camera.dist[k] = float(dist[k])
        self.T_C_B = np.asarray(T_C_B, np.float64).reshape(7).copy()


def compose_T_G_C(T_G_Bs, T_C_B):
    """T_G_C[i] = T_G_B[i] * T_C_B^-1 (ortho-backward-grid.cc:230-233)."""
    T = np.ascontiguousarray(T_G_Bs, np.float64).reshape(-1, 7)
    tcb = np.ascontiguousarray(T_C_B, np.float64).reshape(7)
    out = np.empty_like(T)
    f64p = C.POINTER(C.c_double)
    L.load().amhip_compose_T_G_C(T.ctypes.data_as(f64p), tcb.ctypes.data_as(f64p),
                                 T.shape[0], out.ctypes.data_as(f64p))
    return out


class OrthoBackwardGrid(object):
    """ortho::OrthoBackwardGrid."""

    def __init__(self, ncameras, settings, map=None):
        if ncameras is None:
            raise L.AmhipError(L.ERR_ARG, "CHECK(ncameras_) (ortho-backward-grid.cc:27)")
        if settings.use_multi_threads and map is None:
            # the reference dereferences the null map here (ortho-backward-grid.cc:34)
            raise L.AmhipError(L.ERR_ARG, "map must not be null")
        self.ncameras = ncameras
        self.settings = settings

    def process(self, T_G_Bs, images, map, sync=True):
        """T_G_Bs: (F,7).  images: list of numpy uint8 rasters ((H,W) gray or
        (H,W,3) BGR) for the host path, or one CUDA torch uint8 tensor
        (F,H,W[,3]) for the device-resident path."""
        T_G_Bs = np.ascontiguousarray(T_G_Bs, np.float64).reshape(-1, 7)
        F = T_G_Bs.shape[0]
        if F == 0:
            raise L.AmhipError(L.ERR_ARG, "CHECK(!T_G_Bs.empty())")
        if len(images) != F:
            raise L.AmhipError(L.ERR_ARG, "CHECK(T_G_Bs.size() == images.size())")
        if map is None:
            raise L.AmhipError(L.ERR_ARG, "CHECK(map)")
        lib = L.load()
        map._touched("elevation_angle", "observation_index", "num_observations", "ortho",
                     "colored_ortho")
        cam = self.ncameras.camera
        colored = bool(self.settings.colored_ortho)
        T_G_C = compose_T_G_C(T_G_Bs, self.ncameras.T_C_B)
        f64p = C.POINTER(C.c_double)
        if _is_torch(images):
            assert images.is_cuda and images.is_contiguous() and images.element_size() == 1
            ch = 3 if images.dim() == 4 else 1
            row = cam.width * ch
            frame = row * cam.height
            assert images.shape[1] == cam.height and images.shape[2] == cam.width
            map.wait_for_torch(images)
            L.check(lib.amhip_ortho_backward_process_dev(
                map.handle, C.byref(cam), T_G_C.ctypes.data_as(f64p), F,
                C.c_void_p(images.data_ptr()), frame, row, ch, int(colored)))
            if sync:
                map.synchronize()
            return
        ch = 3 if colored else 1
        ptrs = (C.c_void_p * F)()
        steps = (C.c_size_t * F)()
        keep = []
        for k, im in enumerate(images):
            im = np.asarray(im)
            if im.dtype != np.uint8 or im.strides[-1] != 1 or \
                    (im.ndim == 3 and (im.shape[2] != 3 or im.strides[1] != 3)):
                im = np.ascontiguousarray(im, np.uint8)
            if (im.ndim == 3) != colored:
                raise L.AmhipError(L.ERR_ARG, "colored_ortho needs 8UC3 frames, gray needs 8UC1")
            keep.append(im)
            ptrs[k] = im.ctypes.data
            steps[k] = im.strides[0]
        # layers are already device resident: pass NULL for all of them
        L.check(lib.amhip_ortho_backward_process(
            map.handle, C.byref(cam), T_G_C.ctypes.data_as(f64p), F, ptrs, steps, ch,
            int(colored), None, None, None, None, None, None))


LAYER_INIT = {"ortho": 255.0, "elevation": float("nan"), "elevation_angle": 0.0,
              "num_observations": 0.0, "observation_index": float("nan"),
              "colored_ortho": float("nan")}


class HostSession(object):
    """amhip_session: the entry points the C++ drop-in classes use, on HOST matrices.

    `layers[name]` are numpy float32 arrays of shape (cols, rows) -- the memory of the GridMap's
    column-major Eigen matrices -- initialised like AerialGridMap::initialize()
    (aerial-mapper-grid-map.cc:40-48).  dsm_process / ortho_process read and write them like
    dsm::Dsm::process / ortho::OrthoBackwardGrid::process do; between calls the caller may
    change them freely (the session notices by content).  tiles = (tiles_i, tiles_j) windows on
    `devices` (one entry per window; default: all on device 0)."""

    def __init__(self, settings, tiles=(1, 1), devices=None):
        lib = L.load()
        self._lib = lib
        self.settings = settings
        self.grid = L.make_grid(settings.delta_easting, settings.delta_northing,
                                settings.resolution, settings.center_easting,
                                settings.center_northing)
        nw = int(tiles[0]) * int(tiles[1])
        devs = (C.c_int32 * nw)(*([0] * nw if devices is None else [int(d) for d in devices]))
        h = C.c_void_p()
        L.check(lib.amhip_session_create(C.byref(self.grid), int(tiles[0]), int(tiles[1]), devs,
                                         C.byref(h)))
        self._h = h
        shape = (self.grid.cols, self.grid.rows)
        self.layers = {n: np.full(shape, v, np.float32) for n, v in LAYER_INIT.items()}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.amhip_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def num_windows(self):
        return int(self._lib.amhip_session_num_windows(self._h))

    def window(self, k):
        w = (C.c_int32 * 4)()
        L.check(self._lib.amhip_session_window(self._h, int(k), w))
        return tuple(int(v) for v in w)

    def set_always_copy(self, on):
        L.check(self._lib.amhip_session_set_always_copy(self._h, int(bool(on))))

    def set_dsm_precision(self, exact):
        L.check(self._lib.amhip_session_set_dsm_precision(self._h, 1 if exact else 0))

    def transfer_stats(self):
        """(bytes of layer data uploaded, downloaded) since the session was created."""
        up, down = C.c_uint64(), C.c_uint64()
        L.check(self._lib.amhip_session_transfer_stats(self._h, C.byref(up), C.byref(down)))
        return int(up.value), int(down.value)

    def last_profile(self):
        """amhip_session_last_profile: where the last host-buffer call spent its time (ms / bytes)."""
        out = (C.c_double * 8)()
        L.check(self._lib.amhip_session_last_profile(self._h, out))
        keys = ("total_ms", "h2d_ms", "host_sum_ms", "kernel_ms", "dev_sum_wait_ms", "d2h_ms", "bytes_up",
                "bytes_down")
        return {k: (round(float(v), 3) if k.endswith("_ms") else int(v)) for k, v in zip(keys, out)}

    def dsm_process(self, dsm_settings, points):
        pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        s = dsm_settings
        L.check(self._lib.amhip_session_dsm_process(
            self._h, pts.ctypes.data, pts.shape[0], s.interpolation_radius, s.center_easting,
            s.center_northing, self.layers["elevation"].ctypes.data))

    def ortho_from_pcl_process(self, settings, points, intensities):
        pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        inten = np.ascontiguousarray(intensities, np.int32).reshape(-1)
        assert inten.shape[0] == pts.shape[0]
        L.check(self._lib.amhip_session_ortho_from_pcl_process(
            self._h, pts.ctypes.data, inten.ctypes.data, pts.shape[0], settings.interpolation_radius,
            int(bool(settings.use_adaptive_interpolation)), self.layers["ortho"].ctypes.data))

    def ortho_process(self, ncameras, ortho_settings, T_G_Bs, images):
        T_G_Bs = np.ascontiguousarray(T_G_Bs, np.float64).reshape(-1, 7)
        F = T_G_Bs.shape[0]
        colored = bool(ortho_settings.colored_ortho)
        T_G_C = compose_T_G_C(T_G_Bs, ncameras.T_C_B)
        ptrs = (C.c_void_p * F)()
        steps = (C.c_size_t * F)()
        keep = []
        for k, im in enumerate(images):
            im = np.asarray(im)
            if im.dtype != np.uint8 or im.strides[-1] != 1 or \
                    (im.ndim == 3 and (im.shape[2] != 3 or im.strides[1] != 3)):
                im = np.ascontiguousarray(im, np.uint8)
            keep.append(im)
            ptrs[k] = im.ctypes.data
            steps[k] = im.strides[0]
        lay = self.layers
        L.check(self._lib.amhip_session_ortho_backward_process(
            self._h, C.byref(ncameras.camera), T_G_C.ctypes.data_as(C.POINTER(C.c_double)), F, ptrs,
            steps, 3 if colored else 1, int(colored), lay["elevation"].ctypes.data,
            lay["elevation_angle"].ctypes.data, lay["observation_index"].ctypes.data,
            lay["num_observations"].ctypes.data, lay["ortho"].ctypes.data,
            lay["colored_ortho"].ctypes.data))


class OrthoFromPclSettings(object):
    """ortho::Settings of ortho-from-pcl.h:28-35."""

    def __init__(self, show_orthomosaic_opencv=False, interpolation_radius=2,
                 use_adaptive_interpolation=False, save_orthomosaic_jpg=False,
                 orthomosaic_jpg_filename=""):
        self.show_orthomosaic_opencv = show_orthomosaic_opencv
        self.interpolation_radius = int(interpolation_radius)
        self.use_adaptive_interpolation = use_adaptive_interpolation
        self.save_orthomosaic_jpg = save_orthomosaic_jpg
        self.orthomosaic_jpg_filename = orthomosaic_jpg_filename


class OrthoFromPcl(object):
    """ortho::OrthoFromPcl (ortho-from-pcl.h:37-52): intensity of the cloud's
    points interpolated into the 'ortho' layer."""

    def __init__(self, settings):
        self.settings = settings

    def process(self, pointcloud, intensities, map, sync=True):
        """pointcloud (N,3) float64 + intensities (N,) int32: numpy (host path) or
        CUDA torch tensors (device-resident path)."""
        if map is None:
            raise L.AmhipError(L.ERR_ARG, "CHECK(map)")
        lib = L.load()
        s = self.settings
        if _is_torch(pointcloud):
            assert pointcloud.is_cuda and pointcloud.is_contiguous() and pointcloud.element_size() == 8
            assert intensities.is_cuda and intensities.is_contiguous() and \
                intensities.element_size() == 4 and not intensities.dtype.is_floating_point
            n = pointcloud.numel() // 3
            if n == 0 or intensities.numel() < n:
                raise L.AmhipError(L.ERR_ARG, "CHECK(!pointcloud.empty()) / CHECK(i < intensities.size())")
            map.wait_for_torch(pointcloud)
            map._touched("ortho")
            L.check(lib.amhip_ortho_from_pcl_process_dev(
                map.handle, C.c_void_p(pointcloud.data_ptr()), C.c_void_p(intensities.data_ptr()),
                n, s.interpolation_radius, int(bool(s.use_adaptive_interpolation))))
            if sync:
                map.synchronize()
            return
        pts = np.ascontiguousarray(pointcloud, np.float64).reshape(-1, 3)
        inten = np.ascontiguousarray(intensities, np.int32).reshape(-1)
        if pts.shape[0] == 0 or inten.shape[0] < pts.shape[0]:
            raise L.AmhipError(L.ERR_ARG, "CHECK(!pointcloud.empty()) / CHECK(i < intensities.size())")
        if "ortho" in getattr(map, "_external", ()):
            map._touched("ortho")
        ortho = map.host_mirror("ortho")
        L.check(lib.amhip_ortho_from_pcl_process(
            map.handle, pts.ctypes.data, inten.ctypes.data, pts.shape[0],
            s.interpolation_radius, int(bool(s.use_adaptive_interpolation)), ortho.ctypes.data))


def densify(map, disparity, image_left, K, baseline, R_G_C, t_G_C1):
    """stereo::Densifier::computePointCloud's reprojection (densifier.cpp:48-107) on
    the GPU of `map`: CUDA torch tensors disparity (H,W) float32 and image_left
    (H,W) uint8 -> (points (n,3) float64, intensities (n,) int32), both CUDA
    tensors in raster order, ready for Dsm.process / OrthoFromPcl.process."""
    import torch
    assert disparity.is_cuda and disparity.dtype == torch.float32 and disparity.stride(1) == 1
    assert image_left.is_cuda and image_left.dtype == torch.uint8 and image_left.stride(1) == 1
    H, W = disparity.shape
    assert tuple(image_left.shape) == (H, W)
    f64p = C.POINTER(C.c_double)
    Kc = np.ascontiguousarray(K, np.float64).reshape(9)
    Rc = np.ascontiguousarray(R_G_C, np.float64).reshape(9)
    tc = np.ascontiguousarray(t_G_C1, np.float64).reshape(3)
    xyz = torch.empty((H * W, 3), dtype=torch.float64, device=disparity.device)
    inten = torch.empty(H * W, dtype=torch.int32, device=disparity.device)
    count = torch.zeros(1, dtype=torch.int64, device=disparity.device)
    map.wait_for_torch(disparity)
    L.check(L.load().amhip_densify_dev(
        map.handle, C.c_void_p(disparity.data_ptr()), disparity.stride(0) * 4,
        C.c_void_p(image_left.data_ptr()), image_left.stride(0), W, H, Kc.ctypes.data_as(f64p),
        float(baseline), Rc.ctypes.data_as(f64p), tc.ctypes.data_as(f64p),
        C.c_void_p(xyz.data_ptr()), C.c_void_p(inten.data_ptr()), H * W,
        C.c_void_p(count.data_ptr())))
    map.synchronize()
    n = int(count.item())
    return xyz[:n], inten[:n]


def rectify_stereo_pair(map, K, R_G_C1, R_G_C2, t_G_C1, t_G_C2, image_left, image_right,
                        want_maps=False):
    """stereo::Rectifier::rectifyStereoPair + computeMask (rectifier.cpp:34-128) on the GPU of
    `map`: image_left / image_right CUDA torch uint8 tensors (H, W).  Returns a dict: R_G_C
    (3,3) and baseline (host), image_left, image_right, mask (CUDA uint8 (H, W)) and, with
    want_maps, maps (4, H, W) float32 -- the inputs of the block matcher and of densify()."""
    import torch
    assert image_left.is_cuda and image_left.dtype == torch.uint8 and image_left.stride(1) == 1
    assert image_right.is_cuda and image_right.dtype == torch.uint8 and image_right.stride(1) == 1
    H, W = image_left.shape
    assert tuple(image_right.shape) == (H, W)
    f64p = C.POINTER(C.c_double)
    arr = lambda a, n: np.ascontiguousarray(a, np.float64).reshape(n)
    Kc, R1, R2, t1, t2 = arr(K, 9), arr(R_G_C1, 9), arr(R_G_C2, 9), arr(t_G_C1, 3), arr(t_G_C2, 3)
    dev = image_left.device
    out_l = torch.empty((H, W), dtype=torch.uint8, device=dev)
    out_r = torch.empty((H, W), dtype=torch.uint8, device=dev)
    mask = torch.empty((H, W), dtype=torch.uint8, device=dev)
    maps = torch.empty((4, H, W), dtype=torch.float32, device=dev) if want_maps else None
    R = np.zeros(9)
    b = C.c_double()
    map.wait_for_torch(image_left)
    L.check(L.load().amhip_rectify_stereo_pair_dev(
        map.handle, Kc.ctypes.data_as(f64p), R1.ctypes.data_as(f64p), R2.ctypes.data_as(f64p),
        t1.ctypes.data_as(f64p), t2.ctypes.data_as(f64p), W, H,
        C.c_void_p(image_left.data_ptr()), image_left.stride(0),
        C.c_void_p(image_right.data_ptr()), image_right.stride(0), R.ctypes.data_as(f64p),
        C.byref(b), C.c_void_p(maps.data_ptr()) if want_maps else None,
        C.c_void_p(out_l.data_ptr()), C.c_void_p(out_r.data_ptr()), C.c_void_p(mask.data_ptr())))
    map.synchronize()
    return {"R_G_C": R.reshape(3, 3), "baseline": b.value, "image_left": out_l,
            "image_right": out_r, "mask": mask, "maps": maps}


# ---------------------------------------------------------------------------
# ortho::OrthoForwardHomography
# ---------------------------------------------------------------------------
class OrthoForwardHomographySettings(object):
    """ortho::Settings of ortho-forward-homography.h:33-42."""

    def __init__(self, batch=True, ground_plane_elevation_m=414.0, width_mosaic_pixels=1000,
                 height_mosaic_pixels=1000, origin=(0.0, 0.0, 0.0), nframe_id="map",
                 filename_mosaic_output="/tmp/result.jpg"):
        self.batch = batch
        self.ground_plane_elevation_m = float(ground_plane_elevation_m)
        self.width_mosaic_pixels = int(width_mosaic_pixels)
        self.height_mosaic_pixels = int(height_mosaic_pixels)
        self.origin = tuple(float(v) for v in origin)
        self.nframe_id = nframe_id
        self.filename_mosaic_output = filename_mosaic_output


class OrthoForwardHomography(object):
    """ortho::OrthoForwardHomography (ortho-forward-homography.h:44-52): the
    mosaic (result_, CV_16SC3) and its mask live on the device; `result()`
    downloads them.  imshow / imwrite / ROS publishing are the caller's."""

    def __init__(self, ncameras, settings, device=0):
        if ncameras is None:
            raise L.AmhipError(L.ERR_ARG, "CHECK(ncameras_) (ortho-forward-homography.cc:26)")
        lib = L.load()
        self.ncameras = ncameras
        self.settings = settings
        self.desc = L.MosaicDesc()
        self.desc.width_mosaic_pixels = settings.width_mosaic_pixels
        self.desc.height_mosaic_pixels = settings.height_mosaic_pixels
        self.desc.ground_plane_elevation_m = settings.ground_plane_elevation_m
        for k in range(3):
            self.desc.origin[k] = settings.origin[k]
        self.handle = C.c_void_p()
        L.check(lib.amhip_mosaic_create(C.byref(self.desc), C.byref(ncameras.camera), int(device),
                                        C.byref(self.handle)))
        self._torch_stream = None

    def close(self):
        if getattr(self, "handle", None):
            L.load().amhip_mosaic_destroy(self.handle)
            self.handle = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream):
        L.check(L.load().amhip_mosaic_set_stream(self.handle, C.c_void_p(hip_stream or 0)))
        self._torch_stream = hip_stream

    def synchronize(self):
        L.check(L.load().amhip_mosaic_synchronize(self.handle))

    def reset(self):
        L.check(L.load().amhip_mosaic_reset(self.handle))

    def _wait_for_torch(self, tensor):
        import torch
        cur = torch.cuda.current_stream(tensor.device).cuda_stream
        if self._torch_stream != cur:
            torch.cuda.current_stream(tensor.device).synchronize()

    def homography(self, T_G_B, batch=True):
        """The image -> mosaic homography of one frame (3x3, float64)."""
        T_G_C = compose_T_G_C(np.asarray(T_G_B, np.float64).reshape(1, 7), self.ncameras.T_C_B)
        M = np.zeros(9)
        f64p = C.POINTER(C.c_double)
        L.check(L.load().amhip_mosaic_homography(C.byref(self.desc), C.byref(self.ncameras.camera),
                                                 T_G_C.ctypes.data_as(f64p), int(bool(batch)),
                                                 M.ctypes.data_as(f64p)))
        return M.reshape(3, 3)

    @staticmethod
    def _host_image(im):
        im = np.asarray(im)
        if im.dtype != np.uint8 or im.strides[-1] != 1 or \
                (im.ndim == 3 and (im.shape[2] != 3 or im.strides[1] != 3)):
            im = np.ascontiguousarray(im, np.uint8)
        return im

    def batch(self, T_G_Bs, images, sync=True):
        """OrthoForwardHomography::batch.  images: list of numpy uint8 rasters
        ((H,W) or (H,W,3)) or one CUDA torch uint8 tensor (F,H,W[,3])."""
        T_G_C = compose_T_G_C(T_G_Bs, self.ncameras.T_C_B)
        F = T_G_C.shape[0]
        lib = L.load()
        cam = self.ncameras.camera
        f64p = C.POINTER(C.c_double)
        if _is_torch(images):
            assert images.is_cuda and images.is_contiguous() and images.element_size() == 1
            ch = 3 if images.dim() == 4 else 1
            assert images.shape[0] >= F and images.shape[1] == cam.height and images.shape[2] == cam.width
            row = cam.width * ch
            self._wait_for_torch(images)
            L.check(lib.amhip_mosaic_batch_dev(self.handle, T_G_C.ctypes.data_as(f64p), F,
                                               C.c_void_p(images.data_ptr()), row * cam.height, row, ch))
            if sync:
                self.synchronize()
            return
        if len(images) != F:
            raise L.AmhipError(L.ERR_ARG, "poses and images differ in number")
        ims = [self._host_image(im) for im in images]
        ch = 3 if (F and ims[0].ndim == 3) else 1
        ptrs = (C.c_void_p * max(F, 1))()
        steps = (C.c_size_t * max(F, 1))()
        for k, im in enumerate(ims):
            if (3 if im.ndim == 3 else 1) != ch:
                raise L.AmhipError(L.ERR_ARG, "mixed gray / colour frames")
            ptrs[k] = im.ctypes.data
            steps[k] = im.strides[0]
        L.check(lib.amhip_mosaic_batch(self.handle, T_G_C.ctypes.data_as(f64p), F, ptrs, steps, ch,
                                       None, None))

    def updateOrthomosaic(self, T_G_B, image, sync=True):
        """OrthoForwardHomography::updateOrthomosaic (one frame, incremental)."""
        T_G_C = compose_T_G_C(np.asarray(T_G_B, np.float64).reshape(1, 7), self.ncameras.T_C_B)
        lib = L.load()
        cam = self.ncameras.camera
        f64p = C.POINTER(C.c_double)
        if _is_torch(image):
            assert image.is_cuda and image.is_contiguous() and image.element_size() == 1
            ch = 3 if image.dim() == 3 else 1
            self._wait_for_torch(image)
            L.check(lib.amhip_mosaic_update_dev(self.handle, T_G_C.ctypes.data_as(f64p),
                                                C.c_void_p(image.data_ptr()), cam.width * ch, ch))
            if sync:
                self.synchronize()
            return
        im = self._host_image(image)
        ch = 3 if im.ndim == 3 else 1
        L.check(lib.amhip_mosaic_update(self.handle, T_G_C.ctypes.data_as(f64p),
                                        C.c_void_p(im.ctypes.data), im.strides[0], ch, None, None))

    def result(self):
        """(result_ as (H,W,3) int16, result_mask_ as (H,W) uint8)."""
        h, w = self.desc.height_mosaic_pixels, self.desc.width_mosaic_pixels
        res = np.empty((h, w, 3), np.int16)
        mask = np.empty((h, w), np.uint8)
        L.check(L.load().amhip_mosaic_download(self.handle, C.c_void_p(res.ctypes.data),
                                               C.c_void_p(mask.ctypes.data)))
        return res, mask
