"""What leaves the map: layer images in grid_map_cv's orientation, the
GeoTiff container of io::AerialMapperIO::toGeoTiff / writeDataToDEMGeoTiffColor
(aerial-mapper-io.cc:349-509), the grid_map_msgs/GridMap message of AerialGridMap::publishOnce
(aerial-mapper-grid-map.cc:66-72) and binary point clouds.  Thin ctypes wrappers over the C ABI
(amhip_export.hip, amhip_session.hip)."""
import ctypes as C

import numpy as np

from . import hip_lib as L
from .io import DeviceCloud

# AerialGridMap::initialize's layers in construction order (aerial-mapper-grid-map.cc:25-28) and
# the amhip layer behind each (-1: never touched by the path, NaN)
GRID_MAP_LAYERS = ["ortho", "elevation", "elevation_angle", "num_observations",
                   "elevation_angle_first_view", "delta", "observation_index",
                   "observation_index_first", "colored_ortho"]


def _layer_id(name):
    return L.LAYER_NAMES.index(name) if name in L.LAYER_NAMES else -1


def layer_to_image(map_, layer, lower=0.0, upper=255.0, bgr=False):
    """amhip_layer_to_image of an AerialGridMap (its window of the map): (rows, cols) uint8, or
    (rows, cols, 3) B, G, R for the packed colour layer."""
    lib = L.load()
    rows, cols = int(map_.window[2]), int(map_.window[3])
    lid = L.LAYER_NAMES.index(layer) if isinstance(layer, str) else int(layer)
    img = np.zeros((rows, cols, 3) if bgr else (rows, cols), np.uint8)
    L.check(lib.amhip_layer_to_image(map_._h, lid, int(bool(bgr)), float(lower), float(upper),
                                     C.c_void_p(img.ctypes.data), img.strides[0]))
    return img


def session_layer_to_image(session, layer, lower=0.0, upper=255.0, bgr=False):
    """amhip_session_layer_to_image: the whole map of a HostSession."""
    lib = L.load()
    g = session.grid
    lid = L.LAYER_NAMES.index(layer) if isinstance(layer, str) else int(layer)
    img = np.zeros((g.rows, g.cols, 3) if bgr else (g.rows, g.cols), np.uint8)
    L.check(lib.amhip_session_layer_to_image(session._h, lid, int(bool(bgr)), float(lower),
                                             float(upper), C.c_void_p(img.ctypes.data),
                                             img.strides[0]))
    return img


def write_geotiff(filename, image, geotransform, utm_zone=32, northern=True):
    """amhip_geotiff_write_u8: image (H, W) or (H, W, 3) uint8, bands written in the order given."""
    lib = L.load()
    image = np.asarray(image)
    assert image.dtype == np.uint8 and image.ndim in (2, 3)
    bands = 1 if image.ndim == 2 else image.shape[2]
    if image.strides[-1] != 1 or (image.ndim == 3 and image.strides[1] != bands):
        image = np.ascontiguousarray(image)
    gt = (C.c_double * 6)(*[float(v) for v in geotransform])
    L.check(lib.amhip_geotiff_write_u8(str(filename).encode(), C.c_void_p(image.ctypes.data),
                                       image.shape[1], image.shape[0], image.strides[0], bands, gt,
                                       int(utm_zone), int(bool(northern))))


def to_geotiff(orthomosaic, xy, filename):
    """io::AerialMapperIO::toGeoTiff (aerial-mapper-io.cc:349-431): one byte band; the
    geotransform is the one the reference hard-codes (`xy` is ignored there too)."""
    write_geotiff(filename, np.asarray(orthomosaic), (464499.00, 1.0, 0.0, 5.2727e+06, 0.0, -1.0))


def write_data_to_dem_geotiff_color(ortho_image, xy, filename):
    """io::AerialMapperIO::writeDataToDEMGeoTiffColor (:433-509): bands 1, 2, 3 = channels 2, 0, 1
    of the cv::Vec3b pixel (the reference's `TODO: Fix color bands`), unit pixels at `xy`."""
    img = np.asarray(ortho_image)
    write_geotiff(filename, np.ascontiguousarray(img[:, :, [2, 0, 1]]),
                  (float(xy[0]), 1.0, 0.0, float(xy[1]), 0.0, -1.0))


def grid_map_msg_layout(grid, stamp_ns, frame_id="world", layers=GRID_MAP_LAYERS):
    """(buffer, payload offsets): the message with empty payloads (host only)."""
    lib = L.load()
    names = (C.c_char_p * len(layers))(*[n.encode() for n in layers])
    n = lib.amhip_grid_map_msg_bytes(C.byref(grid), frame_id.encode(), len(layers), names)
    buf = np.zeros(n, np.uint8)
    offs = (C.c_size_t * len(layers))()
    L.check(lib.amhip_grid_map_msg_layout(C.byref(grid), int(stamp_ns), frame_id.encode(),
                                          len(layers), names, C.c_void_p(buf.ctypes.data), n, offs))
    return buf, list(offs)


def session_grid_map_msg(session, stamp_ns, frame_id="world", layers=GRID_MAP_LAYERS,
                         host_layers=None, out=None):
    """amhip_session_grid_map_msg: the serialized grid_map_msgs/GridMap of a HostSession's map,
    resident layers straight from the devices.  out: a uint8 buffer to reuse (a publisher's)."""
    lib = L.load()
    names = (C.c_char_p * len(layers))(*[n.encode() for n in layers])
    ids = (C.c_int32 * len(layers))(*[_layer_id(n) for n in layers])
    hosts = (C.c_void_p * len(layers))()
    keep = []
    for k, n in enumerate(layers):
        if host_layers and n in host_layers and ids[k] < 0:
            a = np.ascontiguousarray(host_layers[n], np.float32)
            keep.append(a)
            hosts[k] = a.ctypes.data
    n = lib.amhip_grid_map_msg_bytes(C.byref(session.grid), frame_id.encode(), len(layers), names)
    buf = np.empty(n, np.uint8) if out is None else out[:n]
    assert buf.dtype == np.uint8 and buf.shape[0] == n and buf.flags.c_contiguous
    written = C.c_size_t()
    L.check(lib.amhip_session_grid_map_msg(session._h, int(stamp_ns), frame_id.encode(),
                                           len(layers), names, ids, hosts,
                                           C.c_void_p(buf.ctypes.data), n, C.byref(written)))
    assert written.value == n
    return buf


def write_point_cloud_binary(filename, xyz, intensities=None):
    lib = L.load()
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    inten = None if intensities is None else np.ascontiguousarray(intensities, np.int32)
    assert inten is None or inten.shape[0] == xyz.shape[0]
    L.check(lib.amhip_io_write_point_cloud_binary(
        str(filename).encode(), C.c_void_p(xyz.ctypes.data),
        C.c_void_p(inten.ctypes.data) if inten is not None else None, xyz.shape[0]))


def load_point_cloud_binary(filename, device=0):
    """-> DeviceCloud resident in HBM (pinned double-buffered staging)."""
    lib = L.load()
    xyz, inten, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
    L.check(lib.amhip_io_load_point_cloud_binary(int(device), str(filename).encode(),
                                                 C.byref(xyz), C.byref(inten), C.byref(n)))
    return DeviceCloud(xyz.value, inten.value, n.value, int(device), 0)
