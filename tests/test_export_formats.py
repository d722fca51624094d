"""The formats behind the hot path, host side: the GeoTiff container, the
grid_map_msgs/GridMap wire layout and the binary cloud file, each against the independent
restatement / reader of oracle/amo_export.py.  No GPU needed: these entry points are host code."""
import ctypes as C
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import amo_export as X  # noqa: E402

from aerial_mapper_amd import export as E  # noqa: E402
from aerial_mapper_amd import hip_lib as L  # noqa: E402


def test_gray_geotiff_is_a_readable_georeferenced_tiff(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    f = tmp_path / "o.tif"
    E.to_geotiff(img, (1.0, 2.0), f)          # (xy is ignored, like the reference does)
    tags, px = X.read_tiff(f.read_bytes())
    assert np.array_equal(px, img)
    assert tags[262] == (1,) and tags[277] == (1,) and tags[339] == (1,)
    # adfGeoTransform = {464499.00, 1.0, 0.0, 5.2727e+06, 0.0, -1.0} (aerial-mapper-io.cc:381)
    assert tags[33550] == (1.0, 1.0, 0.0)
    assert tags[33922] == (0.0, 0.0, 0.0, 464499.00, 5.2727e+06, 0.0)
    k = X.geokeys(tags)
    assert k[1024] == 1 and k[1025] == 1 and k[2048] == 4326 and k[3072] == 32632 and k[3076] == 9001
    assert k[1026] == "UTM 32 (WGS84) in northern hemisphere." and k[2049] == "WGS 84"


def test_colour_geotiff_keeps_the_references_band_order(tmp_path):
    rng = np.random.default_rng(2)
    bgr = rng.integers(0, 256, (21, 30, 3), dtype=np.uint8)
    f = tmp_path / "c.tif"
    E.write_data_to_dem_geotiff_color(bgr, (464736.27, 5272359.16), f)
    tags, px = X.read_tiff(f.read_bytes())
    # pdata = tmp(2), pdata2 = tmp(0), pdata3 = tmp(1)  (aerial-mapper-io.cc:487-491)
    assert np.array_equal(px[..., 0], bgr[..., 2])
    assert np.array_equal(px[..., 1], bgr[..., 0])
    assert np.array_equal(px[..., 2], bgr[..., 1])
    assert tags[262] == (2,) and tags[277] == (3,) and tags[258] == (8, 8, 8)
    assert tags[33922] == (0.0, 0.0, 0.0, 464736.27, 5272359.16, 0.0)


def test_geotiff_rows_with_padding_and_several_strips(tmp_path):
    rng = np.random.default_rng(3)
    big = rng.integers(0, 256, (9000, 8192), dtype=np.uint8)      # 70 MB: two strips
    view = big[:, :8000]                                            # step 8192 > row 8000
    f = tmp_path / "s.tif"
    E.write_geotiff(f, view, (10.0, 0.5, 0.0, 20.0, 0.0, -0.25), utm_zone=33, northern=False)
    tags, px = X.read_tiff(f.read_bytes())
    assert len(tags[273]) == 2 and np.array_equal(px, view)
    assert tags[33550] == (0.5, 0.25, 0.0) and X.geokeys(tags)[3072] == 32733


def test_geotiff_argument_errors(tmp_path):
    lib = L.load()
    img = np.zeros((4, 4), np.uint8)
    gt = (C.c_double * 6)(0, 1, 0.1, 0, 0, -1)                     # rotated: refused
    assert lib.amhip_geotiff_write_u8(str(tmp_path / "x.tif").encode(), C.c_void_p(img.ctypes.data),
                                      4, 4, 4, 1, gt, 32, 1) == L.ERR_ARG
    gt = (C.c_double * 6)(0, 1, 0, 0, 0, -1)
    assert lib.amhip_geotiff_write_u8(str(tmp_path / "x.tif").encode(), C.c_void_p(img.ctypes.data),
                                      4, 4, 4, 2, gt, 32, 1) == L.ERR_ARG
    assert lib.amhip_geotiff_write_u8(b"/nonexistent-dir/x.tif", C.c_void_p(img.ctypes.data),
                                      4, 4, 4, 1, gt, 32, 1) == L.ERR_ARG


@pytest.mark.parametrize("rows,cols", [(7, 5), (64, 33)])
def test_grid_map_message_layout_matches_the_ros_serialization(rows, cols):
    res = 0.25
    g = L.make_grid(rows * res, cols * res, res, 12.5, -3.0)
    assert (g.rows, g.cols) == (rows, cols)
    stamp = 1506593812 * 10**9 + 123456789
    rng = np.random.default_rng(rows)
    mats = [(n, rng.standard_normal((cols, rows)).astype(np.float32)) for n in E.GRID_MAP_LAYERS]
    buf, offs = E.grid_map_msg_layout(g, stamp, "world")
    for (_, a), at in zip(mats, offs):
        buf[at:at + a.nbytes] = np.frombuffer(a.tobytes(), np.uint8)
    want = X.grid_map_msg(rows, cols, g.resolution, g.length_x, g.length_y, g.pos_x, g.pos_y, stamp,
                          "world", mats)
    assert bytes(buf) == want
    # spot checks of the wire format itself
    assert struct.unpack_from("<III", want, 0) == (0, 1506593812, 123456789)
    assert want[12:21] == struct.pack("<I", 5) + b"world"
    assert want[-4:] == b"\x00\x00\x00\x00"


def test_binary_cloud_file_layout(tmp_path):
    rng = np.random.default_rng(5)
    xyz = rng.standard_normal((1000, 3))
    inten = rng.integers(0, 256, 1000).astype(np.int32)
    f = tmp_path / "cloud.ampc"
    E.write_point_cloud_binary(f, xyz, inten)
    raw = f.read_bytes()
    assert raw[:8] == b"AMPCLD01" and struct.unpack_from("<QII", raw, 8) == (1000, 1, 0)
    assert raw[32:32 + 24000] == xyz.tobytes() and raw[32 + 24000:] == inten.tobytes()
    E.write_point_cloud_binary(f, xyz)                               # without intensities
    raw = f.read_bytes()
    assert struct.unpack_from("<QI", raw, 8) == (1000, 0) and len(raw) == 32 + 24000


def test_to_image_restatement_known_answers():
    # (v - lower) / (upper - lower) * 255, truncated; clamped; NaN / inf -> 0; image = transpose
    layer = np.array([[0.0, 255.0, 127.5], [np.nan, -5.0, 300.0]], np.float32)   # (cols=2, rows=3)
    img = X.to_image_u8(layer, 0.0, 255.0)
    assert img.shape == (3, 2)
    assert img.tolist() == [[0, 0], [255, 0], [127, 255]]
    packed = np.array([[np.nan, 0.0]], np.float32)
    packed.view(np.uint32)[0, 1] = (3 << 16) | (2 << 8) | 1           # R = 3, G = 2, B = 1
    assert X.colored_to_bgr(packed).tolist() == [[[0, 0, 0]], [[1, 2, 3]]]


def test_cpp_dropin_writers(tmp_path):
    """io::AerialMapperIO::toGeoTiff / writeDataToDEMGeoTiffColor / savePointCloudToBinaryFile and
    AerialGridMap::serializeMessage of the drop-in headers (tests/cpp/shim_export.cc)."""
    import subprocess
    from aerial_mapper_amd import build
    build.build_all()
    lib = os.path.join(ROOT, "aerial_mapper_amd", "lib")
    exe = str(tmp_path / "shim_export")
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_export.cc"), "-o", exe,
                           "-L" + lib, "-laerial_mapper_shim", "-laerial_mapper_hip",
                           "-Wl,-rpath," + lib])
    out = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=120)
    assert out.returncode == 0, out.stdout.decode()
    H, W = 19, 23
    raw = (tmp_path / "inputs.bin").read_bytes()
    gray = np.frombuffer(raw, np.uint8, H * W).reshape(H, W)
    colour = np.frombuffer(raw, np.uint8, H * W * 3, H * W).reshape(H, W, 3)
    tags, px = X.read_tiff((tmp_path / "gray.tif").read_bytes())
    assert np.array_equal(px, gray) and tags[33922][3:5] == (464499.00, 5.2727e+06)
    tags, px = X.read_tiff((tmp_path / "colour.tif").read_bytes())
    assert np.array_equal(px, colour[..., [2, 0, 1]]) and tags[33922][3:5] == (464736.27, 5272359.16)
    raw = (tmp_path / "cloud.ampc").read_bytes()
    assert raw[:8] == b"AMPCLD01" and struct.unpack_from("<QI", raw, 8) == (100, 1)
    xyz = np.frombuffer(raw, np.float64, 300, 32).reshape(100, 3)
    assert xyz[7].tolist() == [3.5, -1.75, 407.0]
    # the message: a 6 x 4 map around (10, -4) with two edited cells
    rows, cols = 6, 4
    init = {"ortho": 255.0, "elevation": np.nan, "elevation_angle": 0.0, "num_observations": 0.0,
            "elevation_angle_first_view": np.nan, "delta": np.nan, "observation_index": np.nan,
            "observation_index_first": np.nan, "colored_ortho": np.nan}
    mats = [(n, np.full((cols, rows), init[n], np.float32)) for n in E.GRID_MAP_LAYERS]
    dict(mats)["elevation"][1, 2] = 412.5      # (i = 2, j = 1) of the column-major matrix
    dict(mats)["ortho"][3, 0] = 17.0
    want = X.grid_map_msg(rows, cols, 0.5, 3.0, 2.0, 10.0, -4.0, 1506593812123456789, "world", mats)
    assert (tmp_path / "map.msg").read_bytes() == want
