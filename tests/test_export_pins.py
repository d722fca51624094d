"""Independent pins of the export formats.

tests/test_export_formats.py checks the writers against oracle/amo_export.py -- a reader and a
serializer by the same author as the product.  Here the same outputs go through readers that
share NOTHING with it:
  * Pillow (PIL, in the image) opens every GeoTiff the C ABI and the drop-in C++ classes write:
    pixels, band layout, padded rows / several strips, and the GeoTIFF tags 33550 / 33922 / 34735
    decoded by the GeoTIFF key rules into scale, tie point and EPSG code;
  * tests/rosmsg_decode.py walks the grid_map_msgs/GridMap bytes by the `.msg` definitions alone
    and returns the fields by name.
GDAL / grid_map_ros themselves are not in the image: what GDAL would make of the file and what
GridMapRosConverter::fromMessage would make of the message stays "parity unpinned" (DESIGN.md 2).
"""
import ctypes as C
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from aerial_mapper_amd import export as E  # noqa: E402
from aerial_mapper_amd import hip_lib as L  # noqa: E402
import rosmsg_decode as R  # noqa: E402

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def pil_read(path):
    im = Image.open(str(path))
    im.load()
    tags = {k: im.tag_v2[k] for k in im.tag_v2}
    return np.array(im), tags, im


def geokeys_from_tags(tags):
    """GeoTIFF 1.0 section 2.4: GeoKeyDirectoryTag = header (version, revision, minor, count) +
    count x (KeyID, TIFFTagLocation, Count, Value_Offset)."""
    d = tuple(tags[34735])
    assert d[0] == 1 and d[1] == 1 and d[2] == 0 and len(d) == 4 + 4 * d[3]
    ascii_params = tags.get(34737, "")
    keys = {}
    for k in range(d[3]):
        key, loc, cnt, val = d[4 + 4 * k: 8 + 4 * k]
        if loc == 0:
            assert cnt == 1
            keys[key] = val
        elif loc == 34737:
            keys[key] = ascii_params[val:val + cnt].rstrip("|")
        else:
            raise AssertionError("GeoKey %d stored in tag %d" % (key, loc))
    return keys


def georeference(tags):
    """(x of pixel column c, y of pixel row r) functions from ModelPixelScale + ModelTiepoint."""
    sx, sy, _sz = tags[33550]
    i, j, _k, x, y, _z = tags[33922]
    return (lambda c: x + (c - i) * sx), (lambda r: y - (r - j) * sy)


def test_pil_reads_the_gray_geotiff(tmp_path):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    f = tmp_path / "g.tif"
    E.write_geotiff(f, img, (464499.0, 0.25, 0.0, 5272700.0, 0.0, -0.25), utm_zone=32, northern=True)
    px, tags, im = pil_read(f)
    assert im.mode == "L" and im.size == (53, 37) and np.array_equal(px, img)
    assert tags[259] == 1 and tags[262] == 1 and tags[277] == 1 and tags[258] == (8,)
    assert tuple(tags[33550]) == (0.25, 0.25, 0.0)
    assert tuple(tags[33922]) == (0.0, 0.0, 0.0, 464499.0, 5272700.0, 0.0)
    k = geokeys_from_tags(tags)
    # GTModelTypeGeoKey = projected, GTRasterTypeGeoKey = PixelIsArea, ProjectedCSTypeGeoKey =
    # EPSG:32632 (WGS 84 / UTM zone 32N), linear unit metre, geographic CS EPSG:4326
    assert k[1024] == 1 and k[1025] == 1 and k[3072] == 32632 and k[3076] == 9001 and k[2048] == 4326
    X, Y = georeference(tags)
    assert X(0) == 464499.0 and X(52) == 464499.0 + 13.0 and Y(36) == 5272700.0 - 9.0


def test_pil_reads_three_bands_and_the_southern_zone(tmp_path):
    rng = np.random.default_rng(12)
    rgb = rng.integers(0, 256, (21, 30, 3), dtype=np.uint8)
    f = tmp_path / "c.tif"
    E.write_geotiff(f, rgb, (500000.0, 1.0, 0.0, 6000000.0, 0.0, -1.0), utm_zone=33, northern=False)
    px, tags, im = pil_read(f)
    assert im.mode == "RGB" and px.shape == (21, 30, 3) and np.array_equal(px, rgb)
    assert tags[262] == 2 and tags[277] == 3 and tags[258] == (8, 8, 8) and tags[284] == 1
    assert geokeys_from_tags(tags)[3072] == 32733          # WGS 84 / UTM zone 33S


def test_pil_reads_padded_rows_and_several_strips(tmp_path):
    rng = np.random.default_rng(13)
    big = rng.integers(0, 256, (9000, 8192), dtype=np.uint8)      # 70 MB: two strips of <= 64 MB
    view = big[:, :8000]                                            # step 8192 > row 8000
    f = tmp_path / "s.tif"
    E.write_geotiff(f, view, (10.0, 0.5, 0.0, 20.0, 0.0, -0.25))
    px, tags, im = pil_read(f)
    assert len(tags[273]) == 2 and len(tags[279]) == 2 and sum(tags[279]) == view.size
    assert np.array_equal(px, view)
    # colour with padded rows
    c = rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)
    E.write_geotiff(f, c[:, :50], (0.0, 1.0, 0.0, 0.0, 0.0, -1.0))
    px, _, _ = pil_read(f)
    assert np.array_equal(px, c[:, :50])


@pytest.fixture(scope="module")
def dropin_outputs(tmp_path_factory):
    """tests/cpp/shim_export.cc: the drop-in io::AerialMapperIO / AerialGridMap methods."""
    from aerial_mapper_amd import build
    build.build_all()
    d = tmp_path_factory.mktemp("dropin")
    lib = os.path.join(ROOT, "aerial_mapper_amd", "lib")
    exe = str(d / "shim_export")
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_export.cc"), "-o", exe,
                           "-L" + lib, "-laerial_mapper_shim", "-laerial_mapper_hip",
                           "-Wl,-rpath," + lib])
    out = subprocess.run([exe, str(d)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert out.returncode == 0, out.stdout.decode()
    return d


def test_pil_reads_what_the_dropin_classes_write(dropin_outputs):
    d = dropin_outputs
    H, W = 19, 23
    raw = (d / "inputs.bin").read_bytes()
    gray = np.frombuffer(raw, np.uint8, H * W).reshape(H, W)
    bgr = np.frombuffer(raw, np.uint8, H * W * 3, H * W).reshape(H, W, 3)
    # io::AerialMapperIO::toGeoTiff (aerial-mapper-io.cc:349-431): one byte band, the hard-coded
    # adfGeoTransform {464499.00, 1, 0, 5.2727e+06, 0, -1} (:381), UTM 32 N / WGS 84 (:392-399)
    px, tags, im = pil_read(d / "gray.tif")
    assert im.mode == "L" and np.array_equal(px, gray)
    assert tuple(tags[33550]) == (1.0, 1.0, 0.0)
    assert tuple(tags[33922]) == (0.0, 0.0, 0.0, 464499.00, 5.2727e+06, 0.0)
    assert geokeys_from_tags(tags)[3072] == 32632
    # writeDataToDEMGeoTiffColor (:433-509): bands 1, 2, 3 = Vec3b channels 2, 0, 1 (:487-491),
    # unit pixels at the xy handed in (:462-467)
    px, tags, im = pil_read(d / "colour.tif")
    assert im.mode == "RGB"
    assert np.array_equal(px[..., 0], bgr[..., 2]) and np.array_equal(px[..., 1], bgr[..., 0])
    assert np.array_equal(px[..., 2], bgr[..., 1])
    assert tuple(tags[33922])[3:5] == (464736.27, 5272359.16) and geokeys_from_tags(tags)[3072] == 32632


def test_definition_driven_decoder_reads_the_dropin_message(dropin_outputs):
    """AerialGridMap::serializeMessage (the message publishOnce hands to ros::Publisher,
    aerial-mapper-grid-map.cc:66-72) decoded by the .msg definitions alone."""
    msg = R.decode("grid_map_msgs/GridMap", (dropin_outputs / "map.msg").read_bytes())
    info = msg["info"]
    assert info["header"] == {"seq": 0, "stamp": (1506593812, 123456789), "frame_id": "world"}
    assert (info["resolution"], info["length_x"], info["length_y"]) == (0.5, 3.0, 2.0)
    assert info["pose"]["position"] == {"x": 10.0, "y": -4.0, "z": 0.0}
    assert info["pose"]["orientation"] == {"x": 0.0, "y": 0.0, "z": 0.0, "w": 1.0}
    assert msg["layers"] == E.GRID_MAP_LAYERS and msg["basic_layers"] == []
    assert msg["outer_start_index"] == 0 and msg["inner_start_index"] == 0
    assert len(msg["data"]) == len(E.GRID_MAP_LAYERS)
    init = {"ortho": 255.0, "elevation": np.nan, "elevation_angle": 0.0, "num_observations": 0.0,
            "elevation_angle_first_view": np.nan, "delta": np.nan, "observation_index": np.nan,
            "observation_index_first": np.nan, "colored_ortho": np.nan}
    for name, arr in zip(msg["layers"], msg["data"]):
        m = R.multiarray_to_matrix(arr)
        assert m.shape == (6, 4)
        want = np.full((6, 4), init[name], np.float32)      # aerial-mapper-grid-map.cc:40-48
        if name == "elevation":
            want[2, 1] = 412.5
        if name == "ortho":
            want[0, 3] = 17.0
        assert np.array_equal(m, want, equal_nan=True), name


@pytest.mark.parametrize("rows,cols", [(7, 5), (64, 33)])
def test_definition_driven_decoder_reads_the_c_abi_layout(rows, cols):
    res = 0.25
    g = L.make_grid(rows * res, cols * res, res, 12.5, -3.0)
    stamp = 1506593812 * 10**9 + 123456789
    rng = np.random.default_rng(rows + 100)
    mats = {n: rng.standard_normal((rows, cols)).astype(np.float32) for n in E.GRID_MAP_LAYERS}
    buf, offs = E.grid_map_msg_layout(g, stamp, "map_frame")
    for n, at in zip(E.GRID_MAP_LAYERS, offs):
        cm = np.asfortranarray(mats[n])                     # Eigen::MatrixXf storage
        buf[at:at + cm.nbytes] = np.frombuffer(cm.tobytes(order="F"), np.uint8)
    msg = R.decode("grid_map_msgs/GridMap", bytes(buf))
    assert msg["info"]["header"]["frame_id"] == "map_frame"
    assert msg["info"]["header"]["stamp"] == (1506593812, 123456789)
    assert msg["info"]["resolution"] == g.resolution
    assert (msg["info"]["length_x"], msg["info"]["length_y"]) == (g.length_x, g.length_y)
    assert (msg["info"]["pose"]["position"]["x"], msg["info"]["pose"]["position"]["y"]) == (12.5, -3.0)
    for n, arr in zip(msg["layers"], msg["data"]):
        d0, d1 = arr["layout"]["dim"]
        assert (d0["label"], d0["size"], d0["stride"]) == ("column_index", cols, rows * cols)
        assert (d1["label"], d1["size"], d1["stride"]) == ("row_index", rows, rows)
        assert np.array_equal(R.multiarray_to_matrix(arr), mats[n])
    # a truncated / padded buffer is not a message
    with pytest.raises((ValueError, struct.error)):
        R.decode("grid_map_msgs/GridMap", bytes(buf[:-3]))
    with pytest.raises(ValueError):
        R.decode("grid_map_msgs/GridMap", bytes(buf) + b"\0")


def test_binary_cloud_header_is_not_trusted(tmp_path):
    """AMPCLD01 loader: the point count of the header is checked against the file's size BEFORE
    it is multiplied (n = 2^61 wrapped 24 * n to 0), counts the DSM cannot index and
    unknown flag bits are refused -- all before a device is touched."""
    lib = L.load()
    f = tmp_path / "bad.ampc"

    def load():
        xyz, inten, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        rc = lib.amhip_io_load_point_cloud_binary(0, str(f).encode(), C.byref(xyz), C.byref(inten),
                                                  C.byref(n))
        return rc, n.value, xyz.value

    body = np.zeros(10 * 3).tobytes()
    for n64, flags in [(1 << 61, 0), ((1 << 61) + 1, 1), (1 << 63, 0), (0xFFFFFFFFFFFFFFFF, 1),
                       (11, 0), (10, 1), (10, 2), (10, 0x80000001)]:
        f.write_bytes(b"AMPCLD01" + struct.pack("<QII", n64, flags, 0) + b"\0" * 8 + body)
        rc, n, ptr = load()
        assert rc == L.ERR_ARG and n == 0 and not ptr, (n64, flags)
    # 2^31 - 1 points and more: refused even when the file is (sparsely) long enough
    with open(f, "wb") as fh:
        fh.write(b"AMPCLD01" + struct.pack("<QII", 0x7FFFFFFF, 0, 0) + b"\0" * 8)
        fh.truncate(32 + 24 * 0x7FFFFFFF)
    rc, n, ptr = load()
    assert rc == L.ERR_ARG and n == 0 and not ptr
