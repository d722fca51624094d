"""CPU restatement of the formats behind the hot path -- TEST INFRASTRUCTURE (imported by tests/
only; the product path never touches it).

The libraries that define these formats are NOT in the reference tree (grid_map_cv, grid_map_ros,
roscpp serialization, GDAL's GTiff driver): what follows restates their published definitions as
adopted here -- "parity unpinned", like the other external-library formulas.
Call sites in the reference: aerial_mapper_grid_map/src/aerial-mapper-grid-map.cc:10,51-72;
aerial_mapper_io/src/aerial-mapper-io.cc:349-509.
"""
import struct

import numpy as np


# ---- grid_map_cv::GridMapCvConverter::toImage<unsigned char, 1> ----------------------------
def to_image_u8(layer_cm, lower, upper):
    """layer_cm: (cols, rows) float32 = the column-major Eigen matrix as it lies in memory.
    toImage: image = zeros(size(0), size(1)); the layer is clamped to [lower, upper]; for every
    cell with a finite value image(i, j) = (uchar)(((v - lower) / (upper - lower)) * 255.0f),
    evaluated in float."""
    m = np.asarray(layer_cm, np.float32).T          # (rows, cols): element (i, j)
    lower, upper = np.float32(lower), np.float32(upper)
    img = np.zeros(m.shape, np.uint8)
    ok = np.isfinite(m)
    v = np.minimum(np.maximum(m[ok], lower), upper)
    t = ((v - lower) / (upper - lower)).astype(np.float32) * np.float32(255.0)
    img[ok] = t.astype(np.float32).astype(np.int32).astype(np.uint8)
    return img


def colored_to_bgr(layer_cm):
    """grid_map's packed colours (colorVectorToValue: the float's bits are R << 16 | G << 8 | B)
    -> (rows, cols, 3) B, G, R; NaN cells -> 0."""
    m = np.asarray(layer_cm, np.float32).T
    bits = np.ascontiguousarray(m).view(np.uint32)
    bits = np.where(np.isnan(m), np.uint32(0), bits)
    out = np.zeros(m.shape + (3,), np.uint8)
    out[..., 0] = bits & 0xFF
    out[..., 1] = (bits >> 8) & 0xFF
    out[..., 2] = (bits >> 16) & 0xFF
    return out


# ---- grid_map_msgs/GridMap, ROS 1 serialization -----------------------------------------------
def _s(x):
    b = x.encode()
    return struct.pack("<I", len(b)) + b


def grid_map_msg(rows, cols, resolution, length_x, length_y, pos_x, pos_y, stamp_ns, frame_id,
                 layers):
    """layers: list of (name, (cols, rows) float32 array).  GridMapRosConverter::toMessage +
    ros::serialization: header (seq 0, stamp.fromNSec, frame_id), resolution, lengths, pose
    (position x, y, 0; orientation 0, 0, 0, 1), layers[], basic_layers[] (empty), data[] with
    matrixEigenCopyToMultiArrayMessage's layout for a column-major matrix (dim[0] = column_index:
    size cols, stride rows * cols; dim[1] = row_index: size rows, stride rows), start indices."""
    out = [struct.pack("<III", 0, stamp_ns // 10**9, stamp_ns % 10**9), _s(frame_id),
           struct.pack("<3d", resolution, length_x, length_y),
           struct.pack("<7d", pos_x, pos_y, 0.0, 0.0, 0.0, 0.0, 1.0),
           struct.pack("<I", len(layers))]
    out += [_s(n) for n, _ in layers]
    out.append(struct.pack("<I", 0))
    out.append(struct.pack("<I", len(layers)))
    for _, a in layers:
        a = np.ascontiguousarray(a, np.float32)
        assert a.shape == (cols, rows)
        out += [struct.pack("<I", 2), _s("column_index"), struct.pack("<II", cols, rows * cols),
                _s("row_index"), struct.pack("<II", rows, rows), struct.pack("<I", 0),
                struct.pack("<I", rows * cols), a.tobytes()]
    out.append(struct.pack("<HH", 0, 0))
    return b"".join(out)


# ---- a minimal TIFF / GeoTIFF reader (classic, little-endian, uncompressed strips) ------------
_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 12: ("d", 8)}


def read_tiff(data):
    """-> (tags: {tag: tuple of values | bytes for ASCII}, pixels: (H, W[, bands]) uint8)."""
    assert data[:4] == b"II*\x00"
    (ifd,) = struct.unpack_from("<I", data, 4)
    (n,) = struct.unpack_from("<H", data, ifd)
    tags = {}
    for k in range(n):
        tag, typ, cnt = struct.unpack_from("<HHI", data, ifd + 2 + 12 * k)
        fmt, size = _TYPES[typ]
        at = ifd + 2 + 12 * k + 8
        if size * cnt > 4:
            (at,) = struct.unpack_from("<I", data, at)
        if typ == 2:
            tags[tag] = bytes(data[at:at + cnt])
        else:
            tags[tag] = struct.unpack_from("<%d%s" % (cnt, fmt), data, at)
    (nxt,) = struct.unpack_from("<I", data, ifd + 2 + 12 * n)
    assert nxt == 0
    w, h = tags[256][0], tags[257][0]
    bands = tags[277][0]
    assert tags[259] == (1,) and tags[284] == (1,) and all(b == 8 for b in tags[258])
    rps = tags[278][0]
    rows = []
    for s, (off, cnt) in enumerate(zip(tags[273], tags[279])):
        nr = min(rps, h - s * rps)
        assert cnt == nr * w * bands
        rows.append(np.frombuffer(data, np.uint8, cnt, off).reshape(nr, w, bands))
    px = np.concatenate(rows, 0)
    return tags, (px[..., 0] if bands == 1 else px)


def geokeys(tags):
    """GeoKeyDirectory -> {key: value}; ASCII keys resolved through GeoAsciiParams."""
    d = tags[34735]
    assert d[0] == 1 and d[1] == 1
    out = {}
    for k in range(d[3]):
        key, loc, cnt, val = d[4 + 4 * k: 8 + 4 * k]
        if loc == 0:
            out[key] = val
        elif loc == 34737:
            out[key] = tags[34737][val:val + cnt].decode().rstrip("|")
    return out
