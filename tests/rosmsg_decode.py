"""A ROS 1 message DECODER driven by message definitions -- TEST INFRASTRUCTURE.

Second, independent reader for grid_map_msgs/GridMap: where
oracle/amo_export.py WRITES the message with hand-placed struct.pack calls (and so shares an
author's reading of the layout with the product), this module knows nothing about GridMap: it
parses `.msg` definition text (the published definitions of grid_map_msgs, std_msgs and
geometry_msgs, quoted below field for field) and walks a byte buffer by the ROS 1 serialization
rules alone -- little-endian primitives, `string` = uint32 length + bytes, `T[]` = uint32 count +
elements, `T[n]` = n elements, `time` = uint32 secs + uint32 nsecs, nested messages in field order.
What it returns is a dict of the fields by NAME, so a test can compare `info.pose.position.x` or
`data[3].layout.dim[0].label` with what went in.

Parity status of the format itself stays "unpinned" (grid_map_ros / roscpp are outside the
reference tree); what this pins is that the bytes ARE a well-formed grid_map_msgs/GridMap whose
named fields carry the map.
"""
import struct

import numpy as np

# The definitions, as published (comments dropped).  grid_map_msgs @ the ROS 1 (kinetic / melodic)
# releases the reference's CI builds against; std_msgs / geometry_msgs are frozen.
DEFINITIONS = {
    "grid_map_msgs/GridMap": """
        GridMapInfo info
        string[] layers
        string[] basic_layers
        std_msgs/Float32MultiArray[] data
        uint16 outer_start_index
        uint16 inner_start_index
    """,
    "grid_map_msgs/GridMapInfo": """
        Header header
        float64 resolution
        float64 length_x
        float64 length_y
        geometry_msgs/Pose pose
    """,
    "std_msgs/Header": """
        uint32 seq
        time stamp
        string frame_id
    """,
    "geometry_msgs/Pose": """
        Point position
        Quaternion orientation
    """,
    "geometry_msgs/Point": """
        float64 x
        float64 y
        float64 z
    """,
    "geometry_msgs/Quaternion": """
        float64 x
        float64 y
        float64 z
        float64 w
    """,
    "std_msgs/Float32MultiArray": """
        MultiArrayLayout layout
        float32[] data
    """,
    "std_msgs/MultiArrayLayout": """
        MultiArrayDimension[] dim
        uint32 data_offset
    """,
    "std_msgs/MultiArrayDimension": """
        string label
        uint32 size
        uint32 stride
    """,
}

_PRIM = {"bool": "?", "int8": "b", "uint8": "B", "char": "B", "byte": "b", "int16": "h",
         "uint16": "H", "int32": "i", "uint32": "I", "int64": "q", "uint64": "Q",
         "float32": "f", "float64": "d"}


def _resolve(type_name, package):
    """ROS name resolution: `Header` is std_msgs/Header; a bare name is looked up in the package
    of the message that uses it."""
    if type_name == "Header":
        return "std_msgs/Header"
    if "/" in type_name:
        return type_name
    return package + "/" + type_name


def _fields(msg_type):
    out = []
    for line in DEFINITIONS[msg_type].strip().splitlines():
        line = line.split("#")[0].strip()
        if not line:
            continue
        t, name = line.split()
        out.append((t, name))
    return out


class Reader:
    def __init__(self, data):
        self.data = memoryview(bytes(data))
        self.at = 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.data, self.at)
        self.at += struct.calcsize("<" + fmt)
        return v

    def value(self, t, package):
        if t.endswith("]"):
            base, dim = t[:-1].split("[")
            n = int(dim) if dim else self.take("I")[0]
            if base in ("float32", "float64"):   # (bulk payloads: not element by element)
                dt = np.dtype("<f4" if base == "float32" else "<f8")
                if self.at + n * dt.itemsize > len(self.data):
                    raise ValueError("array runs past the end of the buffer")
                a = np.frombuffer(self.data, dt, n, self.at).copy()
                self.at += n * dt.itemsize
                return a
            return [self.value(base, package) for _ in range(n)]
        if t in _PRIM:
            return self.take(_PRIM[t])[0]
        if t == "string":
            n = self.take("I")[0]
            s = bytes(self.data[self.at:self.at + n]).decode()
            self.at += n
            return s
        if t in ("time", "duration"):
            secs, nsecs = self.take("II")
            return (secs, nsecs)
        full = _resolve(t, package)
        return self.message(full)

    def message(self, msg_type):
        package = msg_type.split("/")[0]
        return {name: self.value(t, package) for t, name in _fields(msg_type)}


def decode(msg_type, data):
    """-> dict of the message's fields; raises if the buffer is not consumed exactly."""
    r = Reader(data)
    out = r.message(msg_type)
    if r.at != len(r.data):
        raise ValueError("%d trailing bytes after a %s" % (len(r.data) - r.at, msg_type))
    return out


def multiarray_to_matrix(arr):
    """std_msgs/Float32MultiArray -> the (rows, cols) matrix it carries, through the layout's OWN
    labels / sizes / strides (grid_map_ros's multiArrayMessageCopyToMatrixEigen reads them the
    same way): dim[0] is the OUTER dimension; `column_index` outermost = column-major storage."""
    dims = arr["layout"]["dim"]
    assert len(dims) == 2 and arr["layout"]["data_offset"] == 0
    outer, inner = dims
    assert outer["stride"] == outer["size"] * inner["size"] and inner["stride"] == inner["size"]
    data = np.asarray(arr["data"], np.float32)
    assert data.size == outer["size"] * inner["size"]
    m = data.reshape(outer["size"], inner["size"])
    if outer["label"] == "column_index" and inner["label"] == "row_index":
        return m.T            # m[j][i] -> (i, j)
    if outer["label"] == "row_index" and inner["label"] == "column_index":
        return m
    raise ValueError("unknown layout labels %r / %r" % (outer["label"], inner["label"]))
