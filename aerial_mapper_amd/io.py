"""io::AerialMapperIO's text formats (aerial-mapper-io.cc:103-121,309-347).

`load_point_cloud_text` tokenises and parses the `x y z intensity` file on the
GPU (amhip_io.hip) and leaves the cloud in HBM, ready for Dsm.process /
OrthoFromPcl.process; `load_poses_text` reads the small pose file on the host.
"""
import ctypes as C
import mmap
import os

import numpy as np

from . import hip_lib as L


class DeviceCloud(object):
    """A point cloud resident in HBM: .xyz (n,3) float64 and .intensities (n,)
    int32 as torch CUDA tensors sharing the library's allocation."""

    def __init__(self, xyz_ptr, inten_ptr, n, device, strtod_tokens):
        self._xyz_ptr, self._inten_ptr = xyz_ptr, inten_ptr
        self.n = int(n)
        self.device = device
        self.strtod_tokens = int(strtod_tokens)

    def _view(self, ptr, shape, typestr):
        import torch

        class _Holder(object):
            pass

        h = _Holder()
        h.__cuda_array_interface__ = {"shape": shape, "typestr": typestr,
                                      "data": (int(ptr), False), "version": 2}
        h._keepalive = self
        return torch.as_tensor(h, device="cuda:%d" % self.device)

    @property
    def xyz(self):
        return self._view(self._xyz_ptr, (self.n, 3), "<f8") if self.n else None

    @property
    def intensities(self):
        return self._view(self._inten_ptr, (self.n,), "<i4") if self.n and self._inten_ptr else None

    def to_host(self):
        xyz = np.empty((self.n, 3), np.float64)
        inten = np.empty(self.n, np.int32) if (self._inten_ptr or not self.n) else None
        if self.n:
            L.check(L.load().amhip_io_download_point_cloud(
                C.c_void_p(self._xyz_ptr), C.c_void_p(self._inten_ptr), self.n,
                C.c_void_p(xyz.ctypes.data),
                C.c_void_p(inten.ctypes.data) if inten is not None else None))
        return xyz, inten

    def close(self):
        if L is None or not (getattr(self, "_xyz_ptr", None) or getattr(self, "_inten_ptr", None)):
            return        # (nothing to free, or the interpreter is shutting down)
        lib = L.load()
        for name in ("_xyz_ptr", "_inten_ptr"):
            p = getattr(self, name, None)
            if p:
                lib.amhip_io_free(C.c_void_p(p))
                setattr(self, name, None)

    __del__ = close


def parse_point_cloud_text(text, device=0):
    """text: bytes-like content of a point-cloud file -> DeviceCloud."""
    lib = L.load()
    buf = bytes(text) if not isinstance(text, (bytes, mmap.mmap)) else text
    xyz, inten = C.c_void_p(), C.c_void_p()
    n, slow = C.c_size_t(), C.c_size_t()
    if isinstance(buf, mmap.mmap):
        addr = C.cast((C.c_char * len(buf)).from_buffer(buf), C.c_char_p)
    else:
        addr = buf
    L.check(lib.amhip_io_parse_point_cloud_text(int(device), addr, len(buf), C.byref(xyz),
                                                C.byref(inten), C.byref(n), C.byref(slow)))
    return DeviceCloud(xyz.value, inten.value, n.value, int(device), slow.value)


def load_point_cloud_text(filename, device=0):
    """io::AerialMapperIO::loadPointCloudFromFile -> DeviceCloud (CHECKs like the reference)."""
    if not filename:
        raise L.AmhipError(L.ERR_ARG, 'CHECK(filename_point_cloud != "")')
    if os.path.getsize(filename) == 0:
        raise L.AmhipError(L.ERR_ARG, "CHECK(point_cloud_xyz->size() > 0)")
    with open(filename, "rb") as f:
        data = f.read()
    cloud = parse_point_cloud_text(data, device)
    if cloud.n == 0:
        raise L.AmhipError(L.ERR_ARG, "CHECK(point_cloud_xyz->size() > 0)")
    return cloud


def load_poses_text(filename):
    """io::AerialMapperIO::loadPosesFromFileStandard: records x y z qw qx qy qz
    -> (F,7) float64 in the C boundary's layout."""
    if not filename:
        raise L.AmhipError(L.ERR_ARG, "Empty filename")
    vals = []
    with open(filename, "rb") as f:
        for tok in f.read().split():
            try:
                vals.append(float(tok))
            except ValueError:
                break
    n = len(vals) // 7
    if n == 0:
        raise L.AmhipError(L.ERR_ARG, "No poses loaded.")
    return np.asarray(vals[:7 * n], np.float64).reshape(n, 7)
