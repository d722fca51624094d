"""Load tests/golden/*.npz fixtures (see tests/golden/make_golden.py)."""
import glob
import os

import numpy as np

import oracle_ffi as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def names(kind):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        n = os.path.basename(p)[:-4]
        if n.startswith(kind + "_"):
            out.append(n)
    return out


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def grid_of(d):
    lx, ly, res, px, py = [float(v) for v in d["grid"]]
    return O.make_grid(lx, ly, res, px, py)


def camera_of(d):
    c = O.Camera()
    v = d["camera"]
    c.fu, c.fv, c.cu, c.cv = float(v[0]), float(v[1]), float(v[2]), float(v[3])
    c.width, c.height, c.distortion = int(v[4]), int(v[5]), int(v[6])
    for k in range(4):
        c.dist[k] = float(v[7 + k])
    return c


ORTHO_LAYERS = ["elevation_angle", "observation_index", "num_observations", "ortho",
                "colored_ortho"]


def bits_equal(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
