"""GPU: what leaves the map -- layer -> image kernels, the grid_map_msgs
message filled straight from the devices, the binary cloud loader -- against the restatements of
oracle/amo_export.py, and the whole chain cloud file -> DSM -> mosaic -> GeoTiff."""
import os
import sys

import numpy as np
import pytest

import oracle_ffi as O
import scenarios as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import amo_export as X  # noqa: E402

pytestmark = pytest.mark.gpu


def _settings(A, g):
    return A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)


@pytest.mark.parametrize("rows,cols", [(64, 64), (130, 71), (33, 200)])
def test_layer_to_image_is_grid_map_cvs_toimage(rows, cols):
    import aerial_mapper_amd as A
    from aerial_mapper_amd import export as E
    res = 0.5
    rng = np.random.default_rng(rows * 1000 + cols)
    layer = (rng.random((cols, rows)) * 300.0 - 20.0).astype(np.float32)   # beyond [0, 255] too
    layer[rng.random((cols, rows)) < 0.1] = np.nan
    layer[0, 0], layer[-1, -1] = np.inf, -np.inf
    layer[1, 1], layer[2, 2] = 255.0, 0.0
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, rows * res, cols * res, res), device=0) as m:
        assert (m.rows, m.cols) == (rows, cols)
        m.set("ortho", layer)
        for lo, hi in ((0.0, 255.0), (10.0, 200.0), (-20.0, 280.0)):
            got = E.layer_to_image(m, "ortho", lo, hi)
            want = X.to_image_u8(layer, lo, hi)
            assert got.shape == (rows, cols) and np.array_equal(got, want), (lo, hi)
        # a fresh (lazily reset) layer is its initial constant
        m.reset()
        assert (E.layer_to_image(m, "ortho", 0.0, 255.0) == 255).all()
        assert (E.layer_to_image(m, "elevation", 0.0, 255.0) == 0).all()      # NaN -> 0
        # packed colours
        packed = np.full((cols, rows), np.nan, np.float32)
        bits = rng.integers(0, 1 << 24, (cols, rows), dtype=np.uint32)
        keep = rng.random((cols, rows)) < 0.8
        packed.view(np.uint32)[keep] = bits[keep]
        m.set("colored_ortho", packed)
        got = E.layer_to_image(m, "colored_ortho", bgr=True)
        assert np.array_equal(got, X.colored_to_bgr(packed))


@pytest.mark.parametrize("tiles", [(1, 1), (2, 2)])
def test_session_message_and_image_come_straight_from_the_devices(tiles):
    import aerial_mapper_amd as A
    from aerial_mapper_amd import export as E
    sc = S.Scene(96.0, 64.0, 0.5, 40000, seed=77, num_frames=6)
    g = sc.grid
    with A.HostSession(_settings(A, g), tiles=tiles) as hs:
        hs.dsm_process(A.DsmSettings(1), sc.points)
        c = sc.cam
        hs.ortho_process(A.NCamera(c.fu, c.fv, c.cu, c.cv, c.width, c.height), A.OrthoSettings(),
                         sc.poses, sc.frames)
        stamp = 1234567890123456789
        extra = {"delta": np.full((g.cols, g.rows), 3.5, np.float32)}
        msg = E.session_grid_map_msg(hs, stamp, "world", host_layers=extra)
        nan = np.full((g.cols, g.rows), np.nan, np.float32)
        mats = [(n, hs.layers[n] if n in hs.layers else extra.get(n, nan)) for n in E.GRID_MAP_LAYERS]
        want = X.grid_map_msg(g.rows, g.cols, g.resolution, g.length_x, g.length_y, g.pos_x, g.pos_y,
                              stamp, "world", mats)
        assert bytes(msg) == want
        img = E.session_layer_to_image(hs, "ortho", 0.0, 255.0)
        assert np.array_equal(img, X.to_image_u8(hs.layers["ortho"], 0.0, 255.0))
        assert img.min() < 255                       # (the mosaic did write pixels)


def test_binary_cloud_goes_to_hbm_through_pinned_staging(tmp_path):
    import aerial_mapper_amd as A
    from aerial_mapper_amd import export as E
    rng = np.random.default_rng(9)
    n = 3_000_000                                    # 72 MB: three 32 MB staging chunks
    xyz = rng.standard_normal((n, 3))
    inten = rng.integers(0, 256, n).astype(np.int32)
    f = tmp_path / "cloud.ampc"
    E.write_point_cloud_binary(f, xyz, inten)
    cloud = E.load_point_cloud_binary(f)
    assert cloud.n == n
    gx, gi = cloud.to_host()
    assert np.array_equal(gx.view(np.uint64), xyz.view(np.uint64)) and np.array_equal(gi, inten)
    cloud.close()
    E.write_point_cloud_binary(f, xyz[:1000])
    cloud = E.load_point_cloud_binary(f)
    assert cloud.n == 1000 and cloud.intensities is None
    assert np.array_equal(cloud.to_host()[0], xyz[:1000])
    # a truncated file is refused
    raw = f.read_bytes()
    f.write_bytes(raw[:-8])
    with pytest.raises(A.AmhipError):
        E.load_point_cloud_binary(f)


def test_cloud_file_to_geotiff_end_to_end(tmp_path):
    """binary cloud -> HBM -> Dsm -> OrthoBackwardGrid -> image -> GeoTiff, nothing but the file
    reads / writes on the host; the raster equals toImage of the oracle's ortho layer."""
    import torch
    import aerial_mapper_amd as A
    from aerial_mapper_amd import export as E
    sc = S.Scene(80.0, 60.0, 0.25, int(8 * 90 * 70), seed=5, num_frames=8)
    rc, elevation, _ = O.dsm_process(sc.points, sc.grid)
    assert rc == O.OK
    f = tmp_path / "cloud.ampc"
    E.write_point_cloud_binary(f, sc.points)
    g = sc.grid
    with A.AerialGridMap(_settings(A, g), device=0) as m:
        cloud = E.load_point_cloud_binary(f)
        A.Dsm(A.DsmSettings(), m).process(cloud.xyz, m)
        got_elev = m.get("elevation")
        S.assert_dsm_close(got_elev, elevation, tol=1e-4)
        layers = O.new_layers(g)
        layers["elevation"] = got_elev.copy()
        assert O.ortho_process(g, sc.cam, sc.poses, sc.T_C_B, sc.frames, layers) == O.OK
        imgs = torch.from_numpy(np.stack(sc.frames)).to("cuda:0")
        c = sc.cam
        A.OrthoBackwardGrid(A.NCamera(c.fu, c.fv, c.cu, c.cv, c.width, c.height), A.OrthoSettings(),
                            m).process(sc.poses, imgs, m)
        img = E.layer_to_image(m, "ortho", 0.0, 255.0)
    assert np.array_equal(img, X.to_image_u8(layers["ortho"], 0.0, 255.0))
    tif = tmp_path / "ortho.tif"
    E.to_geotiff(img, (0.0, 0.0), tif)
    tags, px = X.read_tiff(tif.read_bytes())
    assert np.array_equal(px, img) and X.geokeys(tags)[3072] == 32632
