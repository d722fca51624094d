"""GPU: the HIP path against the committed golden fixtures (written in the CPU container by
the reference's own translation units compiled against oracle/refkit/ -- tests/golden/
make_golden.py)."""
import numpy as np
import pytest

import golden_io as G
import scenarios as S

pytestmark = pytest.mark.gpu


def _map(A, d):
    lx, ly, res, px, py = [float(v) for v in d["grid"]]
    return A.AerialGridMap(A.GridMapSettings(px, py, lx, ly, res))


@pytest.mark.parametrize("mode", ["default", "fast"])
@pytest.mark.parametrize("name", G.names("dsm"))
def test_hip_dsm_matches_golden(name, mode):
    """default = AMHIP_DSM_EXACT (what a new map / the drop-in classes start in): the reference's
    floats, >= 99.9 % of the cells bit for bit (only the order of the double sums may move a
    value that sits on a float rounding boundary); fast (opt-in): within the contract's 1e-4 m."""
    import aerial_mapper_amd as A
    d = G.load(name)
    with _map(A, d) as m:
        if mode == "fast":
            m.set_dsm_precision(False)
        if d["elevation_init"].size:
            m.set("elevation", d["elevation_init"])
        st = A.DsmSettings(int(d["radius_sq"]), False, float(d["center_easting"]),
                           float(d["center_northing"]))
        A.Dsm(st, m).process(d["points"], m)
        got = m.get("elevation")
    same = S.assert_dsm_close(got, d["elevation"], tol=1e-4 if mode == "fast" else 1e-6)
    if mode == "default":
        assert same >= 0.999, same


@pytest.mark.parametrize("name", G.names("ortho"))
def test_hip_ortho_matches_golden(name):
    import aerial_mapper_amd as A
    d = G.load(name)
    cam = G.camera_of(d)
    with _map(A, d) as m:
        m.set("elevation", d["elevation"])
        m.set("num_observations", np.full_like(d["elevation"], float(d["num_observations_init"])))
        nc = A.NCamera(cam.fu, cam.fv, cam.cu, cam.cv, cam.width, cam.height, cam.distortion,
                       tuple(cam.dist), d["T_C_B"])
        mosaic = A.OrthoBackwardGrid(nc, A.OrthoSettings(colored_ortho=bool(d["colored"])), m)
        frames = [np.ascontiguousarray(f) for f in d["frames"]]
        for lo, hi in d["batches"]:
            mosaic.process(d["T_G_B"][lo:hi], frames[lo:hi], m)
        got = {n: m.get(n) for n in G.ORTHO_LAYERS}
    S.assert_layers_equal(got, d, G.ORTHO_LAYERS)


@pytest.mark.parametrize("name", G.names("pcl"))
def test_hip_from_pcl_matches_golden(name):
    import aerial_mapper_amd as A
    d = G.load(name)
    with _map(A, d) as m:
        st = A.OrthoFromPclSettings(interpolation_radius=int(d["radius_sq"]),
                                    use_adaptive_interpolation=bool(d["adaptive"]))
        A.OrthoFromPcl(st).process(d["points"], d["intensities"], m)
        got = m.get("ortho")
    err = np.abs(got.astype(np.float64) - d["ortho"].astype(np.float64)).max()
    assert err <= 1e-4, err


@pytest.mark.parametrize("name", G.names("densify"))
def test_hip_densify_matches_golden(name):
    """The fixture was written by the reference's own densifier.cpp (tests/golden/make_golden.py)."""
    import torch
    import aerial_mapper_amd as A
    d = G.load(name)
    with A.AerialGridMap(A.GridMapSettings(0, 0, 8, 8, 1.0)) as m:
        pts, inten = A.densify(m, torch.from_numpy(d["disparity"]).cuda(),
                               torch.from_numpy(d["image_left"]).cuda(), d["K"], float(d["baseline"]),
                               d["R_G_C"], d["t_G_C1"])
        pts, inten = pts.cpu().numpy(), inten.cpu().numpy()
    assert np.array_equal(pts.view(np.uint64), d["points"].view(np.uint64))
    assert np.array_equal(inten, d["intensities"])
