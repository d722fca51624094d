"""GPU: amhip_rectify_stereo_pair_dev (stereo::Rectifier::rectifyStereoPair + computeMask,
rectifier.cpp:34-128) bit for bit against the oracle -- maps (float), both remapped images,
mask, rectified rotation, baseline; and, where it was built, against the reference's own
rectifier.cpp compiled over oracle/refkit.  Then the whole front of the dense pipeline stays on
the GPU: rectify -> (a synthetic disparity) -> densify -> Dsm."""
import numpy as np
import pytest

import oracle_ffi as O
from test_oracle_rectify import rig

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,W,H", [(11, 160, 120), (12, 752, 480), (13, 333, 211)])
def test_gpu_rectifier_is_bit_exact(seed, W, H):
    import torch
    import aerial_mapper_amd as A
    K, R1, R2, t1, t2, left, right = rig(seed, W=W, H=H)
    which = "loops" if O.have_loops() else "port"
    rc, want = O.rectify_stereo_pair(K, R1, R2, t1, t2, left, right, which=which)
    assert rc == O.OK
    with A.AerialGridMap(A.GridMapSettings(0.0, 0.0, 32.0, 32.0, 1.0)) as m:
        # (row steps larger than the width: views of wider rasters)
        wide_l = torch.zeros((H, W + 24), dtype=torch.uint8, device="cuda")
        wide_r = torch.zeros((H, W + 8), dtype=torch.uint8, device="cuda")
        wide_l[:, :W] = torch.from_numpy(left).cuda()
        wide_r[:, :W] = torch.from_numpy(right).cuda()
        got = A.rectify_stereo_pair(m, K, R1, R2, t1, t2, wide_l[:, :W], wide_r[:, :W], want_maps=True)
    assert got["baseline"] == want["baseline"]
    assert np.array_equal(got["R_G_C"], want["R_G_C"])
    assert np.array_equal(got["maps"].cpu().numpy().view(np.uint32), want["maps"].view(np.uint32))
    assert np.array_equal(got["image_left"].cpu().numpy(), want["left"])
    assert np.array_equal(got["image_right"].cpu().numpy(), want["right"])
    assert np.array_equal(got["mask"].cpu().numpy(), want["mask"])
    assert 0.2 < (want["mask"] == 255).mean() < 1.0


def test_rectify_densify_dsm_stays_on_the_gpu():
    import torch
    import aerial_mapper_amd as A
    K, R1, R2, t1, t2, left, right = rig(14, W=320, H=240)
    with A.AerialGridMap(A.GridMapSettings(12.0, -4.0, 160.0, 120.0, 0.5)) as m:
        r = A.rectify_stereo_pair(m, K, R1, R2, t1, t2, torch.from_numpy(left).cuda(),
                                  torch.from_numpy(right).cuda())
        # the block matcher (OpenCV) is outside the path: a plane of constant disparity stands in
        disp = torch.full((240, 320), 11.0, dtype=torch.float32, device="cuda")
        disp[r["mask"] == 0] = -1.0
        pts, inten = A.densify(m, disp, r["image_left"], K, r["baseline"], r["R_G_C"], t1)
        assert pts.shape[0] > 1000 and pts.is_cuda
        A.Dsm(A.DsmSettings(1), m).process(pts, m)
        elev = m.get("elevation")
    assert (~np.isnan(elev)).sum() > 100
