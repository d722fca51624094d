"""The register / occupancy budgets of the hot kernels, from the compiler's own remarks
(aerial_mapper_amd/lib/kernel_resources.txt, written by every build): code added to a cold
path of a kernel raises the register allocation of the WHOLE kernel, silently (round 2: an
optional mode hooked into cell_global took the FP64 gather from 7 to 4 waves per SIMD)."""
import os
import re

import pytest

from aerial_mapper_amd import build


def _kernels():
    build.build_hip()
    if not os.path.exists(build.RESOURCES_PATH):
        build.build_hip(force=True)
    out, cur = {}, None
    for line in open(build.RESOURCES_PATH):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def _one(kernels, needle):
    hits = {k: v for k, v in kernels.items() if needle in k}
    assert len(hits) == 1, (needle, sorted(hits))
    return next(iter(hits.values()))


BUDGETS = [
    # (mangled-name fragment, min waves/SIMD, max spilled VGPRs)
    ("k_dsm_gather_f32ILi512ELi16ELi1024ELi0E", 8, 0),   # the default DSM gather
    ("k_dsm_gather_f32ILi512ELi16ELi2048ELi0E", 8, 0),
    ("k_dsm_gather_tiledILi512ELi16ELi1024E", 8, 0),      # FP64 mode: the default DSM gather (four workgroups per CU)
    ("k_dsm_p3_countILb0E", 8, 0),
    # (the scatter passes: 70+ KB of LDS per workgroup allow two of them = 4 waves per SIMD; all of a
    # thread's loads are in flight at once, which takes more than 64 registers in the first pass)
    ("k_dsm_p3_scatterILb0E", 4, 0),
    ("k_dsm_p3_scatterILb1E", 4, 0),
    ("14k_dsm_p3_placeE", 4, 0),
    ("20k_dsm_p3_reduce_scanE", 4, 0),                    # (one workgroup of 1024 threads scans: 128 registers at most, none spilled)
    ("k_dsm_p3_scatter_recILb0E", 4, 0),                  # the record pipeline (single-precision mode)
    ("k_dsm_p3_scatter_recILb1E", 4, 0),
    ("18k_dsm_p3_place_recE", 4, 0),
    ("22k_ortho_backward_fast4E", 4, 12),                 # the default mosaic kernel (two cells per lane)
    # denser clouds in single precision: 4096-point images run two workgroups per CU (4 waves per
    # SIMD), the wave-per-block kernel three waves per SIMD
    ("k_dsm_gather_f32_wideILi512ELi16ELi4096E", 4, 0),
    ("k_dsm_gather_f32_listILi512ELi16ELi4096E", 4, 0),
    ("k_dsm_gather_f32_wideILi512ELi16ELi7680E", 2, 0),
    ("18k_dsm_gather_denseILb0E", 3, 0),
    ("18k_dsm_gather_denseILb1E", 3, 26),                 # (cold spills, all outside the candidate loop: DESIGN 4.2)
]


@pytest.mark.parametrize("needle,min_waves,max_spill", BUDGETS)
def test_hot_kernel_budget(needle, min_waves, max_spill):
    k = _one(_kernels(), needle)
    assert k["Occupancy"] >= min_waves, k
    assert k["VGPRs Spill"] <= max_spill, k
