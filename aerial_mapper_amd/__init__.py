"""aerial_mapper_amd -- MI355X-native DSM rasterisation + grid-based orthomosaic.

The hot path of ethz-asl/aerial_mapper (dsm::Dsm::process and
ortho::OrthoBackwardGrid::process) as hand-written HIP kernels for gfx950 behind
a C ABI (include/aerial_mapper_hip.h), with host-side mirrors of the reference's
class API.  See DESIGN.md / INTEGRATION.md.
"""
from .hip_lib import (AmhipError, Camera, GridDesc, DIST_EQUIDISTANT, DIST_NONE,  # noqa: F401
                      DIST_RADTAN, LAYER_NAMES, make_grid, cell_position)
from .mapper import (AerialGridMap, Dsm, DsmSettings, GridMapSettings, HostSession, NCamera,  # noqa: F401
                     OrthoBackwardGrid, OrthoForwardHomography, OrthoForwardHomographySettings,
                     OrthoFromPcl, OrthoFromPclSettings, OrthoSettings, compose_T_G_C, densify,
                     rectify_stereo_pair)

__all__ = ["AerialGridMap", "GridMapSettings", "HostSession", "Dsm", "DsmSettings", "OrthoBackwardGrid",
           "OrthoSettings", "OrthoForwardHomography", "OrthoForwardHomographySettings", "OrthoFromPcl", "OrthoFromPclSettings", "NCamera", "compose_T_G_C", "densify", "rectify_stereo_pair", "AmhipError", "Camera", "GridDesc",
           "make_grid", "cell_position", "LAYER_NAMES", "DIST_NONE", "DIST_RADTAN",
           "DIST_EQUIDISTANT"]
