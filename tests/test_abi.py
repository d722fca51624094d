"""CPU tests of the C-ABI library: it loads, exports every symbol the header
declares, its host-side helpers agree with the oracle, and it fails loudly
without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L(hip_built):
    from aerial_mapper_amd import hip_lib
    hip_lib.load()
    return hip_lib


def test_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "aerial_mapper_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(amhip_[a-z_A-Z0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(L.EXPORTS) == declared


def test_struct_layouts_match_oracle(L):
    assert C.sizeof(L.GridDesc) == C.sizeof(O.Grid) == 48
    assert C.sizeof(L.Camera) == C.sizeof(O.Camera) == 80
    assert L.load().amhip_abi_version() == L.ABI_VERSION


def test_geometry_helpers_match_oracle(L):
    for args in [(2500.0, 2500.0, 0.25, 0.0, 0.0), (200.3, 199.6, 0.5, 464980.0, 5272690.0),
                 (10.2, 7.6, 0.3, -3.0, 2.0)]:
        a = L.make_grid(*args)
        b = O.make_grid(*args)
        assert bytes(a) == bytes(b)
        for i, j in [(0, 0), (a.rows - 1, a.cols - 1), (a.rows // 3, a.cols // 2)]:
            assert L.cell_position(a, i, j) == O.cell_position(b, i, j)


def test_pose_composition_matches_oracle(L):
    from aerial_mapper_amd import compose_T_G_C
    rng = np.random.default_rng(0)
    T = rng.normal(size=(16, 7))
    T[:, 3:] /= np.linalg.norm(T[:, 3:], axis=1, keepdims=True)
    tcb = rng.normal(size=7)
    tcb[3:] /= np.linalg.norm(tcb[3:])
    assert np.array_equal(compose_T_G_C(T, tcb), O.compose_T_G_C(T, tcb))


def test_fails_loudly_without_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import aerial_mapper_amd as A
    with pytest.raises(A.AmhipError) as ei:
        A.AerialGridMap(A.GridMapSettings(0, 0, 10, 10, 1.0))
    assert ei.value.status == L.ERR_NO_DEVICE
    # the widened entry points have no CPU path either
    with pytest.raises(A.AmhipError) as ei:
        A.OrthoForwardHomography(A.NCamera(100.0, 100.0, 31.5, 23.5, 64, 48),
                                 A.OrthoForwardHomographySettings(width_mosaic_pixels=32,
                                                                  height_mosaic_pixels=32))
    assert ei.value.status == L.ERR_NO_DEVICE
    from aerial_mapper_amd import io as AIO
    with pytest.raises(A.AmhipError) as ei:
        AIO.parse_point_cloud_text(b"1 2 3 4\n")
    assert ei.value.status == L.ERR_NO_DEVICE


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "aerial_mapper_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                src = open(os.path.join(base, f), errors="replace").read()
                assert "oracle_ffi" not in src and "liboracle" not in src and \
                    "amo_" not in src.replace("amo_compat.h", ""), os.path.join(base, f)


def test_argument_errors_are_reported_without_a_gpu(L):
    """Null / empty arguments fail with AMHIP_ERR_ARG before any device work
    (the reference's CHECK(map), CHECK(!T_G_Bs.empty()), ...)."""
    lib = L.load()
    assert lib.amhip_dsm_process_dev(None, None, 10, 1, 0.0, 0.0) == L.ERR_ARG
    assert b"null context" in lib.amhip_last_error()
    assert lib.amhip_dsm_process(None, None, 10, 1, 0.0, 0.0, None) == L.ERR_ARG
    assert lib.amhip_ortho_backward_process_dev(None, None, None, 0, None, 0, 0, 1, 0) == L.ERR_ARG
    assert lib.amhip_ortho_from_pcl_process_dev(None, None, None, 0, 2, 0) == L.ERR_ARG
    assert lib.amhip_layer_upload(None, 0, None) == L.ERR_ARG
    assert lib.amhip_ctx_synchronize(None) == L.ERR_ARG
    assert lib.amhip_layer_device_ptr(None, 1) is None
    g = L.make_grid(10.0, 10.0, 1.0)
    h = C.c_void_p()
    assert lib.amhip_ctx_create_window(C.byref(g), 5, 5, 10, 10, 0, C.byref(h)) in \
        (L.ERR_ARG, L.ERR_NO_DEVICE)
    bad = L.GridDesc()
    assert lib.amhip_ctx_create(C.byref(bad), 0, C.byref(h)) == L.ERR_ARG
    assert lib.amhip_kernel_name(3) == b"k_dsm_gather"
    assert lib.amhip_mosaic_batch_dev(None, None, 0, None, 0, 0, 1) == L.ERR_ARG
    assert lib.amhip_mosaic_update(None, None, None, 0, 1, None, None) == L.ERR_ARG
    assert lib.amhip_mosaic_create(None, None, 0, C.byref(h)) == L.ERR_ARG
    n = C.c_size_t()
    assert lib.amhip_io_parse_point_cloud_text(0, b"1 2 3 4", 7, None, None, C.byref(n), None) == L.ERR_ARG
    # the homography helper is host arithmetic: usable (and checked) without a GPU
    desc = L.MosaicDesc()
    desc.width_mosaic_pixels, desc.height_mosaic_pixels, desc.ground_plane_elevation_m = 200, 120, 400.0
    cam = O.Camera()
    cam.fu = cam.fv = 70.0
    cam.cu, cam.cv, cam.width, cam.height = 47.5, 26.5, 96, 54
    T = np.array([11.0, 7.0, 470.0, 0.0, 1.0, 0.0, 0.0])
    M = np.zeros(9)
    f64p = C.POINTER(C.c_double)
    lcam = L.Camera.from_buffer_copy(bytes(cam))
    assert lib.amhip_mosaic_homography(C.byref(desc), C.byref(lcam), T.ctypes.data_as(f64p), 1,
                                       M.ctypes.data_as(f64p)) == L.OK
    odesc = O.mosaic_desc(200, 120, 400.0)
    rc, want = O.fwd_homography(cam, odesc, T, True)
    assert rc == O.OK and np.array_equal(M.reshape(3, 3).view(np.uint64), want.view(np.uint64))


def test_every_entry_point_is_mapped_in_integration_md():
    """INTEGRATION.md says for every exported function which reference lines it replaces."""
    import re
    from aerial_mapper_amd import hip_lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    # rows abbreviate families: `amhip_layer_upload/_download/_device_ptr`, `amhip_x` / `_dev`
    missing = []
    for name in hip_lib.EXPORTS:
        if name in doc:
            continue
        parts = name.split("_")
        found = False
        for cut in range(2, len(parts)):
            stem, tail = "_".join(parts[:cut]), "_" + "_".join(parts[cut:])
            if re.search(re.escape(stem) + r"[a-z_]*`?[^|\n]*" + re.escape(tail) + r"\b", doc):
                found = True
                break
        if not found:
            missing.append(name)
    assert not missing, missing


def test_catkin_package_builds_the_same_sources():
    """catkin/aerial_mapper_hip/CMakeLists.txt compiles the sources aerial_mapper_amd/build.py
    compiles (a new .hip / .cc file must reach the workspace build too)."""
    from aerial_mapper_amd import build
    cm = open(os.path.join(ROOT, "catkin", "aerial_mapper_hip", "CMakeLists.txt")).read()
    for src in build.HIP_SOURCES:
        assert "/" + src in cm, src
    cpp = os.path.join(ROOT, "aerial_mapper_amd", "cpp")
    for f in sorted(os.listdir(cpp)):
        if f.endswith(".cc"):
            assert f in cm or "*.cc" in cm or "GLOB" in cm, f


def test_catkin_package_configures_and_emits_a_sound_hipcc_command(tmp_path):
    """A real CMake configure of catkin/aerial_mapper_hip (catkin_simple replaced by a three-macro
    shim: there is no ROS in the image) and a dry run of the Makefile it generates: the hipcc
    command must reach the shell with nothing for it to mangle (an escaped-quote define
    lost its quotes on the way and broke amhip_build_id.cc)."""
    import shutil
    import subprocess
    if not shutil.which("cmake") or not shutil.which("make"):
        pytest.skip("cmake / make not available")
    shim = tmp_path / "catkin_simple"
    shim.mkdir()
    (shim / "catkin_simpleConfig.cmake").write_text(
        "set(CATKIN_DEVEL_PREFIX ${CMAKE_BINARY_DIR}/devel)\n"
        "set(CATKIN_PACKAGE_LIB_DESTINATION lib)\n"
        "set(CATKIN_GLOBAL_INCLUDE_DESTINATION include)\n"
        "file(MAKE_DIRECTORY ${CATKIN_DEVEL_PREFIX}/lib)\n"
        "macro(catkin_simple)\nendmacro()\n"
        "macro(cs_add_library name)\n  add_library(${name} SHARED ${ARGN})\nendmacro()\n"
        "macro(cs_install)\nendmacro()\nmacro(cs_export)\nendmacro()\n")
    bld = tmp_path / "build"
    cfg = subprocess.run(["cmake", "-S", os.path.join(ROOT, "catkin", "aerial_mapper_hip"), "-B", str(bld),
                          "-G", "Unix Makefiles", "-Dcatkin_simple_DIR=%s" % shim,
                          "-DAERIAL_MAPPER_AMD_ROOT=%s" % ROOT],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert cfg.returncode == 0, cfg.stdout[-3000:]
    dry = subprocess.run(["make", "-C", str(bld), "-n", "aerial_mapper_hip_kernels"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert dry.returncode == 0, dry.stdout[-3000:]
    cmds = [ln for ln in dry.stdout.splitlines() if "hipcc" in ln and "--offload-arch=gfx950" in ln]
    assert len(cmds) == 1, dry.stdout[-3000:]
    cmd = cmds[0]
    # nothing the shell would re-interpret: no quotes, no backslashes, no defines carrying strings
    assert '"' not in cmd and "'" not in cmd and "\\" not in cmd, cmd
    assert "-ffp-contract=off" in cmd and "-DAMHIP_BUILD_ID" not in cmd
    from aerial_mapper_amd import build
    for src in build.HIP_SOURCES:
        assert "/" + src in cmd, src


def test_tuning_knobs_go_through_one_door():
    """amhip_set_tuning / AMHIP_TUNING (include/aerial_mapper_hip.h): known keys only, process-wide,
    and the library's own getenv sites stay the documented handful."""
    import re
    import subprocess
    import sys
    from aerial_mapper_amd import hip_lib
    lib = hip_lib.load()
    assert lib.amhip_get_tuning(b"p3_target", 1536.0) == 1536.0
    hip_lib.set_tuning("p3_target", 48)
    assert lib.amhip_get_tuning(b"p3_target", 1536.0) == 48.0
    hip_lib.set_tuning("p3_target", None)
    assert lib.amhip_get_tuning(b"p3_target", 1536.0) == 1536.0
    assert lib.amhip_set_tuning(b"no_such_knob", 1.0) == hip_lib.ERR_ARG
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from aerial_mapper_amd import hip_lib\n"
            "l = hip_lib.load()\n"
            "print(l.amhip_get_tuning(b'p3_cap', 2048.0), l.amhip_get_tuning(b'sort_one_level', 0.0), "
            "l.amhip_get_tuning(b'p3_target', 7.0), l.amhip_default_dsm_precision())\n" % ROOT)
    env = dict(os.environ, AMHIP_TUNING="p3_cap=64, sort_one_level", AMHIP_DSM_FAST="1")
    env.pop("AMHIP_DSM_EXACT", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                         universal_newlines=True, check=True).stdout.split()
    assert out == ["64.0", "1.0", "7.0", "0"], out
    # every environment variable the shipped sources read
    seen = set()
    for base in ("csrc", "cpp"):
        d = os.path.join(ROOT, "aerial_mapper_amd", base)
        for f in sorted(os.listdir(d)):
            seen |= set(re.findall(r'getenv\("([A-Z_0-9a-z]+)"\)', open(os.path.join(d, f)).read()))
    assert seen == {"AMHIP_TUNING", "AMHIP_DSM_FAST", "AMHIP_DSM_EXACT", "AERIAL_MAPPER_HIP_DEVICE",
                    "AERIAL_MAPPER_HIP_DEVICES"}, seen
    hdr = open(os.path.join(ROOT, "include", "aerial_mapper_hip.h")).read()
    for name in seen:
        assert name in hdr, name


def test_committed_pmc_evidence_is_of_this_build():
    """The newest rocprofv3 counter summaries under profiles/ (what bench.py copies `roofline.traffic`
    and the VALU figures from) carry the build id of the library they were collected from: the SHA-256
    prefix aerial_mapper_amd/build.py computes over the library's sources, headers and flags -- the
    same string amhip_build_id() returns.  The BUILT library must be of these sources (asserted).
    Evidence of another build is not an error of the code: bench.py then prints `traffic: null` with
    the reason (`traffic_stale`) instead of a stale figure -- this test is then SKIPPED with that
    reason, so that a kernel change without a new tools/collect_profiles.sh run stays visible in
    the test report without turning the suite red half-way through a round."""
    import glob
    import json
    import re
    from aerial_mapper_amd import build, hip_lib
    want = build.source_build_id()
    assert hip_lib.build_id() == want, "the built library is not of these sources: rebuild"

    def newest(pattern):
        fs = glob.glob(os.path.join(ROOT, "profiles", pattern))
        return sorted(fs, key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))[-1]
    stale = []
    for mode in ("exact", "fast"):
        for pattern in ("r*_%s_pmc_traffic.json" % mode, "r*_%s_cfg3_pmc_sq.json" % mode):
            f = newest(pattern)
            got = json.load(open(f)).get("build_id")
            assert isinstance(got, str) and re.fullmatch(r"[0-9a-f]{16}", got), (os.path.basename(f), got)
            if got != want:
                stale.append("%s (build %s)" % (os.path.basename(f), got))
    if stale:
        pytest.skip("counter evidence of another build than %s: %s -- bench.py refuses it (traffic: null)"
                    % (want, ", ".join(stale)))
