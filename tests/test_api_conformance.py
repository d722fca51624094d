"""The drop-in's public C++ surface IS the reference's: one source full of
static_asserts about namespaces, Settings fields, constructor and member signatures
(tests/cpp/api_conformance.cc) is compiled against the reference's own headers (externals from
oracle/refkit/) and against include/ of this repository.  CPU only, syntax only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SRC = os.path.join(ROOT, "tests", "cpp", "api_conformance.cc")
APIS = ["API_DSM", "API_BACKWARD", "API_FROM_PCL", "API_FORWARD", "API_IO"]


def _compile(includes, api):
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-D" + api] + ["-I" + i for i in includes] + [SRC]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


@pytest.mark.parametrize("api", APIS)
def test_drop_in_headers_satisfy_the_api_statements(api):
    _compile([os.path.join(ROOT, "include")], api)


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")
@pytest.mark.parametrize("api", APIS)
def test_reference_headers_satisfy_the_same_statements(api):
    inc = [os.path.join(ROOT, "oracle", "refkit"), os.path.join(ROOT, "oracle")]
    inc += [os.path.join(REF, d, "include") for d in
            ("aerial_mapper_utils", "aerial_mapper_thirdparty", "aerial_mapper_dsm", "aerial_mapper_ortho",
             "aerial_mapper_io", "aerial_mapper_grid_map")]
    _compile(inc, api)


@pytest.mark.parametrize("camera,refused", [("ok", False), ("fisheye", True), ("unified", True)])
def test_dropin_refuses_camera_models_it_does_not_implement(tmp_path, camera, refused):
    """ADVICE r1 / VERDICT r1 #9: describe_camera() used to map every unknown distortion to
    'none' and never looked at the camera type."""
    from aerial_mapper_amd import build
    build.build_all()
    exe = str(tmp_path / "refusal")
    lib = os.path.join(ROOT, "aerial_mapper_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_camera_refusal.cc"), "-o", exe,
                           "-L" + lib, "-laerial_mapper_hip", "-Wl,-rpath," + lib])
    r = subprocess.run([exe, camera], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=60)
    if refused:
        assert r.returncode != 0 and "NOT REFUSED" not in r.stdout
        assert "FATAL describe_camera" in r.stderr
    else:
        assert r.returncode == 0, r.stdout + r.stderr


SHIM_SOURCES = ["dsm.cc", "ortho-backward-grid.cc", "ortho-from-pcl.cc", "ortho-forward-homography.cc",
                "aerial-mapper-io.cc", "shim_common.cc"]


@pytest.mark.parametrize("source", SHIM_SOURCES)
def test_shim_compiles_in_its_real_dependencies_branch(source):
    """include/aerial-mapper-deps.h picks the REAL grid_map_core / aslam_cv2 / minkindr / Eigen /
    OpenCV headers when they are on the include path (a catkin workspace).  None of them exists
    in this image; oracle/refkit/ holds stand-ins under the externals' own include paths with
    the member names the real classes have, so the AERIAL_MAPPER_REAL_DEPS=1 branch of every
    shim source is at least compiled (syntax only)."""
    src = os.path.join(ROOT, "aerial_mapper_amd", "cpp", source)
    probe = ('#include "aerial-mapper-deps.h"\n#if !AERIAL_MAPPER_REAL_DEPS\n'
             '#error compat branch taken\n#endif\n')
    inc = ["-I" + os.path.join(ROOT, "oracle", "refkit"), "-I" + os.path.join(ROOT, "oracle"),
           "-I" + os.path.join(ROOT, "include")]
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++", "-"] + inc,
                       input=probe.encode(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only"] + inc + [src], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
