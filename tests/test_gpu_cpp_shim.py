"""GPU: the drop-in C++ classes (dsm::Dsm, ortho::OrthoBackwardGrid,
ortho::OrthoForwardHomography) driven like the reference's demos, checked
against the oracle by tests/cpp/shim_parity.cc / shim_forward_parity.cc."""
import os
import subprocess

import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path_factory, source):
    from aerial_mapper_amd import build
    build.build_all()
    O.build()
    out = str(tmp_path_factory.mktemp("shim") / source[:-3])
    lib = os.path.join(ROOT, "aerial_mapper_amd", "lib")
    cmd = ["g++", "-O2", "-std=c++11", "-pthread", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
           os.path.join(ROOT, "tests", "cpp", source), "-o", out,
           "-L" + lib, "-laerial_mapper_shim", "-laerial_mapper_hip",
           "-L" + os.path.join(ROOT, "oracle"), "-loracle",
           "-Wl,-rpath," + lib, "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)
    return out


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    return _compile(tmp_path_factory, "shim_parity.cc")


@pytest.fixture(scope="module")
def exe_forward(tmp_path_factory):
    return _compile(tmp_path_factory, "shim_forward_parity.cc")


@pytest.mark.parametrize("mode", ["gray", "colored"])
@pytest.mark.parametrize("devices", [None, "0,0", "0,0,0,0,0,0"])
def test_cpp_dropin_classes_match_oracle(exe, mode, devices):
    """devices: AERIAL_MAPPER_HIP_DEVICES -- ONE host process, the map cut into one window per
    listed device (here all on device 0; on a node: different GPUs, halo points over xGMI).
    Dsm and OrthoBackwardGrid share the map's session: the layers stay on the device(s)."""
    env = dict(os.environ)
    env.pop("AERIAL_MAPPER_HIP_DEVICES", None)
    if devices:
        env["AERIAL_MAPPER_HIP_DEVICES"] = devices
    r = subprocess.run([exe, mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                       env=env)
    print(r.stdout.decode())
    assert r.returncode == 0, r.stdout.decode()[-2000:]


@pytest.mark.parametrize("mode", [("batch", "gray"), ("batch", "colored"),
                                  ("incremental", "gray")])
def test_cpp_forward_homography_class_matches_oracle(exe_forward, mode):
    r = subprocess.run([exe_forward] + list(mode), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    print(r.stdout.decode())
    assert r.returncode == 0, r.stdout.decode()[-2000:]


def test_cpp_io_loaders_match_the_reference_loops(tmp_path_factory, tmp_path):
    exe_io = _compile(tmp_path_factory, "shim_io_parity.cc")
    r = subprocess.run([exe_io, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=300)
    print(r.stdout.decode())
    assert r.returncode == 0, r.stdout.decode()[-2000:]
