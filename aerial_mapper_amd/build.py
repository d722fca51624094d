"""Build libaerial_mapper_hip.so (hand-written HIP kernels + C ABI) for gfx950.

Usage:  python -m aerial_mapper_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the CPU-only build
container; the resulting .so lives in-tree (aerial_mapper_amd/lib/) and travels
to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libaerial_mapper_hip.so")
SHIM_PATH = os.path.join(LIB_DIR, "libaerial_mapper_shim.so")
RESOURCES_PATH = os.path.join(LIB_DIR, "kernel_resources.txt")
OBJ_DIR = os.path.join(ROOT, "build", "hip_obj")

HIP_SOURCES = ["amhip_api.hip", "amhip_sort.hip", "amhip_dsm.hip", "amhip_ortho.hip", "amhip_densify.hip",
               "amhip_forward.hip", "amhip_io.hip", "amhip_session.hip", "amhip_rectify.hip", "amhip_export.hip",
               "amhip_hostsum.cc",   # (.cc: host-only, the AVX-512 loop of the session's content sums)
               "amhip_tuning.cc",    # (host-only: the tuning knobs' store, amhip_set_tuning / AMHIP_TUNING)
               "amhip_build_id.cc"]  # (host-only: amhip_build_id(), recompiled whenever anything else is)
HIP_HEADERS = ["amhip_common.h", "amhip_tuning.h", "amhip_device.h", "amhip_ortho_fold.h", "amhip_pow5_table.h", "amhip_atan_cr.h", "amhip_atan_table.h", "amhip_content_sum.h", os.path.join(ROOT, "include", "aerial_mapper_hip.h")]

# -ffp-contract=off: every decision of the path (inside-radius test, image-box
# test, pixel rounding, best-view comparison) must see the same doubles as the
# reference's baseline-x86-64 build, i.e. no fused multiply-add unless written
# explicitly with fma().
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def source_build_id(extra_flags=()):
    """First 16 hex digits of the SHA-256 over the library's sources, headers and flags."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, s) for s in HIP_SOURCES if s != "amhip_build_id.cc"] + \
        [f if os.path.isabs(f) else os.path.join(CSRC, f) for f in HIP_HEADERS]
    for f in sorted(files):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    h.update(" ".join(list(HIPCC_FLAGS) + list(extra_flags)).encode())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_hip(force=False, verbose=False):
    """One object per .hip source (compiled in parallel, only the stale ones), then one link.
    No device code crosses a translation unit, so no -fgpu-rdc."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    # AMHIP_BUILD_DEFINES="-DAMHIP_TIMING_PROBES": lab builds only (timing probes that give
    # wrong heights, guard-threshold knobs); the shipped library is built without
    extra = os.environ.get("AMHIP_BUILD_DEFINES", "").split()
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    flags_now = " ".join(HIPCC_FLAGS + extra)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    jobs = []
    build_id = source_build_id(extra)
    id_stamp = os.path.join(OBJ_DIR, "build_id.txt")
    id_changed = not os.path.exists(id_stamp) or open(id_stamp).read() != build_id
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s + ".o")
        if force or _stale(obj, [src] + hdrs) or not os.path.exists(obj + ".remarks") or \
                (s == "amhip_build_id.cc" and id_changed):
            jobs.append((src, obj))
    objs = [os.path.join(OBJ_DIR, s + ".o") for s in HIP_SOURCES]
    if not jobs and not _stale(LIB_PATH, objs) and os.path.exists(RESOURCES_PATH):
        return LIB_PATH

    def compile_one(job):
        src, obj = job
        host_only = src.endswith(".cc")
        cmd = [_hipcc()] + [f for f in HIPCC_FLAGS if f != "-shared" and
                            not (host_only and f.startswith("--offload-arch"))] + extra + [
            "-c"] + ([] if host_only else ["-Rpass-analysis=kernel-resource-usage"]) + [
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", obj] + (
            ['-DAMHIP_BUILD_ID="%s"' % build_id] if src.endswith("amhip_build_id.cc") else []) + (
            ["-x", "c++"] if host_only else []) + [src]   # (hipcc takes any source for HIP otherwise)
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        # the compiler's per-kernel register / spill / occupancy remarks are kept next to the
        # library (tests/test_kernel_resources.py holds the hot kernels to their budgets)
        remarks, other = [], []
        import re as _re
        after_remark = False
        for line in res.stderr.splitlines():
            if "-Rpass-analysis=kernel-resource-usage" in line:
                remarks.append(line)
                after_remark = True
            elif after_remark and _re.match(r"^\s*\d*\s*\|", line):
                pass   # (the remark's source excerpt and caret)
            else:
                other.append(line)
                after_remark = False
        if other:
            sys.stderr.write("\n".join(other) + "\n")
        if res.returncode != 0:
            raise subprocess.CalledProcessError(res.returncode, cmd)
        with open(obj + ".remarks", "w") as fh:
            fh.write("\n".join(remarks) + "\n")

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(compile_one, jobs))
    open(stamp, "w").write(flags_now)
    open(id_stamp, "w").write(build_id)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(RESOURCES_PATH, "w") as fh:
        for o in objs:
            fh.write(open(o + ".remarks").read())
    return LIB_PATH


def build_shim(force=False, verbose=False):
    """The drop-in C++ classes (dsm::Dsm, ortho::OrthoBackwardGrid) over the C ABI,
    compiled against the compat headers (the real grid_map / aslam / minkindr /
    OpenCV are absent from this image)."""
    cpp = os.path.join(PKG, "cpp")
    srcs = [os.path.join(cpp, f) for f in sorted(os.listdir(cpp)) if f.endswith(".cc")] \
        if os.path.isdir(cpp) else []
    if not srcs:
        return None
    inc = os.path.join(ROOT, "include")
    deps = list(srcs)
    for base, _, files in os.walk(inc):
        deps += [os.path.join(base, f) for f in files]
    build_hip(force=False, verbose=verbose)
    if not force and not _stale(SHIM_PATH, deps + [LIB_PATH]):
        return SHIM_PATH
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-O2", "-std=c++11", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
           "-Wall", "-fvisibility-inlines-hidden", "-I" + inc, "-o", SHIM_PATH] + srcs + \
          ["-L" + LIB_DIR, "-laerial_mapper_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # Like the reference, three headers define a struct ortho::Settings; that is
    # only safe while none of its (implicit) members is emitted out of line.
    nm = subprocess.run(["nm", SHIM_PATH], stdout=subprocess.PIPE, universal_newlines=True).stdout
    if "_ZN5ortho8Settings" in nm:
        raise RuntimeError("ortho::Settings member emitted out of line in the shim: ODR clash")
    return SHIM_PATH


def build_all(force=False, verbose=False):
    out = [build_hip(force, verbose)]
    s = build_shim(force, verbose)
    if s:
        out.append(s)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    for p in build_all(force=force, verbose=True):
        print("built", p)
