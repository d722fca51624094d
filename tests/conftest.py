import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


# `-m gpu` on a box without a GPU fails loudly (AMHIP_ERR_NO_DEVICE), it is never
# skipped silently; without `-m gpu` the marker expression deselects those tests.


@pytest.fixture(scope="session")
def hip_built():
    from aerial_mapper_amd import build
    build.build_hip()
    return build.LIB_PATH


# Tuning knobs (include/aerial_mapper_hip.h: amhip_set_tuning / AMHIP_TUNING).
def tuning_env(base=None, **knobs):
    """environment for a CHILD process with AMHIP_TUNING="key=value,..." (merged into `base`)"""
    env = dict(os.environ if base is None else base)
    items = [s for s in env.get("AMHIP_TUNING", "").split(",") if s]
    items += ["%s=%s" % (k, v) for k, v in knobs.items()]
    if items:
        env["AMHIP_TUNING"] = ",".join(items)
    return env


@pytest.fixture
def tuning():
    """tuning(key=value, ...) sets process-wide knobs of the loaded library for this test (None clears
    one); everything it touched is cleared again afterwards."""
    from aerial_mapper_amd import hip_lib
    touched = set()

    def setter(**knobs):
        for k, v in knobs.items():
            hip_lib.set_tuning(k, v)
            touched.add(k)
    yield setter
    for k in touched:
        hip_lib.set_tuning(k, None)
