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
