import os
import sys

import pytest

# the engine reads its WANN_* switches once per index; tests flip them between batches on one index, and some force rare
# paths through test-only hooks: both need WANN_TEST_HOOKS=1 (rangefilteredann_amd/csrc/wann_tuning.h)
os.environ.setdefault("WANN_TEST_HOOKS", "1")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def wa():
    import rangefilteredann_amd  # noqa: F401  (raises loudly when the extension is not built)
    import window_ann
    return window_ann


@pytest.fixture(scope="session")
def gpu(wa):
    if wa.device_count() < 1:
        pytest.fail("gpu-marked test started without a usable gfx950 device")
    return 0
