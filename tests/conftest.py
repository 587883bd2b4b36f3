import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `-m gpu` on the GPU box")
    config.addinivalue_line("markers", "slow: GPU tests that spend minutes of HOST time in the CPU oracle or in subprocesses at SD v1-4 size; "
                                       "`-m 'gpu and not slow'` is the quick iteration suite, `-m gpu` (what the driver runs) is everything")


# the SD-v1-4-size tests recompute the CPU oracle live (most of the ~7 min of `-m gpu`); they are tagged here, by name, so that the
# selection `-m gpu` is unchanged and a kernel iteration can run the rest in about a minute
_SLOW = ("sd14", "run_config5_tool", "main_cli", "b16_largest_batch", "txt2img_script")


def pytest_collection_modifyitems(config, items):
    for item in items:
        if item.get_closest_marker("gpu") is not None and any(tag in item.name for tag in _SLOW):
            item.add_marker(pytest.mark.slow)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built_lib():
    """build (if stale) and load libcelebbasis_hip.so"""
    from celebbasis_amd import build as cb_build
    from celebbasis_amd import lib
    cb_build.build(verbose=False)
    return lib.load()
