import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    import harness
    if not harness.have("orc"):
        harness.build_oracle()
    yield


@pytest.fixture(autouse=True)
def _no_context_outlives_its_test():
    """GPU tests: every context a test made is destroyed when the test ends, whether it passed or not (a failed test's locals live on in its
    traceback).  The number of contexts alive on the device decides whether a new context's streams get hardware queues of their own, which the
    persistent all-gather form of avbd_group_step needs: leftovers of one test must not change the schedule of the next."""
    yield
    for name, mod in list(sys.modules.items()):
        if name.replace("-", "_").endswith("demo3d_amd") and hasattr(mod, "close_all_worlds"):
            mod.close_all_worlds()
    import gc
    gc.collect()
