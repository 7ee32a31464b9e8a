import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def sift():
    import sift_amd

    sift_amd.lib()
    # The suite runs the queue route (pre-filter, candidate verification) whatever the size of the
    # job; tests of the dense route that small jobs take by default switch it on themselves.
    sift_amd.lib().sift_set_dense_route_pixels(0)
    return sift_amd


# sift_set_dense_route_pixels' default (include/sift_amd.h)
DENSE_ROUTE_DEFAULT = 2200000
