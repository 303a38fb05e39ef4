import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def built_lib():
    """libpolar_amd.so, (re)built in-tree for gfx950 if the sources are newer."""
    from polar_amd import build
    return build.build()


@pytest.fixture(scope="session")
def oracle_built():
    import oracle_lib
    oracle_lib.build_oracle()
    return True
