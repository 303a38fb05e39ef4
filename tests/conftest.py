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


@pytest.fixture(scope="module")
def hooks_lib(built_lib):
    """The TEST build of the library (libpolar_amd_test.so: the product's objects with the fault-injection hooks of
    include/polar_amd_debug.h compiled in) for the module that asks for it; the product library again afterwards."""
    import polar_amd
    from polar_amd import build
    path = build.build_test()
    polar_amd.use_library(path)
    yield path
    polar_amd.use_library(None)
