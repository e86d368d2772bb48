import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # a fresh checkout has no built libraries (they are git-ignored): build them once, the way the
    # driver's build check does (hipcc cross-compiles without a GPU)
    if not os.path.exists(os.path.join(ROOT, "rdis_amd", "lib", "librdis_hip.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_golden.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def gctx():
    """one rdis_hip context on cuda:0; raises (loudly) if the HIP library or GPU is missing"""
    from rdis_amd import capi
    return capi.Context(0)
