import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    os.chdir(ROOT)  # texture paths in the scene files are relative to the cwd, as in the reference


@pytest.fixture(scope="session")
def ref_outputs():
    return json.load(open(os.path.join(GOLDEN, "ref_outputs.json")))


@pytest.fixture(scope="session")
def golden_hits():
    return np.load(os.path.join(GOLDEN, "hit_ids_300x200.npz"))


@pytest.fixture(scope="session")
def renderer():
    """One rt_context on cuda:0 for the whole GPU session.  No fallback: fails if the HIP library
    or the device is missing."""
    import raytracer_amd as rt
    r = rt.Renderer(0)
    yield r
    r.close()
