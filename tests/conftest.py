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


def _make(env):
    import raytracer_amd as rt
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)  # RT_TRAVERSAL / RT_PIPELINE are read by rt_create
    try:
        return rt.Renderer(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="session")
def renderer_walk():
    """Wavefront pipeline with the visit-for-visit walk of the reference's BVH (no culling, closest-hit
    shadow rays): its box / primitive test counts must equal the reference's."""
    r = _make({"RT_TRAVERSAL": "walk"})
    yield r
    r.close()


@pytest.fixture(scope="session")
def renderer_levels():
    """The same level-synchronous evaluation as grid-wide kernels (one per phase and depth)."""
    r = _make({"RT_PIPELINE": "levels"})
    yield r
    r.close()


@pytest.fixture(scope="session")
def renderer_mega():
    """The persistent per-pixel kernel (explicit frame stack), an independent second GPU implementation."""
    r = _make({"RT_PIPELINE": "mega", "RT_TRAVERSAL": "walk"})
    yield r
    r.close()


@pytest.fixture(scope="session")
def renderer_literal():
    """Literal restatement of BVHNode::intersect with true divisions, per-pixel kernel."""
    r = _make({"RT_TRAVERSAL": "ref"})
    yield r
    r.close()


@pytest.fixture(scope="session")
def renderer_grid():
    """Per-level dealing over the whole grid: cooperative launch with grid barriers (gridwave_kernels.inc)."""
    r = _make({"RT_PIPELINE": "grid"})
    yield r
    r.close()


@pytest.fixture(scope="session")
def renderer_block():
    """Three-phase (trace | shadow | shade) per-workgroup pipeline for Whitted frames too."""
    r = _make({"RT_PIPELINE": "block"})
    yield r
    r.close()


@pytest.fixture(scope="session")
def renderer_fused_whole():
    """Fused pipeline with level 0 inside the frame kernel (no fw_roots_kernel), also for path tracing."""
    r = _make({"RT_PIPELINE": "fused", "RT_SPLIT_ROOTS": "0"})
    yield r
    r.close()


@pytest.fixture(scope="session")
def renderer():
    """One rt_context on cuda:0 for the whole GPU session.  No fallback: fails if the HIP library
    or the device is missing."""
    import raytracer_amd as rt
    r = rt.Renderer(0)
    yield r
    r.close()
