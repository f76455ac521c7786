"""Pins the oracle (oracle/oracle.cpp) to the reference.

The committed sha256 values are those of the UNMODIFIED reference's P3 output (SURVEY.md 8(c),
re-measured by tests/golden/make_golden.py through oracle/_ref).  The oracle, fed by the product's
host ingest and written out by the product's PPM writer, must reproduce them byte for byte, and
must make exactly as many Scene::intersect calls as the reference.
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt
from tests import oracle_lib, scene_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref")

CASES = [("binary_primitives", None, None), ("blinn_phong", None, None), ("mirror_glass_cylinder", None, None),
         ("clutter_500", None, 0), ("clutter_500", None, 1), ("textured", None, None), ("simple_phong", None, None),
         ("tone_mapping", None, None), ("glass", None, None), ("mirror_image", None, None),
         ("mirror_glass_cylinder", "reinhard", None), ("mirror_glass_cylinder", "ward", None),
         ("mirror_glass_cylinder", "uncharted2", None)]


def sha_of(rgb, tmp_path):
    p = os.path.join(tmp_path, "o.ppm")
    rt.write_ppm(p, rgb)
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


@pytest.mark.parametrize("name,tonemap,bvh", CASES)
def test_oracle_ppm_is_byte_identical_to_reference(name, tonemap, bvh, ref_outputs, tmp_path):
    sf = scene_util.load(name, tonemap=tonemap)
    desc = sf.desc
    if bvh is not None:
        desc.use_bvh = bvh  # the reference renders clutter_500 identically with and without the BVH
    rgb, _, cnt, _ = oracle_lib.render(desc, sf.camera, sf.params, want_hit=False)
    key = name + (":" + tonemap if tonemap else "")
    assert sha_of(rgb, str(tmp_path)) == ref_outputs[key]["sha256"]
    if "rays" in ref_outputs[key]:
        assert cnt["rays"] == ref_outputs[key]["rays"]


def test_oracle_traversal_counts_match_survey():
    # 29.0 box tests, 2.85 sphere + 0.61 triangle tests per ray, 511 nodes for 483 prims
    sf = scene_util.load("clutter_500")
    desc = sf.desc
    desc.use_bvh = 1
    _, _, cnt, _ = oracle_lib.render(desc, sf.camera, sf.params, want_hit=False)
    assert abs(cnt["boxes"] / cnt["rays"] - 29.0) < 0.1
    assert abs(cnt["prims"] / cnt["rays"] - (2.85 + 0.61)) < 0.02
    order = np.zeros(desc.n_prims, dtype=np.int32)
    n_nodes = np.zeros(1, dtype=np.int32)
    k = oracle_lib.load().oracle_bvh_leaf_order(desc, order.ctypes.data, n_nodes.ctypes.data)
    assert k == 483 and int(n_nodes[0]) == 511


@pytest.mark.skipif(not (os.path.isdir("/root/reference/src") and os.path.exists(REF_BIN)),
                    reason="container only: the reference (and oracle/_ref, built from it) does not travel to the GPU box")
def test_committed_hashes_are_what_the_reference_binary_produces(ref_outputs, tmp_path):
    out = os.path.join(str(tmp_path), "r.ppm")
    subprocess.run([REF_BIN, "scenes/mirror_glass_cylinder.json", out], cwd=ROOT, check=True,
                   stdout=subprocess.DEVNULL)
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == ref_outputs["mirror_glass_cylinder"]["sha256"]


def test_golden_hit_maps_are_the_oracles(golden_hits):
    sf = scene_util.load("mirror_glass_cylinder", 300, 200)
    _, hit, _, _ = oracle_lib.render(sf.desc, sf.camera, sf.params)
    assert np.array_equal(hit, golden_hits["mirror_glass_cylinder:bvh1"].astype(np.int32))
