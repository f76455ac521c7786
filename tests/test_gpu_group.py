"""rt_group_*: one frame over several GPUs from one process (SURVEY.md 8(b) "create context (takes device
list)", 8(e)).  On the one-GPU box the device list names device 0 more than once: the members are separate contexts
with their own streams and buffers, the partition, the strided band copies and the reduction are the ones an 8-GPU
node runs (there the copies cross xGMI)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt
from tests import oracle_lib, scene_util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_whitted_frame_of_a_group_has_the_reference_sha256(devices, ref_outputs, golden_hits, tmp_path):
    g = rt.Group(devices)
    try:
        sf = scene_util.load("mirror_glass_cylinder")  # 1200 x 800 as shipped
        g.load(sf)
        u8, _ = g.render_u8()
        out = tmp_path / "g.ppm"
        rt.write_ppm_u8(out, u8)
        assert hashlib.sha256(out.read_bytes()).hexdigest() == ref_outputs["mirror_glass_cylinder"]["sha256"]
        assert g.stats().rays == ref_outputs["mirror_glass_cylinder"]["rays"]
        sf = scene_util.load("mirror_glass_cylinder", 300, 200)
        g.load(sf)
        rgb, hit = g.render()
        assert np.array_equal(hit, golden_hits["mirror_glass_cylinder:bvh1"].astype(np.int32))
        orgb, _, _, _ = oracle_lib.render(sf.desc, sf.camera, sf.params)
        assert np.abs(rgb - orgb).max() <= 1e-12
    finally:
        g.close()


def test_group_bands_with_a_ragged_last_band_and_more_members_than_bands(renderer):
    for devices, (w, h) in (([0, 0, 0], (150, 101)), ([0, 0, 0, 0, 0], (64, 20))):  # 101 = 12 bands + 5 rows; 20 rows = 3 bands for 5 members
        g = rt.Group(devices)
        try:
            sf = scene_util.load("clutter_500", w, h)
            g.load(sf)
            rgb, hit = g.render()
            u8, hit8 = g.render_u8(want_hit=True)
            renderer.load(sf)
            one, one_hit = renderer.render()
            assert np.array_equal(rgb, one) and np.array_equal(hit, one_hit) and np.array_equal(hit8, one_hit)
            assert np.array_equal(u8, (255.999 * one).astype(np.int32).astype(np.uint8))
            assert g.stats().rays == renderer.stats().rays
        finally:
            g.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_path_traced_frame_of_a_group_is_the_one_device_frame(renderer, devices):
    """Strata ranges per member, sums added on the first device before the non-linear stages: equal to the
    one-context frame up to the order of the fp additions."""
    g = rt.Group(devices)
    try:
        sf = scene_util.load("pathtrace", 96, 64)
        P = sf.params
        P.pixel_samples, P.light_samples = 16, 4
        g.load(sf, P)
        rgb, hit = g.render()
        renderer.load(sf, P)
        one, _ = renderer.render(want_hit=False)
        assert np.abs(rgb - one).max() < 1e-12
        assert (hit == -1).all()
        assert g.stats().rays == renderer.stats().rays
        u8, _ = g.render_u8()
        assert np.abs(u8.astype(int) - (255.999 * one).astype(np.int32)).max() <= 1  # a byte may flip on a 1e-16 difference
        # a sample count that does not split evenly, and more members than strata
        P.pixel_samples = 4
        g5 = rt.Group([0] * 5)
        g5.load(sf, P)
        a, _ = g5.render(want_hit=False)
        renderer.set_params(P)
        b, _ = renderer.render(want_hit=False)
        assert np.abs(a - b).max() < 1e-12
        g5.close()
    finally:
        g.close()


def test_group_rejects_partitioned_params():
    g = rt.Group([0, 0])
    try:
        sf = scene_util.load("blinn_phong", 64, 48)
        P = sf.params
        P.band_world = 2
        with pytest.raises(rt.RtError, match="partitions the frame itself"):
            g.load(sf, P)
    finally:
        g.close()


def test_cli_splits_the_frame_over_the_listed_devices(ref_outputs, tmp_path):
    exe = os.path.join(ROOT, "raytracer_amd", "csrc", "raytracer")
    out = tmp_path / "o.ppm"
    res = subprocess.run([exe, os.path.join("scenes", "clutter_500.json"), str(out)], cwd=ROOT, capture_output=True, text=True,
                         timeout=120, env=dict(os.environ, RT_DEVICES="0,0,0,0"))
    assert res.returncode == 0, res.stderr
    assert hashlib.sha256(out.read_bytes()).hexdigest() == ref_outputs["clutter_500"]["sha256"]


def test_first_frame_of_four_members_is_not_four_sizing_frames_in_a_row(ref_outputs, tmp_path):
    """The first frame of a setup sizes every member's node store, and a sizing frame waits for its GPU: the group drives
    the members from one host thread each for that frame.  First frame = the reference's bytes; and once buffers and code
    objects exist (second setup of the same scene), a first frame takes less than twice a later one."""
    import time
    g = rt.Group([0] * 4)
    try:
        sf = scene_util.load("clutter_500", 1920, 1080)
        small = scene_util.load("clutter_500")
        g.load(small)
        u8, _ = g.render_u8()  # the very first frame of the group
        out = tmp_path / "g.ppm"
        rt.write_ppm_u8(out, u8)
        assert hashlib.sha256(out.read_bytes()).hexdigest() == ref_outputs["clutter_500"]["sha256"]
        g.load(sf)
        ref, _ = g.render_u8()
        firsts, laters = [], []
        for _ in range(3):
            g.load(sf)  # new setup generation: every context sizes its store again
            t0 = time.perf_counter()
            a, _ = g.render_u8()
            t1 = time.perf_counter()
            b, _ = g.render_u8()
            t2 = time.perf_counter()
            assert np.array_equal(a, ref) and np.array_equal(b, ref)
            firsts.append(t1 - t0); laters.append(t2 - t1)
        print("first frames %s later frames %s (s)" % (firsts, laters))
        assert min(firsts) < 2.0 * min(laters)
    finally:
        g.close()
