"""Node-store overflow is checked on every frame (it used to be checked only while the store was
being sized).  Debug knobs read by rt_create force the cases:
  RT_WF_DEBUG_SHRINK=f  the capacity measured on the first frame of a setup is replaced, once, by f x what that
                        frame needed -> the NEXT frame overflows although the store counts as sized;
  RT_WF_DEBUG_CAP=n     a setup's first frame starts with n nodes -> the sizing loop has to grow, in path tracing
                        also for later batches of strata.
Blocking entry points must return the correct frame regardless; asynchronous ones report the void frame at
the next call on the context."""
import numpy as np
import pytest

import raytracer_amd as rt
from tests import oracle_lib, scene_util
from tests.conftest import _make

pytestmark = pytest.mark.gpu


def whitted_case():
    sf = scene_util.load("mirror_glass_cylinder", 300, 200)
    P = sf.params
    orgb, ohit, cnt, _ = oracle_lib.render(sf.desc, sf.camera, P)
    return sf, P, orgb, ohit, cnt


def pt_case(spp=16, ls=4):
    sf = scene_util.load("pathtrace", 90, 60)
    P = sf.params
    P.pixel_samples, P.light_samples = spp, ls
    orgb, _, cnt, _ = oracle_lib.render(sf.desc, sf.camera, P, want_hit=False)
    return sf, P, orgb, cnt


@pytest.mark.parametrize("pipeline", ["", "block", "grid"])
def test_blocking_render_recovers_from_an_overflow_after_sizing(pipeline):
    r = _make({"RT_WF_DEBUG_SHRINK": "0.05", "RT_PIPELINE": pipeline} if pipeline else {"RT_WF_DEBUG_SHRINK": "0.05"})
    try:
        sf, P, orgb, ohit, cnt = whitted_case()
        r.load(sf, P)
        for k in range(3):  # frame 0 sizes the store (then shrunk), frame 1 overflows inside and is rendered again
            rgb, hit = r.render()
            assert np.array_equal(hit, ohit), k
            assert np.abs(rgb - orgb).max() <= 1e-12, k
            assert r.stats().rays == cnt["rays"], k
        u8, _ = r.render_u8()
        assert np.array_equal(u8, np.floor(255.999 * orgb).astype(np.uint8))
    finally:
        r.close()


def test_blocking_path_trace_recovers_from_an_overflow_after_sizing():
    r = _make({"RT_WF_DEBUG_SHRINK": "0.05"})
    try:
        sf, P, orgb, cnt = pt_case()
        r.load(sf, P)
        first, _ = r.render(want_hit=False)
        for k in range(2):
            again, _ = r.render(want_hit=False)
            assert np.array_equal(again, first), k
        assert (np.abs(first - orgb).max(axis=2) > 1e-9).mean() <= 1e-3
        assert r.stats().rays == cnt["rays"]
    finally:
        r.close()


@pytest.mark.parametrize("generic", ["0", "1"])
def test_sizing_grows_for_any_batch_of_strata(generic, monkeypatch):
    """A tiny first capacity and a node budget that cuts the strata into several batches: every batch, not only
    the first, may have to grow its arenas, and a batch is added to the frame only once it was valid."""
    monkeypatch.setenv("RT_WF_NODE_BUDGET", "70000")
    r = _make({"RT_WF_DEBUG_CAP": "2048", "RT_PT_GENERIC": generic})
    try:
        sf, P, orgb, cnt = pt_case(spp=36, ls=2)
        r.load(sf, P)
        rgb, _ = r.render(want_hit=False)
        assert (np.abs(rgb - orgb).max(axis=2) > 1e-9).mean() <= 1e-3
        assert r.stats().rays == cnt["rays"]
        sfw, Pw, owrgb, owhit, wcnt = whitted_case()
        r.load(sfw, Pw)
        rgb, hit = r.render()
        assert np.array_equal(hit, owhit) and np.abs(rgb - owrgb).max() <= 1e-12
    finally:
        r.close()


def test_asynchronous_frames_report_a_void_frame_at_the_next_call():
    import torch
    r = _make({"RT_WF_DEBUG_SHRINK": "0.05"})
    try:
        sf, P, orgb, ohit, cnt = whitted_case()
        r.load(sf, P)
        out = torch.zeros(200 * 300 * 3, dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        r.render_device(out.data_ptr(), stream=stream)   # sizes the store (synchronously), capacity then shrunk
        torch.cuda.synchronize()
        assert np.abs(out.cpu().numpy().reshape(200, 300, 3) - orgb).max() <= 1e-12
        r.render_device(out.data_ptr(), stream=stream)   # overflows; nobody has looked yet
        torch.cuda.synchronize()
        with pytest.raises(rt.RtError, match="earlier frame"):
            r.render_device(out.data_ptr(), stream=stream)
        r.render_device(out.data_ptr(), stream=stream)   # sized anew
        torch.cuda.synchronize()
        assert np.abs(out.cpu().numpy().reshape(200, 300, 3) - orgb).max() <= 1e-12
        assert r.stats().rays == cnt["rays"]
        # rt_get_stats reports a void frame too
        r2 = _make({"RT_WF_DEBUG_SHRINK": "0.05"})
        r2.load(sf, P)
        r2.render_device(out.data_ptr(), stream=stream)
        r2.render_device(out.data_ptr(), stream=stream)
        with pytest.raises(rt.RtError, match="overflowed"):
            r2.stats()
        r2.close()
    finally:
        r.close()
