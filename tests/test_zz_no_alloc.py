"""rt_render_device_u8_static: the frame of rt_render_device_u8 out of the library's static workspace, without any
allocation (include/raytracer_amd.h).  Two calls in a row on one context, the second with a smaller input: the
array keeps the first call's contents, so whatever must start at zero has to be cleared by the call itself."""
import numpy as np
import pytest

import raytracer_amd as rt
from tests import oracle_lib, scene_util

pytestmark = pytest.mark.gpu


def both(r, sf, P, w, h):
    import torch
    r.load(sf, P)
    stream = torch.cuda.current_stream().cuda_stream
    ref8 = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
    ref_hit = torch.full((h * w,), -7, dtype=torch.int32, device="cuda")
    r.render_device_u8(ref8.data_ptr(), None, ref_hit.data_ptr(), stream=stream)   # the existing entry point
    torch.cuda.synchronize()
    ref_rays = r.stats().rays
    got8 = torch.full((h * w * 3,), 77, dtype=torch.uint8, device="cuda")
    got64 = torch.full((h * w * 3,), -1.0, dtype=torch.float64, device="cuda")
    got_hit = torch.full((h * w,), -7, dtype=torch.int32, device="cuda")
    r.render_device_u8_static(got8.data_ptr(), got64.data_ptr(), got_hit.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    st = r.stats()
    assert st.rays == ref_rays
    assert torch.equal(got8, ref8)
    assert torch.equal(got_hit, ref_hit)
    return got8.cpu().numpy().reshape(h, w, 3), got64.cpu().numpy().reshape(h, w, 3), got_hit.cpu().numpy().reshape(h, w)


def test_static_workspace_frames_equal_the_allocating_entry_point(renderer):
    # first call: glass + mirrors, depth 8, 300 x 200; second call: a smaller frame of another scene (fewer rays, fewer
    # levels) right after it, on the same context and therefore over the first call's leftovers
    for name, (w, h) in (("mirror_glass_cylinder", (300, 200)), ("clutter_500", (150, 100)), ("binary_primitives", (64, 48))):
        sf = scene_util.load(name, w, h)
        P = sf.params
        u8, rgb, hit = both(renderer, sf, P, w, h)
        orgb, ohit, cnt, _ = oracle_lib.render(sf.desc, sf.camera, P)
        assert np.array_equal(hit, ohit)
        assert np.abs(rgb - orgb).max() <= 1e-12
        assert np.array_equal(u8, (255.999 * rgb).astype(np.int32).astype(np.uint8))


def test_static_workspace_limits(renderer):
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    sf = scene_util.load("blinn_phong", 1920, 1080)   # 2 M pixels x 3 records do not fit in 256 MiB
    renderer.load(sf)
    out = torch.zeros(1920 * 1080 * 3, dtype=torch.uint8, device="cuda")
    with pytest.raises(rt.RtError, match="static workspace"):
        renderer.render_device_u8_static(out.data_ptr(), stream=stream)
    pt = scene_util.load("pathtrace", 64, 48)
    renderer.load(pt)
    with pytest.raises(rt.RtError, match="binary / phong"):
        renderer.render_device_u8_static(out.data_ptr(), stream=stream)
    # and the context still renders through the allocating path afterwards
    sf = scene_util.load("blinn_phong", 96, 64)
    renderer.load(sf)
    rgb, hit = renderer.render()
    orgb, ohit, _, _ = oracle_lib.render(sf.desc, sf.camera, sf.params)
    assert np.array_equal(hit, ohit) and np.abs(rgb - orgb).max() <= 1e-12


def test_static_workspace_under_stream_capture(renderer):
    """The entry point inside a captured HIP graph (torch.cuda.graph: the stream handed over is capturing), replayed twice
    into a cleared buffer: the frame of the allocating entry point both times.  One call outside capture comes first -- it
    makes the setup's attribute / occupancy queries -- and rt_get_stats after the replays reports no overflow."""
    import torch
    w, h = 200, 120
    sf = scene_util.load("mirror_glass_cylinder", w, h)
    renderer.load(sf, sf.params)
    ref8 = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
    renderer.render_device_u8(ref8.data_ptr(), None, None, stream=torch.cuda.current_stream().cuda_stream)
    got8 = torch.zeros_like(ref8)
    renderer.render_device_u8_static(got8.data_ptr(), None, None, stream=torch.cuda.current_stream().cuda_stream)  # warm-up, outside capture
    torch.cuda.synchronize()
    assert torch.equal(got8, ref8)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        renderer.render_device_u8_static(got8.data_ptr(), None, None, stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        got8.fill_(77)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(got8, ref8)
    renderer.stats()  # (raises if a replay had overflowed the workspace)
