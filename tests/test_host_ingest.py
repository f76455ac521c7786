"""Host ingest (JSON -> rt_scene_desc/rt_camera/rt_params) and the PPM writer, against the reference's
documented behaviour (RayTracer.cpp:817-1035, 352-367)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import capi
from tests import scene_util


def write_scene(tmp_path, obj, name="s.json"):
    p = os.path.join(str(tmp_path), name)
    json.dump(obj, open(p, "w"))
    return p


BASE = {"camera": {"width": 30, "height": 20, "position": [0, 0, 0], "lookAt": [0, 0, 1], "upVector": [0, 1, 0], "fov": 45.0},
        "scene": {"backgroundcolor": [0.1, 0.2, 0.3], "shapes": []}}


def test_defaults(tmp_path):
    sf = rt.SceneFile(write_scene(tmp_path, BASE))
    P = sf.params
    assert (P.max_depth, P.mode, P.tonemap) == (5, rt.RT_MODE_PHONG, 0)      # nbounces=5, rendermode="phong"
    assert (P.exposure, P.shadow_bias) == (1.0, 1e-4)
    assert sf.desc.use_bvh == 1 and sf.desc.n_lights == 0 and sf.desc.n_prims == 0
    assert sf.camera.lens_radius == 0.0
    # focusDistance defaults to |lookAt - position| = 1 -> lower-left z = 1 (w points backwards)
    assert sf.camera.lower_left[2] == 1.0


def test_sample_keys_are_swapped(tmp_path):
    s = dict(BASE, rendermode="pathtrace", pixelsample=10, lightsample=3)
    P = rt.SceneFile(write_scene(tmp_path, s)).params
    assert (P.pixel_samples, P.light_samples) == (3, 10)  # RayTracer.cpp:138-139
    P = rt.SceneFile(scene_util.scene_path("pathtrace")).params
    assert (P.pixel_samples, P.light_samples) == (4, 16)  # "nspp" is never read; defaults 16/4, swapped


def test_cylinder_parse_transform(tmp_path):
    s = json.loads(json.dumps(BASE))
    s["scene"]["shapes"] = [{"type": "cylinder", "center": [1, 2, 3], "axis": [0, 2, 0], "radius": 0.5, "height": 0.25}]
    sf = rt.SceneFile(write_scene(tmp_path, s))  # keep the owner alive while reading through desc
    d = sf.desc
    g = [d.prim_geom[k] for k in range(8)]
    # height doubled; base shifted by the UN-normalized axis * height / 2; axis normalized
    assert g == [1.0, 2.0 - 2 * 0.5 / 2.0, 3.0, 0.0, 1.0, 0.0, 0.5, 0.5]


def test_unknown_shape_light_mode_continue(tmp_path, capfd):
    s = json.loads(json.dumps(BASE))
    s["rendermode"] = "sketch"
    s["scene"]["shapes"] = [{"type": "torus"}, {"type": "sphere", "center": [0, 0, 3], "radius": 1}]
    s["scene"]["lightsources"] = [{"type": "spot"}, {"type": "pointlight", "position": [0, 1, 0], "intensity": [1, 1, 1]}]
    sf = rt.SceneFile(write_scene(tmp_path, s), tonemap="bogus")
    assert sf.desc.n_prims == 1 and sf.desc.n_lights == 1
    assert sf.params.mode == rt.RT_MODE_PHONG and sf.params.tonemap == 0
    err = capfd.readouterr().err
    assert "Unsupported shape type 'torus'" in err and "Unsupported light type 'spot'" in err
    assert "Unsupported rendermode 'sketch'" in err and "Unsupported tonemapping 'bogus'" in err


def test_missing_file_and_bad_json(tmp_path):
    with pytest.raises(rt.RtError):
        rt.SceneFile(os.path.join(str(tmp_path), "nope.json"))
    p = os.path.join(str(tmp_path), "bad.json")
    open(p, "w").write('{"camera": [1, 2,')
    with pytest.raises(rt.RtError):
        rt.SceneFile(p)


def test_missing_texture_disables_it(tmp_path):
    s = json.loads(json.dumps(BASE))
    mat = {"diffusecolor": [1, 0, 0], "specularcolor": [1, 1, 1], "texturepath": "texture/does_not_exist.ppm"}
    s["scene"]["shapes"] = [{"type": "sphere", "center": [0, 0, 3], "radius": 1, "material": mat}]
    sf = rt.SceneFile(write_scene(tmp_path, s))
    d = sf.desc
    assert d.materials[0].texture == -1 and d.n_textures == 0  # src/Material.cpp:24-28


def test_textures_load_and_dedupe():
    sf = scene_util.load("textured")
    d = sf.desc
    assert d.n_textures == 3  # paper, tree, checkerboard (used twice)
    dims = sorted((d.textures[i].width, d.textures[i].height) for i in range(3))
    assert dims == [(275, 183), (512, 512), (680, 460)]


def test_json_numbers_and_escapes(tmp_path):
    s = json.loads(json.dumps(BASE))
    text = json.dumps(s).replace('"fov": 45.0', '"fov": 4.5e1').replace('"width": 30', '"width": 30.9')
    p = os.path.join(str(tmp_path), "n.json")
    open(p, "w").write("﻿" + text.replace("backgroundcolor", "background\\u0063olor"))
    sf = rt.SceneFile(p)
    assert sf.params.width == 30  # float -> int truncates like nlohmann's get<int>
    assert sf.desc.background[2] == 0.3


def test_resize_rebuilds_camera():
    a = scene_util.load("clutter_500")
    b = scene_util.load("clutter_500", 3840, 2160)
    assert (b.params.width, b.params.height) == (3840, 2160)
    # vertical extent is unchanged, horizontal scales with the aspect ratio
    assert np.allclose(list(a.camera.vertical), list(b.camera.vertical), rtol=0, atol=0)
    ha, hb = np.linalg.norm(list(a.camera.horizontal)), np.linalg.norm(list(b.camera.horizontal))
    assert abs(hb / ha - (3840 / 2160) / 1.5) < 1e-12


def test_ppm_writer_format(tmp_path):
    rgb = np.zeros((2, 3, 3))
    rgb[0, 0] = [1.0, 0.5, 0.0]       # bottom-left pixel
    rgb[1, 2] = [0.999, 0.001, 1.0]   # top-right pixel
    p = os.path.join(str(tmp_path), "o.ppm")
    rt.write_ppm(p, rgb)
    lines = open(p).read().split("\n")
    assert lines[:3] == ["P3", "3 2", "255"]
    assert lines[3 + 2] == "255 0 255"   # top row first (j = H-1), third pixel; int(255.999*0.999) = 255
    assert lines[3 + 3] == "255 127 0"   # then the bottom row
    assert lines[-1] == "" and len(lines) == 3 + 6 + 1


def test_ppm_writer_survives_out_of_range_values(tmp_path):
    """a 1-pixel-wide image of NaN / huge values prints 11 characters per component, as the
    reference's operator<< would ("-2147483648" for int(255.999 * NaN) on x86-64); the line buffer must hold it."""
    rgb = np.array([[[np.nan, -1e300, 1e300]], [[-3.0, 2.0, 0.5]]])
    p = os.path.join(str(tmp_path), "w1.ppm")
    rt.write_ppm(p, rgb)
    lines = open(p).read().split("\n")
    assert lines[:3] == ["P3", "1 2", "255"]
    assert lines[3] == "-767 511 127"                      # top row first: int(255.999 * c), no clamping in the writer
    assert lines[4] == "-2147483648 -2147483648 -2147483648"


def test_json_nesting_depth_is_capped(tmp_path):
    p = os.path.join(str(tmp_path), "deep.json")
    open(p, "w").write("[" * 100000 + "]" * 100000)
    with pytest.raises(rt.RtError):
        rt.SceneFile(p)


def test_cli_usage_and_missing_file_messages(tmp_path):
    """Exit codes and messages of main() before the seam (RayTracer.cpp:47-60); no GPU needed to get there."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "csrc", "raytracer")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert res.returncode == 1 and "Usage: raytracer.exe path_to_JSON output_filename.ppm" in res.stderr
    res = subprocess.run([exe, os.path.join(str(tmp_path), "nope.json"), os.path.join(str(tmp_path), "o.ppm")],
                         capture_output=True, text=True, timeout=60)
    assert res.returncode == 1
    assert res.stderr.count("Could not open JSON file") + res.stderr.count("could not open JSON file") == 1  # once, as the reference
