"""Generates the committed fixtures under tests/golden/ -- run in the build container only.

Needs /root/reference (read-only) and oracle/_ref/raytracer_ref built from it by
`make -C oracle ref`.  Nothing of the reference's source text is stored: the fixtures are
sha256 values, ray counts, hit-ID maps and two small rendered images (outputs).

    python tests/golden/make_golden.py [--pathtrace]

  ref_outputs.json       sha256 of the reference's P3 output and its Scene::intersect call
                         counts for the deterministic scenes (1200x800 as shipped); the hashes are
                         re-measured here and must equal SURVEY.md 8(c).
  hit_ids_300x200.npz    primary-hit primitive IDs of the pinned oracle (the reference exposes
                         no primitive ID), int16, one array per scene/tie rule.
  pt_ref_*.npy           (--pathtrace) uint8 images rendered by the REFERENCE BINARY at 120x80
                         with many samples: the statistical anchor for path tracing.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref")
OUT = os.path.dirname(os.path.abspath(__file__))

# counts obtained there with a throw-away instrumented copy of the reference.
SURVEY_RAYS = {"binary_primitives": 960000, "blinn_phong": 1194892, "mirror_glass_cylinder": 3429116,
               "clutter_500": 3909615}
DETERMINISTIC = ["binary_primitives", "blinn_phong", "mirror_glass_cylinder", "clutter_500", "textured",
                 "simple_phong", "tone_mapping", "glass", "mirror_image"]


def run_ref(json_path, tonemap=None):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.ppm")
        cmd = [REF_BIN, json_path, out] + ([tonemap] if tonemap else [])
        subprocess.run(cmd, cwd=REF, check=True, stdout=subprocess.DEVNULL)
        return open(out, "rb").read()


def read_p3(data):
    tok = data.split()
    assert tok[0] == b"P3"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int32).reshape(h, w, 3).astype(np.uint8)


def main():
    assert os.path.exists(REF_BIN), "build oracle/_ref first: make -C oracle ref"
    if "--pathtrace-only" in sys.argv:
        return pathtrace_fixtures()
    outputs = {}
    for name in DETERMINISTIC:
        data = run_ref(os.path.join(REF, "scenes", name + ".json"))
        outputs[name] = {"sha256": hashlib.sha256(data).hexdigest(), "width": 1200, "height": 800}
        if name in SURVEY_RAYS:
            outputs[name]["rays"] = SURVEY_RAYS[name]
        print(name, outputs[name]["sha256"])
    # tone-mapped variants of one scene exercise toneMap()
    for tm in ["reinhard", "ward", "uncharted2"]:
        data = run_ref(os.path.join(REF, "scenes", "mirror_glass_cylinder.json"), tm)
        outputs["mirror_glass_cylinder:" + tm] = {"sha256": hashlib.sha256(data).hexdigest(), "width": 1200,
                                                  "height": 800}
    json.dump(outputs, open(os.path.join(OUT, "ref_outputs.json"), "w"), indent=1, sort_keys=True)

    # hit-ID maps from the pinned oracle
    import raytracer_amd as rt
    from tests import oracle_lib
    maps = {}
    for name in ["binary_primitives", "blinn_phong", "mirror_glass_cylinder", "clutter_500", "textured", "final"]:
        sf = rt.SceneFile(os.path.join(ROOT, "scenes", name + ".json"), width=300, height=200)
        P = sf.params
        P.mode = rt.RT_MODE_PHONG if P.mode == rt.RT_MODE_PATHTRACE else P.mode
        for bvh in (0, 1):
            desc = sf.desc
            desc.use_bvh = bvh
            _, hit, _, _ = oracle_lib.render(desc, sf.camera, P)
            assert hit.min() >= -1 and hit.max() < 32767
            maps[f"{name}:bvh{bvh}"] = hit.astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "hit_ids_300x200.npz"), **maps)

    if "--pathtrace" in sys.argv:
        pathtrace_fixtures()


def pathtrace_fixtures():
    if True:  # (one block, kept at its indentation)
        only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
        # final_textured_lens: BASELINE config 5's workload, derived from final.json by scripts/make_config5_scene.py
        # (our file, under ROOT/scenes; the reference binary renders it from its own directory, where texture/ lies)
        for name, w, h, spp, ls, tm in [("pathtrace", 120, 80, 1024, 16, None), ("final", 120, 80, 256, 4, "uncharted2"),
                                        ("final_textured_lens", 120, 80, 1024, 16, "uncharted2")]:
            if only and name not in only:
                continue
            src = os.path.join(REF, "scenes", name + ".json")
            j = json.load(open(src if os.path.exists(src) else os.path.join(ROOT, "scenes", name + ".json")))
            j["camera"]["width"], j["camera"]["height"] = w, h
            j["lightsample"] = spp  # -> RayTracer::pixelSamples (the swap at RayTracer.cpp:138-139)
            j["pixelsample"] = ls   # -> RayTracer::lightSamples
            with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
                json.dump(j, f)
            img = read_p3(run_ref(f.name, tm))
            np.save(os.path.join(OUT, f"pt_ref_{name}_{w}x{h}_{spp}spp_{ls}ls.npy"), img)
            os.unlink(f.name)


if __name__ == "__main__":
    main()
