#!/usr/bin/env python3
"""BASELINE.json configs[4]: "scenes/final.json textured + lens_sampling + Uncharted2".  final.json as shipped has
no texture and aperture 0; the features live in scenes/textured.json:41,61,80 ("texturepath") and
scenes/lens_sampling.json:16-17 ("aperture", "focusDistance").  This writes the combination,
scenes/final_textured_lens.json: final.json's Cornell box with the back wall textured by checkerboard.ppm, the
floor by paper.ppm, the cylinder by tree.ppm, and a thin lens of aperture 0.3 focused on the glass sphere.
Everything else (lights, geometry, nbounces, sample keys) is final.json's.  Tone map: argv[3] = uncharted2."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
j = json.load(open(os.path.join(ROOT, "scenes", "final.json")))
shapes = j["scene"]["shapes"]
for i in (8, 9):
    shapes[i]["material"]["texturepath"] = "texture/checkerboard.ppm"   # back wall
for i in (10, 11):
    shapes[i]["material"]["texturepath"] = "texture/paper.ppm"          # floor
shapes[12]["material"]["texturepath"] = "texture/tree.ppm"              # cylinder
assert shapes[12]["type"] == "cylinder" and shapes[13]["type"] == "sphere"
j["camera"]["aperture"] = 0.3                                            # scenes/lens_sampling.json:16
j["camera"]["focusDistance"] = 3.2                                       # camera z = 3.5, sphere centre z = 0.3
json.dump(j, open(os.path.join(ROOT, "scenes", "final_textured_lens.json"), "w"), indent=4)
print("wrote scenes/final_textured_lens.json")
