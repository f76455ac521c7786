#!/usr/bin/env python3
"""Seeded generator of large "clutter" scenes in the reference's scene-file format.

    python scripts/gen_clutter.py --spheres 5000 --seed 1 --out /tmp/clutter_5000.json [--width 3840 --height 2160]

Same ingredients as the shipped clutter_*.json scenes (a two-triangle ground plane, a jittered square
grid of radius-0.2 spheres, about 90 % diffuse / 5 % metal / 5 % glass, one point light high above,
"phong" mode, 8 bounces), but the count is a parameter and the random stream is numpy's PCG64 seeded
by --seed, so a scene can be regenerated anywhere instead of shipped (500 k spheres are ~250 MB of
JSON).  The camera backs away with the grid so that the whole field stays in view.
"""
import argparse
import json
import math
import sys

import numpy as np


def material(kd, ks, exp, diffuse, specular, reflective=False, reflectivity=0.0, refractive=False, ior=1.0):
    return {"ks": ks, "kd": kd, "specularexponent": exp, "diffusecolor": diffuse, "specularcolor": specular,
            "isreflective": reflective, "reflectivity": reflectivity, "isrefractive": refractive, "refractiveindex": ior}


def clutter(n_spheres, seed, width=1200, height=800, bvh=True):
    rng = np.random.default_rng(seed)
    side = max(1, math.ceil(math.sqrt(n_spheres)))
    half = side / 2.0
    ground = material(1.0, 0.0, 0, [0.25, 0.6, 0.25], [0.0, 0.0, 0.0])
    ext = 1000.0 + 2.0 * half
    shapes = [
        {"type": "triangle", "v0": [-ext, -0.5, -ext], "v1": [ext, -0.5, -ext], "v2": [-ext, -0.5, ext], "material": ground},
        {"type": "triangle", "v0": [ext, -0.5, -ext], "v1": [ext, -0.5, ext], "v2": [-ext, -0.5, ext], "material": ground},
    ]
    cells = rng.permutation(side * side)[:n_spheres]  # which grid cells hold a sphere
    for cell in np.sort(cells):
        a, b = divmod(int(cell), side)
        x = a - half + 0.9 * rng.random()
        z = b - half + 0.9 * rng.random()
        pick = rng.random()
        if pick < 0.90:
            c = (rng.random(3) * rng.random(3)).tolist()
            m = material(1.0, 0.0, 0, c, [0.0, 0.0, 0.0])
        elif pick < 0.95:
            fuzz = 0.5 * rng.random()
            c = (0.5 * (1.0 + rng.random(3))).tolist()
            m = material(fuzz, 1.0 - fuzz, int(100 * (1.0 - fuzz)), [0.0, 0.0, 0.0], c, True, 1.0 - fuzz)
        else:
            m = material(0.2, 0.7, 20, [1.0, 1.0, 1.0], [1.0, 1.0, 1.0], True, 0.2, True, 1.5)
        shapes.append({"type": "sphere", "center": [x, -0.3, z], "radius": 0.2, "material": m})
    dist = 1.7 * half + 5.0
    return {
        "nbounces": 8, "rendermode": "phong", "bvh": bool(bvh),
        "camera": {"type": "pinhole", "width": width, "height": height,
                   "position": [dist, 0.12 * dist + 0.4, 0.23 * dist], "lookAt": [0.0, -0.5, 0.0],
                   "upVector": [0.0, 1.0, 0.0], "fov": 40.0, "exposure": 1, "focalLength": 10.0, "aperture": 0.0},
        "scene": {"backgroundcolor": [0.53, 0.80, 0.92],
                  "lightsources": [{"type": "pointlight", "intensity": [1, 1, 1], "position": [0, 100 + half, 0]}],
                  "shapes": shapes},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spheres", type=int, required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--no-bvh", action="store_true", help='"bvh": false (linear-scan tie rule), as the shipped scenes')
    ap.add_argument("--out", default="-")
    a = ap.parse_args()
    scene = clutter(a.spheres, a.seed, a.width, a.height, not a.no_bvh)
    out = sys.stdout if a.out == "-" else open(a.out, "w")
    json.dump(scene, out)
    out.write("\n")


if __name__ == "__main__":
    main()
