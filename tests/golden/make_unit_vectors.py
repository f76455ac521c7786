#!/usr/bin/env python3
"""Generates tests/golden/unit_vectors.npz from the REFERENCE'S OWN OBJECTS (container only).

Compiles tests/golden/make_unit_vectors.cpp (our driver) against /root/reference/include and links it with
the reference's translation units, compiled from the sources where they lie (main renamed; outputs under
oracle/_ref/, git- and gpurun-ignored), runs it, and packs inputs + the reference's answers into one .npz.
Nothing of the reference's source enters the repository: the fixture is data."""
import glob
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(ROOT, "oracle", "_ref")


def main():
    if not os.path.isdir(os.path.join(REF, "src")):
        sys.exit("reference sources not present: this script only runs in the build container")
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for src in sorted(glob.glob(os.path.join(REF, "src", "*.cpp"))):
        o = os.path.join(OUT, "uv_" + os.path.basename(src)[:-4] + ".o")
        # -O0 -ffp-contract=off: the reference's own Makefile uses no -O flag; x86-64 has no contraction at any level
        subprocess.check_call(["g++", "-O0", "-std=c++17", "-fopenmp", "-w", "-ffp-contract=off", "-I" + os.path.join(REF, "include"),
                               "-Dmain=reference_main", "-c", src, "-o", o])
        objs.append(o)
    exe = os.path.join(OUT, "unit_vectors")
    subprocess.check_call(["g++", "-O0", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-I" + os.path.join(REF, "include"),
                           os.path.join(ROOT, "tests", "golden", "make_unit_vectors.cpp")] + objs + ["-o", exe])
    raw = os.path.join(OUT, "unit_vectors.bin")
    subprocess.check_call([exe, raw], cwd=REF)
    arrays = {}
    with open(raw, "rb") as f:
        while True:
            head = f.read(4)
            if not head:
                break
            (n,) = struct.unpack("<I", head)
            name = f.read(n).decode()
            rows, cols = struct.unpack("<II", f.read(8))
            arrays[name] = np.frombuffer(f.read(rows * cols * 8), dtype=np.float64).reshape(rows, cols).copy()
    path = os.path.join(ROOT, "tests", "golden", "unit_vectors.npz")
    np.savez_compressed(path, **arrays)
    for k, v in arrays.items():
        print(f"{k:20s} {v.shape}")
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
