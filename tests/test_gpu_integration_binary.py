"""INTEGRATION.md section 2 run for real: oracle/_ref/integration_check is the binding compiled against the reference's own
headers and linked with the reference's own classes (built in the build container by `make -C oracle integration-check`).
Code built from the reference does not travel to this project's GPU box (oracle/_ref is listed in .gpurunignore), so the
test is SKIPPED there; it is for a machine that has both a gfx950 GPU and the binary.  It builds scenes from the reference's objects, renders
them through the C-ABI on the GPU and with the reference's own RayTracer::render() on the CPU, and compares the two P3 files
byte for byte: binary and Phong modes, mirror and glass bounces, both tie rules, two tone maps."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "integration_check")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/integration_check is built where /root/reference exists (make -C oracle integration-check)")
def test_reference_objects_through_the_binding_give_the_references_own_frame(tmp_path):
    res = subprocess.run([EXE, str(tmp_path)], capture_output=True, text=True, timeout=300)
    lines = [l for l in res.stdout.splitlines() if ": identical" in l or ": DIFFERENT" in l]
    assert res.returncode == 0, (res.returncode, lines, res.stderr[-2000:])
    assert len(lines) == 4 and all(": identical" in l for l in lines), lines
