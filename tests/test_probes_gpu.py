"""The hardware behaviour conv3x3_stage_kernel relies on, checked on the device the suite runs on: `buffer_load ... lds`
writes lane l's 16 bytes at the LDS base + 16 l and ZERO-FILLS lanes whose offset is out of range; raw buffer stores DROP such
lanes, with a scalar offset in use (scripts/probe/*.hip, stand-alone HIP programs compiled here with hipcc)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["buffer_lds_probe", "buffer_store_probe"])
def test_buffer_addressing_probe(name, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / name)
    src = os.path.join(ROOT, "scripts", "probe", name + ".hip")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", src, "-o", exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0 and "PROBE OK" in run.stdout, run.stdout[-2000:] + run.stderr[-1000:]
