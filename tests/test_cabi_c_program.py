"""A plain C host against the C ABI (no Python or JS between the program and libbmx.so), checked against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    import __graft_entry__ as g
    g.build()
    exe = os.path.join(tmp, "cabi_parity")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi_parity.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "bullet-js_amd"), "-lbmx", "-L", os.path.join(ROOT, "oracle"), "-lbmx_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "bullet-js_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return exe


def test_c_program_links_against_the_abi(tmp_path):
    exe = _build(str(tmp_path))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    import torch
    if not torch.cuda.is_available():
        assert out.returncode == 2 and "no HIP device" in out.stderr     # loud failure, no CPU fallback
    else:
        assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_program_parity_on_gpu(tmp_path):
    exe = _build(str(tmp_path))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cabi_parity ok" in out.stdout
