"""The host side in the reference's language (bullet-js_amd/js): GpuCRT / GpuQuery.
CPU part: single-operation semantics vs the reference's golden vectors (and, when a reference checkout is
mounted, through the real Bullet facade). GPU part: batch merge and device indices through the N-API addon."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "bullet-js_amd", "js", "test")
GOLD = os.path.join(ROOT, "tests", "golden")
NODE = shutil.which("node")

needs_node = pytest.mark.skipif(NODE is None, reason="node is not installed on this box")


@needs_node
def test_host_semantics_match_reference_golden():
    out = subprocess.run([NODE, os.path.join(JS, "host_semantics.js"), GOLD], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_semantics ok" in out.stdout


@needs_node
def test_addon_builds_and_refuses_without_gpu():
    import __graft_entry__ as g
    g.build()
    addon = os.path.join(ROOT, "bullet-js_amd", "bmx.node")
    assert os.path.exists(addon)
    code = ("const b=require(%r); if (b.abiVersion()!==3) process.exit(2);"
            "const need=['create','destroy','mergeBatch','mergeBatchAsync','reserve','loadRows','getRows','rowCount','dumpRows','indexBuild','indexDrop','indexSize','scanRange','scanCount','scanFilter','info'];"
            "for (const k of need) if (typeof b[k]!=='function') { console.log('missing',k); process.exit(3); } console.log('addon ok');" % addon)
    out = subprocess.run([NODE, "-e", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "addon ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@needs_node
def test_device_parity_through_napi():
    out = subprocess.run([NODE, os.path.join(JS, "device_parity.js"), GOLD], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "device_parity ok" in out.stdout
