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
    code = ("const b=require(%r); if (b.abiVersion()!==4) process.exit(2);"
            "const need=['create','destroy','mergeBatch','mergeBatchAsync','reserve','loadRows','getRows','rowCount','dumpRows','indexBuild','indexDrop','indexSize','scanRange','scanCount','scanFilter','info'];"
            "for (const k of need) if (typeof b[k]!=='function') { console.log('missing',k); process.exit(3); } console.log('addon ok');" % addon)
    out = subprocess.run([NODE, "-e", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "addon ok" in out.stdout, out.stdout + out.stderr


@needs_node
def test_writer_index_hands_out_free_components_in_order_of_appearance():
    """vector-clock mode without a writer list (writers: "auto"): the table's writer index gives a writer nobody named before the next free component,
    never an integer-like id (a JS object reorders such keys: src/bullet-crt.js:200-203 compares clocks with their key ORDER), never more than the width"""
    code = ("const { WriterIndex } = require(%r); const { clockKeyset } = require(%r);"
            "const t = { writers: ['w'], K: 4 }; const ix = new WriterIndex(t); ix.set('w', 0);"
            "const a = require('assert'); const comps = new Uint32Array(4);"
            "a.strictEqual(ix.get('w'), 0); a.strictEqual(ix.get('p1'), 1); a.strictEqual(ix.get('p1'), 1); a.strictEqual(ix.get('42'), undefined); a.strictEqual(ix.get(''), undefined);"
            "a.strictEqual(ix.get('p2'), 2); a.strictEqual(ix.get('p3'), 3); a.strictEqual(ix.get('p4'), undefined); a.deepStrictEqual(t.writers, ['w', 'p1', 'p2', 'p3']);"
            "const ks = clockKeyset({ p3: 5, w: 2 }, ix, comps); a.ok(ks >= 0); a.deepStrictEqual(Array.from(comps), [2, 0, 0, 5]); a.strictEqual(ks & 0xff, 0x03);"   # key order p3 (3), w (0)
            "a.strictEqual(clockKeyset({ p9: 1 }, ix, comps), -1); a.strictEqual(clockKeyset({ w: -1 }, ix, comps), -1); console.log('writer index ok');"
            % (os.path.join(ROOT, "bullet-js_amd", "js", "device-graph.js"), os.path.join(ROOT, "bullet-js_amd", "js", "hash.js")))
    out = subprocess.run([NODE, "-e", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "writer index ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@needs_node
def test_device_parity_through_napi():
    out = subprocess.run([NODE, os.path.join(JS, "device_parity.js"), GOLD], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "device_parity ok" in out.stdout
