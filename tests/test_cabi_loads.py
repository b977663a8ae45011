"""CPU-only checks of the drop-in boundary: libbmx.so builds, loads, exports every symbol that
include/bmx.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

import bmx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    return bmx.load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bmx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bmx_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), "libbmx.so does not export " + name
    assert declared == set(bmx.EXPORTS)


def test_abi_version(lib):
    assert lib.bmx_abi_version() == 4


def test_owner_of_matches_oracle(lib):
    from oracle.oracle import owner_of as o_owner
    from oracle import streams
    import numpy as np
    ids = streams.splitmix64_np(np.arange(1, 2000, dtype=np.uint64))
    for g in (1, 2, 3, 8, 16):
        a = bmx.owner_of(ids, g); b = o_owner(ids, g)
        assert (a == b).all() and a.max() < g
        if g == 8:
            assert np.bincount(a, minlength=8).min() > 150


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bmx.BmxError) as ei:
        bmx.Engine(1000)
    assert ei.value.code == bmx.ERR_NO_DEVICE


def test_selfcheck_and_deferral_switch_need_a_gpu_or_a_context(lib):
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(bmx.BmxError) as ei:
            bmx.selfcheck(0)
        assert ei.value.code == bmx.ERR_NO_DEVICE
    assert lib.bmx_set_deferred_compaction(None, 1) == bmx.ERR_INVALID and lib.bmx_merge_fence(None) == bmx.ERR_INVALID
    assert lib.bmx_get_deferred_counts(None, None, None) == bmx.ERR_INVALID


def test_key_set_words_and_page_locked_memory_without_a_gpu(lib):
    """Pure helpers work anywhere (the key-set word of a vector clock: include/bmx.h); page-locked memory comes from the HIP runtime, so without a GPU
    bmx_host_alloc answers with an error code and a message — it never hands out ordinary memory in its place."""
    import ctypes as C
    import numpy as np
    import torch
    for keys in ([], [0], [2, 1, 0], [7, 0, 3, 1]):
        a = np.array(keys, np.uint8)
        assert lib.bmx_vc_keyset(C.c_void_p(a.ctypes.data) if len(a) else None, len(a)) == bmx.keyset(keys)
        assert bmx.keyset_writers(bmx.keyset(keys)) == keys
    assert lib.bmx_vc_keyset_dense(3) == bmx.keyset([0, 1, 2]) and lib.bmx_vc_keyset_dense(8) == bmx.keyset(range(8))
    p = C.c_void_p()
    assert lib.bmx_host_alloc(0, C.byref(p)) == bmx.ERR_INVALID and lib.bmx_host_free(None) == 0
    if not torch.cuda.is_available():
        with pytest.raises(bmx.BmxError) as ei:
            bmx.HostBuffer(4096)
        assert ei.value.code in (bmx.ERR_HIP, bmx.ERR_NOMEM, bmx.ERR_NO_DEVICE) and str(ei.value)


def test_oversize_tables_are_refused_before_any_device_work(lib):
    """Slot indices in the per-batch workspace are 32-bit: a table that would need more than 2^32 slots (137 GB of 32-byte slots would
    fit the 288 GB of HBM) is refused with BMX_ERR_INVALID, on any machine, instead of corrupting rows silently."""
    import ctypes as C
    h = C.c_void_p()
    for cap, pct in ((1 << 31, 0), ((1 << 32) + 5, 90), (1 << 50, 50)):
        rc = lib.bmx_create_ex(0, cap, pct, 0, C.byref(h))
        assert rc == bmx.ERR_INVALID and not h.value, (cap, pct, rc)
        assert b"2^32 slots" in lib.bmx_last_error(None)
    assert lib.bmx_create_ex(0, 1000, 95, 0, C.byref(h)) == bmx.ERR_INVALID      # load factor out of range
    assert lib.bmx_vc_create(0, 1 << 31, 3, 0, C.byref(h)) == bmx.ERR_INVALID
    assert b"2^32 slots" in lib.bmx_vc_last_error(None)
