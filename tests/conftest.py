import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "bullet-js_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# The merge parity modules run twice: on the default kernels and on the bucketed path (BMX_CTX_BUCKETED_MERGE).
_BOTH_PATHS = ("test_gpu_merge", "test_gpu_fuzz", "test_gpu_fullsize")


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in _BOTH_PATHS:
        metafunc.fixturenames.append("merge_path")
        metafunc.parametrize("merge_path", ["default", "bucketed"], indirect=True)


@pytest.fixture
def merge_path(request):
    import bmx
    old = bmx.DEFAULT_CTX_FLAGS
    bmx.DEFAULT_CTX_FLAGS = old | (bmx.CTX_BUCKETED_MERGE if request.param == "bucketed" else 0)
    yield request.param
    bmx.DEFAULT_CTX_FLAGS = old
