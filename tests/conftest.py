import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    return {name: np.load(os.path.join(GOLDEN, name + ".npz")) for name in ("glm_probe", "oracle_kat", "stages", "ref_tables", "thrust_probe")}


@pytest.fixture(scope="session")
def mmgen_pkg():
    return importlib.import_module("mega-minecraft_amd")


@pytest.fixture(scope="session")
def gen(mmgen_pkg):
    """The HIP product on cuda:0.  Fails loudly (no CPU fallback) when the library or the GPU is missing."""
    return mmgen_pkg.MMGen(0)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bit_equal(a, b, what=""):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    bad = bits(a) != bits(b)
    if bad.any():
        idx = np.argwhere(bad)[:5]
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} entries differ, first at {idx.tolist()}: "
                             f"{a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}")
