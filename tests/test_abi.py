"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/expann_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from expann_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "expann_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(expann_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from expann_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 14
    assert sorted(_lib.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version(lib):
    assert lib.expann_abi_version() == 1


def test_argument_validation_without_touching_a_gpu(lib):
    h = C.c_void_p()
    assert lib.expann_create(100, 0, 0, 0, C.byref(h)) == 1          # dim % 16 != 0
    assert b"multiple of 16" in lib.expann_last_error(None)
    assert lib.expann_create(128, 0, 7, 0, C.byref(h)) == 1          # bad metric
    assert lib.expann_create(128, 0, 0, 0, None) == 1                # out == NULL


def test_fails_loudly_without_a_device(lib):
    """No CPU fallback: on a box without a HIP device create() must report NO_DEVICE."""
    if lib.expann_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    rc = lib.expann_create(128, 0, 0, 0, C.byref(h))
    assert rc == 2 and not h.value
    assert b"no CPU fallback" in lib.expann_last_error(None)
    from expann_amd import GpuBruteForceEngine
    from expann_amd._lib import ExpannError
    with pytest.raises(ExpannError):
        GpuBruteForceEngine(128)


def test_product_never_imports_the_oracle():
    """The package and the C ABI sources must not reference oracle/ in any way."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "expann_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle_ctypes|liboracle|expann_oracle|oracle/", txt):
                    bad.append(os.path.join(dirpath, f))
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p) and re.search(r"oracle", open(p).read()):
            bad.append(p)
    assert not bad, bad
