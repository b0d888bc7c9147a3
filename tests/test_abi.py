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
    assert lib.expann_abi_version() == 2


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


_ORDER_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r})
from expann_amd import _lib
L = _lib.load()                       # the library FIRST ...
{use}
import torch                          # ... torch second
mapped = _lib.hip_runtimes_mapped()
assert len(mapped) == 1, mapped       # one HIP runtime in the process, not two
{after}
print("ok", mapped[0])
"""


def test_one_hip_runtime_whatever_the_import_order():
    """Round 1 observed 'No HIP GPUs are available' when torch initialised after the library had
    been used.  Cause: torch's wheels bundle their own libamdhip64.so; loaded second it is mapped
    as a SECOND runtime next to /opt/rocm's (expann_amd/_lib.py::_one_hip_runtime).  The loader
    mechanics need no GPU to check: library first, torch second -> exactly one runtime mapped."""
    import subprocess
    import sys
    pytest.importorskip("torch")
    out = subprocess.run([sys.executable, "-c", _ORDER_SCRIPT.format(root=ROOT, use="", after="")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stderr[-2000:]


@pytest.mark.gpu
def test_torch_initialises_after_heavy_use_of_the_library():
    """The round-1 scenario itself, on the GPU: engines created, searched and destroyed through the
    library first; torch's HIP context comes up afterwards in the same process and both keep working."""
    import subprocess
    import sys
    use = ("import numpy as np\n"
           "from expann_amd import GpuBruteForceEngine\n"
           "rng = np.random.RandomState(0)\n"
           "base = rng.standard_normal((6000, 64)).astype(np.float32)\n"
           "q = rng.standard_normal((40, 64)).astype(np.float32)\n"
           "first = None\n"
           "for i in range(40):\n"
           "    e = GpuBruteForceEngine(64, 'l2'); e.store_many_vectors(base); e.build()\n"
           "    ids, _ = e.query_k_batch(q, 10); e.close()\n"
           "    first = ids if first is None else first\n"
           "    assert (ids == first).all()\n")
    after = ("assert torch.cuda.is_available()\n"
             "x = torch.randn(1000, 64, device='cuda'); assert float((x * x).sum()) > 0\n"
             "e = GpuBruteForceEngine(64, 'l2'); e.store_many_vectors(base); e.build()\n"
             "ids, _ = e.query_k_batch(q, 10); e.close(); assert (ids == first).all()\n")
    out = subprocess.run([sys.executable, "-c", _ORDER_SCRIPT.format(root=ROOT, use=use, after=after)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stderr[-2000:]
