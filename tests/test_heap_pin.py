"""The graph path's bit-exactness rests on the order in which EQUAL distances leave libstdc++'s
heaps (the reference's priority queues compare .first only, src/antitopo_engine.h:540-558).
tests/golden/heap_ref.json holds tie-heavy queue traces answered by the image's real
std::priority_queue (oracle/ref/heap_ref.cpp via oracle/gen_heap_golden.py); here the three hand
restatements are checked against it:
  * the oracle's pq_* (oracle/expann_oracle_graph.c)          -- CPU, through oracle_heap_trace
  * expann::std_heap (include/expann/antitopo_index.h)        -- CPU, through tests/native/std_heap_hook.cpp
  * the device heap (expann_amd/csrc/graph_search.hpp)        -- GPU, a uint8 traversal over binary
    rows (integer distances, massive ties) against the oracle's walk that the traces pin."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "heap_ref.json")


@pytest.fixture(scope="module")
def cases():
    return json.load(open(GOLDEN))["cases"]


def _check(cases, fn, oracle):
    assert len(cases) >= 20
    for c in cases:
        init = [(d, i) for d, i in c["init"]]
        ops = [(k, d, i) for k, d, i in c["ops"]]
        states, drain = oracle.heap_trace(c["max_heap"], init, ops, fn)
        assert states == [tuple(s) for s in c["states"]], c["name"]
        assert drain == [tuple(x) for x in c["drain"]], c["name"]


def test_oracle_heap_equals_libstdcxx(cases, oracle):
    _check(cases, None, oracle)


def test_builder_std_heap_equals_libstdcxx(cases, oracle, tmp_path):
    so = tmp_path / "std_heap_hook.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "std_heap_hook.cpp"), "-o", str(so)])
    lib = C.CDLL(str(so))
    lib.std_heap_trace.restype = C.c_size_t
    lib.std_heap_trace.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 8
    _check(cases, lib.std_heap_trace, oracle)


def test_golden_is_what_this_box_libstdcxx_does(cases, tmp_path):
    """Recompile the generator's driver against the libstdc++ of the box the tests run on and
    replay a few traces: the fixture is not specific to the container that generated it."""
    exe = tmp_path / "heap_ref"
    subprocess.check_call(["g++", "-O2", "-std=c++20", os.path.join(ROOT, "oracle", "ref", "heap_ref.cpp"),
                           "-o", str(exe)])
    for c in cases[::3]:
        txt = [str(c["max_heap"]), str(len(c["init"]))] + [f"{d!r} {i}" for d, i in c["init"]]
        txt += [str(len(c["ops"]))] + [f"{k} {d!r} {i}" for k, d, i in c["ops"]]
        out = subprocess.run([str(exe)], input="\n".join(txt) + "\n", capture_output=True, text=True, check=True)
        lines = [[int(x) for x in ln.split()] for ln in out.stdout.strip().splitlines()]
        n = len(c["ops"]) + 1
        assert lines[:n] == c["states"] and lines[n:] == c["drain"], c["name"]


@pytest.mark.gpu
def test_device_coop_heap_equals_libstdcxx(cases, oracle):
    """The walk's queues themselves: every golden trace replayed through the device's wave-cooperative
    push / pop (csrc/graph_search.hpp coop_push, coop_pop; expann_device_heap_trace) must leave the states
    and the drain order the image's std::priority_queue left -- ties, range construction, long queues."""
    from expann_amd import _lib
    lib = _lib.load()
    lib.expann_device_heap_trace.restype = C.c_size_t
    lib.expann_device_heap_trace.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 8
    _check(cases, lib.expann_device_heap_trace, oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("levels,M", [(2, 16), (4, 40)])
def test_device_heap_on_massive_ties(tmp_path, oracle, levels, M):
    """uint8 rows with `levels` distinct values per component: every distance is a small integer
    and most comparisons inside the two queues are ties, so ids / distances / distcomps can only
    match the oracle if the device heap moves elements exactly like libstdc++'s."""
    from graph_helpers import build_engines, check_against_oracle
    rng = np.random.RandomState(77 + levels)
    n, d, m, k = 2500, 128, 96, 10
    base = rng.randint(0, levels, size=(n, d)).astype(np.float32)
    q = rng.randint(0, levels, size=(m, d)).astype(np.float32)
    engs, idx = build_engines(base, tmp_path, M=M, ef_construction=3 * M)
    check_against_oracle(oracle, engs, idx, q, k, efs=(10, 40, 150))
    # ties really are massive: a typical query sees each distance value many times
    one = ((base - q[0]) ** 2).sum(1)
    assert len(np.unique(one)) < n // 10
