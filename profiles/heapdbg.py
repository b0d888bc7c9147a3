import json, sys, os, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import oracle_ctypes as oc
from expann_amd import _lib
lib = _lib.load()
lib.expann_device_heap_trace.restype = C.c_size_t
lib.expann_device_heap_trace.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 8
cases = json.load(open("/root/repo/tests/golden/heap_ref.json"))["cases"]
for c in cases:
    init = [(d, i) for d, i in c["init"]]
    ops = [(k, d, i) for k, d, i in c["ops"]]
    states, drain = oc.heap_trace(c["max_heap"], init, ops, lib.expann_device_heap_trace)
    want = [tuple(s) for s in c["states"]]
    wd = [tuple(x) for x in c["drain"]]
    ok = states == want and drain == wd
    msg = ""
    if states != want:
        j = next(i for i in range(len(want)) if i >= len(states) or states[i] != want[i])
        msg = f" first state mismatch at {j}: got {states[j] if j < len(states) else None} want {want[j]}; op {ops[j-1] if j else None}; size before {want[j-1][0] if j else None}"
    elif drain != wd:
        j = next(i for i in range(len(wd)) if i >= len(drain) or drain[i] != wd[i])
        msg = f" first drain mismatch at {j} of {len(wd)}: got {drain[j] if j < len(drain) else None} want {wd[j]}"
    print(("ok  " if ok else "BAD ") + c["name"], "max" if c["max_heap"] else "min", len(init), len(ops), msg)
