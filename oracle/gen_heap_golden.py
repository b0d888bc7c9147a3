"""Generate tests/golden/heap_ref.json by running tie-heavy queue traces through the image's OWN
libstdc++ std::priority_queue (oracle/ref/heap_ref.cpp, compiled here with g++ into
oracle/_ref/heap_ref).  The reference's graph search keeps its two queues in
std::priority_queue<std::pair<float, size_t>> with comparators that look at .first only
(upstream src/antitopo_engine.h:540-558), so the order among equal distances -- and with it the
walk -- is whatever libstdc++'s make_heap / push_heap / pop_heap do.  The fixture holds the
traces (inputs) and libstdc++'s answers (state after every operation, final drain order).

    make -C oracle heap_ref && python oracle/gen_heap_golden.py
"""
import json
import os
import subprocess
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "_ref", "heap_ref")
OUT = os.path.join(HERE, "..", "tests", "golden", "heap_ref.json")


def f32(x):
    return float(np.float32(x))


def run(max_heap, init, ops):
    txt = [str(int(max_heap)), str(len(init))]
    txt += [f"{d!r} {i}" for d, i in init]
    txt.append(str(len(ops)))
    txt += [f"{k} {d!r} {i}" for k, d, i in ops]
    out = subprocess.run([EXE], input="\n".join(txt) + "\n", capture_output=True, text=True, check=True)
    lines = out.stdout.strip().splitlines()
    states = [tuple(int(x) for x in ln.split()) for ln in lines[:len(ops) + 1]]
    drain = [tuple(int(x) for x in ln.split()) for ln in lines[len(ops) + 1:]]
    return states, drain


def make_case(rng, name, max_heap, n_init, n_ops, levels, cap=None, p_push=0.6):
    """Random trace.  levels > 0: distances are small integers (massive ties, the uint8 path's
    regime); cap: the bounded `nearest` protocol -- push, then pop while size > cap."""
    def dist():
        if levels:
            return f32(rng.randint(0, levels))
        return f32(rng.standard_normal() ** 2)
    nid = 0
    init = []
    for _ in range(n_init):
        init.append((dist(), nid))
        nid += 1
    ops = []
    size = n_init
    while len(ops) < n_ops:
        if cap is not None:
            ops.append((1, dist(), nid))
            nid += 1
            size += 1
            if size > cap:
                ops.append((0, 0.0, 0))
                size -= 1
        elif size == 0 or rng.rand() < p_push:
            ops.append((1, dist(), nid))
            nid += 1
            size += 1
        else:
            ops.append((0, 0.0, 0))
            size -= 1
    states, drain = run(max_heap, init, ops)
    return dict(name=name, max_heap=int(max_heap), init=[[d, i] for d, i in init],
                ops=[[k, d, i] for k, d, i in ops], states=[list(s) for s in states],
                drain=[list(x) for x in drain])


def main():
    subprocess.check_call(["make", "-s", "-C", HERE, "heap_ref"])
    rng = np.random.RandomState(20261005)
    cases = []
    for mh in (0, 1):
        tag = "max" if mh else "min"
        cases.append(make_case(rng, f"{tag}_all_equal", mh, 0, 200, 1))
        cases.append(make_case(rng, f"{tag}_two_levels", mh, 3, 400, 2))
        cases.append(make_case(rng, f"{tag}_five_levels_init7", mh, 7, 600, 5))
        cases.append(make_case(rng, f"{tag}_int_levels_40", mh, 1, 800, 40))
        cases.append(make_case(rng, f"{tag}_floats_no_ties", mh, 4, 300, 0))
        cases.append(make_case(rng, f"{tag}_init_only_make_heap_33", mh, 33, 0, 3))
        cases.append(make_case(rng, f"{tag}_init_only_make_heap_64_even", mh, 64, 0, 4))
        cases.append(make_case(rng, f"{tag}_mostly_pops", mh, 50, 300, 3, p_push=0.35))
    # the bounded `nearest` queue (max-heap, push then pop above ef): ef = 10, 60, 480
    for ef in (10, 60, 480):
        cases.append(make_case(rng, f"nearest_cap{ef}_levels6", 1, 1, 4 * ef + 50, 6, cap=ef))
    cases.append(make_case(rng, "nearest_cap20_floats", 1, 1, 200, 0, cap=20))
    with open(OUT, "w") as f:
        json.dump(dict(generator="oracle/gen_heap_golden.py",
                       source="the image's libstdc++ std::priority_queue<std::pair<float,size_t>> with "
                              ".first-only comparators (g++ " +
                              subprocess.check_output(["g++", "-dumpfullversion"], text=True).strip() + ")",
                       format="states[i] = [size, top distance bits, top id] after construction (i = 0) "
                              "and after op i-1; ops [1,d,id] push / [0,0,0] pop; drain = [bits, id] per pop",
                       cases=cases), f)
    print("wrote", OUT, len(cases), "cases", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
