"""Generate tests/golden/topk_ref.json by running the REFERENCE's own topk_t<float>
(upstream src/topk_t.h compiled unmodified into oracle/_ref/libtopk_ref.so by
oracle/Makefile).  Run in the build container only (needs /root/reference):

    make -C oracle && python oracle/gen_golden.py

The fixture holds inputs and the reference's outputs (data only).  The brute-force
admission loop of src/brute_force_engine.h:28-46 is the same rule applied to unique,
increasing ids, so the "scan" cases (ids = 0..n-1) also pin the brute-force selection,
tie order and output order.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_ctypes as oc  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                   "topk_ref.json")


def case(name, k, d, v, discard_goal=-1):
    r = oc.ref_topk_run(k, d, v, discard_goal)
    return dict(name=name, k=int(k), discard_goal=int(discard_goal),
                d=[float(x) for x in np.asarray(d, np.float32)],
                v=[int(x) for x in v],
                is_good=[int(x) for x in r["is_good"]],
                size_after=[int(x) for x in r["size_after"]],
                worst_after=[int(x) for x in r["worst_after"]],
                worst_val_after=[float(x) for x in r["worst_val_after"]],
                at_capacity_after=[int(x) for x in r["at_capacity_after"]],
                out_ids=[int(x) for x in r["out_ids"]],
                out_dists=[float(x) for x in r["out_dists"]])


def main():
    cases = []
    # the survey's known-answer case
    cases.append(case("survey_kat", 3, [5, 5, 5, 3, 3, 9, 1], list(range(7))))
    rng = np.random.RandomState(20261004)
    # scans with unique increasing ids (= brute_force_engine admission loop)
    for n, k, levels in [(50, 1, 0), (200, 10, 0), (200, 10, 7), (64, 100, 5), (1000, 10, 0),
                         (1000, 100, 16), (300, 17, 3)]:
        if levels:  # heavy ties: distances drawn from a few levels
            d = rng.randint(0, levels, size=n).astype(np.float32)
        else:
            d = rng.standard_normal(n).astype(np.float32) ** 2
        cases.append(case(f"scan_n{n}_k{k}_lv{levels}", k, d, list(range(n))))
    # all-equal distances, descending and ascending distances
    cases.append(case("scan_all_equal", 5, np.full(40, 2.5, np.float32), list(range(40))))
    cases.append(case("scan_descending", 8, np.arange(100, 0, -1, dtype=np.float32),
                      list(range(100))))
    cases.append(case("scan_ascending", 8, np.arange(100, dtype=np.float32), list(range(100))))
    # repeated ids (the `known` set) in arbitrary order, with ties
    for n, k, idr, levels in [(120, 6, 30, 5), (400, 10, 50, 0), (90, 4, 8, 3)]:
        v = rng.randint(0, idr, size=n)
        d = (rng.randint(0, levels, size=n).astype(np.float32) if levels
             else rng.standard_normal(n).astype(np.float32) ** 2)
        cases.append(case(f"dedup_n{n}_k{k}_ids{idr}_lv{levels}", k, d, [int(x) for x in v]))
    # discard_until_size
    d = rng.standard_normal(60).astype(np.float32) ** 2
    cases.append(case("discard_to_3", 10, d, list(range(60)), discard_goal=3))
    cases.append(case("discard_noop", 10, d, list(range(60)), discard_goal=50))
    with open(OUT, "w") as f:
        json.dump(dict(generator="oracle/gen_golden.py",
                       source="reference src/topk_t.h via oracle/_ref/libtopk_ref.so",
                       cases=cases), f)
    print("wrote", OUT, len(cases), "cases", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
