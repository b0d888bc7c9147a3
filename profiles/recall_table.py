#!/usr/bin/env python3
"""profiles/recall_table.py <index file> <queries file> [label] -- recall@10 of the reference walk (the CPU
oracle's restatement of antitopo_engine::_query_k, oracle/expann_oracle_graph.c) over an index file in the
reference's layout, ef_search = 10 ... 60, fp32 rows and uint8 rows + re-score, against the exact brute-force
answer over the vectors the file holds (src/basic_bench.h:116-121,143).  Used to compare the serial host
builder's graph with the batched GPU builder's graph of the same rows (profiles/r03_c4_recall_100k.txt)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_ctypes as oc  # noqa: E402


def main():
    idx, qfile = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(idx)
    g = oc.Graph(idx)
    base = g.vectors()
    q = np.fromfile(qfile, dtype=np.float32).reshape(-1, g.dim)
    k = 10
    gt, _ = oc.brute_force(base, q, k, oc.METRIC_L2_F32, n_threads=os.cpu_count() or 1)
    row = [f"{label}: n = {g.n}, {len(q)} queries, recall@{k} (distance evaluations per query)"]
    for comp in (False, True):
        cells = []
        for ef in (10, 20, 30, 40, 50, 60):
            ids, _, dc = g.query_k(q, k, ef, comp)
            cells.append(f"ef {ef}: {oc.recall(ids, gt):.4f} ({dc.mean():.0f})")
        row.append(("  uint8 + re-score: " if comp else "  fp32:             ") + ", ".join(cells))
    print("\n".join(row))


if __name__ == "__main__":
    main()
