"""A handful of single-query searches (m from argv, default 1) for a kernel trace of the latency path."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from expann_amd import GpuBruteForceEngine  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.RandomState(1234)
base = rng.standard_normal((1_000_000, 128)).astype(np.float32)
queries = rng.standard_normal((64, 128)).astype(np.float32)
eng = GpuBruteForceEngine(128, "l2")
eng.store_many_vectors(base)
eng.build()
for i in range(12):
    eng.query_k_batch(queries[i:i + m], 10)
eng.close()
