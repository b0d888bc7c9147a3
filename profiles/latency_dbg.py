"""debug=16 on the single-query path: in-kernel timestamps of the fused sampled pass."""
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
for i in range(4):
    eng.query_k_batch(queries[i:i + m], 10)
eng.set_option("debug", 16)
for i in range(4):
    eng.query_k_batch(queries[i:i + m], 10)
eng.close()
