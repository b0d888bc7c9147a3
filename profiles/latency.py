"""Serial-query latency (the reference's own timing mode: basic_bench.h:82-126 calls query_k one
query at a time): microseconds per expann_search call for m = 1 / 4 / 8 host-buffer queries,
with the pinned latency mode on and off.  Usage: python profiles/latency.py [n] [d]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from expann_amd import GpuBruteForceEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.RandomState(1234)
base = rng.standard_normal((n, d)).astype(np.float32)
queries = rng.standard_normal((512, d)).astype(np.float32)
eng = GpuBruteForceEngine(d, "l2")
eng.store_many_vectors(base)
eng.build()
for mode in (0, 1):
    eng.set_option("latency_mode", mode)
    for m in (1, 4, 8, 32):
        for i in range(20):                                    # warm-up (fp16 copy, buffers)
            eng.query_k_batch(queries[i * m % 256:i * m % 256 + m], 10)
        reps = 300
        t0 = time.perf_counter()
        for i in range(reps):
            o = (i * m) % 256
            eng.query_k_batch(queries[o:o + m], 10)
        dt = (time.perf_counter() - t0) / reps
        print(f"latency_mode={mode} m={m}: {dt * 1e6:.1f} us per call ({m / dt:.0f} queries/s)", flush=True)
eng.close()
