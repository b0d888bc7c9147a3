#!/bin/bash
# C3's per-GPU shape (1.25 M rows x d128, 10 k queries, k = 100): rows read by the sampled pass = 1/frac
for f in 0 4 6 8 10 12 16 24; do bash profiles/bench_variants.sh --rows 1250000 --k 100 --steps 10 --sample-frac $f; done
