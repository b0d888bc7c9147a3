#!/bin/bash
# rows read by the sampled pass = 1/frac at the large shapes (the planner's sample_frac_for was tuned at C2)
for f in 0 24 32 48 64 96; do bash profiles/bench_variants.sh --workload c5 --steps 3 --sample-frac $f; done
for f in 0 24 32 48 64; do bash profiles/bench_variants.sh --n 1000000 --d 960 --steps 3 --sample-frac $f; done
for f in 0 24 32 48; do bash profiles/bench_variants.sh --n 1000000 --d 512 --steps 5 --sample-frac $f; done
for f in 0 24 32 48; do bash profiles/bench_variants.sh --n 10000000 --steps 3 --sample-frac $f; done
