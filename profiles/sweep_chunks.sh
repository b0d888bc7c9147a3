#!/bin/bash
# row-chunk count of the C2 scan (pick_row_chunks' choice vs forced counts): tail imbalance vs prologue
for c in 0 64 96 128 160 192 256 320; do
  bash profiles/bench_variants.sh --steps 20 --warmup 3 --scan-chunks $c
done
