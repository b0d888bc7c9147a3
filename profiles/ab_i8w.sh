#!/bin/bash
# A/B at d = 128 / 256, 8-bit rows: scan_gemm_i8w (EXPANN_I8W=1, default) vs scan_gemm_i8q (0)
for args in "--dtype u8" "--sift-like" "--dtype i8 --metric ip" "--dtype u8 --k 100 --rows 1250000" "--dtype u8 --queries 1000" "--d 256 --dtype i8" "--d 256 --dtype u8"; do
  bash profiles/bench_variants.sh --steps 10 $args
  EXPANN_I8W=0 bash profiles/bench_variants.sh --steps 10 $args | sed 's/^/   i8q: /'
done
