#!/bin/bash
# A/B at d = 64 fp32: scan_gemm_f16x<64> at three workgroups per CU (EXPANN_F16X=1, default) vs scan_gemm_f16<64> (0)
for args in "--n 1000000 --d 64 --steps 10" "--n 1000000 --d 64 --k 100 --steps 10" "--n 1000000 --d 64 --queries 1000 --steps 20" "--n 4000000 --d 64 --steps 5"; do
  bash profiles/bench_variants.sh $args
  EXPANN_F16X=0 bash profiles/bench_variants.sh $args | sed 's/^/   f16<64>: /'
done
