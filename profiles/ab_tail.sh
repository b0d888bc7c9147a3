#!/bin/bash
# A/B: two-size row chunks of scan_gemm_f16x (EXPANN_TAIL_CHUNKS=1, default) vs equal chunks (0), same box
for args in "--steps 20" "--clustered 1000 --steps 20" "--rows 1250000 --k 100 --steps 10" "--n 500000 --steps 20" "--n 750000 --steps 20" "--n 2000000 --steps 10" "--n 4000000 --steps 5" "--n 10000000 --steps 3" "--n 4000000 --d 64 --steps 5"; do
  bash profiles/bench_variants.sh $args
  EXPANN_TAIL_CHUNKS=0 bash profiles/bench_variants.sh $args | sed 's/^/   equal: /'
done
