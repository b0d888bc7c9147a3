#!/bin/bash
# sweep of the sampled-pass fraction (bench.py --sample-frac F: the pass reads 1/F of the rows)
for r in ${@:-8 12 16 24 32}; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --sample-frac $r 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frac',$r,'ms/step',j['ms_per_step'],'scan_ms',j['roofline']['kernel_ms'],'cands',j['roofline']['candidates_per_query'])"
done
