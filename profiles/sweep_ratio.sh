#!/bin/bash
# sweep of the threshold-level sampling ratio (bench.py --sample-ratio)
for r in ${@:-8 10 16 32 64}; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --sample-ratio $r 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ratio',$r,'ms/step',j['ms_per_step'],'scan_ms',j['roofline']['kernel_ms'],'cands',j['roofline']['candidates_per_query'])"
done
