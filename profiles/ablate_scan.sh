#!/bin/bash
# ablation of the default scan kernel (bench.py --debug bits: 1 no staging, 2 no MFMA, 4 no epilogue)
for d in ${@:-0 4 5 6}; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --debug $d 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug',$d,'ms/step',j['ms_per_step'],'scan_ms',j['roofline']['kernel_ms'],'cands',j['roofline']['candidates_per_query'])"
done
