#!/bin/bash
# ablation of the default scan kernel (bench.py --debug bits, scan_gemm_f16.hpp: 1 no barrier / wait,
# 2 no staging, 4 no epilogue, 8 no candidate path, 32 no LDS fragment reads; 16 = in-kernel clock)
for d in ${@:-0 8 12 14 46}; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --debug $d --no-cpu-baseline --no-verify 2>gpurun_out/ablate_$d.err | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug',$d,'ms/step',j['ms_per_step'],'scan_ms',j['roofline']['kernel_ms'],'frac',j['roofline']['frac'],'cands',j['roofline']['candidates_per_query'])"
  grep -h "MHz\|resident" gpurun_out/ablate_$d.err | sort | uniq -c | head -3
done
