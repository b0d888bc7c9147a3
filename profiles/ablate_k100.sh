#!/bin/bash
# ablation of the default scan on C3's per-GPU shape (1.25 M rows, k = 100): bench.py --debug bits of the
# debug instance (scan_gemm_f16x.hpp: 8 no candidate path, 64 flushes drop their hits, 16 in-kernel clock)
mkdir -p gpurun_out
for d in ${@:-0 64 8 16}; do
  timeout -k 10 200 python bench.py --rows 1250000 --k 100 --steps 20 --warmup 3 --debug $d --no-cpu-baseline --no-verify 2>gpurun_out/ablk_$d.err | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug',$d,'ms/step',j['ms_per_step'],'scan_ms',j['roofline']['kernel_ms'],'frac',j['roofline']['frac'],'cands',j['roofline']['candidates_per_query'])"
  grep -h "MHz\|resident\|seg" gpurun_out/ablk_$d.err | sort | uniq -c | head -6
done
