#!/bin/bash
# profiles/ab_i8w_sample.sh -- the sampled pass of the 8-bit d = 128 / 256 scans: scan_gemm_i8w_kernel<D, L2F, true>
# (EXPANN_I8W_SAMPLE=1, default) vs round 1's scan_gemm_i8q_kernel<D, L2F, true> (=0), same box, alternating
mkdir -p gpurun_out
line() { python -c "
import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1]); r=j['roofline']; print('%-10s sample_w=%s QPS %10.0f  ms/step %.3f  scan %.4f ms' % ('$1','$2',j['value'],j['ms_per_step'],r['kernel_ms']))"; }
for i in 1 2; do
  for v in 0 1; do
    export EXPANN_I8W_SAMPLE=$v
    timeout -k 10 200 python bench.py --dtype u8 --steps 30 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | line u8d128 $v
    timeout -k 10 200 python bench.py --sift-like --steps 30 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | line sift $v
    timeout -k 10 200 python bench.py --dtype u8 --k 100 --rows 1250000 --steps 20 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | line u8k100 $v
    timeout -k 10 200 python bench.py --dtype i8 --dim 256 --steps 20 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | line i8d256 $v
  done
done
