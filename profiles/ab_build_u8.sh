#!/bin/bash
# profiles/ab_build_u8.sh [rounds] -- as ab_build.sh, on the shapes scan_gemm_i8w<128> serves: uint8 d128 rows, SIFT-like fp32 rows
mkdir -p gpurun_out
R=${1:-3}
line() { python -c "
import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1]); r=j['roofline']; print('%-8s %-5s QPS %10.0f  ms/step %.3f  %s %.4f ms  frac %.4f' % ('$1','$2',j['value'],j['ms_per_step'],r['kernel'][:28],r['kernel_ms'],r['frac']))"; }
for i in $(seq $R); do
  for which in ${AB_LIBS:-base new}; do
    if [ $which = new ]; then unset EXPANN_LIB; else export EXPANN_LIB=$PWD/expann_amd/libexpann_hip_$which.so; fi
    timeout -k 10 200 python bench.py --dtype u8 --steps 30 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | line u8d128 $which
    timeout -k 10 200 python bench.py --sift-like --steps 30 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | line sift $which
    timeout -k 10 200 python bench.py --dtype u8 --k 100 --rows 1250000 --steps 20 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | line u8k100 $which
  done
done
