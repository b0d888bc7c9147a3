#!/bin/bash
# profiles/ab_build_c5.sh [rounds] -- as ab_build.sh, on the shapes scan_gemm_i8x serves: C5 (int8 IP d768; 2 M rows here), uint8 d = 832 / 960
mkdir -p gpurun_out
R=${1:-2}
line() { python -c "
import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1]); r=j['roofline']; print('%-8s %-5s QPS %10.0f  ms/step %.3f  %s %.4f ms  frac %.4f' % ('$1','$2',j['value'],j['ms_per_step'],r['kernel'][:28],r['kernel_ms'],r['frac']))"; }
for i in $(seq $R); do
  for which in ${AB_LIBS:-base new}; do
    if [ $which = new ]; then unset EXPANN_LIB; else export EXPANN_LIB=$PWD/expann_amd/libexpann_hip_$which.so; fi
    timeout -k 10 300 python bench.py --workload c5 --rows 2000000 --dim 768 --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | line c5-2M $which
    for d in 768 832 960; do
      timeout -k 10 200 python bench.py --dim $d --dtype u8 --steps 10 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | line u8d$d $which
    done
  done
done
