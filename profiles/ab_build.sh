#!/bin/bash
# profiles/ab_build.sh [rounds] -- two builds of the library on ONE box, alternating: expann_amd/libexpann_hip_base.so
# (a copy of the build to compare with, kept by hand; AB_LIBS="base new x" adds libexpann_hip_x.so) vs the current one, on C2, C3's per-GPU shape and d = 64.
mkdir -p gpurun_out
R=${1:-3}
line() { python -c "
import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1]); r=j['roofline']; print('%-6s %-5s QPS %10.0f  ms/step %.3f  scan %.4f ms  frac %.4f' % ('$1','$2',j['value'],j['ms_per_step'],r['kernel_ms'],r['frac']))"; }
for i in $(seq $R); do
  for which in ${AB_LIBS:-base new}; do
    if [ $which = new ]; then unset EXPANN_LIB; else export EXPANN_LIB=$PWD/expann_amd/libexpann_hip_$which.so; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | line c2 $which
    timeout -k 10 200 python bench.py --rows 1250000 --k 100 --steps 20 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | line k100 $which
    timeout -k 10 200 python bench.py --dim 64 --steps 30 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | line d64 $which
  done
done
