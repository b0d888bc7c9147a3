#!/bin/bash
# profiles/sweep_frac_r3.sh -- the sampled pass reads 1/frac of the rows: frac re-swept on the round-3 build (hits got
# cheaper: lighter flush, keys made by the gather, pruned selects) at k = 100 (10 M and 1.25 M rows) and k = 10 (1 M, 10 M)
line() { python -c "
import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')][-1]); r=j['roofline']; print('%-22s frac %-3s QPS %10.0f  ms/step %.3f  scan %.4f ms  cands %.0f' % ('$1','$2',j['value'],j['ms_per_step'],r['kernel_ms'],r['candidates_per_query']))"; }
for f in 0 6 8 12 16 24; do
  python bench.py --workload c3 --steps 3 --warmup 1 --sample-frac $f --no-cpu-baseline --no-verify 2>/dev/null | line "10M k100" $f
done
for f in 0 6 8 10 12 16; do
  python bench.py --rows 1250000 --k 100 --steps 20 --warmup 3 --sample-frac $f --no-cpu-baseline --no-verify 2>/dev/null | line "1.25M k100" $f
done
for f in 0 12 16 24 32; do
  python bench.py --steps 20 --warmup 3 --sample-frac $f --no-cpu-baseline --no-verify 2>/dev/null | line "1M k10" $f
done
for f in 0 16 32 48 64; do
  python bench.py --rows 10000000 --steps 3 --warmup 1 --sample-frac $f --no-cpu-baseline --no-verify 2>/dev/null | line "10M k10" $f
done
