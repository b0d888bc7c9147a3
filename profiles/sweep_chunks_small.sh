#!/bin/bash
# profiles/sweep_chunks_small.sh -- forced row-chunk counts of the fp16 scan on the shard sizes of C2 over 8 / 4 / 2 GPUs
# (125 k / 250 k / 500 k rows x 10 k queries) against the planner's choice (0)
for n in ${ROWS:-125000 250000 500000}; do
  for c in 0 8 12 16 24 25 32 48 64; do
    python bench.py --rows $n --steps 50 --warmup 5 --scan-chunks $c --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=j['roofline']; print('rows',j['config']['n'],'chunks',$c,'ms/step',j['ms_per_step'],'scan',r['kernel_ms'],'frac',r['frac'])"
  done
done
