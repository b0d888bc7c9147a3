#!/bin/bash
# same box, back to back: scan_gemm_f16x<128> as a persistent launch (resident workgroups pull (query tile, row
# chunk) items from per-XCD counters) against the plain launch, C2 and larger shapes
B="python bench.py --no-cpu-baseline --steps 30 --warmup 5"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'QPS', d['value'], 'ms/step', d['ms_per_step'], 'scan ms', r['kernel_ms'], 'frac', r['frac'], 'bit_exact', d.get('bit_exact'))"; }
for rep in 1 2; do
for p in 1 0; do
EXPANN_PERSIST=$p $B | show "c2 persist=$p"
done; done
for p in 1 0; do
EXPANN_PERSIST=$p $B --rows 4000000 --steps 10 | show "4M rows persist=$p"
EXPANN_PERSIST=$p $B --rows 1250000 --k 100 --steps 15 | show "1.25M k=100 persist=$p"
EXPANN_PERSIST=$p $B --clustered 1000 --steps 15 | show "clustered persist=$p"
done
