#!/bin/bash
# regenerate the secondary bench lines quoted in DESIGN.md (one JSON line each) into gpurun_out/lines/
OUT=gpurun_out/lines; mkdir -p $OUT
B="python bench.py --no-cpu-baseline"
$B --workload c5 --steps 3 --warmup 2 > $OUT/r01_bench_c5_int8_ip.json 2>/dev/null
$B --dim 256 --steps 5 --warmup 2 > $OUT/r01_bench_d256_f32.json 2>/dev/null
$B --dim 512 --steps 5 --warmup 2 > $OUT/r01_bench_d512_f32.json 2>/dev/null
$B --dim 960 --steps 5 --warmup 1 > $OUT/r01_bench_d960_f32.json 2>/dev/null
$B --metric ip --steps 10 --warmup 2 > $OUT/r01_bench_ip_f32.json 2>/dev/null
$B --rows 1250000 --k 100 --steps 5 --warmup 2 > $OUT/r01_bench_k100_n1250k.json 2>/dev/null
$B --m 8 --steps 200 --warmup 10 > $OUT/r01_bench_m8.json 2>/dev/null
$B --m 1 --steps 50 --warmup 5 > $OUT/r01_bench_m1_latency.json 2>/dev/null
$B --sift-like --steps 10 --warmup 2 > $OUT/r01_bench_sift_like_f32.json 2>/dev/null
$B --dtype u8 --steps 10 --warmup 2 > $OUT/r01_bench_u8_d128.json 2>/dev/null
$B --dtype u8 --dim 960 --steps 5 --warmup 1 > $OUT/r01_bench_u8_d960.json 2>/dev/null
for f in $OUT/*.json; do python -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1]); r=j['roofline']
print('$f'.split('/')[-1], j['value'], j['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'])"; done
