#!/bin/bash
# secondary configurations next to the default C2 line (one JSON line each, abbreviated)
run() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print(' '.join(sys.argv[1:]), '| QPS', j['value'], 'ms/step', j['ms_per_step'], 'scan_ms', r['kernel_ms'], r['kernel'], 'frac', r['frac'], 'cands', r['candidates_per_query'])" "$@"; }
if [ $# -gt 0 ]; then run "$@"; exit 0; fi
# round 3: persistent launch of the d = 128 scan (resident workgroups pull items) next to the plain launch
for args in "" "--clustered 1000" "--k 100 --rows 1250000" "--metric ip" "--queries 1000" "--queries 256" "--sift-like"; do
  run --steps 10 $args
  EXPANN_PERSIST=0 run --steps 10 $args | sed 's/^/   plain launch: /'
done
run --n 1250000 --k 100 --steps 5
run --n 10000000 --k 100 --steps 3
run --n 1000000 --k 10 --m 1000 --steps 20
run --n 1000000 --k 10 --m 256 --steps 20
run --n 1000000 --d 64 --k 10 --steps 10
run --n 1000000 --dtype u8 --steps 10
run --n 1000000 --dtype i8 --metric ip --steps 10
run --n 1000000 --d 256 --dtype i8 --steps 10
run --n 1000000 --d 256 --steps 10
run --n 1000000 --d 512 --steps 5
run --n 1000000 --d 768 --steps 3
run --n 1000000 --d 832 --steps 3
run --n 1000000 --d 960 --steps 3
run --n 1000000 --d 960 --metric ip --steps 3
run --n 1000000 --d 960 --dtype u8 --steps 3
run --n 1000000 --d 832 --dtype u8 --steps 3
run --n 1000000 --k 10 --m 1 --steps 50
run --n 1000000 --k 10 --m 4 --steps 50
run --workload c3 --steps 3
run --workload c5 --steps 3
