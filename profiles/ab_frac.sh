#!/bin/bash
# same box, back to back: the size-scaled sample fraction (default) vs round 1's (by k only)
ab() { old=$1; shift; bash profiles/bench_variants.sh "$@"; bash profiles/bench_variants.sh "$@" --sample-frac $old | sed 's/^/   old: /'; }
ab 16 --workload c5 --steps 3
ab 16 --n 1000000 --d 960 --steps 3
ab 16 --n 1000000 --d 768 --steps 3
ab 16 --n 10000000 --steps 3
ab 8 --n 10000000 --k 100 --steps 3
ab 16 --n 1000000 --d 960 --dtype u8 --steps 3
ab 16 --n 4000000 --steps 5
