#!/bin/bash
# profiles/run_rocprof.sh <tag> -- collect the rocprofv3 evidence for bench.py on the GPU box.
# Writes raw output under gpurun_out/prof_<tag>/ (scratch); profiles/summarize.py turns it
# into the small per-round summaries that are committed under profiles/.
# Separate passes: (1) --kernel-trace --stats of the default bench (C2), (2) the same for the
# single-query pass (m=1: one pass over the base, HBM-bound), the d = 960 k-split kernel, k = 100, C5,
# (3)+(4) PMC FETCH_SIZE passes.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c2_trace" -- python3 $B --steps 20 --warmup 3 > "$OUT/c2_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/m1_trace" -- python3 $B --steps 50 --warmup 2 --m 1 > "$OUT/m1_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/d960_trace" -- python3 $B --steps 5 --warmup 1 --rows 1000000 --dim 960 > "$OUT/d960_trace.log" 2>&1
# round 2: k = 100 on a 1.25 M-row shard (C3 per GPU) and C5 (10 M x d768 int8 inner product)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/k100_trace" -- python3 $B --steps 10 --warmup 2 --rows 1250000 --k 100 > "$OUT/k100_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_trace" -- python3 $B --workload c5 --steps 3 --warmup 1 --no-verify > "$OUT/c5_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/c5_fetch" -- python3 $B --workload c5 --steps 2 --warmup 1 --no-verify > "$OUT/c5_fetch.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/c2_fetch" -- python3 $B --steps 2 --warmup 1 > "$OUT/c2_fetch.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/m1_fetch" -- python3 $B --steps 5 --warmup 1 --m 1 > "$OUT/m1_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/c2_write" -- python3 $B --steps 2 --warmup 1 > "$OUT/c2_write.log" 2>&1
find "$OUT" -name "*.csv" | head -40
