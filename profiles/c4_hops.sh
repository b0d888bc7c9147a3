#!/bin/bash
# profiles/c4_hops.sh [rows] -- the graph walk at C4's size with the instrumented kernel instance
# (EXPANN_GRAPH_STAMPS=1: shader clocks per phase of a hop), next to the production instance's timing.
ROWS=${1:-1000000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out
TOOL="$ROOT/expann_amd/host/expann_graph_tool --M 60 --ef_construction 480 --n $ROWS --m 10000 --d 128 --k 10 --data sift --batched 1024 --index /tmp/c4hops.index --ef 10,30,60"
$TOOL > $ROOT/gpurun_out/c4_plain.json 2> $ROOT/gpurun_out/c4_plain.err || exit 1
EXPANN_GRAPH_STAMPS=1 $TOOL --read-index 1 > $ROOT/gpurun_out/c4_stamps.json 2> $ROOT/gpurun_out/c4_stamps.txt || exit 1
python3 - $ROOT/gpurun_out/c4_plain.json <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if d.get("phase") == "query":
        print("plain: compression %d ef %3d: kernel %.3f ms, %.0f distcomps/query, recall %.4f" % (d["use_compression"], d["ef_search"], d["kernel_ms"], d["distcomps_per_query"], d["recall"]))
    else:
        print({k: d[k] for k in d if k in ("phase", "builder", "time_to_build_ns", "batches", "dropped_reverse_edges", "rows_repruned")})
PY
cat $ROOT/gpurun_out/c4_stamps.txt
