#!/bin/bash
# profiles/quick_stats.sh <tag> [bench args...] -- one rocprofv3 --kernel-trace --stats pass of bench.py with
# the given arguments; prints the per-kernel table (top 14 rows) and leaves the raw output under gpurun_out/.
set -e
TAG=${1:-q}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
grep '^{"metric"' "$OUT/bench.log" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'], r['candidates_per_query'])"
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %5s avg %10.1f us  %5s%%" % (r["Name"].replace("void ", "").replace("expann::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
