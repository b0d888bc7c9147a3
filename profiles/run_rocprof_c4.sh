#!/bin/bash
# profiles/run_rocprof_c4.sh <tag> [rows]: rocprofv3 evidence for the graph path (C4): kernel trace +
# stats of the batched GPU build and the traversal sweep, then FETCH_SIZE of the same command
# (separate --pmc pass).  Summaries: profiles/<tag>_kernel_stats_c4.csv, <tag>_c4_pmc.txt
set -e
TAG=${1:-r02}; ROWS=${2:-200000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_c4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TOOL="$ROOT/expann_amd/host/expann_graph_tool --M 60 --ef_construction 480 --n $ROWS --m 10000 --d 128 --k 10 --data sift --batched 1024 --index /tmp/c4prof.index --ef 10,60"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $TOOL > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $TOOL > "$OUT/fetch.log" 2>&1
python3 - "$OUT" "$ROOT/profiles/${TAG}" <<'PY'
import csv, glob, sys, collections
out, dst = sys.argv[1], sys.argv[2]
st = glob.glob(out + "/trace/*/*_kernel_stats.csv")
rows = list(csv.DictReader(open(st[0])))
with open(dst + "_kernel_stats_c4.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"].replace("void ", "").replace("expann::", "")[:90], r["Calls"], r["TotalDurationNs"],
                    r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(out + "/fetch/*/*_counter_collection.csv")[0])):
    k = r["Kernel_Name"].replace("void ", "").replace("expann::", "")[:60]
    if "graph_search" in k or "build_" in k:
        acc[k].append(float(r["Counter_Value"]))
with open(dst + "_c4_pmc.txt", "w") as f:
    f.write("FETCH_SIZE (KiB as reported; x2 on gfx950 for streamed bytes, see r02_summary.json calibration) per launch\n")
    for k, v in acc.items():
        f.write(f"{k}: launches {len(v)}, mean {sum(v)/len(v):.0f} KiB, max {max(v):.0f} KiB, total {sum(v):.0f} KiB\n")
print(open(dst + "_c4_pmc.txt").read())
PY
grep phase "$OUT/trace.log" | cut -c1-260
