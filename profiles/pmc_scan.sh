#!/bin/bash
# profiles/pmc_scan.sh <tag> <bench args...>: SQ counters of the scan kernels (one --pmc pass; no trace
# options beside it), condensed to one line per kernel: where the waves' cycles go.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --no-cpu-baseline --no-verify --steps 3 --warmup 1 "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")
if not f:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("void ", "").replace("expann::", "")[:60]
    if "scan_gemm" not in k and "graph_search" not in k:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    n = len(c["SQ_WAVE_CYCLES"])
    # keep the long launches only (the full scans)
    top = sorted(range(n), key=lambda i: -c["SQ_WAVE_CYCLES"][i])[:max(1, n // 2)]
    m = {name: sum(v[i] for i in top) / len(top) for name, v in c.items()}
    wc = m["SQ_WAVE_CYCLES"]
    print(k, "launches", len(top), " ".join(f"{name}={val:.4g}" for name, val in sorted(m.items())))
    print("   of wave cycles: wait_any %.3f  wait_inst_any %.3f (lds %.3f)  active_inst %.3f ; mfma_busy/busy_cycles %.3f ; gui_active/8 %.4g" % (
        m["SQ_WAIT_ANY"] / wc, m["SQ_WAIT_INST_ANY"] / wc, m["SQ_WAIT_INST_LDS"] / wc, m["SQ_ACTIVE_INST_ANY"] / wc,
        m["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, m["SQ_BUSY_CYCLES"]), m["GRBM_GUI_ACTIVE"] / 8))
PY
