#!/bin/bash
# the committed round-end evidence in one go (writes under gpurun_out/final/; copy what is to be judged into
# profiles/).  TAG = round (r03): bench lines of C2 (the headline), C3's whole index on one GPU, C5 and C4;
# rocprofv3 kernel stats + PMC traffic of the same commands (run_rocprof.sh + summarize.py, run_rocprof_c4.sh at
# C4's real size); the graph walk's per-phase stamps; the secondary shapes.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2>$OUT/bench_line.err
echo "c2 done"
python bench.py --workload c3 --steps 3 --warmup 1 > $OUT/${TAG}_bench_c3_10m_k100.json 2>$OUT/bench_c3.err
python bench.py --workload c5 --steps 3 --warmup 1 > $OUT/${TAG}_bench_c5_int8_ip.json 2>$OUT/bench_c5.err
echo "c3 c5 done"
python bench.py --workload c4 > $OUT/${TAG}_bench_c4_graph_1m.json 2>$OUT/bench_c4.err
echo "c4 done"
bash profiles/run_rocprof.sh $TAG > $OUT/run_rocprof.log 2>&1 && python profiles/summarize.py $TAG > $OUT/summarize.log 2>&1
echo "rocprof done"
bash profiles/run_rocprof_c4.sh $TAG 1000000 > $OUT/run_rocprof_c4.log 2>&1
echo "rocprof c4 done"
bash profiles/c4_hops.sh 1000000 > $OUT/${TAG}_c4_hop_breakdown.txt 2>&1
bash profiles/bench_variants.sh > $OUT/${TAG}_bench_variants.txt 2>&1
cp profiles/${TAG}_kernel_stats_*.csv profiles/${TAG}_summary.json profiles/${TAG}_c4_pmc.txt $OUT/ 2>/dev/null
for f in $OUT/${TAG}_bench_*.json; do python - "$f" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["frac"], j.get("bit_exact"), j.get("recall_measured"))
PY
done
