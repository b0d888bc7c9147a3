#!/bin/bash
# the committed round-end evidence in one go (writes under gpurun_out/final/)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_line.json 2>$OUT/bench_line.err
python bench.py --workload c5 --steps 3 --warmup 1 > $OUT/${TAG}_bench_c5_int8_ip.json 2>$OUT/bench_c5.err
python bench.py --workload c4 > $OUT/${TAG}_bench_c4_graph_1m.json 2>$OUT/bench_c4.err
bash profiles/run_rocprof.sh $TAG > $OUT/run_rocprof.log 2>&1 && python profiles/summarize.py $TAG > $OUT/summarize.log 2>&1
cp profiles/${TAG}_kernel_stats_*.csv profiles/${TAG}_summary.json $OUT/ 2>/dev/null
for f in $OUT/${TAG}_bench_*.json; do python - "$f" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["frac"], j.get("bit_exact"), j.get("recall_measured"))
PY
done
