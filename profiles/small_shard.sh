cd $GRAFT_REPO_ROOT
for args in "--n 125000" "--n 250000" "--n 500000" "--m 1250" "--m 2500" "--m 5000"; do
  bash profiles/bench_variants.sh --steps 50 --warmup 5 $args
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_n125k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 5 --n 125000 > $GRAFT_REPO_ROOT/gpurun_out/prof_n125k.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_n125k -name "*kernel_stats.csv" -exec cat {} \; | cut -d, -f1-8
