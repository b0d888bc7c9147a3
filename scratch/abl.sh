for dbg in 1 9 11; do
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --debug $dbg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('debug', $dbg, d['roofline']['kernel_ms'])" || exit 1
done
