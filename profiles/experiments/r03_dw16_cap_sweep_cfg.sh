for wl in cfg2 cfg4 cfg5 r50; do
for v in "v1 X=1" "cap2 X=1" "cap3 X=1"; do
  set -- $v
  env CARTPOLEPP_ABLATION=$1 $2 python bench.py --quick --steps 100 --workload $wl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl $v', d['value'], [(r['layer'], r['avg_launch_us']) for r in d['layers'][:2]])"
done; done
