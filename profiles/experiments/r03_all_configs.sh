for wl in cfg3 cfg2 cfg4 cfg5 r50; do
  python bench.py --quick --steps 100 --workload $wl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], [(r['layer'], r['avg_launch_us']) for r in d['layers'][:2]])"
done
CARTPOLEPP_ABLATION=1 CPP_DW16_PAIR=0 python bench.py --quick --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 nopair', d['value'], [(r['layer'], r['avg_launch_us']) for r in d['layers'][:2]])"
