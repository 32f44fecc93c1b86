for i in 1 2; do for c in 4 2 3 1; do for o in 1 0; do
  env CARTPOLEPP_ABLATION=1 CPP_DWB16_CAP=$c CPP_PAIR_ORDER=$o python bench.py --quick --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cap $c order $o', d['value'], [(r['layer'], r['avg_launch_us']) for r in d['layers'][3:4]])"
done; done; done
