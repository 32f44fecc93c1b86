for i in 1 2; do for nb in 2 4; do for wl in cfg3 cfg4; do
  env CARTPOLEPP_ABLATION=1 CPP_CONV_NBANDS=$nb python bench.py --quick --steps 200 --workload $wl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl nbands $nb', d['value'], [(r['layer'], r['avg_launch_us']) for r in d['layers'][3:]])"
done; done; done
