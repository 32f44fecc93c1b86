# per-configuration A/B of round 3's switches (ablation library): two-network conv1-dW workgroups, conv bands
for wl in r50 cfg2 cfg4; do
  for v in "X=1" "CPP_DW16_PAIR=0" "CPP_CONV_BANDS=0"; do
    env CARTPOLEPP_ABLATION=1 $v python bench.py --quick --steps 200 --workload $wl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl $v', d['value'], [(r['layer'], r['avg_launch_us']) for r in d['layers']])"
  done
done
