for i in 1 2; do
for v in "v1 X=1" "v1 CPP_DW16_PAIR=0" "cap2 CPP_DW16_PAIR=0" "cap3 CPP_DW16_PAIR=0" "cap3 X=1"; do
  set -- $v
  env CARTPOLEPP_ABLATION=$1 $2 python bench.py --quick --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], [(r['layer'], r['avg_launch_us']) for r in d['layers'][:2]])"
done; done
