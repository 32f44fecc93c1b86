#!/bin/bash
# conv1 dW: units per image / grid per network / where the sample pass's workgroups sit in the grid (experimental build:
# -DDW16_EXPERIMENT objects linked into the ablation library).  Prints steps/s and the two conv1-dW kernels' launch times.
export CARTPOLEPP_ABLATION=1
run() {
  echo "== $*"
  env "$@" python bench.py --quick --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        k = d['kernels']
        row = [r for r in d['layers'] if r['layer'] == 'conv1 dW'][0]
        print(d['value'], 'steps/s', d['ms_per_step'], 'ms;  conv1 dW avg us', row['avg_launch_us'], {n: k[n] for n in k if 'conv1_dw' in n or n in ('dw_reduce', 'gather_stats')})
"
}
run X=0
run DW16_UPI=3 DW16_GRID=384
run DW16_UPI=3 DW16_GRID=384 DW16_RIDER=1
run DW16_UPI=3 DW16_GRID=384 DW16_RIDER=3
run DW16_UPI=3 DW16_GRID=384 DW16_RIDER=4
run DW16_UPI=2 DW16_RIDER=1
run DW16_UPI=2 DW16_RIDER=3
run DW16_UPI=3 DW16_GRID=512
run DW16_UPI=6 DW16_GRID=384
run DW16_UPI=4 DW16_GRID=512
run X=0
