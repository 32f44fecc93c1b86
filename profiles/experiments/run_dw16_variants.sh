#!/bin/bash
# bench.py --quick on each experimental library (build_dw16_variants.sh); conv1 dW time per launch is what is read
for v in "$@"; do
  env CARTPOLEPP_ABLATION=$v python bench.py --quick --steps 200 --warmup 20 2>/tmp/err_$v.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['value'], [(r['layer'], r['avg_launch_us']) for r in d['layers'][:2]])
"
done
grep -h DW16CLK /tmp/err_clock.txt 2>/dev/null | head -6
