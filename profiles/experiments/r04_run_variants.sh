#!/bin/bash
# bench.py --quick on each experimental library; prints the value and the five conv layers' launch times
for v in "$@"; do
  env CARTPOLEPP_ABLATION=$v python bench.py --quick --steps 200 --warmup 20 2>/tmp/err_$v.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['value'], [(r['layer'], r['avg_launch_us']) for r in d['layers'][:5]], d.get('non_conv_us_per_step'))
"
done
grep -h "K16CLK\|K16PRE" /tmp/err_k16clk.txt 2>/dev/null | sort | uniq -c | sort -rn | head -12
