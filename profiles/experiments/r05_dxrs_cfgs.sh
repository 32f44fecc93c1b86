#!/bin/bash
show() { python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d.get('kernels',{})
print('$1', d['value'], {n:round(v['ms_per_step']*1000,1) for n,v in k.items() if 'conv2' in n or 'conv3' in n})"; }
export CARTPOLEPP_ABLATION=1
for w in cfg2 cfg4; do
python bench.py --quick --workload $w 2>/dev/null | show $w
CPP_CONV_DXRS=0 python bench.py --quick --workload $w 2>/dev/null | show ${w}_old
done
