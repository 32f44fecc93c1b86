#!/bin/bash
show() { python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d.get('kernels',{})
print('$1', d['value'], {n:round(v['ms_per_step']*1000,1) for n,v in k.items() if 'conv2' in n})"; }
export CARTPOLEPP_ABLATION=1
for i in 1 2; do
python bench.py --quick 2>/dev/null | show base
CPP_CONV2_PAIR=0 python bench.py --quick 2>/dev/null | show nopair2
CPP_PAIR_ORDER=0 python bench.py --quick 2>/dev/null | show order0
CPP_PAIR_ORDER=2 python bench.py --quick 2>/dev/null | show order2
CPP_DWB16_CAP=2 python bench.py --quick 2>/dev/null | show dwb16cap2
CPP_DWB16_CAP=8 python bench.py --quick 2>/dev/null | show dwb16cap8
done
python bench.py --quick --workload cfg5 2>/dev/null | show cfg5
CPP_CONV_DXRS=0 python bench.py --quick --workload cfg5 2>/dev/null | show cfg5_old
