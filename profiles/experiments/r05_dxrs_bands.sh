#!/bin/bash
show() { python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); k=d.get('kernels',{})
print('$1', d['value'], {n:round(v['ms_per_step']*1000,1) for n,v in k.items() if 'conv2' in n})"; }
export CARTPOLEPP_ABLATION=1
for i in 1 2; do
python bench.py --quick --workload cfg4 2>/dev/null | show cfg4_bands
CPP_DXRS_BANDS=0 python bench.py --quick --workload cfg4 2>/dev/null | show cfg4_whole
python bench.py --quick 2>/dev/null | show cfg3_whole
CPP_DXRS_BANDS=2 python bench.py --quick 2>/dev/null | show cfg3_bands
CPP_DXRS_BANDS=2 CPP_PAIR_ORDER=1 python bench.py --quick 2>/dev/null | show cfg3_bands_dwfirst
CPP_DXRS_BANDS=2 CPP_PAIR_ORDER=2 python bench.py --quick 2>/dev/null | show cfg3_bands_mixed
done
