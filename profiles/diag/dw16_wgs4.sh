#!/bin/bash
# conv1 dW at four workgroups per CU (one round of 1024), with the sample pass riding in it or in the reductions' launch
L=cartpoleplusplus_amd/lib
cp $L/libcartpolepp_hip_ablation.so /tmp/abl_keep.so
export CARTPOLEPP_ABLATION=1
run() { lib=$1; shift; cp $L/$lib $L/libcartpolepp_hip_ablation.so; echo "== $lib $*"; env "$@" python bench.py --quick --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels']; print(d['value'], d['ms_per_step'], {n: k[n]['ms_per_step'] for n in k if 'conv1_dw' in n or n in ('dw_reduce', 'gather_stats', 'reduce_gather')})
"; }
for r in 1 2; do
run libexp_v0.so X=0
run libexp_w4.so X=0
run libexp_v0.so CPP_RIDE_DW=0
run libexp_w4.so CPP_RIDE_DW=0
done
cp /tmp/abl_keep.so $L/libcartpolepp_hip_ablation.so
