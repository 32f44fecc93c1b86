#!/bin/bash
# lib_variant_sweep.sh for another workload: $1 = workload, $2 = steps
L=cartpoleplusplus_amd/lib
cp $L/libcartpolepp_hip_ablation.so /tmp/abl_keep.so
export CARTPOLEPP_ABLATION=1
for round in 1 2; do
for v in $L/libexp_*.so; do
  cp $v $L/libcartpolepp_hip_ablation.so
  echo "== $(basename $v)"
  python bench.py --quick --workload $1 --steps $2 --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], [(r['layer'], r['avg_launch_us']) for r in d['layers']][:2])
"
done
done
cp /tmp/abl_keep.so $L/libcartpolepp_hip_ablation.so
