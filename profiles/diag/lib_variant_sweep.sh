#!/bin/bash
# A/B of experimental library builds: every cartpoleplusplus_amd/lib/libexp_*.so is copied over the ablation library in turn and
# bench.py --quick is run twice on it (CARTPOLEPP_ABLATION=1 selects that library; no CPP_* switch is set).
L=cartpoleplusplus_amd/lib
cp $L/libcartpolepp_hip_ablation.so /tmp/abl_keep.so
export CARTPOLEPP_ABLATION=1
for round in 1 2 3; do
for v in $L/libexp_*.so; do
  cp $v $L/libcartpolepp_hip_ablation.so
  echo "== $(basename $v)"
  python bench.py --quick --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], [(r['layer'], r['avg_launch_us']) for r in d['layers']])
"
done
done
cp /tmp/abl_keep.so $L/libcartpolepp_hip_ablation.so
