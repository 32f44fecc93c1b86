# the fully connected layers' dW GEMMs behind conv3's backward pair (default) against the GEMM levels carrying them (CPP_RIDE_FC_DW=0),
# ablation build, one box, alternating
for w in cfg3 cfg2; do
for i in 1 2 3; do
  for v in 1 0; do
    CARTPOLEPP_ABLATION=1 CPP_RIDE_FC_DW=$v python bench.py --quick --workload $w --steps 200 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('$w ride=$v', d['value'], 'non_conv', d['non_conv_us_per_step'], 'gemm', k['gemm']['ms_per_step'], 'conv3_bwd', k['conv3_bwd']['ms_per_step'])"
  done
done
done
# Result (one MI355X, steps/s; HIP-event pass per launch): riding 3061 / 3054 / 3062 against 3087 / 3117 / 3106 at cfg3, 3514 / 3506 / 3525
# against 3569 / 3570 / 3572 at cfg2.  The four GEMM levels lose 6 us (44.5 -> 38.5) and conv3's backward launch gains 9.5 (16.0 -> 25.6):
# the tiles inherit the pair kernel's 80 KB of LDS and its two workgroups per CU, where a GEMM level runs them eight to a CU -- the
# latency of a tile (one operand round trip per 64 k) is hidden by occupancy, not by the chip being idle.  Not kept (r06_fc_dw_ride.diff).
