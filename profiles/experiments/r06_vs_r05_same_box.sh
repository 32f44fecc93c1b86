# round 6's library against round 5's (lib/libcartpolepp_hip_r05.so: the sources of commit 9432a27 built by the same toolchain, loaded by
# CARTPOLEPP_ABLATION=r05 under this round's Python -- the ABI did not change), alternating on ONE box
for w in cfg3 cfg2 cfg4; do
for i in 1 2 3; do
  for v in "" r05; do
    CARTPOLEPP_ABLATION=$v python bench.py --quick --workload $w --steps 200 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w lib=%-4s' % ('$v' or 'r06'), d['value'], [(l['layer'][:13], l['avg_launch_us']) for l in d['layers'][:4]], 'non_conv', d['non_conv_us_per_step'])"
  done
done
done
# Result (one MI355X, steps/s, three alternations each):
#   cfg3  r06 3164 / 3171 / 3160   r05 3151 / 3143 / 3158   (+0.5 %: conv1 dW 56.1 -> 53.3 us, the optimiser's launch -2; conv1 forward +0.4)
#   cfg2  r06 3624 / 3604 / 3627   r05 3498 / 3505 / 3505   (+3.3 %: conv1 forward 82.6 -> 62.8 us)
#   cfg4  r06 4429 / 4437 / 4429   r05 4405 / 4390 / 4412   (+0.7 %)
