# conv_dw16.h's 2^S from the bounds conv2's dX leaves (as conv_dw16_rs.h since earlier in round 6) instead of a scan of the unit's pooled rows:
# cfg2 (conv_dw16_kernel<9>) and cfg4 (NAF: conv1_dw_gather_kernel<2>) against the previous commit, alternating on one box
for w in cfg2 cfg4; do
for i in 1 2 3; do
  for v in "" prev; do
    CARTPOLEPP_ABLATION=$v python bench.py --quick --workload $w --steps 200 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('$w variant=%-5s' % '$v', d['value'], ' '.join('%s %.4f' % (n, k[n]['ms_per_step']) for n in sorted(k) if 'conv1_dw' in n))"
  done
done
done
