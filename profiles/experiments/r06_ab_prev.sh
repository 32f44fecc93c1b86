for i in 1 2 3; do
  for v in "" prev; do
    CARTPOLEPP_ABLATION=$v python bench.py --quick --steps 200 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('variant=%-5s' % '$v', d['value'], 'non_conv', d['non_conv_us_per_step'], 'gemm', k['gemm']['ms_per_step'], 'clip_sgd', k['clip_sgd']['ms_per_step'], 'dw_reduce', k.get('dw_reduce', {}).get('ms_per_step'), 'heads', k['heads']['ms_per_step'])"
  done
done
