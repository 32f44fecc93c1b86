# conv1 dW at cfg2 (9 channels): two networks per workgroup (CPP_DW16_PAIR9=1, ablation build) against one (shipped), alternating
for i in 1 2 3; do
  for v in 0 1; do
    CARTPOLEPP_ABLATION=1 CPP_DW16_PAIR9=$v python bench.py --quick --workload cfg2 --steps 200 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('pair9=$v', d['value'], ' '.join('%s %.4f' % (n, k[n]['ms_per_step']) for n in sorted(k, key=lambda n: -k[n]['ms_per_step'])[:5]))"
  done
done
