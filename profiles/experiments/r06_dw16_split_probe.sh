for i in 1 2 3; do
  for v in "" probe; do
    CARTPOLEPP_ABLATION=$v python bench.py --quick --workload cfg5 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('variant=%-5s' % '$v', d['value'], ' '.join('%s %.4f' % (n, k[n]['ms_per_step']) for n in sorted(k, key=lambda n: -k[n]['ms_per_step'])[:3]))"
  done
done
