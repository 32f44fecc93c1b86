# the 128-bit row stores of the row-streaming kernels with their data registers held for four wait states (buffer_store_b128_held) against
# the previous commit (lib/libcartpolepp_hip_prev.so), alternating on one box: cfg3 and cfg5
for w in cfg3 cfg5; do
for i in 1 2 3; do
  for v in "" prev; do
    CARTPOLEPP_ABLATION=$v python bench.py --quick --workload $w --steps 200 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('$w variant=%-5s' % '$v', d['value'], ' '.join('%s %.4f' % (n, k[n]['ms_per_step']) for n in sorted(k, key=lambda n: -k[n]['ms_per_step'])[:6]))"
  done
done
done
