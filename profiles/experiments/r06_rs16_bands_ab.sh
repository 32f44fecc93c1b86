# conv1 forward of NAF's two trunks (cfg4: 256 workgroups of conv_fwd_rs16_kernel = one wave per SIMD) as two bands of rows per image
# (512 workgroups) against the previous commit's whole images (lib/libcartpolepp_hip_prev.so), alternating on one box
for i in 1 2 3; do
  for v in "" prev; do
    CARTPOLEPP_ABLATION=$v python bench.py --quick --workload cfg4 --steps 200 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('variant=%-5s' % '$v', d['value'], ' '.join('%s %.4f' % (n, k[n]['ms_per_step']) for n in sorted(k, key=lambda n: -k[n]['ms_per_step'])[:4]))"
  done
done
