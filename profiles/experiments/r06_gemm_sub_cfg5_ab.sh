# GEMM 2 x 2 sub-tiles for K >= 2000 (cfg5's first fully connected level: K = 2561) against the shipped 1 x 1 tiling, alternating on one box.
# variant "sub": make OBJDIR=/tmp/objsub OUT=../lib/libcartpolepp_hip_sub.so CXXFLAGS="... -DGEMM_SUB_MIN_K=2000" ../lib/libcartpolepp_hip_sub.so
for i in 1 2 3; do
  for v in "" sub; do
    CARTPOLEPP_ABLATION=$v python bench.py --quick --workload cfg5 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('variant=%-5s' % '$v', d['value'], 'non_conv', d['non_conv_us_per_step'], ' '.join('%s %.4f' % (n, k[n]['ms_per_step']) for n in sorted(k, key=lambda n: -k[n]['ms_per_step'])[:9]))"
  done
done
