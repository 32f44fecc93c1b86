# GEMM 2 x 2 sub-tiles for problems with >= 4096 tiles of 16 x 16 whatever their K (cfg5's dX of the first fully connected layer: 512 x 2560
# outputs = 5120 tiles of K = 100 / 200) against the shipped 1 x 1 tiling, alternating on one box.  variant "subt": common.h's gemm_sub() with
# the tile-count rule, whole library built with -DGEMM_SUB_MIN_K=99990 (which instantiates the 2 x 2 path).
for i in 1 2 3; do
  for v in "" subt; do
    CARTPOLEPP_ABLATION=$v python bench.py --quick --workload cfg5 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('variant=%-5s' % '$v', d['value'], 'gemm', k['gemm']['ms_per_step'], 'non_conv', d['non_conv_us_per_step'])"
  done
done
