# conv1 dW at cfg5 (30 channels, 128 wide): conv_dw16.h with TWO networks per workgroup (one resident workgroup per CU, every A fragment
# multiplied with both networks' dY) against the shipped one network per workgroup (two per CU); ablation build, CPP_DW16_PAIR30=1.
for i in 1 2 3; do
  for v in 0 1; do
    CARTPOLEPP_ABLATION=1 CPP_DW16_PAIR30=$v python bench.py --quick --workload cfg5 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('pair30=$v', d['value'], 'non_conv', d['non_conv_us_per_step'], ' '.join('%s %.4f' % (n, k[n]['ms_per_step']) for n in sorted(k, key=lambda n: -k[n]['ms_per_step'])[:4]))"
  done
done
CARTPOLEPP_ABLATION=1 CPP_DW16_PAIR30=1 python -m pytest tests/test_gpu_fused_fullsize.py -q -k cfg5 -s 2>&1 | tail -4
