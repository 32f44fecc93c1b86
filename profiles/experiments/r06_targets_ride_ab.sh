# both target updates in the outer step's last optimiser launch (default) against soft_update_kernel's own launch (CPP_RIDE_TARGETS=0),
# ablation build, one box, alternating; cfg3 with 5 minibatches per outer step and with 1 (SURVEY 8d's second variant)
for bps in 5 1; do
for i in 1 2 3; do
  for v in 1 0; do
    CARTPOLEPP_ABLATION=1 CPP_RIDE_TARGETS=$v python bench.py --quick --batches-per-step $bps --steps 200 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batches_per_step=$bps ride=$v', d['value'], 'non_conv', d['non_conv_us_per_step'])"
  done
done
done
