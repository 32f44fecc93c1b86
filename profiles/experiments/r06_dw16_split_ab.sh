# conv1 dW at cfg5 (conv_dw16_kernel<30, 5, 4>): the (MT / 2) x 2 division of the accumulator tiles over a workgroup's waves against the
# MT x 1 one of the previous commit (lib/libcartpolepp_hip_prev.so), alternating on one box; and that the gradients are the SAME BITS.
for i in 1 2 3; do
  for v in "" prev; do
    CARTPOLEPP_ABLATION=$v python bench.py --quick --workload cfg5 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('variant=%-5s' % '$v', d['value'], ' '.join('%s %.4f' % (n, k[n]['ms_per_step']) for n in sorted(k, key=lambda n: -k[n]['ms_per_step'])[:3]))"
  done
done
for v in "" prev; do
CARTPOLEPP_ABLATION=$v python - <<'PY'
import hashlib, numpy as np
from tests.helpers import make_pair
agent, _ref, _ = make_pair((128, 128, 3, 2, 5), 64, True, replay_size=256)
agent.replay_memory.fill_synthetic(192, seed=33)
agent.train_step(64, 1, idxs=np.arange(64, dtype=np.int32))
g = np.concatenate([agent.actor.get_grads(), agent.critic.get_grads()])
print("gradient digest", hashlib.sha256(g.tobytes()).hexdigest()[:16], float(np.abs(g).max()))
agent.close()
PY
done
