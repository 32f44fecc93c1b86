"""conv1 dW at 30 channels (cfg5's geometry): the (MT / 2) x 2 division of conv_dw16.h's accumulator tiles against the previous build's
MT x 1 one (lib/libcartpolepp_hip_prev.so) on the same minibatch: where do the gradients differ, and by how much?
usage: python profiles/diag/dw16_split_diff.py   (runs itself twice, CARTPOLEPP_ABLATION = '' / 'prev')"""
import os, subprocess, sys
import numpy as np
if len(sys.argv) > 1:
    from tests.helpers import make_pair
    B = int(os.environ.get("DIFF_B", "64"))
    agent, _ref, _ = make_pair((128, 128, 3, 2, 5), B, True, replay_size=B + 192)
    agent.replay_memory.fill_synthetic(B + 128, seed=33)
    agent.train_step(B, 1, idxs=np.arange(B, dtype=np.int32))
    np.savez(sys.argv[1], actor=agent.actor.get_grads(), critic=agent.critic.get_grads())
    agent.close()
    sys.exit(0)
out = {}
for v in ("", "prev"):
    f = "/tmp/dw16_split_%s.npz" % (v or "new")
    r = subprocess.run([sys.executable, __file__, f], env=dict(os.environ, CARTPOLEPP_ABLATION=v, PYTHONPATH="."), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    out[v] = dict(np.load(f))
for net in ("actor", "critic"):
    a, b = out[""][net], out["prev"][net]
    d = np.abs(a.astype(np.float64) - b)
    nz = np.nonzero(d)[0]
    print(net, "size", a.size, "differing", nz.size, "max abs diff", d.max(), "max |g|", np.abs(b).max(), "first / last differing index", (nz[:5], nz[-5:]) if nz.size else None)
    # conv1's weights are the first 5 * 5 * 30 * 10 = 7500 floats (+ 10 biases) of a network's flat list
    print("   inside conv1's block (first 7510):", int((nz < 7510).sum()), "outside:", int((nz >= 7510).sum()))
    if nz.size:
        rel = d[nz] / np.maximum(np.abs(b[nz]), 1e-30)
        print("   relative differences: max", rel.max(), "median", np.median(rel))
    if net == "actor":
        print("   conv1 biases new :", a[7500:7510])
        print("   conv1 biases prev:", b[7500:7510])
