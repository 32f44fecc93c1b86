import os, subprocess, sys, numpy as np, tempfile
sys.path.insert(0, os.getcwd())
from tests.test_gpu_conv1_rs16_channels import _SNIPPET
def run(shape, B, extra):
    out = tempfile.mktemp(suffix=".npz")
    r = subprocess.run([sys.executable, "-c", _SNIPPET, repr(shape), str(B), out], env=dict(os.environ, CARTPOLEPP_ABLATION="1", **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return dict(np.load(out))
shape, B = (64, 64, 3, 1, 3), 5
n, o = run(shape, B, {}), run(shape, B, {"CPP_CONV_RS16_CH": "0"})
d = np.abs(n["actor_pool1"] - o["actor_pool1"])
print("by col", d.max(axis=(0, 1, 3)).round(3))
print("by row", d.max(axis=(0, 2, 3)).round(3))
print("by img", d.max(axis=(1, 2, 3)).round(3))
print("by filter", d.max(axis=(0, 1, 2)).round(3))
