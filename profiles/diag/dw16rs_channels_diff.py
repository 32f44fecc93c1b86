"""conv_dw16_rs.h at a channel count other than 18 against conv_dw16.h (CPP_CONV1_DWRS_CH=0): where do conv1's weight gradients differ?"""
import os, subprocess, sys, numpy as np, tempfile
sys.path.insert(0, os.getcwd())
from tests.test_gpu_conv1_rs16_channels import _SNIPPET
def run(shape, B, extra):
    out = tempfile.mktemp(suffix=".npz")
    r = subprocess.run([sys.executable, "-c", _SNIPPET, repr(shape), str(B), out], env=dict(os.environ, CARTPOLEPP_ABLATION="1", **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return dict(np.load(out))
for shape in ((64, 64, 3, 1, 2), (64, 64, 3, 1, 3)):
    B, C = 5, int(np.prod(shape[2:]))
    n, o = run(shape, B, {})["grads"], run(shape, B, {"CPP_CONV1_DWRS_CH": "0"})["grads"]
    nw = 25 * C * 10
    wa, wb = n[:nw].reshape(5, 5, C, 10).astype(np.float64), o[:nw].reshape(5, 5, C, 10).astype(np.float64)
    d = np.abs(wa - wb)
    print("C", C, "max", d.max(), "of", np.abs(wb).max(), "bias", np.abs(n[nw:nw + 10] - o[nw:nw + 10]).max())
    print("  by ky", d.max(axis=(1, 2, 3)).round(5), "\n  by kx", d.max(axis=(0, 2, 3)).round(5), "\n  by c", d.max(axis=(0, 1, 3)).round(5), "\n  by o", d.max(axis=(0, 1, 2)).round(5))
    r = wa / np.where(np.abs(wb) > 1e-9, wb, 1)
    print("  ratio quantiles", np.quantile(r, [0.05, 0.5, 0.95]))
