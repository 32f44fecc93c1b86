import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k, name in enumerate(("actor", "critic")):
    wa, wb = a[k][:4500].reshape(5, 5, 18, 10), b[k][:4500].reshape(5, 5, 18, 10)
    d = np.abs(wa.astype(np.float64) - wb)
    print(name, "w max abs diff", d.max(), "rel l2", np.linalg.norm(wa.astype(np.float64) - wb) / np.linalg.norm(wb), "bias", a[k][4500:], b[k][4500:])
    print("  by ky", d.max(axis=(1, 2, 3)), "\n  by kx", d.max(axis=(0, 2, 3)), "\n  by c", d.max(axis=(0, 1, 3)), "\n  by o", d.max(axis=(0, 1, 2)))
    r = wa / np.where(np.abs(wb) > 1e-12, wb, 1)
    print("  ratio quantiles", np.quantile(r, [0.05, 0.5, 0.95]))
