import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
d = np.abs(a.astype(np.float64) - b)
print(sys.argv[1], "vs", sys.argv[2], "max abs diff", d.max(), "rel l2", np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
if d.max() > 0:
    idx = np.argwhere(d > 0.1 * d.max())
    print("count > 10% of max:", len(idx))
    import collections
    print("nets", collections.Counter(idx[:, 0]).most_common(3))
    print("images", collections.Counter(idx[:, 1]).most_common(6))
    print("rows", sorted(collections.Counter(idx[:, 2]).items()))
    print("cols", sorted(collections.Counter(idx[:, 3]).items()))
    print("chans", sorted(collections.Counter(idx[:, 4]).items()))
