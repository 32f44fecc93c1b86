import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    from helpers import make_pair, O
    shape, Bs = (128, 128, 3, 2, 5), 2
    agent, ref, (aspec, cspec) = make_pair(shape, Bs, True)
    rng = np.random.default_rng(6)
    t = O.synthetic_batch(rng, Bs, shape, 2, True)
    class HB(object): pass
    hb = HB(); hb.state_1, hb.action, hb.reward, hb.terminal_mask, hb.state_2 = t
    print("state dtype", t[0].dtype)
    agent.critic.train(hb)
    g = agent.critic.get_grads()
    p1 = agent.critic.pool1.eval(Bs)
    from cartpoleplusplus_amd._lib import lib, check, ptr
    codes = np.empty(p1.shape, np.float32)
    check(lib.cpp_net_get_pool(agent.critic.handle, 11, Bs, ptr(codes)))
    np.savez(sys.argv[1], g=g, p1=p1, codes=codes)
else:
    for k in ("0", "1"):
        env = dict(os.environ, CARTPOLEPP_ABLATION="1", CPP_CONV_KYO=k)
        subprocess.check_call([sys.executable, __file__, "/tmp/kyo%s.npz" % k], env=env, stderr=subprocess.DEVNULL)
    a, b = np.load("/tmp/kyo0.npz"), np.load("/tmp/kyo1.npz")
    print("pool1 maxabs diff", np.abs(a["p1"] - b["p1"]).max())
    cd = a["codes"] != b["codes"]
    print("codes differ", cd.sum(), "of", cd.size, "by o", cd.sum(axis=(0, 1, 2)), "where pool>0:", (cd & (a["p1"] > 0)).sum())
    idx = np.argwhere(cd & (a["p1"] > 0))
    print(idx[:30].tolist())
    print("by px", cd.sum(axis=(0, 1, 3)))
    print("by py", cd.sum(axis=(0, 2, 3)))
    for (b_, y_, x_, o_) in idx[:8]:
        print((b_, y_, x_, o_), "old", a["codes"][b_, y_, x_, o_], "new", b["codes"][b_, y_, x_, o_])
    d = np.abs(a["g"] - b["g"])
    print("grad diff max", d.max(), "argmax", d.argmax(), "n>1e-4", (d > 1e-4).sum(), "of", d.size)
    nz = np.nonzero(d > 1e-4)[0]
    print(nz[:40])
    w = d[:5*5*30*10].reshape(5, 5, 30, 10)
    print("by ky", w.max(axis=(1,2,3)), "by kx", w.max(axis=(0,2,3)), "by o", w.max(axis=(0,1,2)))
    print("by c", w.max(axis=(0,1,3)))
