"""cfg3's geometry at the reference's DEFAULT batch size (--batch-size 128, ddpg_cartpole.py:32) and smaller: training steps/s of the fused
step with conv1 forward's images walked as two bands of rows (default when the launch has at most one workgroup per CU) and as whole
images (CPP_CONV_BANDS=0, ablation build).  usage: python profiles/diag/small_batch_bands.py   (runs itself per setting)"""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    import numpy as np
    from cartpoleplusplus_amd import ddpg_cartpole as D
    B = int(sys.argv[1])
    class Env(object):
        class S(object):
            def __init__(self, s): self.shape = tuple(s)
        observation_space, action_space = S((64, 64, 3, 2, 3)), S((1, 2))
    D.set_opts(D.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2, action_repeats=3, batch_size=B, replay_memory_size=8000))
    agent = D.DeepDeterministicPolicyGradientAgent(Env())
    agent.initialise_variables(seed=42); agent.post_var_init_setup()
    agent.replay_memory.fill_synthetic(8000, seed=1234)
    for _ in range(100): agent.train_step(B, 5)
    agent.actor.ctx.sync()
    best = 0.0
    for _rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200): agent.train_step(B, 5)
        agent.actor.ctx.sync()
        best = max(best, 1000 / (time.perf_counter() - t0))
    print("RESULT B=%d %.1f minibatches/s" % (B, best))
    agent.close()
    sys.exit(0)
for B in (128, 64, 32):
    for bands in ("1", "0"):
        env = dict(os.environ, CARTPOLEPP_ABLATION="1", PYTHONPATH=".")
        if bands == "0": env["CPP_CONV_BANDS"] = "0"
        r = subprocess.run([sys.executable, __file__, str(B)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        out = [l for l in r.stdout.decode().splitlines() if l.startswith("RESULT")]
        print("bands=%s" % bands, out[-1] if out else r.stdout.decode()[-800:], flush=True)
