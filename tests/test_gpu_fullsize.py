"""GPU tests at BASELINE.json's full sizes (B = 256, 64x64x18; replay in HBM), through size-independent
properties, plus the geometry paths the small cases do not reach (128-wide images: partial-width tiles,
30 input channels)."""
import numpy as np
import pytest

from oracle import ddpg_np as O
from tests.helpers import make_pair, assert_flat_close, assert_grads_close_modulo_pool_ties

pytestmark = pytest.mark.gpu

SHAPE, B = (64, 64, 3, 2, 3), 256


def test_full_size_gather_is_exact_and_statistics_match_f64():
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    rm = ReplayMemory(3000, SHAPE, 2)
    rm.fill_synthetic(2500, seed=11)
    b = rm.sample_on_device(B, seed=99, counter=7)
    assert b.idxs.min() >= 0 and b.idxs.max() < 2500
    # property: every gathered row is bit-for-bit the stored row of its index (double indirection)
    rows = np.random.default_rng(0).choice(B, 24, replace=False)
    s1 = rm.state[rm.state_1_idx[b.idxs[rows]]]
    s2 = rm.state[rm.state_2_idx[b.idxs[rows]]]
    assert np.array_equal(b.state_1[rows], s1) and np.array_equal(b.state_2[rows], s2)
    assert np.array_equal(b.terminal_mask[:, 0], rm.terminal_mask[b.idxs, 0])
    # property: pixel values are the 256 f16(k/255) levels
    lv = (np.arange(256).astype(np.float16) / np.float16(255))
    assert np.isin(b.state_1[:8].ravel(), lv).all()
    rm.close()


def test_full_size_whitening_statistics_and_forward_against_f64_oracle_slice():
    """whitening uses the statistics of the whole 256-row batch; check them (and a forward through all three
    convs) by recomputing in f64 on the host for the same batch."""
    agent, ref, _ = make_pair(SHAPE, B, True, replay_size=600)
    try:
        agent.replay_memory.fill_synthetic(500, seed=5)
        b = agent.replay_memory.sample_on_device(B, seed=3, counter=0)
        s1 = b.state_1
        x = s1.reshape(B, 64, 64, 18)
        scale, shift = O.whiten_stats(x, np.float64)
        # conv1 of the first 4 images with the FULL batch statistics
        sub = ref.actor.forward(s1[:4], white=(scale, shift))
        got = agent.actor.forward(s1)            # device: batch statistics of all 256 rows
        assert np.abs(got[:4] - sub["out"]).max() < 1e-5
        pool1 = agent.actor.pool1.eval(B)
        assert np.abs(pool1[:4] - sub["conv1"][1]).max() < 1e-5
    finally:
        agent.close()


def test_full_size_fused_step_equals_unfused_ops_and_is_deterministic():
    """the hipGraph / batched-GEMM / fused-heads / deferred-reduction path and the op-by-op path are different launch
    sequences (and, for the MLP heads, different summation orders): they must agree to rounding; two fused runs must
    agree bit for bit.  After ONE minibatch the bar is 1e-5; the second minibatch starts from parameters that differ in
    the last bits, which moves a few max-pool routes of the 2.6 M pooling windows -- a looser bar there."""
    results = {}
    for mode, nb in (("fused", 2), ("fused_again", 2), ("unfused", 2), ("fused", 1), ("unfused", 1)):
        agent, ref, (aspec, cspec) = make_pair(SHAPE, B, True, replay_size=600)
        try:
            agent.replay_memory.fill_synthetic(500, seed=5)
            idxs = np.random.default_rng(1).integers(0, 500, 2 * B)
            if mode.startswith("fused"):
                agent.train_step(B, nb, idxs=idxs[:nb * B])
            else:
                for i in range(nb):
                    batch = agent.replay_memory.batch(idxs=idxs[i * B:(i + 1) * B])
                    agent.actor.train(batch)
                    agent.critic.train(batch)
                agent.target_actor.update_weights(); agent.target_critic.update_weights()
            results[(mode, nb)] = (agent.actor.get_params(), agent.critic.get_params(),
                                   agent.target_actor.get_params(), agent.target_critic.get_params())
        finally:
            agent.close()
    for k in range(4):
        assert np.array_equal(results[("fused", 2)][k], results[("fused_again", 2)][k])
    assert_flat_close(aspec, results[("fused", 1)][0], results[("unfused", 1)][0], rel=1e-5, what="actor fused vs unfused, 1 minibatch")
    assert_flat_close(cspec, results[("fused", 1)][1], results[("unfused", 1)][1], rel=1e-5, what="critic fused vs unfused, 1 minibatch")
    assert_flat_close(aspec, results[("fused", 2)][0], results[("unfused", 2)][0], rel=2e-4, what="actor fused vs unfused")
    assert_flat_close(cspec, results[("fused", 2)][1], results[("unfused", 2)][1], rel=2e-4, what="critic fused vs unfused")
    assert np.isfinite(results[("fused", 2)][0]).all() and np.isfinite(results[("fused", 2)][1]).all()


def test_full_size_train_ops_against_f64_oracle():
    """one full-size minibatch through the op-by-op train ops (actor.train, critic.train: ddpg_cartpole.py:140-145, :230-237)
    against the float64 oracle at north_star's bar: Q / TD / actions / dQ/da within 1e-5, both pre-clip gradient lists per
    variable at 2e-5 (the pool routes and ReLU decisions are the device's; one that differs from the oracle's own must be a
    rounding-level tie: with 2.6 M pooling windows per network a minibatch holds O(1) of them)."""
    from tests.helpers import device_pool_codes, pool_flips_are_near_ties, device_relu_active, relu_flips_are_at_the_boundary
    agent, _ref, (aspec, cspec) = make_pair(SHAPE, B, True, replay_size=600)
    try:
        agent.replay_memory.fill_synthetic(500, seed=8)
        batch = agent.replay_memory.batch(idxs=np.random.default_rng(4).integers(0, 500, B))
        t = (batch.state_1, batch.action, batch.reward, batch.terminal_mask, batch.state_2)
        P = [n.get_params() for n in agent.networks()]
        loss, td, q = agent.critic.check_loss(batch)
        agent.actor.train(batch)
        actions, dq_da, _q, _td = agent.trainer.last_values(B)
        g_a, codes_a, relu_a = agent.actor.get_grads(), device_pool_codes(agent.actor, B), device_relu_active(agent.actor, B)
        agent.critic.train(batch)
        g_c, codes_c, relu_c = agent.critic.get_grads(), device_pool_codes(agent.critic, B), device_relu_active(agent.critic, B)
    finally:
        agent.close()
    ref = O.DDPG(aspec, cspec, P[0], P[1], np.float64)
    ref.set_targets(P[2], P[3])
    ref.actor.amax_override, ref.critic.amax_override = codes_a, codes_c
    ref.actor.relu_override, ref.critic.relu_override = relu_a, relu_c
    ag, cg = ref.actor_gradients(t[0]), ref.critic_gradients(t)
    pool_flips_are_near_ties(ag["cache_actor"], codes_a, what="actor")
    pool_flips_are_near_ties(cg["cache_critic"], codes_c, what="critic")
    relu_flips_are_at_the_boundary(ag["cache_actor"], relu_a, what="actor")
    relu_flips_are_at_the_boundary(cg["cache_critic"], relu_c, what="critic")
    assert np.abs(q - cg["q"]).max() < 1e-5 and np.abs(td - cg["td"]).max() < 1e-5
    assert abs(loss - cg["loss"]) < 1e-5 * max(1.0, abs(cg["loss"]))
    assert np.abs(actions - ag["actions"]).max() < 1e-5 and np.abs(dq_da - ag["dq_da"]).max() < 1e-5
    assert_flat_close(aspec, g_a, ag["grads"], rel=2e-5, what="actor grads (f64 oracle)")
    assert_flat_close(cspec, g_c, cg["grads"], rel=2e-5, what="critic grads (f64 oracle)")


@pytest.mark.parametrize("shape,Bs", [((128, 128, 3, 2, 5), 2), ((64, 64, 3, 1, 3), 3)],
                         ids=["128x128x30-cfg5-shape", "64x64x9-cfg2-shape"])
def test_wide_image_geometry_parity(shape, Bs):
    """cfg5 image shape: two 64-column tiles per row (partial-width tiles with real halo columns), 30 input
    channels (one workgroup per CU), conv2 on 64x64, conv3 on 32x32."""
    agent, ref, (aspec, cspec) = make_pair(shape, Bs, True)
    rng = np.random.default_rng(6)
    t = O.synthetic_batch(rng, Bs, shape, 2, True)
    try:
        class HB(object):
            pass
        hb = HB(); hb.state_1, hb.action, hb.reward, hb.terminal_mask, hb.state_2 = t
        ag, cg = ref.actor_gradients(t[0]), ref.critic_gradients(t)
        assert np.abs(agent.actor.forward(t[0]) - ag["actions"]).max() < 1e-5
        loss, td, q = agent.critic.check_loss(hb)
        assert np.abs(q - cg["q"]).max() < 1e-5
        pa = agent.actor.get_params()
        agent.actor.train(hb)
        # B is tiny here, so a single rounding-level tie-break in a 2x2 pooling window shows in the conv1 gradient
        assert_grads_close_modulo_pool_ties(
            aspec, agent.actor, Bs, ref.actor, lambda: ref.actor.forward(t[0]),
            lambda: ref.actor_gradients(t[0])["grads"], agent.actor.get_grads(), what="actor grads")
        agent.actor.set_params(pa)
        agent.critic.train(hb)
        assert_grads_close_modulo_pool_ties(
            cspec, agent.critic, Bs, ref.critic, lambda: ref.critic.forward(t[0], action=np.asarray(t[1])),
            lambda: ref.critic_gradients(t)["grads"], agent.critic.get_grads(), what="critic grads")
    finally:
        agent.close()


def test_earlier_conv_kernels_stay_parity_green_when_selected():
    """CPP_CONV_KYO=0 routes every conv layer through the (ky,(kx,c)) x o kernels (normally only the fallback for
    shapes whose rows cannot be staged as 16-byte chunks): the cfg3 / cfg2-shape parity cases must pass on them too."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CARTPOLEPP_ABLATION="1", CPP_CONV_KYO="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "64x64 and (forward or gradients or fused)"], cwd=root, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0 and " passed" in tail, tail


_CONV1_ERR_SNIPPET = r"""
import numpy as np
from oracle import ddpg_np as O
from tests.helpers import make_pair
SHAPE, B = (64, 64, 3, 2, 3), 256
agent, ref, _ = make_pair(SHAPE, B, True, replay_size=600)
agent.replay_memory.fill_synthetic(500, seed=5)
b = agent.replay_memory.sample_on_device(B, seed=3, counter=0)
s1 = b.state_1
scale, shift = O.whiten_stats(s1.reshape(B, 64, 64, 18), np.float64)
sub = ref.actor.forward(s1[:6], white=(scale, shift))          # float64 ground truth
agent.actor.forward(s1)
pool1 = agent.actor.pool1.eval(B)[:6]
want = sub["conv1"][1]
print("CONV1ERR %.3e %.3e" % (np.abs(pool1 - want).max(), np.abs(want).max()))
agent.close()
"""


_F16_BUILDS = (("two", {}), ("three", {"TEST_EXACT_PRODUCTS": "1"}),      # (release library: cpp_ctx_set_precision fast / exact)
               ("f32", {"CARTPOLEPP_ABLATION": "1", "CPP_CONV_K16": "0"}))


def test_f16_piece_conv1_is_as_close_to_the_f64_oracle_as_the_f32_mfma_kernel():
    """conv_k16.h claims f32-grade results (f16 x f16 products, exact; f32 accumulation) from two f16 pieces of each weight (the
    shipped kernels: the weight to within one f32 ulp) and from three (--exact-products / cpp_ctx_set_precision: the weight itself): the pooled conv1
    output must sit as close to the float64 oracle as the f32-MFMA kernel's (CPP_CONV_K16=0), far inside 1e-5."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    errs = {}
    for name, env in _F16_BUILDS:
        r = subprocess.run([sys.executable, "-c", _CONV1_ERR_SNIPPET], cwd=root, env=dict(os.environ, **env),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        m = re.search(r"CONV1ERR (\S+) (\S+)", r.stdout.decode())
        assert r.returncode == 0 and m, r.stdout.decode()[-1500:]
        errs[name] = (float(m.group(1)), float(m.group(2)))
    (e32, mag) = errs["f32"]
    assert mag > 0.5                                   # outputs of order one and more
    assert e32 < 1e-5, errs                            # (measured: two 3.2e-6, three 3.7e-6, f32 7.0e-6 at |z| up to 7.1)
    for name in ("two", "three"):
        assert errs[name][0] < 1e-5 and errs[name][0] <= 1.25 * e32 + 1e-7, (name, errs)
    assert errs["two"][0] <= 1.25 * errs["three"][0] + 1e-7, errs


def test_f32_mfma_conv1_stays_parity_green_when_selected():
    """CPP_CONV_K16=0 keeps conv1 forward on the f32-input MFMA kernel (the path of f32 states, odd channel counts and
    per-image whitening): the full-size parity cases must pass on it too."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "64x64 and (forward or gradients or fused)"], cwd=root, env=dict(os.environ, CARTPOLEPP_ABLATION="1", CPP_CONV_K16="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0 and " passed" in tail, tail


@pytest.mark.parametrize("switch", ["CPP_FUSED_HEADS", "CPP_HEADS_PRE"])
def test_gemm_level_heads_stay_parity_green_when_selected(switch):
    """CPP_FUSED_HEADS=0 runs the MLP heads as GEMM levels + TD kernel, CPP_HEADS_PRE=0 keeps the actors' last hidden layer
    out of the heads kernel (the paths of networks the kernel does not cover: dropout, wide layers): same parity cases."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", "fused or gradients or train_ops"], cwd=root, env=dict(os.environ, CARTPOLEPP_ABLATION="1", **{switch: "0"}),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0 and " passed" in tail, tail


_CONV1_DW_ERR_SNIPPET = r"""
import numpy as np
from oracle import ddpg_np as O
from tests.helpers import make_pair, device_pool_codes, per_var_report
SHAPE, B = (64, 64, 3, 2, 3), 16
agent, ref, (aspec, cspec) = make_pair(SHAPE, B, True)
class HB(object): pass
hb = HB()
t = O.synthetic_batch(np.random.default_rng(17), B, SHAPE, 2, True)       # f16 pixel states
hb.state_1, hb.action, hb.reward, hb.terminal_mask, hb.state_2 = t
agent.critic.train(hb)
got = agent.critic.get_grads()
ref.critic.amax_override = device_pool_codes(agent.critic, B)             # same pooling routes: rounding is all that is left
want = ref.critic_gradients(t)["grads"]                                    # float64
for name, amax, rel in per_var_report(cspec, got, want):
    if name.endswith("conv1/weights") or name.endswith("conv1/biases"):
        print("DWERR %s %.3e" % (name.split("/")[-1], rel))
agent.close()
"""


def test_f16_piece_conv1_dw_is_as_close_to_the_f64_oracle_as_the_f32_mfma_kernel():
    """conv_dw16.h: conv1's weight gradient from f16 x f16 products of the raw pixels and two (shipped) / three (--exact-products) f16
    pieces of dY must match the float64 oracle (with the device's pooling routes) at least as well as the f32-MFMA kernel does."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    errs = {}
    for name, env in _F16_BUILDS:
        r = subprocess.run([sys.executable, "-c", _CONV1_DW_ERR_SNIPPET], cwd=root, env=dict(os.environ, **env),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        m = dict(re.findall(r"DWERR (\S+) (\S+)", r.stdout.decode()))
        assert r.returncode == 0 and "weights" in m, r.stdout.decode()[-1500:]
        errs[name] = float(m["weights"])
    assert max(errs.values()) < 5e-6, errs             # (measured: two 1.22e-6, three 1.19e-6, f32 1.69e-6)
    assert errs["two"] <= 1.25 * errs["f32"] + 1e-8 and errs["three"] <= 1.25 * errs["f32"] + 1e-8, errs
    assert errs["two"] <= 1.25 * errs["three"] + 1e-8, errs


_CONV2_ERR_SNIPPET = r"""
import numpy as np
from oracle import ddpg_np as O
from tests.helpers import make_pair, device_pool_codes, per_var_report
SHAPE, B = (64, 64, 3, 2, 3), 16
agent, ref, (aspec, cspec) = make_pair(SHAPE, B, True)
class HB(object): pass
hb = HB()
t = O.synthetic_batch(np.random.default_rng(23), B, SHAPE, 2, True)       # f16 pixel states
hb.state_1, hb.action, hb.reward, hb.terminal_mask, hb.state_2 = t
agent.critic.train(hb)
got = agent.critic.get_grads()
pool2 = agent.critic.pool2.eval(B)
ref.critic.amax_override = device_pool_codes(agent.critic, B)             # same pooling routes: rounding is all that is left
cg = ref.critic_gradients(t)                                               # float64
fw = ref.critic.forward(t[0], action=t[1])
print("C2FWD %.3e %.3e" % (np.abs(pool2 - fw["conv2"][1]).max(), np.abs(fw["conv2"][1]).max()))
for name, amax, rel in per_var_report(cspec, got, cg["grads"]):
    if name.endswith("conv2/weights"):
        print("C2DW %.3e" % rel)
agent.close()
"""


def test_bf16_conv2_is_as_close_to_the_f64_oracle_as_the_f32_mfma_kernels():
    """conv2 forward / dW from three bf16 pieces per operand (conv_k16.h B16 mode, conv_dwb16.h) -- the six largest piece products
    (the shipped kernels; the dropped three are at most 2^-24 of the product), all nine (CPP_B16_PRODUCTS=9, ablation build) -- against CPP_CONV_B16=0 (f32-input
    MFMA): pooled conv2 output and conv2 weight gradient vs the float64 oracle."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, extra in (("six", {"CARTPOLEPP_ABLATION": "1"}), ("nine", {"CARTPOLEPP_ABLATION": "1", "CPP_B16_PRODUCTS": "9"}),
                        ("f32", {"CARTPOLEPP_ABLATION": "1", "CPP_CONV_B16": "0"}), ("exact", {"TEST_EXACT_PRODUCTS": "1"})):
        r = subprocess.run([sys.executable, "-c", _CONV2_ERR_SNIPPET], cwd=root, env=dict(os.environ, **extra),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        out = r.stdout.decode()
        f, d = re.search(r"C2FWD (\S+) (\S+)", out), re.search(r"C2DW (\S+)", out)
        assert r.returncode == 0 and f and d, out[-1500:]
        res[name] = (float(f.group(1)), float(f.group(2)), float(d.group(1)))
    (f32e, mag, d32e) = res["f32"]
    assert f32e < 1e-5 * max(1.0, mag) and d32e < 5e-6, res
    for name in ("six", "nine", "exact"):      # (exact: nine products behind a three-piece conv1)
        fe, _, de = res[name]
        assert fe < 1e-5 * max(1.0, mag) and fe <= 1.5 * f32e + 1e-7 * max(1.0, mag), (name, res)
        assert de < 5e-6 and de <= 1.5 * d32e + 1e-8, (name, res)
    # dropping the three smallest products costs nothing measurable against the oracle
    assert res["six"][0] <= 1.1 * res["nine"][0] + 1e-8 * max(1.0, mag) and res["six"][2] <= 1.1 * res["nine"][2] + 1e-9, res


_PAIR_DW_SNIPPET = r"""
import hashlib, os, sys
import numpy as np
from tests.helpers import make_pair
for SHAPE, B in (((64, 64, 3, 2, 3), 256), ((64, 64, 3, 1, 3), 64), ((32, 32, 3, 2, 3), 32), ((50, 50, 3, 2, 3), 16)):
    agent, _ref, _ = make_pair(SHAPE, B, True, replay_size=400, seed=2)
    agent.replay_memory.fill_synthetic(300, seed=5)
    idx = np.random.default_rng(1).integers(0, 300, 2 * B)
    agent.train_step(B, 2, idxs=idx)
    flat = np.concatenate([net.get_params() for net in agent.networks()] + [agent.actor.get_grads(), agent.critic.get_grads()])
    tag = "x".join(map(str, SHAPE))
    np.save(os.path.join(sys.argv[1], "pairdw_%s.npy" % tag), flat)
    print("PAIRDW %s %s" % (tag, hashlib.sha256(flat.tobytes()).hexdigest()))
    agent.close()
"""


def _run_pair_snippet(tmp_path, env_extra):
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, extra in env_extra.items():
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([sys.executable, "-c", _PAIR_DW_SNIPPET, str(d)], cwd=root, env=dict(os.environ, CARTPOLEPP_ABLATION="1", **extra),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        txt = r.stdout.decode()
        found = re.findall(r"PAIRDW (\S+) (\S+)", txt)
        assert r.returncode == 0 and len(found) == 4, txt[-1500:]
        out[tag] = {shape: (h, np.load(str(d / ("pairdw_%s.npy" % shape)))) for shape, h in found}
    return out


def test_two_network_conv1_dw_workgroups_agree_with_the_single_network_kernel(tmp_path):
    """conv_dw16.h with NNET = 2 (the actor's and the critic's conv1 dW from one staged image row; selected at cfg3's geometry) against
    the one-network-per-workgroup kernel (CPP_DW16_PAIR=0): the same exact products and the same MFMA order per accumulator; the two
    choose different row bands (one resident round each), so partial sums are grouped differently: equal to f32 rounding, and
    bit-identical wherever the pair kernel is not selected."""
    got = _run_pair_snippet(tmp_path, {"pair": {"CPP_DW16_PAIR": "1"}, "single": {"CPP_DW16_PAIR": "0"}})
    for shape in got["pair"]:
        (h1, x1), (h0, x0) = got["pair"][shape], got["single"][shape]
        if shape == "64x64x3x2x3":
            scale = np.abs(x0).max()
            assert np.abs(x1 - x0).max() <= 5e-6 * scale, (shape, float(np.abs(x1 - x0).max()), float(scale))
        else:
            assert h1 == h0, shape


