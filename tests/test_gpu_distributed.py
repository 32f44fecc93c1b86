"""The data-parallel learner behind the C ABI (cpp_comm_*, cpp_ddpg_dp_train_step, cpp_naf_dp_train_step) on ONE GPU: a
communicator of world size 1 runs the real RCCL calls, and every mode must walk the same minibatches to the same parameters
as the fused single-learner step.  (World sizes > 1 cannot run on the 1-GPU test box; the protocol itself is covered at world
size 2 on CPU by tests/test_distributed_gloo.py.)"""
import numpy as np
import pytest

from tests.helpers import make_pair

pytestmark = pytest.mark.gpu


def _params(agent):
    return [n.get_params() for n in agent.networks()]


def _close_params(a, b, tol=1e-6):
    # (the fused step's dW reductions may sum in 4 slices instead of 16: last-bit differences only)
    for x, y in zip(a, b):
        assert np.abs(x - y).max() < tol, float(np.abs(x - y).max())


@pytest.mark.parametrize("mode,shape,B", [("gradient-allreduce", (16, 16, 3, 2, 3), 16), ("overlap", (16, 16, 3, 2, 3), 16),
                                          ("periodic-3", (16, 16, 3, 2, 3), 16), ("no-communicator", (16, 16, 3, 2, 3), 16),
                                          ("gradient-allreduce", (64, 64, 3, 2, 3), 256), ("overlap", (64, 64, 3, 2, 3), 256)],
                         ids=["gradient-allreduce", "overlap", "periodic-3", "no-communicator", "gradient-allreduce-cfg3-B256", "overlap-cfg3-B256"])
def test_native_learner_at_world_size_one_equals_the_fused_step(mode, shape, B):
    from cartpoleplusplus_amd import ddpg_cartpole as D
    from cartpoleplusplus_amd.distributed import Communicator, NativeLearner
    res = []
    for which in ("fused", "dp"):
        agent, _ref, _ = make_pair(shape, B, True, replay_size=max(300, 4 * B))
        try:
            agent.replay_memory.fill_synthetic(max(200, 3 * B), seed=11)
            if which == "fused":
                for _ in range(5):
                    agent.train_step(B, 3)
            else:
                comm = None if mode == "no-communicator" else Communicator.single(agent.trainer.ctx)
                if comm is not None:
                    assert (comm.rank, comm.world) == (0, 1) and comm.max_over_ranks(3.5) == 3.5
                    comm.barrier()
                learner = NativeLearner(agent, B, int(D.opts.sample_seed), comm, sync_every=3 if mode == "periodic-3" else 1,
                                        overlap=mode == "overlap")
                for _ in range(5):
                    learner.train_step(3)
                assert "dp1" in learner.describe()
                learner.close()
            agent.actor.ctx.sync()
            res.append(_params(agent))
        finally:
            agent.close()
    _close_params(res[0], res[1])


def test_half_step_graph_variants_survive_batch_size_changes_and_replay_growth():
    """the half step keeps three cached graph variants (own sample / presampled into either slot set) keyed on (batch size, seed,
    replay): walk them through a batch-size change, episodes added between steps (the sampler's range is a device word: no
    recapture) and a second replay memory, against the fused step doing the same."""
    import ctypes
    from cartpoleplusplus_amd import _lib, ddpg_cartpole as D
    from cartpoleplusplus_amd.distributed import NativeLearner
    shape = (16, 16, 3, 2, 3)
    res = []
    for which in ("fused", "dp-overlap-graphs", "dp"):
        agent, _ref, _ = make_pair(shape, 32, True, replay_size=600)
        try:
            rm = agent.replay_memory
            rm.fill_synthetic(150, seed=3)
            seed = int(D.opts.sample_seed)
            learners = {}

            def step(B, n):
                if which == "fused":
                    agent.train_step(B, n)
                else:
                    if B not in learners:
                        from cartpoleplusplus_amd.distributed import Communicator
                        comm = Communicator.single(agent.trainer.ctx) if which == "dp-overlap-graphs" else None
                        learners[B] = NativeLearner(agent, B, seed, comm, overlap=comm is not None)
                    learners[B].train_step(n)
            rows_seen = []
            for B, n, grow in ((32, 3, 0), (32, 2, 0), (16, 3, 0), (32, 4, 400), (32, 3, 0), (32, 2, 500), (16, 2, 600), (16, 5, 0)):
                if grow:
                    rm.fill_synthetic(grow, seed=3)
                step(B, n)
                idxs = np.empty(B, np.int32)
                _lib.check(_lib.lib.cpp_replay_last_indexes(rm.handle, B, idxs.ctypes.data_as(ctypes.c_void_p)))
                rows_seen.append(idxs.copy())
            agent.actor.ctx.sync()
            res.append((_params(agent), rows_seen))
            for l in learners.values():
                l.close()
        finally:
            agent.close()
    # (the rows of the last draw are not comparable: a half step has already presampled the NEXT minibatch behind conv1's dW;
    # identical parameters after 24 minibatches are only possible if every minibatch drew the same rows -- including the first
    # one after the memory grew at an unchanged batch size (32, 2, 500): the minibatch presampled before the write is dropped)
    for k in (1, 2):
        _close_params(res[0][0], res[k][0], tol=5e-5)         # 24 minibatches of last-bit differences (dW reduction slices)
    assert max(r.max() for r in res[0][1][3:5]) >= 150    # the grown memory is sampled without a recapture


@pytest.mark.parametrize("share", [True, False], ids=["shared-trunk", "own-trunks"])
@pytest.mark.parametrize("mode", ["gradient-allreduce", "periodic-2"])
def test_naf_native_learner_at_world_size_one_equals_the_fused_step(share, mode):
    import json
    from cartpoleplusplus_amd import naf_cartpole as F
    from cartpoleplusplus_amd.distributed import Communicator, NativeLearner
    from tests.helpers import FakeEnv
    shape, B = (16, 16, 3, 1, 2), 8
    res = []
    for which in ("fused", "dp"):
        F.set_opts(F.default_opts(use_raw_pixels=True, render_height=16, render_width=16, num_cameras=1, action_repeats=2,
                                  batch_size=B, replay_memory_size=200, share_input_state_representation=share,
                                  optimiser="Adam", optimiser_args=json.dumps({"learning_rate": 0.001})))
        agent = F.NormalizedAdvantageFunctionAgent(FakeEnv(shape))
        try:
            agent.initialise_variables(seed=4)
            agent.post_var_init_setup()
            agent.replay_memory.fill_synthetic(150, seed=9)
            if which == "fused":
                for _ in range(4):
                    agent.train_step(B, 3)
            else:
                learner = NativeLearner(agent, B, int(F.opts.sample_seed), Communicator.single(agent.naf.ctx),
                                        sync_every=2 if mode == "periodic-2" else 1)
                for _ in range(4):
                    learner.train_step(3)
                learner.close()
            agent.naf.ctx.sync()
            res.append([n.get_params() for n in agent.networks()])
            assert agent.naf.last_stats()[2] == 0
        finally:
            agent.close()
    _close_params(res[0], res[1])


@pytest.mark.parametrize("torch_first", [True, False], ids=["torch-imported-first", "library-alone"])
def test_communicator_in_a_fresh_process(torch_first):
    """torch ships its own HIP runtime and RCCL: the library's communicator must come up whether or not torch was imported
    (and initialised its CUDA state) before the library was loaded."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch; torch.cuda.init(); x = torch.ones(4, device='cuda');\n" if torch_first else "") + (
        "import numpy as np, ctypes\n"
        "from cartpoleplusplus_amd import _lib\n"
        "from cartpoleplusplus_amd.distributed import Communicator\n"
        "ctx = _lib.default_context(); c = Communicator.single(ctx)\n"
        "assert c.max_over_ranks(2.25) == 2.25; c.barrier(); c.close(); print('COMM_OK')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and "COMM_OK" in r.stdout.decode(), r.stdout.decode()[-1500:]


@pytest.mark.parametrize("which,extra", [("ddpg", []), ("ddpg", ["--sync-every", "3"]), ("ddpg", ["--overlap-allreduce"]), ("naf", [])],
                         ids=["ddpg", "ddpg-periodic", "ddpg-overlap", "naf"])
def test_cli_data_parallel_mode_as_a_world_of_one(which, extra, capsys):
    """`--data-parallel` of the agents' own main(): rollouts on the stand-in env, the replay shard in HBM, every inner step through the
    collective learner (a plain run is a world of one: real RCCL calls, no torch.distributed)."""
    import json
    if which == "ddpg":
        from cartpoleplusplus_amd import ddpg_cartpole as M
    else:
        from cartpoleplusplus_amd import naf_cartpole as M
    M.main(["--synthetic-env", "--use-raw-pixels", "--render-width", "16", "--render-height", "16", "--batch-size", "8",
            "--replay-memory-size", "120", "--replay-memory-burn-in", "20", "--max-episode-len", "12", "--max-num-actions", "70",
            "--data-parallel"] + extra)
    out = capsys.readouterr().out
    stats = [json.loads(l.split("\t", 1)[1]) for l in out.splitlines() if l.startswith("STATS")]
    assert len(stats) >= 4 and any(np.isfinite(s["mean_losses"]) for s in stats), out[-500:]
    assert stats[-1]["replay_memory_stats"][">batch"] > 0


def test_the_fused_cfg3_step_is_bit_reproducible_from_run_to_run():
    """No kernel of the step may depend on timing (no float atomics, no reads that race a write): the same 30 minibatches from the same
    state give the same parameter BITS, three times.  (Round 4: conv_fwd_k16.hip built under another LLVM scheduling strategy broke its
    hand-counted vmcnt waits: the fused-vs-data-parallel comparison above differed in 4 of 10 runs, this test caught it in 1 of 6 at half
    the length -- profiles/experiments/r04_sched_strategy.txt.)"""
    runs = []
    for _ in range(3):
        agent, _ref, _ = make_pair((64, 64, 3, 2, 3), 256, True, replay_size=1024)
        try:
            agent.replay_memory.fill_synthetic(768, seed=11)
            for _ in range(10):
                agent.train_step(256, 3)
            agent.actor.ctx.sync()
            runs.append(_params(agent))
        finally:
            agent.close()
    for other in runs[1:]:
        for x, y in zip(runs[0], other):
            assert np.array_equal(x, y), float(np.abs(x - y).max())


@pytest.mark.parametrize("shape,B,kw", [((64, 64, 3, 1, 3), 256, {}), ((128, 128, 3, 2, 5), 512, {}), ((64, 64, 3, 2, 3), 64, {}),
                                        ((64, 64, 3, 2, 3), 128, {"use_batch_norm": True}), ((64, 64, 3, 2, 3), 128, {"replay_store": "u8"}),
                                        ((50, 50, 3, 2, 3), 128, {})],
                         ids=["cfg2", "cfg5", "cfg3-B64-banded", "batch-norm", "u8-store", "50x50x18"])
def test_the_other_configurations_steps_are_bit_reproducible_too(shape, B, kw):
    """the same property at cfg2 (the 9-channel instances), cfg5 (the 30-channel ring forward, conv_dw16.h's re-divided dW, the 64-wide
    row-streaming conv2 backward: round 6 met a build of that dX instance -- its per-row bound made live -- whose output differed from run
    to run in the odd channels by 1e-3 of conv1's gradient, which only the f64-oracle test at B = 512 noticed; profiles/NOTEBOOK_r06.md 10)
    and a small batch (conv1 forward walked as two bands of rows per image); the batch-norm networks (bn.hip, the f32-input kernels), the
    8-bit replay store and the reference's default 50 x 50 render (the ring kernel, conv_dw16.h at 25-pixel conv2 rows) as well."""
    runs = []
    for _ in range(3):
        agent, _ref, _ = make_pair(shape, B, True, replay_size=2 * B + 64, **kw)
        try:
            agent.replay_memory.fill_synthetic(2 * B, seed=11)
            for _ in range(6):
                agent.train_step(B, 3)
            agent.actor.ctx.sync()
            runs.append(_params(agent))
        finally:
            agent.close()
    for other in runs[1:]:
        for x, y in zip(runs[0], other):
            assert np.array_equal(x, y), float(np.abs(x - y).max())
