"""Data-parallel actor-learners: one process per GPU, replicated weights, an own replay shard and an own
minibatch per learner, collectives on RCCL over xGMI.

The reference is single-process (its only nod to this is the TODO "switch back to async training with
multiple replicas", ddpg_cartpole.py:259 / naf_cartpole.py:294).  Two modes:

  sync_every = 1   per minibatch every learner leaves [actor grads | critic grads] in ONE flat f32 buffer, the buffer is
                   summed across ranks, and every rank applies clip + SGD to the mean -- identical inputs on every rank, so
                   the replicas stay bit-identical with no parameter broadcast.
  sync_every = k   ("periodic") k local minibatch updates, then the parameters (targets and optimiser slots included) are
                   averaged across ranks.
Whitening statistics and target soft updates are local in both.

The product path is `NativeLearner`: the whole step -- hipGraph-captured half steps, the ncclAllReduce calls, the optimiser
kernels -- runs behind the C ABI (cpp_ddpg_dp_train_step / cpp_naf_dp_train_step with a cpp_comm); torch.distributed is only
used to hand the 128-byte communicator id (and, once, rank 0's initial parameters) to the other ranks.  The ranks agree that
every one of them formed the communicator, or all raise (`make_learner`): there is no per-rank fallback.  The agents' outer
loops decide "train this iteration" / "stop" collectively through `LoopAgreement` (training_loop.py).
`DataParallelLearner` is the same protocol written out on the host with pluggable pieces (unit-tested with gloo on CPU);
`TorchCollectiveLearner` (the half-step entry points + a torch.distributed all-reduce of the library's gradient buffer) is a
DIAGNOSTIC learner for `bench.py --diag-backend gloo`, selected explicitly.

Learners are synchronous in their collective step.  What BASELINE configs[2] calls "async actor-learners" is `--async-rollouts`
(training_loop.py): every learner's environment runs on a rollout thread and never gates the collective step.
"""
import ctypes as C
import os


def _abi():
    """(lib, check) of the C ABI.  The world-size-2 CPU tests replace this function with a Python double of the few entry points
    this module calls (tests/test_distributed_gloo.py), so that everything ABOVE the ABI -- communicator id hand-over, learner
    selection, replica sync, the agents' loop agreement -- runs there as it does over RCCL."""
    from ._lib import lib, check
    return lib, check


# ---------------------------------------------------------------------------------------------------------------------
# the library's communicator
# ---------------------------------------------------------------------------------------------------------------------
class Communicator(object):
    """cpp_comm: this rank's end of an RCCL communicator on a Context's GPU."""
    ID_BYTES = 128

    def __init__(self, ctx, unique_id, rank, world):
        lib, check = _abi()
        assert len(unique_id) == self.ID_BYTES
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), self.ID_BYTES)
        check(lib.cpp_comm_create(ctx.handle, buf, int(rank), int(world), C.byref(h)))
        self.handle, self.ctx, self.rank, self.world = h, ctx, int(rank), int(world)

    @staticmethod
    def new_unique_id():
        lib, check = _abi()
        buf = C.create_string_buffer(Communicator.ID_BYTES)
        check(lib.cpp_comm_unique_id(buf, Communicator.ID_BYTES))
        return bytes(buf.raw)

    @classmethod
    def from_torch_distributed(cls, ctx, group=None):
        """rank / world from the initialised torch.distributed group (any backend: only one small object is broadcast)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        kw = {}
        if dist.get_backend(group) == "nccl":
            import torch
            kw["device"] = torch.device("cuda", ctx.device_id)
        dist.broadcast_object_list(box, src=0, group=group, **kw)
        return cls(ctx, box[0], rank, world)

    @classmethod
    def single(cls, ctx):
        return cls(ctx, cls.new_unique_id(), 0, 1)

    def barrier(self):
        lib, check = _abi()
        check(lib.cpp_comm_barrier(self.handle))

    def max_over_ranks(self, *values):
        """element-wise max over the ranks of up to 8 host doubles (one value: a float; several: a list)."""
        lib, check = _abi()
        v = (C.c_double * len(values))(*[float(x) for x in values])
        check(lib.cpp_comm_max_doubles(self.handle, v, len(values)))
        return v[0] if len(values) == 1 else list(v)

    def close(self):
        if self.handle:
            lib, _ = _abi()
            lib.cpp_comm_destroy(self.handle)
            self.handle = None


class LoopAgreement(object):
    """collective yes/no decisions of the agents' outer loop under --data-parallel (training_loop.py): `all(flag)` is True on
    every rank iff `flag` is True on every rank.  One cpp_comm_max_doubles per call; `calls` counts them (tests compare ranks)."""

    def __init__(self, comm):
        self.comm, self.calls = comm, 0

    def all(self, flag):
        self.calls += 1
        if self.comm is None or self.comm.world == 1:
            return bool(flag)
        return self.comm.max_over_ranks(0.0 if flag else 1.0) == 0.0

    def any(self, flag):
        self.calls += 1
        if self.comm is None or self.comm.world == 1:
            return bool(flag)
        return self.comm.max_over_ranks(1.0 if flag else 0.0) != 0.0


class NativeLearner(object):
    """the data-parallel inner step behind the C ABI (ddpg_cartpole.py:331-337 / naf_cartpole.py:367-373 for N learners)."""

    def __init__(self, agent, batch_size, seed, comm=None, sync_every=1, overlap=False):
        self._lib, self._check = _abi()
        self.agent, self.B, self.seed, self.comm = agent, int(batch_size), int(seed), comm
        self.sync_every, self.overlap = int(sync_every), bool(overlap)
        self.is_naf = hasattr(agent, "naf")
        self.world = comm.world if comm is not None else 1

    def train_step(self, batches_per_step):
        rm, ch = self.agent.replay_memory, self.comm.handle if self.comm is not None else None
        rm.stats[">batch"] += batches_per_step
        if self.is_naf:
            self._check(self._lib.cpp_naf_dp_train_step(self.agent.naf.handle, rm.handle, ch, self.B, int(batches_per_step),
                                                        self.seed, self.sync_every))
        else:
            self._check(self._lib.cpp_ddpg_dp_train_step(self.agent.trainer.handle, rm.handle, ch, self.B, int(batches_per_step),
                                                         self.seed, self.sync_every, 1 if self.overlap else 0))

    def describe(self):
        how = ("flat-gradient ncclAllReduce per minibatch%s" % (", fc gradients reduced beside the conv backward" if self.overlap else "")
               if self.sync_every == 1 else "%d local minibatches between parameter averagings (ncclAvg)" % self.sync_every)
        return "dp%d: one learner per GPU behind the C ABI (cpp_%s_dp_train_step), own replay shard, %s over RCCL" % (
            self.world, "naf" if self.is_naf else "ddpg", how)

    def dp_status(self):
        """which form the default step takes on this rank: 'hipgraph' (one graph replay per outer step, the all-reduce inside),
        'stream' (the same launches on the stream: the runtime or RCCL refused the capture -- `reason` says what it said) or
        'none' (no default-mode step has run: --sync-every / --overlap take the half-step path)."""
        import ctypes
        mode, reason = ctypes.c_int(0), ctypes.create_string_buffer(256)
        fn = self._lib.cpp_naf_dp_status if self.is_naf else self._lib.cpp_ddpg_dp_status
        self._check(fn(self.agent.naf.handle if self.is_naf else self.agent.trainer.handle, ctypes.byref(mode), reason, 256))
        return {"path": ("none", "hipgraph", "stream")[mode.value], "reason": reason.value.decode("utf-8", "replace")}

    def agreement(self):
        return LoopAgreement(self.comm)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


# ---------------------------------------------------------------------------------------------------------------------
# the protocol on the host (CPU tests with gloo; fallback with torch's collective)
# ---------------------------------------------------------------------------------------------------------------------
class _DeviceBufferView(object):
    """zero-copy view of a device f32 buffer for torch.as_tensor(..., device='cuda')."""

    def __init__(self, dev_ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (int(n_floats),), "typestr": "<f4",
                                         "data": (int(dev_ptr), False), "version": 2, "strides": None}


class GradAllReducer(object):
    """sum- / mean-all-reduce of flat buffers through torch.distributed.  `tensor` may be any torch tensor (CPU + gloo in
    the unit tests; the HIP library's device buffer + RCCL in the fallback learner)."""

    def __init__(self, tensor, group=None, stream=None, param_tensors=()):
        import torch.distributed as dist
        self.dist, self.tensor, self.group, self.stream = dist, tensor, group, stream
        self.param_tensors = list(param_tensors)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.always = False        # run the collective even at world size 1 (plumbing tests)

    @classmethod
    def for_trainer(cls, trainer, torch_stream, group=None):
        import torch
        p, n = trainer.grad_buffer()
        t = torch.as_tensor(_DeviceBufferView(p, n), device="cuda:%d" % trainer.ctx.device_id)
        assert t.data_ptr() == p and t.numel() == n
        return cls(t, group, torch_stream)

    def _run(self, fn):
        if self.world == 1 and not self.always:
            return
        if self.stream is not None:
            import torch
            with torch.cuda.stream(self.stream):
                fn()
        else:
            fn()

    def allreduce_sum(self):
        self._run(lambda: self.dist.all_reduce(self.tensor, op=self.dist.ReduceOp.SUM, group=self.group))

    def average_params(self):
        def fn():
            for t in self.param_tensors:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
                t.mul_(1.0 / self.world)
        self._run(fn)


class DataParallelLearner(object):
    """the inner train step (ddpg_cartpole.py:331-337) for N synchronous learners -- the same sequence as
    cpp_ddpg_dp_train_step, with the device-side pieces supplied by `ops` so that it can be unit-tested on CPU:
      ops.sample_and_compute()   -> fills the flat gradient buffer
      ops.apply(grad_scale)      -> clip + SGD on grad_scale * buffer
      ops.update_targets()
    sync_every = k > 1: k local updates (grad_scale 1), then reducer.average_params()."""

    def __init__(self, ops, reducer, sync_every=1):
        self.ops, self.reducer, self.sync_every, self._local = ops, reducer, int(sync_every), 0

    def train_step(self, batches_per_step):
        for _ in range(batches_per_step):
            self.ops.sample_and_compute()
            if self.sync_every > 1:
                self.ops.apply(1.0)
                self._local += 1
                if self._local >= self.sync_every:
                    self.reducer.average_params()
                    self._local = 0
            else:
                self.reducer.allreduce_sum()
                self.ops.apply(1.0 / self.reducer.world)
        self.ops.update_targets()


class AgentOps(object):
    """DataParallelLearner ops of a real DDPG agent (HIP path, half-step entry points)."""

    def __init__(self, agent, batch_size, seed):
        self._lib, self._check = _abi()
        self.agent, self.B, self.seed = agent, int(batch_size), int(seed)
        self.trainer = agent.trainer

    def sample_and_compute(self):
        self._check(self._lib.cpp_ddpg_sample_and_compute(self.trainer.handle, self.agent.replay_memory.handle,
                                                          self.B, self.seed))

    def apply(self, grad_scale):
        self._check(self._lib.cpp_ddpg_apply_gradients(self.trainer.handle, C.c_float(grad_scale)))

    def update_targets(self):
        self._check(self._lib.cpp_ddpg_update_targets(self.trainer.handle))


class _TorchAgreement(object):
    """LoopAgreement over the torch.distributed group (the diagnostic learner below has no cpp_comm)."""

    def __init__(self, device=None):
        self.calls, self.device = 0, device

    def all(self, flag):
        import torch
        import torch.distributed as dist
        self.calls += 1
        t = torch.tensor([0.0 if flag else 1.0], device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) == 0.0


class TorchCollectiveLearner(DataParallelLearner):
    """DIAGNOSTIC learner (bench.py --diag-backend gloo: several ranks on a box with fewer GPUs, where an RCCL communicator cannot be
    formed): the library's half steps + torch.distributed's all-reduce of its gradient buffer (zero-copy view).  Chosen explicitly
    (`make_learner(collective="torch")`), never as a silent fallback."""

    def __init__(self, agent, batch_size, seed, torch_stream, always=False):
        reducer = GradAllReducer.for_trainer(agent.trainer, torch_stream)
        reducer.always = always
        super(TorchCollectiveLearner, self).__init__(AgentOps(agent, batch_size, seed), reducer)
        self.world, self.B = reducer.world, int(batch_size)
        self._device = reducer.tensor.device

    def describe(self):
        import torch.distributed as dist
        be = dist.get_backend() if dist.is_initialized() else "none"
        return ("dp%d DIAGNOSTIC: one learner per process, half steps behind the C ABI + torch.distributed all_reduce (backend %s%s) of "
                "the flat gradient buffer per minibatch" % (self.world, be, " = RCCL" if be == "nccl" else ""))

    def agreement(self):
        import torch.distributed as dist
        return _TorchAgreement(self._device if dist.get_backend() == "nccl" else None)

    def close(self):
        pass


def _all_ranks_ok(ok, what, err=None):
    """collective: raise on EVERY rank if `ok` is False on any (one rank failing alone would leave the others blocked in the next
    collective).  Uses the torch.distributed group that carried the communicator id; a world of one decides alone."""
    import torch.distributed as dist
    msgs = [None]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        msgs = [None] * dist.get_world_size()
        dist.all_gather_object(msgs, None if ok else "rank %d: %s" % (dist.get_rank(), err))
    elif not ok:
        msgs = [str(err)]
    bad = [m for m in msgs if m is not None]
    if bad:
        raise RuntimeError("%s failed on %d rank(s): %s" % (what, len(bad), "; ".join(bad)))


def make_learner(agent, batch_size, seed, sync_every=1, overlap=False, always=False, torch_stream=None, collective="rccl"):
    """the learner bench.py / the agents' --data-parallel mode use.  collective = "rccl": NativeLearner over a cpp_comm whose id
    travels through the initialised torch.distributed group (or a world of one) -- the ranks agree that every one of them formed
    the communicator, otherwise ALL of them raise (no per-rank fallback: ranks on different collectives deadlock).
    collective = "torch": the diagnostic TorchCollectiveLearner."""
    import torch.distributed as dist
    if collective == "torch":
        assert not hasattr(agent, "naf") and sync_every == 1, "the diagnostic learner runs DDPG with a per-minibatch all-reduce only"
        return TorchCollectiveLearner(agent, batch_size, seed, torch_stream, always)
    assert collective == "rccl", collective
    ctx = (agent.naf if hasattr(agent, "naf") else agent.trainer).ctx
    comm, err = None, None
    try:
        if dist.is_available() and dist.is_initialized():
            comm = Communicator.from_torch_distributed(ctx)
        elif always:
            comm = Communicator.single(ctx)
    except Exception as e:      # noqa: BLE001 -- reported on every rank by _all_ranks_ok
        err = e
    try:
        _all_ranks_ok(err is None, "cpp_comm_create (RCCL communicator of the data-parallel learners)", err)
    except Exception:
        if comm is not None:
            comm.close()
        raise
    return NativeLearner(agent, batch_size, seed, comm, sync_every, overlap)


def init_process_group_from_env():
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run exports them -> (rank, world, local_rank); initialises
    the default group (RCCL) if nobody has yet.  A plain run is a world of one and touches nothing."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return 0, 1, local_rank
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return dist.get_rank(), dist.get_world_size(), local_rank


def learner_for_agent(agent, opts, batch_size):
    """the learner behind `--data-parallel` of ddpg_cartpole.py / naf_cartpole.py: one actor-learner per process (launched by
    torch.distributed.run; a plain run is a world of one), own environment and own replay shard, the sampler seeded per rank.
    Every rank takes rank 0's parameters first.  torch.distributed only carries the communicator id and that one broadcast."""
    rank, world, local_rank = init_process_group_from_env()
    if world > 1 and not getattr(agent, "_replicas_synced", False):      # once per agent: a learner rebuilt for a new batch size keeps them
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else None
        sync_replicas_from_rank0(agent, dist, device=dev)
        agent._replicas_synced = True
    return make_learner(agent, batch_size, seed=int(opts.sample_seed) + rank, sync_every=int(opts.sync_every),
                        overlap=bool(opts.overlap_allreduce), always=True)


def setup_data_parallel(agent, opts, batch_size=None):
    """main() of both agents under --data-parallel, BEFORE the first rollout: process group, replica sync and the communicator are
    all collective, so every rank does them at the same point of the program (not lazily inside whichever train step comes first
    on that rank).  Returns the learner (also kept as agent._learner)."""
    if getattr(agent, "_learner", None) is None or (batch_size is not None and agent._learner.B != int(batch_size)):
        if getattr(agent, "_learner", None) is not None:
            agent._learner.close()
        agent._learner = learner_for_agent(agent, opts, int(batch_size if batch_size is not None else opts.batch_size))
    return agent._learner


def shutdown_data_parallel(agent=None):
    """after the agreed exit from the training loop: communicator first, then the process group."""
    if agent is not None and getattr(agent, "_learner", None) is not None:
        agent._learner.close()
        agent._learner = None
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:       # noqa: BLE001
        pass


def sync_replicas_from_rank0(agent, dist, device=None):
    """every process built its networks from its own random numbers (or, rank 0, from a checkpoint): the synchronous step keeps
    replicas identical only if they START identical, so rank 0's parameters -- target networks and optimiser slots included -- go to
    every rank once, as host arrays through torch.distributed (a few MB)."""
    rank = dist.get_rank()
    nets = list(agent.networks())
    opt_owner = getattr(agent, "naf", None)
    payload = [None]
    if rank == 0:
        payload = [{"params": [n.get_params() for n in nets],
                    "opt": opt_owner.get_optimiser_state() if opt_owner is not None and hasattr(opt_owner, "get_optimiser_state") else None}]
    kw = {"device": device} if device is not None and dist.get_backend() == "nccl" else {}
    dist.broadcast_object_list(payload, src=0, **kw)
    if rank != 0:
        for n, p in zip(nets, payload[0]["params"]):
            n.set_params(p)
        if payload[0]["opt"] is not None:
            opt_owner.set_optimiser_state(payload[0]["opt"])
