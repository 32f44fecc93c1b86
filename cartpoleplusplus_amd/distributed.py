"""Data-parallel actor-learners: one process per GPU, replicated weights, one flat-gradient all-reduce
per minibatch over RCCL/xGMI (torch.distributed backend "nccl" == RCCL on ROCm).

The reference is single-process (its only nod to this is the TODO "switch back to async training with
multiple replicas", ddpg_cartpole.py:259).  Per minibatch every learner samples from its own replay
shard and leaves [actor grads | critic grads] in ONE flat f32 buffer (cpp_ddpg_sample_and_compute); the
buffer is summed across ranks, and every rank applies clip + SGD to the mean (cpp_ddpg_apply_gradients
with grad_scale = 1/world) -- identical inputs on every rank, so the replicas stay bit-identical with no
parameter broadcast.  Whitening statistics and target soft updates are local.

torch is plumbing here: it owns the process group and the collective; the gradient buffer belongs to the
HIP library and is exposed to torch zero-copy through __cuda_array_interface__.
"""
import ctypes as C


class _DeviceBufferView(object):
    """zero-copy view of a device f32 buffer for torch.as_tensor(..., device='cuda')."""

    def __init__(self, dev_ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (int(n_floats),), "typestr": "<f4",
                                         "data": (int(dev_ptr), False), "version": 2, "strides": None}


class GradAllReducer(object):
    """sum-all-reduce of a flat gradient buffer.  `tensor` may be any torch tensor (CPU + gloo in the
    unit tests; the HIP library's device buffer + RCCL in production)."""

    def __init__(self, tensor, group=None, stream=None):
        import torch.distributed as dist
        self.dist, self.tensor, self.group, self.stream = dist, tensor, group, stream
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.always = False        # run the collective even at world size 1 (plumbing tests)

    @classmethod
    def for_trainer(cls, trainer, torch_stream, group=None):
        import torch
        p, n = trainer.grad_buffer()
        t = torch.as_tensor(_DeviceBufferView(p, n), device="cuda:%d" % trainer.ctx.device_id)
        assert t.data_ptr() == p and t.numel() == n
        return cls(t, group, torch_stream)

    def allreduce_sum(self):
        if self.world == 1 and not self.always:
            return
        if self.stream is not None:
            import torch
            with torch.cuda.stream(self.stream):
                self.dist.all_reduce(self.tensor, op=self.dist.ReduceOp.SUM, group=self.group)
        else:
            self.dist.all_reduce(self.tensor, op=self.dist.ReduceOp.SUM, group=self.group)


class DataParallelLearner(object):
    """the inner train step (ddpg_cartpole.py:331-337) for N synchronous learners.

    `ops` supplies the three device-side pieces so that the protocol can be unit-tested on CPU:
      ops.sample_and_compute()   -> fills the flat gradient buffer
      ops.apply(grad_scale)      -> clip + SGD on grad_scale * buffer
      ops.update_targets()
    """

    def __init__(self, ops, reducer):
        self.ops, self.reducer = ops, reducer

    def train_step(self, batches_per_step):
        for _ in range(batches_per_step):
            self.ops.sample_and_compute()
            self.reducer.allreduce_sum()
            self.ops.apply(1.0 / self.reducer.world)
        self.ops.update_targets()


class AgentOps(object):
    """DataParallelLearner ops of a real agent (HIP path)."""

    def __init__(self, agent, batch_size, seed):
        from ._lib import lib, check
        self._lib, self._check = lib, check
        self.agent, self.B, self.seed = agent, int(batch_size), int(seed)
        self.trainer = agent.trainer

    def sample_and_compute(self):
        self._check(self._lib.cpp_ddpg_sample_and_compute(self.trainer.handle, self.agent.replay_memory.handle,
                                                          self.B, self.seed))

    def apply(self, grad_scale):
        self._check(self._lib.cpp_ddpg_apply_gradients(self.trainer.handle, C.c_float(grad_scale)))

    def update_targets(self):
        self._check(self._lib.cpp_ddpg_update_targets(self.trainer.handle))
