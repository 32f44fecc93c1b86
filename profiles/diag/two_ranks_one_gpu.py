"""diagnostic: can two ranks share the one GPU of a gpurun box (RCCL normally refuses duplicate devices)?  If yes, the N = 2
data-parallel step can be exercised for real: replicas must stay bit-identical and equal the mean-gradient update."""
import os, sys, socket
import numpy as np
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.helpers import make_pair
    from cartpoleplusplus_amd.distributed import Communicator, NativeLearner
    shape, B = (16, 16, 3, 2, 3), 16
    agent, _ref, _ = make_pair(shape, B, True, replay_size=300)
    agent.replay_memory.fill_synthetic(200, seed=11 + rank)          # own shard per rank
    try:
        comm = Communicator.from_torch_distributed(agent.trainer.ctx)
        for mode in (dict(), dict(overlap=True), dict(sync_every=3)):
            learner = NativeLearner(agent, B, 1234 + rank, comm, **mode)
            for _ in range(3):
                learner.train_step(3)
        agent.actor.ctx.sync()
        out[rank] = ("ok", agent.actor.get_params(), agent.critic.get_params())
    except Exception as e:
        out[rank] = ("fail", repr(e)[:300])
    dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    print(r0[0], r1[0], r0[1] if r0[0] == "fail" else "", r1[1] if r1[0] == "fail" else "")
    if r0[0] == r1[0] == "ok":
        print("replicas identical:", np.array_equal(r0[1], r1[1]) and np.array_equal(r0[2], r1[2]))
