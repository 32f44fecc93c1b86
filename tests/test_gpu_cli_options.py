"""The agents' own main() (ddpg_cartpole.py:412-443 / naf_cartpole.py:442-480) with option combinations the other tests do not pair:
rollouts on the stand-in env, replay in HBM, training, STATS lines."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PIXELS = ["--synthetic-env", "--use-raw-pixels", "--render-width", "16", "--render-height", "16", "--batch-size", "8",
          "--replay-memory-size", "120", "--replay-memory-burn-in", "20", "--max-episode-len", "12", "--max-num-actions", "70"]
LOWDIM = ["--synthetic-env", "--batch-size", "8", "--replay-memory-size", "120", "--replay-memory-burn-in", "20",
          "--max-episode-len", "12", "--max-num-actions", "70"]
CASES = [
    ("ddpg", PIXELS + ["--host-rng-sampling"]),                       # the reference's literal loop: batch(), actor.train, critic.train
    ("ddpg", PIXELS + ["--replay-store", "u8", "--use-batch-norm"]),
    ("ddpg", PIXELS + ["--use-dropout"]),
    ("ddpg", LOWDIM),                                                 # configs[0]: the 28-d pose state
    ("naf", PIXELS + ["--optimiser", "Adam", "--optimiser-args", '{"learning_rate": 0.001}', "--share-input-state-representation"]),
    ("naf", PIXELS + ["--use-batch-norm"]),
    ("naf", PIXELS + ["--host-rng-sampling"]),
]


@pytest.mark.parametrize("which,args", CASES, ids=["ddpg-host-rng", "ddpg-u8-bn", "ddpg-dropout", "ddpg-lowdim", "naf-adam-shared", "naf-bn", "naf-host-rng"])
def test_cli_runs_and_trains(which, args, capsys):
    if which == "ddpg":
        from cartpoleplusplus_amd import ddpg_cartpole as M
    else:
        from cartpoleplusplus_amd import naf_cartpole as M
    M.main(list(args))
    out = capsys.readouterr().out
    stats = [json.loads(l.split("\t", 1)[1]) for l in out.splitlines() if l.startswith("STATS")]
    assert len(stats) >= 4 and any(np.isfinite(s["mean_losses"]) for s in stats), out[-400:]
    assert stats[-1]["replay_memory_stats"][">add"] >= 60
