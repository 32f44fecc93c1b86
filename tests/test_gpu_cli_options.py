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
    ("ddpg", PIXELS + ["--async-rollouts"]),                          # episodes on a rollout thread, the learner trains back to back
    ("naf", PIXELS + ["--async-rollouts", "--data-parallel"]),        # ... as one learner of a world of one (RCCL agreement per iteration)
]


@pytest.mark.parametrize("which,args", CASES, ids=["ddpg-host-rng", "ddpg-u8-bn", "ddpg-dropout", "ddpg-lowdim", "naf-adam-shared", "naf-bn", "naf-host-rng",
                              "ddpg-async-rollouts", "naf-async-rollouts-dp1"])
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
    if "--async-rollouts" in args:       # the learner did not wait for episodes: more train calls than STATS lines since burn-in
        assert stats[-1]["train_calls"] >= len([s for s in stats if np.isfinite(s["mean_losses"])])


def test_record_an_event_log_then_train_from_it_offline(tmp_path, capsys):
    """the workflow of exps/run_81-84: one run plays and records (--event-log-out, bullet_cartpole.py:27,90-94), the next trains from
    the log alone (--event-log-in --dont-do-rollouts)."""
    from cartpoleplusplus_amd import ddpg_cartpole as D, event_log as E
    path = str(tmp_path / "played.log")
    D.main(PIXELS + ["--event-log-out", path])
    capsys.readouterr()
    episodes = list(E.EventLogReader(path).entries())
    assert len(episodes) >= 5 and all(len(ep.event) >= 2 for ep in episodes)
    assert len(episodes[0].event[0].action) == 0 and len(episodes[0].event[1].action) == 2          # first event: just the state
    D.main(["--use-raw-pixels", "--render-width", "16", "--render-height", "16", "--batch-size", "8", "--replay-memory-size", "400",
            "--replay-memory-burn-in", "20", "--max-episode-len", "12", "--max-num-actions", "1", "--max-run-time", "1", "--synthetic-env", "--event-log-in", path, "--dont-do-rollouts"])
    out = capsys.readouterr().out
    last = json.loads([l for l in out.splitlines() if l.startswith("STATS")][-1].split("\t", 1)[1])
    assert last["replay_memory_stats"][">add_episode"] == len(episodes) and np.isfinite(last["mean_losses"])
