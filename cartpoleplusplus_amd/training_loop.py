"""The outer training loop of both agents -- rollout an episode, add it to the replay memory, train, print STATS, decide whether
to stop (ddpg_cartpole.py:291-383, naf_cartpole.py:323-389 of the reference) -- in ONE place, with the two things the reference's
single-process loop never needed:

  * rank agreement under `--data-parallel`.  The train step of N learners is a collective (cartpoleplusplus_amd/distributed.py);
    "train this iteration" (the burn-in test, ddpg_cartpole.py:329) and "leave the loop" (:379-383) are facts of ONE learner's
    environment, replay shard and clock.  Decided per process, the first rank to finish would leave the others blocked in
    ncclAllReduce.  Here every outer iteration makes both decisions collectively (`distributed.LoopAgreement`, one 2-word
    all-reduce each): train only when EVERY rank is past burn-in, leave only when EVERY rank's own criterion is met -- every
    learner trains for "at least" its --max-num-actions / --max-run-time, as the flags' help texts promise, and all ranks issue
    exactly the same number of collective steps.

  * `--async-rollouts` (BASELINE configs[2]: "async actor-learners"; the reference's TODO ddpg_cartpole.py:259).  Physics and
    rendering stay on the host, but on a rollout THREAD: episodes are played with the live policy while the learner thread trains
    back to back, finished episodes wait in a bounded queue and are added to the device replay shard between train steps.  A
    synchronous learner waits for its own episode before every train step -- and, data-parallel, for the slowest rank's episode;
    an asynchronous one never waits for an environment once it is past burn-in.  The two threads share the agent's one context
    and stream through a fair lock (`FairLock`): the library's calls on a context are not re-entrant, and one stream keeps every
    device access ordered.

The agents keep `run_training(max_num_actions, max_run_time, batch_size, batches_per_step, saver_util)` and delegate to
`TrainingLoop`; what differs between them (how to act, how to train once, NAF's timing prints) is passed in.
"""
import collections
import contextlib
import datetime
import json
import queue
import sys
import threading
import time

import numpy as np


class FairLock(object):
    """first-come-first-served mutex (threading.Lock may hand the lock back to the thread that just released it: the learner thread
    re-acquires within microseconds and would starve the rollout thread)."""

    def __init__(self):
        self._cv = threading.Condition()
        self._next, self._serving = 0, 0
        self._abandoned = set()

    def __enter__(self):
        with self._cv:
            ticket = self._next
            self._next += 1
            try:
                while self._serving != ticket:
                    self._cv.wait()
            except BaseException:            # KeyboardInterrupt while queued: this ticket would block everybody behind it for ever
                self._abandoned.add(ticket)
                self._skip_abandoned()
                raise
        return self

    def _skip_abandoned(self):
        while self._serving in self._abandoned:
            self._abandoned.discard(self._serving)
            self._serving += 1
        self._cv.notify_all()

    def __exit__(self, *exc):
        with self._cv:
            self._serving += 1
            self._skip_abandoned()
        return False


Episode = collections.namedtuple("Episode", "initial_state sequence rewards seconds")


def play_episode(env, act):
    """one episode with `act(state) -> action` (ddpg_cartpole.py:313-326)."""
    t0 = time.time()
    state_1 = env.reset()
    initial_state = np.copy(state_1)
    sequence, rewards = [], []
    done = False
    while not done:
        action = act(state_1)
        state_2, reward, done, _ = env.step(action)
        rewards.append(reward)
        sequence.append((action, reward, np.copy(state_2)))
        state_1 = state_2
    return Episode(initial_state, sequence, rewards, time.time() - t0)


class AsyncRollouts(object):
    """the rollout thread of --async-rollouts: plays episodes with `act` and, after every `eval_every`-th, one evaluation episode
    through `evaluate()` (the environment belongs to this thread, so the learner thread must not run its own)."""

    def __init__(self, env, act, evaluate=None, eval_every=10, max_queued=8):
        self.env, self.act, self.evaluate, self.eval_every = env, act, evaluate, int(eval_every)
        self.queue = queue.Queue(maxsize=max_queued)
        self._stop = threading.Event()
        self.error = None
        self.episodes_played = 0
        self.thread = threading.Thread(target=self._run, name="cartpolepp-rollouts", daemon=True)

    def start(self):
        self.thread.start()
        return self

    def _run(self):
        try:
            while not self._stop.is_set():
                ep = play_episode(self.env, self.act)
                self.episodes_played += 1
                while not self._stop.is_set():
                    try:
                        self.queue.put(ep, timeout=0.05)
                        break
                    except queue.Full:          # the learner is behind (or below burn-in on another rank): wait, do not drop
                        pass
                if self.evaluate is not None and self.eval_every > 0 and self.episodes_played % self.eval_every == 0 \
                        and not self._stop.is_set():
                    self.evaluate()
        except BaseException as e:      # noqa: BLE001 -- handed to the learner thread, which re-raises it
            self.error = e

    def drain(self, wait_for_one=False, timeout=None):
        """every finished episode (at least one if wait_for_one; the rollout thread's exception is re-raised here)."""
        out = []
        if wait_for_one:
            deadline = None if timeout is None else time.time() + timeout
            while not out:
                if self.error is not None:
                    raise self.error
                try:
                    out.append(self.queue.get(timeout=0.05))
                except queue.Empty:
                    if deadline is not None and time.time() > deadline:
                        break
        while True:
            try:
                out.append(self.queue.get_nowait())
            except queue.Empty:
                break
        if self.error is not None:
            raise self.error
        return out

    def stop(self):
        self._stop.set()
        deadline = time.time() + 30.0           # (an episode in flight finishes first; a thread that is stuck is left to the daemon flag)
        while self.thread.is_alive() and time.time() < deadline:           # unblock a put() on a full queue
            try:
                self.queue.get_nowait()
            except queue.Empty:
                pass
            self.thread.join(timeout=0.05)
        if self.thread.is_alive():
            sys.stderr.write("AsyncRollouts.stop: the rollout thread did not finish within 30 s; leaving it behind\n")


class TrainingLoop(object):
    """agent: replay_memory, env, run_eval(n); act(state) -> action with exploration noise; train(batch_size, batches_per_step)
    -> list of losses (the inner step ddpg_cartpole.py:331-337 / naf_cartpole.py:367-373, whichever way the agent runs it).
    `agreement` (distributed.LoopAgreement or None), `device_lock` (FairLock, --async-rollouts) and `verbose()` are optional."""

    def __init__(self, agent, opts, act, train, agreement=None, verbose=None, after_train=None, timing_prints=False,
                 dump_weights_requested=None, out=None):
        self.agent, self.opts, self.act, self.train = agent, opts, act, train
        self.agreement, self.verbose, self.after_train = agreement, verbose or (lambda: False), after_train
        self.timing_prints, self.dump_weights_requested = timing_prints, dump_weights_requested
        self.out = out or sys.stdout
        self.iterations = self.train_calls = 0

    def _print(self, *a):
        print(*a, file=self.out)

    def run(self, max_num_actions, max_run_time, batch_size, batches_per_step, saver_util):
        agent, opts = self.agent, self.opts
        lock = getattr(agent, "device_lock", None) or contextlib.nullcontext()
        rollouts = None
        if getattr(opts, "async_rollouts", False) and not opts.dont_do_rollouts:
            rollouts = AsyncRollouts(agent.env, self.act, evaluate=lambda: agent.run_eval(1)).start()
        start_time = time.time()
        num_actions_taken = 0
        n = 0
        try:
            while True:
                rewards, losses, episodes = [], [], []
                if rollouts is not None:
                    # below burn-in there is nothing to do but wait for an episode; past it the learner never waits
                    below = agent.replay_memory.size() <= opts.replay_memory_burn_in
                    episodes = rollouts.drain(wait_for_one=below, timeout=1.0)
                elif not opts.dont_do_rollouts:
                    episodes = [play_episode(agent.env, self.act)]          # physics + rendering stay on the host
                with lock:
                    for ep in episodes:
                        if self.timing_prints:
                            self._print("episode_took", ep.seconds, len(ep.rewards))
                        t0 = time.time()
                        agent.replay_memory.add_episode(ep.initial_state, ep.sequence)
                        if self.timing_prints:
                            self._print("replay_took", time.time() - t0)
                        rewards.extend(ep.rewards)

                    # do a training step (after waiting for buffer to fill a bit...) -- collectively, if this is one learner of N
                    ready = agent.replay_memory.size() > opts.replay_memory_burn_in
                    if self.agreement is not None:
                        ready = self.agreement.all(ready)
                    if ready:
                        losses.extend(self.train(batch_size, batches_per_step))
                        self.train_calls += 1
                        if self.after_train is not None:
                            self.after_train(batch_size)

                    if episodes or rollouts is None:
                        stats = collections.OrderedDict()
                        stats["time"] = time.time()
                        stats["n"] = n
                        stats["mean_losses"] = float(np.mean(losses)) if losses else float("nan")
                        stats["total_reward"] = float(np.sum(rewards))
                        stats["episode_len"] = len(rewards)
                        stats["replay_memory_stats"] = agent.replay_memory.current_stats()
                        if rollouts is not None:
                            stats["train_calls"] = self.train_calls
                        self._print("STATS %s\t%s" % (datetime.datetime.now().strftime('%Y-%m-%d %H:%M:%S'), json.dumps(stats)))
                        self.out.flush()
                        n += 1
                    if saver_util is not None:
                        saver_util.save_if_required()
                    if self.dump_weights_requested is not None and self.dump_weights_requested():
                        agent.debug_dump_network_weights()
                if rollouts is None and (self.verbose() or n % 10 == 0):
                    agent.run_eval(1)
                self.iterations += 1

                num_actions_taken += len(rewards)
                stop = ((max_num_actions > 0 and num_actions_taken > max_num_actions)
                        or (max_run_time > 0 and time.time() > start_time + max_run_time)
                        or (opts.dont_do_rollouts and max_num_actions <= 0 and max_run_time <= 0))
                if self.agreement is not None:
                    with lock:
                        stop = self.agreement.all(stop)
                if stop:
                    break
        finally:
            if rollouts is not None:
                rollouts.stop()
