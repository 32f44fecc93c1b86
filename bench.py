#!/usr/bin/env python
"""Headline benchmark: DDPG training steps/sec on synthetic 64x64x18 pixel minibatches, batch 256
(BASELINE.json metric).  One process per GPU; N > 1 runs under torch.distributed.run -- `python bench.py --gpus N` as a
plain command launches its own N ranks that way.

A "step" is one minibatch update of the hot path: fused sample + gather of B transitions from the
HBM-resident replay memory, actor update, critic update (4 conv-trunk forwards, 2 backwards), global-norm
clip + SGD for both nets, with both target soft updates after every 5th minibatch
(ddpg_cartpole.py:331-337).  Inputs are resident in HBM when the timed region starts.  The timed region is exactly --steps
minibatches; in front of it run --warmup untimed ones and as many more as it takes to reach 200 (`warmup_steps_run` in the
JSON line): the chip's clock needs tens of milliseconds of load to settle, and a 20-step region started 10 steps after idle
reports the ramp, not the rate (measured 3-7 % low).

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline       : the dominant kernel (conv1 forward) against the matrix-pipe peak of the instruction it issues, launch
                   durations measured with HIP events on the stream the kernel runs on (a profiled pass of the same step
                   sequence, run right after the timed region; the timed region itself is one hipGraph replay per 5
                   minibatches and cannot carry per-kernel events); `traffic` from the committed rocprofv3 PMC passes
  layers         : every conv launch of the step: algorithmic GFLOP, microseconds, the pipe it runs on, fraction of that
                   pipe's bound; `conv_bound_frac_whole_step` = sum of the per-layer bound times / step time
  cpu_baseline   : the same minibatch update on this host's cores: torch-CPU restatement (oracle/ddpg_torch.py, the
                   stand-in for the reference's TF-CPU kernels) -- and `cpu_baseline_numpy`, the numpy oracle
  control        : the same step with conv1 / conv2 forced onto the f32-input MFMA kernels (ablation build of the library,
                   CPP_CONV_K16=0 CPP_CONV_B16=0): the f32 twin of the f16x2 / bf16x6 numbers
  control_exact_products : the same step with --precision exact (three f16 pieces, nine bf16 products: every product exact)
  extra          : short runs of the other BASELINE configs (cfg2, cfg4 = NAF, cfg5), steps/s each
(N = 1, rank 0 only for the last three; --quick skips them.)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (state_shape (H, W, 3, cameras, repeats), batch, agent)
    "cfg3": ((64, 64, 3, 2, 3), 256, "ddpg"),    # 64x64x18, B=256 -- the configuration the metric is quoted on
    "cfg2": ((64, 64, 3, 1, 3), 256, "ddpg"),    # 64x64x9
    "cfg4": ((64, 64, 3, 2, 3), 256, "naf"),     # NAF, 64x64x18, shared conv trunk (BASELINE configs[3])
    "cfg5": ((128, 128, 3, 2, 5), 512, "ddpg"),  # 128x128x30, B=512 (BASELINE configs[4]; replay rows: see REPLAY_ROWS_BY)
    "r50": ((50, 50, 3, 2, 3), 256, "ddpg"),     # the reference's default 50x50 render (exps/run_98.sh: 2 cameras, 3 repeats)
}
REPLAY_ROWS_BY = {"cfg5": 6000}          # 9000 state slots x 983 KB = 8.8 GB (--replay-rows overrides; 125 000 rows = one GPU's shard of configs[4])
BATCHES_PER_STEP = 5                     # --batches-per-step default (ddpg_cartpole.py:30); main() rebinds it from the flag of the same name
REPLAY_ROWS = 22000                      # --replay-memory-size default (ddpg_cartpole.py:46)
SETTLE_STEPS = 200                       # untimed minibatches in front of the timed region, at least (see main)
PEAK_F32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_F16_MFMA_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak
CONV_DEFS = ((5, 10), (5, 10), (3, 10))
# matrix pipe of a profiled kernel name -> (peak TFLOP/s for ALGORITHMIC flops, description)
SUSTAINED_F16_MFMA_TFLOPS_RANDOM_OPERANDS = 2000      # measured 1914-2056 on two boxes: profiles/r02_mfma_rate_probe.txt (informational, see roofline["sustained"])
# --precision: the release library multiplies two f16 pieces of conv1's f32 operand and six bf16 piece products in conv2 ("fast", the
# default) or three and nine -- every product exact ("exact"; cpp_ctx_set_precision, conv_k16.h)
EXACT_PRODUCTS = "exact" in [a.split("=")[-1] for i, a in enumerate(sys.argv) if a.startswith("--precision=") or (i and sys.argv[i - 1] == "--precision")]
F16_PIPE, B16_PIPE = ("f16x3", "bf16x9") if EXACT_PRODUCTS else ("f16x2", "bf16x6")
PIPES = {
    "f16x2": (PEAK_F16_MFMA_TFLOPS / 2.0, "f16 MFMA, 2 f16 x f16 piece products per f32 product (2500 / 2)"),
    "f16x3": (PEAK_F16_MFMA_TFLOPS / 3.0, "f16 MFMA, 3 exact f16 x f16 products per f32 product (2500 / 3)"),
    "bf16x9": (PEAK_F16_MFMA_TFLOPS / 9.0, "bf16 MFMA, 9 exact bf16 x bf16 products per f32 product (2500 / 9)"),
    "bf16x6": (PEAK_F16_MFMA_TFLOPS / 6.0, "bf16 MFMA, 6 bf16 x bf16 piece products per f32 product (2500 / 6)"),
    "f32": (PEAK_F32_MFMA_TFLOPS, "f32-input MFMA"),
}


def conv_macs(shape):
    """per-image forward MACs F and backward MACs Bk of the trunk (bwd = conv1 dW +
    conv2/3 dW + dX), unpadded."""
    H, W, cin = shape[0], shape[1], int(np.prod(shape[2:]))
    per_layer = []
    for k, cout in CONV_DEFS:
        per_layer.append(H * W * k * k * cin * cout)
        cin, H, W = cout, H // 2, W // 2
    F = sum(per_layer)
    Bk = per_layer[0] + 2 * sum(per_layer[1:])
    return F, Bk, per_layer


def _host_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    except Exception:
        return os.cpu_count() or 1


def _oracle_agent(shape, rng):
    from oracle import ddpg_np as O          # checker / baseline only
    kw = dict(pixel=True, H=shape[0], W=shape[1], C=int(np.prod(shape[2:])))
    aspec, cspec = O.NetSpec("actor", 2, [100, 100, 50], **kw), O.NetSpec("critic", 2, [], **kw)
    return O, aspec, cspec, O.init_params(aspec, rng), O.init_params(cspec, rng)


def cpu_baseline_numpy(shape, B, budget_s=10.0, max_reps=4):
    """full-batch minibatch updates of the numpy oracle (f32) on this host until ~budget_s of CPU work is done."""
    rng = np.random.default_rng(0)
    O, aspec, cspec, af, cf = _oracle_agent(shape, rng)
    agent = O.DDPG(aspec, cspec, af, cf, np.float32)
    batch = O.synthetic_batch(rng, B, shape, 2, True)
    threads = _host_threads()
    reps, t0 = 0, time.time()
    while reps < max_reps and (reps == 0 or time.time() - t0 < budget_s):
        agent.train_minibatch(batch)
        reps += 1
    dt = time.time() - t0
    return {"value": round(reps / dt, 4), "unit": "steps/s", "cores": int(threads), "kind": "port", "impl": "oracle/ddpg_np.py",
            "sample": "%d full minibatch update(s) (B=%d, same shapes as the GPU workload) of oracle/ddpg_np.py, "
                      "numpy f32 with OpenBLAS on %d threads, %.1f s wall" % (reps, B, threads, dt)}


def cpu_baseline_torch(shape, B, budget_s=25.0):
    """the same update on torch's CPU kernels (oneDNN conv, autograd): SURVEY 8(d)'s proxy for the reference's TF-CPU path
    (TensorFlow 0.x cannot be installed).  Thread counts 8, 16, 32 and 64 are tried within the budget (a 10-filter conv
    does not scale to hundreds of threads) and the best is reported."""
    import torch
    from oracle.ddpg_torch import TorchDDPG   # baseline only
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    O, aspec, cspec, af, cf = _oracle_agent(shape, rng)
    agent = TorchDDPG(aspec, cspec, af, cf)
    batch = O.synthetic_batch(rng, B, shape, 2, True)
    tried, t_all = {}, time.time()
    for threads in sorted({min(cores, 8), min(cores, 16), min(cores, 32), min(cores, 64)}):
        if tried and time.time() - t_all > budget_s:
            break
        if len(tried) >= 3 and list(tried.values())[-1] < list(tried.values())[-2] < list(tried.values())[-3]:
            break           # twice slower in a row with more threads (at 256 threads one update of this 10-filter net takes 27 s)
        torch.set_num_threads(threads)
        agent.train_minibatch(batch)          # warm-up (thread pool, oneDNN primitive cache)
        reps, t0 = 0, time.time()
        while reps < 3 and (reps == 0 or time.time() - t0 < budget_s / 6):
            agent.train_minibatch(batch)
            reps += 1
        tried[threads] = reps / (time.time() - t0)
    best = max(tried, key=tried.get)
    return {"value": round(tried[best], 4), "unit": "steps/s", "cores": int(best), "kind": "port",
            "impl": "oracle/ddpg_torch.py (torch %s CPU: oneDNN convolutions + autograd) -- proxy for the reference's TF-CPU kernels" % torch.__version__,
            "sample": "full minibatch updates after 1 warm-up (B=%d, same shapes as the GPU workload), float32; steps/s by thread count "
                      "%s on a %d-core host; %.1f s wall" % (B, {k: round(v, 3) for k, v in tried.items()}, cores, time.time() - t_all)}


def sub_bench(extra_args, env=None, timeout=420):
    """one more bench.py process (another workload, or the ablation library); returns its parsed JSON line or an error."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--quick"] + list(extra_args)
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=timeout)
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "rc %d: %s" % (r.returncode, r.stderr.decode()[-300:])}
        return json.loads(lines[-1])
    except Exception as e:      # noqa: BLE001 -- a failed side run must not take the headline line with it
        return {"error": repr(e)}


def self_launch(n):
    """re-run this command line as n ranks: python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <same flags>."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, cwd=ROOT, env=env)


def main():
    global BATCHES_PER_STEP
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--replay-rows", type=int, default=0, help="rows of the replay shard (default: 22000; cfg5: 6000)")
    ap.add_argument("--batches-per-step", type=int, default=BATCHES_PER_STEP,
                    help="minibatches per train-step call = per target soft update (ddpg_cartpole.py:30; 1: the per-minibatch-target-update variant of SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="headline + roofline only: no cpu baseline, control or extra runs")
    ap.add_argument("--profile-steps", type=int, default=10, help="minibatches of the per-kernel HIP-event pass")
    ap.add_argument("--use-batch-norm", action="store_true", help="informational: the networks of exps/run_8x / run_9x (--use-batch-norm)")
    ap.add_argument("--precision", default="fast", choices=["fast", "exact"],
                    help="arithmetic contract of conv1 / conv2 on the f16 / bf16 pipes (cpp_ctx_set_precision): operands to within one f32 ulp "
                         "(two f16 pieces, six bf16 products) or every operand bit (three, nine)")
    ap.add_argument("--replay-store", default="f16", choices=["f16", "u8"],
                    help="informational: u8 = the 8-bit replay store (same batches, half the gather reads)")
    ap.add_argument("--diag-states", default="random", choices=["random", "blocks", "flat"],
                    help="diagnostic only (never the benchmark): overwrite the synthetic U{0..255} pixels with 8x8 constant blocks / one grey level "
                         "to see how much of a kernel's time is the chip's clock under operand toggling")
    ap.add_argument("--diag-backend", default="nccl", choices=["nccl", "gloo"],
                    help="diagnostic only: torch.distributed backend of an N > 1 run (gloo: several ranks may share one GPU)")
    ap.add_argument("--force-dp", action="store_true", help="use the data-parallel learner path (RCCL all-reduce) even at world size 1")
    ap.add_argument("--overlap", action="store_true", help="data-parallel: reduce the fully connected layers' gradients beside the conv backward")
    ap.add_argument("--sync-every", type=int, default=1, help="data-parallel: k local minibatches between parameter averagings (1: gradient all-reduce per minibatch)")
    args = ap.parse_args()
    BATCHES_PER_STEP = max(1, args.batches_per_step)

    # keep real stdout for the ONE JSON line: RCCL / libraries print banners to fd 1
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as a plain command: start the N ranks ourselves (one process per GPU under torch.distributed.run,
        # exactly the driver's launch form) and hand rank 0's ONE JSON line through
        os.dup2(real_stdout, 1)
        sys.exit(self_launch(args.gpus))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    shape, B, kind = WORKLOADS[args.workload]
    replay_rows = args.replay_rows or REPLAY_ROWS_BY.get(args.workload, REPLAY_ROWS)

    import torch
    if args.diag_backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    use_dp = world > 1 or args.force_dp
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.diag_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:       # diagnostic: the multi-process harness on a box with fewer GPUs than ranks (the library's RCCL communicator cannot be
            dist.init_process_group(args.diag_backend, rank=rank, world_size=world)     # created there: the torch fallback learner runs)

    from cartpoleplusplus_amd import _lib

    stream = torch.cuda.Stream(device=local_rank)
    ctx = _lib.Context(local_rank, stream=stream.cuda_stream)
    _lib.set_default_context(ctx)

    class Env(object):
        class S(object):
            def __init__(self, s): self.shape = tuple(s)
        observation_space, action_space = S(shape), S((1, 2))

    common = dict(use_raw_pixels=True, render_height=shape[0], render_width=shape[1], num_cameras=shape[3],
                  action_repeats=shape[4], batch_size=B, replay_memory_size=replay_rows,
                  use_batch_norm=bool(args.use_batch_norm), replay_store=args.replay_store,
                  exact_products=(args.precision == "exact"))
    if kind == "ddpg":
        from cartpoleplusplus_amd import ddpg_cartpole as D
        D.set_opts(D.default_opts(sample_seed=1234 + rank, **common))
        agent = D.DeepDeterministicPolicyGradientAgent(Env())
    else:       # NAF on the shared trunk, Momentum as in exps/run_93.sh
        from cartpoleplusplus_amd import naf_cartpole as D
        D.set_opts(D.default_opts(share_input_state_representation=True, optimiser="Momentum",
                                  optimiser_args=json.dumps({"learning_rate": 0.01, "momentum": 0.9}), **common))
        agent = D.NormalizedAdvantageFunctionAgent(Env())
    agent.initialise_variables(seed=42)                 # identical replicas on every rank
    agent.post_var_init_setup()
    agent.replay_memory.fill_synthetic(replay_rows, seed=1234 + rank)   # own replay shard per learner
    if args.diag_states != "random":
        import ctypes as C_
        rm = agent.replay_memory
        H_, W_, ch_ = shape[0], shape[1], int(np.prod(shape[2:]))
        yy, xx, cc = np.meshgrid(np.arange(H_), np.arange(W_), np.arange(ch_), indexing="ij")
        used = replay_rows + replay_rows // 50 + 1
        for s0 in range(0, used, 256):
            slots = np.arange(s0, min(used, s0 + 256), dtype=np.int32)
            if args.diag_states == "flat":
                st = np.full((len(slots), H_ * W_ * ch_), 128, np.uint8)
            else:
                st = np.stack([((xx // 8) * 5 + (yy // 8) * 17 + cc * 7 + int(k) * 3) % 256 for k in slots]).astype(np.uint8).reshape(len(slots), -1)
            _lib.check(_lib.lib.cpp_replay_write_states(rm.handle, slots.ctypes.data_as(C_.c_void_p), len(slots), st.ctypes.data_as(C_.c_void_p), _lib.CPP_U8))
        ctx.sync()

    groups, tail = divmod(args.steps, BATCHES_PER_STEP)
    wgroups = max(1, -(-args.warmup // BATCHES_PER_STEP))

    learner = None
    if not use_dp:
        def run(g, t):
            for _ in range(g):
                agent.train_step(B, BATCHES_PER_STEP)
            if t:
                agent.train_step(B, t)
        parallelism = "single learner: fused inner step (one hipGraph replay per %d minibatches), no collective" % BATCHES_PER_STEP
    else:
        from cartpoleplusplus_amd.distributed import make_learner
        learner = make_learner(agent, B, seed=1234 + rank, sync_every=args.sync_every, overlap=args.overlap, always=args.force_dp,
                               torch_stream=stream, collective="rccl" if args.diag_backend == "nccl" else "torch")

        def run(g, t):
            for _ in range(g):
                learner.train_step(BATCHES_PER_STEP)
            if t:
                learner.train_step(t)
        parallelism = learner.describe()

    def full_sync():      # stream idle, barrier, device idle (the barrier's own collective kernel included: with the barrier LAST its tail ran into
        ctx.sync()        # the first steps of the timed region -- 1 ms of a 36 ms region, measured against the same steps without a barrier)
        if use_dp:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up (also captures the hipGraphs for both call shapes)
    run(wgroups, tail)
    run(1, tail)
    # ... and the chip's clock / power state: it takes tens of milliseconds of load to settle (measured, K = 20: 0.435-0.454 ms per
    # step when the timed region starts 10 minibatches after idle, 0.423 after 200 -- the steady state a 200-step region reports
    # whatever its warm-up).  So at least SETTLE_STEPS untimed minibatches precede the timed region whatever --warmup says (the same
    # number on every rank: the data-parallel step is collective); the JSON line says how many were run.
    settle_groups = max(0, SETTLE_STEPS // BATCHES_PER_STEP - wgroups - 1)
    run(settle_groups, 0)
    warmup_steps_run = (wgroups + 1 + settle_groups) * BATCHES_PER_STEP + 2 * tail
    full_sync()
    if use_dp:
        # the barrier is a host-side collective of about a millisecond during which the chip idles and its clock drops: the first steps
        # behind it ran 10-20 % slow (1.2 ms of a 36 ms region; the same steps without a barrier, or in the alternating blocks below,
        # do not show it; one outer step behind the barrier was not enough, the clock takes tens of milliseconds to come back).  The settle
        # steps run again behind the barrier -- themselves collective, so the ranks stay aligned -- and a device sync re-opens the region.
        run(SETTLE_STEPS // BATCHES_PER_STEP, 0)
        warmup_steps_run += SETTLE_STEPS // BATCHES_PER_STEP * BATCHES_PER_STEP
        ctx.sync()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(groups, tail)
    # this rank's K steps are done when its stream is idle; the closing barrier follows the clock read and the job's time is the MAX over
    # the ranks of these local times (all ranks left the opening barrier together).  (Reading the clock behind the barrier added the
    # collective's own host latency, ~1 ms at world size 1 = 2.7 % of a 100-step region, to every rank's time.)
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    full_sync()
    steps = groups * BATCHES_PER_STEP + tail
    per_rank = [round(steps / elapsed, 3)]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda:%d" % local_rank)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank = [round(steps / float(t.item()), 3) for t in every]
        elapsed = max(float(t.item()) for t in every)         # the MAX over the ranks is the job's time
    ms_per_step = 1e3 * elapsed / steps
    value = world * steps / elapsed

    # ---- what a first N > 1 run must tell us without anybody watching it (nobody has had more than one GPU): which form the step took on
    # every rank (one hipGraph with the all-reduce inside, or the same launches on the stream because the capture was refused -- and why),
    # whether the replicas still hold the same bits after the timed region (identical inputs to every update: they must), and what the
    # all-reduce costs where nothing hides it (between the gradient kernels and the update: HIP events around the collective, one
    # profiled pass, per minibatch)
    dp_diag = None
    if use_dp and learner is not None:
        import hashlib
        st = learner.dp_status() if hasattr(learner, "dp_status") else {"path": "torch-collective", "reason": ""}
        nets_ = [n for n in (getattr(agent, "actor", None), getattr(agent, "critic", None), getattr(agent, "target_actor", None),
                             getattr(agent, "target_critic", None)) if n is not None]
        if not nets_ and hasattr(agent, "naf"):
            nets_ = list(agent.naf.networks()) if hasattr(agent.naf, "networks") else []
        digest = hashlib.sha256(b"".join(n.get_params().tobytes() for n in nets_)).hexdigest()[:16] if nets_ else None
        mine = {"rank": rank, "path": st["path"], "reason": st["reason"], "params_sha256_16": digest}
        every = [mine]
        if world > 1:
            every = [None] * world
            dist.all_gather_object(every, mine)
        ar_us = None
        if hasattr(learner, "dp_status") and args.sync_every == 1 and not args.overlap:
            ctx.prof_reset(); ctx.prof_enable(True)
            learner.train_step(BATCHES_PER_STEP)                # (profiled pass: eager launches, every kernel and the collective between HIP events)
            ctx.sync(); ctx.prof_enable(False)
            pr_ = ctx.prof_read()
            if "allreduce" in pr_ and pr_["allreduce"][1] > 0:
                ar_us = round(1e3 * pr_["allreduce"][0] / pr_["allreduce"][1], 2)
        digests = [e["params_sha256_16"] for e in every]
        dp_diag = {"per_rank": every, "paths": sorted(set(e["path"] for e in every)),
                   "replicas_bit_identical": (len(set(digests)) == 1) if (args.sync_every == 1 and digests[0] is not None) else None,
                   "exposed_allreduce_us_per_minibatch": ar_us,
                   "allreduce_bytes": 4 * int(sum(n.get_params().size for n in nets_[:2])) if nets_ else None,
                   "note": "replicas_bit_identical compares sha256 of (actor, critic, targets) parameters over the ranks after the timed region; "
                           "exposed_allreduce: HIP events around ncclAllReduce in an eager pass of the same step (nothing overlaps it in the default mode)"}
        parallelism += "; the default-mode step ran as: %s" % ", ".join(dp_diag["paths"])
        if dp_diag["replicas_bit_identical"] is False:
            sys.stderr.write("bench.py: the data-parallel replicas DIVERGED (parameter digests differ over the ranks): %r\n" % (digests,))

    # ---- --force-dp at world size 1: the data-parallel graph against the single learner's fused graph IN THIS PROCESS, in alternating
    # blocks (the two bench lines of a profile run come from two processes on a chip whose clock state drifts by a few per cent: paired
    # runs gave 0.957 ... 0.990; the kernels differ by one sumsq launch, rocprof: 1782 vs 1770 us per five minibatches)
    dp_vs_fused = None
    if use_dp and world == 1 and learner is not None and args.sync_every == 1 and not args.overlap and hasattr(agent, "train_step"):
        agent.train_step(B, BATCHES_PER_STEP); agent.train_step(B, BATCHES_PER_STEP)      # (captures the fused graph)
        blocks, bg = 8, max(20, groups)
        t_dp, t_fused = [], []
        for blk in range(blocks):
            for which in ((0, 1) if blk % 2 == 0 else (1, 0)):
                ctx.sync()
                tb = time.perf_counter()
                for _ in range(bg):
                    if which == 0:
                        learner.train_step(BATCHES_PER_STEP)
                    else:
                        agent.train_step(B, BATCHES_PER_STEP)
                ctx.sync()
                (t_dp if which == 0 else t_fused).append(time.perf_counter() - tb)
        n_blk = bg * BATCHES_PER_STEP
        dp_vs_fused = {"dp_steps_per_sec": round(n_blk * blocks / sum(t_dp), 1), "fused_steps_per_sec": round(n_blk * blocks / sum(t_fused), 1),
                       "ratio": round(sum(t_fused) / sum(t_dp), 4), "ratio_per_block_pair": [round(f / d, 4) for d, f in zip(t_dp, t_fused)],
                       "blocks": blocks, "minibatches_per_block": n_blk,
                       "how": "alternating blocks of the two hipGraphs on the same trainer, replay and stream, order swapped every pair"}

    # ---- per-kernel HIP-event pass (rank 0 reports) -> roofline of the dominant kernel, per-layer table
    ctx.prof_reset()
    ctx.prof_enable(True)
    pgroups = max(1, args.profile_steps // BATCHES_PER_STEP)
    for _ in range(pgroups):
        agent.train_step(B, BATCHES_PER_STEP)
    ctx.sync()
    ctx.prof_enable(False)
    prof = ctx.prof_read()
    pm = pgroups * BATCHES_PER_STEP

    F, Bk, per_layer = conv_macs(shape)
    nfwd, nbwd = (4, 2) if kind == "ddpg" else (2, 1)      # trunks per minibatch (NAF shared trunk: value on s1, target on s2)
    conv_flops_step = 2.0 * B * (nfwd * F + nbwd * Bk)
    gf = lambda macs, nets: 2.0 * B * macs * nets          # algorithmic FLOPs of one launch over the whole minibatch

    # One row per conv launch of the fused step.  A row may be served by several profiled kernel names (the same kernel with
    # and without the next minibatch's sample pass riding along: conv1_dw_f16 / conv1_dw_gather): their times and launches
    # are merged, and the FLOPs per launch are a constant of the launch (2 * B * MACs * networks), never derived from counts.
    # parts: [(algorithmic flops per launch, pipe)] -- a paired launch (dW + dX) is bounded by the sum of its parts' times.
    def row(label, names, parts):
        ms = sum(prof.get(n, (0.0, 0))[0] for n in names)
        n = sum(prof.get(n, (0.0, 0))[1] for n in names)
        if n == 0:
            return None
        avg_us = 1e3 * ms / n
        flops = sum(p[0] for p in parts)
        bound_us = sum(1e6 * p[0] / (PIPES[p[1]][0] * 1e12) for p in parts)
        return {"layer": label, "kernels": [k for k in names if k in prof], "launches_per_step": round(n / pm, 2),
                "gflop_per_launch": round(flops / 1e9, 3), "avg_launch_us": round(avg_us, 2), "us_per_step": round(1e3 * ms / pm, 2),
                "pipe": "+".join(p[1] for p in parts), "achieved_tflops": round(flops / (avg_us * 1e-6) / 1e12, 2),
                "bound_us": round(bound_us, 2), "frac": round(bound_us / avg_us, 4)}
    L1, L2, L3 = per_layer
    nb = nbwd
    fast = "conv1_fwd_f16" in prof and not os.environ.get("CPP_CONV_DXRS") and not os.environ.get("CPP_CONV_DWRS")      # (the f16 / bf16 pipes are on)
    bn = bool(getattr(args, "use_batch_norm", False))
    rs2 = fast and not bn and shape[1] // 2 in (32, 64) and shape[0] % 4 == 0
    rs3 = fast and not bn and shape[1] // 4 in (16, 32, 64) and shape[0] % 8 == 0 and shape[0] // 4 >= 8
    rows = [r for r in (
        row("conv1 forward", ["conv1_fwd_f16"], [(gf(L1, nfwd), F16_PIPE)]),
        row("conv1 forward (f32 MFMA)", ["conv1_fwd"], [(gf(L1, nfwd), "f32")]),
        row("conv1 dW", ["conv1_dw_f16", "conv1_dw_gather"], [(gf(L1, nb), F16_PIPE)]),
        row("conv1 dW (f32 MFMA)", ["conv1_dw"], [(gf(L1, nb), "f32")]),
        # (when conv3 + pool3 ride as the tail of conv2's workgroups there is no conv3_fwd launch: its FLOPs belong to this row)
        row("conv2 forward" + ("" if "conv3_fwd" in prof else " + conv3 forward"), ["conv2_fwd"],
            [(gf(L2, nfwd), B16_PIPE if "conv1_fwd_f16" in prof else "f32")] + ([] if "conv3_fwd" in prof else [(gf(L3, nfwd), "f32")])),
        # (round 5: conv2's dX -- and conv3's dW / dX -- run on the bf16 pipes' row-streaming bodies, conv_dx_rs.h / conv_dw_rs.h, where
        # the rows are 32 or 64 pixels wide (conv3's pair launch: 16 too); other widths keep the f32-input kernels)
        row("conv2 dW + dX", ["conv2_bwd"], [(gf(L2, nb), B16_PIPE), (gf(L2, nb), B16_PIPE if rs2 else "f32")]),
        # (conv2's dW on its own launch -- cfg5's 64 x 64 conv2 has no pair instance -- runs conv_dwb16_kernel, bf16 pieces, whenever conv1 /
        # conv2 are on the f16 / bf16 pipes; round 4's line priced it against the f32 pipe)
        row("conv2 dW", ["conv2_dw"], [(gf(L2, nb), B16_PIPE if "conv1_fwd_f16" in prof else "f32")]),
        row("conv2 dX", ["conv2_dx"], [(gf(L2, nb), B16_PIPE if rs2 else "f32")]),
        row("conv3 forward", ["conv3_fwd"], [(gf(L3, nfwd), B16_PIPE if rs3 and shape[1] // 4 >= 32 else "f32")]),      # (conv_fw_rs.h at rows of 32 / 64 pixels)
        row("conv3 dW + dX", ["conv3_bwd"], [(gf(L3, nb), B16_PIPE if rs3 else "f32"), (gf(L3, nb), B16_PIPE if rs3 else "f32")]),
        row("conv3 dW", ["conv3_dw"], [(gf(L3, nb), B16_PIPE if rs3 and shape[1] // 4 >= 32 else "f32")]),
        row("conv3 dX", ["conv3_dx"], [(gf(L3, nb), B16_PIPE if rs3 and shape[1] // 4 >= 32 else "f32")])) if r]
    for r in rows:
        assert r["frac"] <= 1.0, "roofline accounting error: %r" % (r,)
    mapped = {k for r in rows for k in r["kernels"]}
    # ("conv1_image": conv1's operand images as a launch of their own -- the first minibatch of an outer step; the others' ride in the
    # optimiser's launch.  No convolution FLOPs: it counts under non_conv_us_per_step)
    unmapped_conv = sorted(k for k in prof if k.startswith("conv") and k not in mapped and k != "conv1_image")
    # (reported, not asserted: a kernel id without a `layers` row understates conv_bound_frac_whole_step, it does not invalidate the line)
    conv_us = sum(r["us_per_step"] for r in rows)
    bound_us_step = sum(r["bound_us"] * r["launches_per_step"] for r in rows)
    kernels = {k: {"ms_per_step": round(v[0] / pm, 4), "launches_per_step": round(v[1] / pm, 2)}
               for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    other_us = 1e3 * sum(v[0] for v in prof.values()) / pm - conv_us

    dom = max(rows, key=lambda r: r["us_per_step"]) if rows else None
    roofline = None
    if dom:
        pipe = dom["pipe"].split("+")[0]
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; they come from separate
        # rocprofv3 --pmc passes of this same command (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), committed under profiles/
        traffic, traffic_src = None, None
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            try:
                with open(os.path.join(ROOT, "profiles", "%s_pmc.json" % rnd)) as f:
                    pmc = json.load(f)
                if args.workload == "cfg3" and dom["kernels"][0] in pmc["kernels"]:
                    traffic = pmc["kernels"][dom["kernels"][0]]["hbm_bytes_per_launch"]
                    traffic_src = "profiles/%s_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, bytes per launch)" % rnd
                    break
            except Exception:
                pass
        roofline = {"bound": "mfma", "kernel": dom["kernels"][0], "layer": dom["layer"], "achieved": dom["achieved_tflops"],
                    "peak": round(PIPES[pipe][0], 1), "unit": "TFLOP/s", "frac": dom["frac"], "peak_basis": PIPES[pipe][1],
                    # SURVEY 8(d) named the f32-input MFMA peak as the bounding roofline before conv1 moved to exact f16 pieces: the
                    # same achieved rate against THAT denominator (> 1 is possible and means nothing but "not an f32-MFMA kernel")
                    "frac_vs_f32_mfma": round(dom["achieved_tflops"] / PEAK_F32_MFMA_TFLOPS, 4),
                    "flops_per_launch": dom["gflop_per_launch"] * 1e9, "avg_launch_ms": round(dom["avg_launch_us"] / 1e3, 5),
                    "launches_per_step": dom["launches_per_step"], "traffic": traffic, "traffic_source": traffic_src}
        if pipe == F16_PIPE:
            # informational, next to (never instead of) `peak` / `frac`: what the f16 pipes sustain on operands that toggle like image
            # data -- the chip clocks to its power budget (profiles/diag/mfma_rate_probe.hip, profiles/r02_mfma_rate_probe.txt)
            npieces = 3 if EXACT_PRODUCTS else 2
            sus = SUSTAINED_F16_MFMA_TFLOPS_RANDOM_OPERANDS / float(npieces)
            roofline["sustained"] = {"peak": round(sus, 1), "frac": round(dom["achieved_tflops"] / sus, 4),
                                     "basis": "v_mfma_f32_16x16x32_f16 issued back to back by every SIMD with pseudo-random operands: 1914-2056 TFLOP/s at "
                                              "1.9-2.0 GHz on two boxes, %d taken (2354-2453 at 2.4 GHz with constant operands; 32x32x16: 1595-1719), "
                                              "/ %d pieces; profiles/r02_mfma_rate_probe.txt" % (SUSTAINED_F16_MFMA_TFLOPS_RANDOM_OPERANDS, npieces)}

    # the fully connected heads (reported separately, never in a roofline numerator): per image and minibatch the actor's
    # layers run forward once and backward twice (dX, dW), the critic's forward twice and backward twice, each target network forward once
    mlp_gflop = None
    if kind == "ddpg":
        hp, wp = shape[0], shape[1]
        for _ in CONV_DEFS:
            hp, wp = hp // 2, wp // 2
        flat = hp * wp * CONV_DEFS[-1][1]
        actor_macs = flat * 100 + 100 * 100 + 100 * 50 + 50 * 2            # ddpg_cartpole.py:95-100 ("100,100,50", action_dim 2)
        critic_macs = flat * 200 + 200 * 50 + (50 + 2) * 50 + 50 * 1       # :166-184
        mlp_gflop = round(2.0 * B * (4 * actor_macs + 5 * critic_macs) / 1e9, 3)
    ch = int(np.prod(shape[2:]))
    out = {
        "metric": "%s training steps/sec, %dx%dx%d pixel obs, batch=%d" % ("DDPG" if kind == "ddpg" else "NAF", shape[0], shape[1], ch, B),
        "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": steps, "warmup": wgroups * BATCHES_PER_STEP,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32-acc/f16x3,bf16x9" if EXACT_PRODUCTS else "f32-acc/f16x2,bf16x6", "data": "synthetic",
        "timed_region_ms": round(1e3 * elapsed, 2), "warmup_steps_run": warmup_steps_run,
        "dtype_note": ("f32 accumulation everywhere; conv1 multiplies exact f16 operands (the replay store's pixels x three-piece f16 splits "
                       "of the f32 weights / gradients), conv2 forward / dW / dX (round 5; and conv3's dW / dX at rows of 16+ pixels) "
                       "three-piece bf16 splits of both f32 operands with all nine "
                       "products: every product exact (--precision exact)" if EXACT_PRODUCTS else
                       "f32 accumulation everywhere; conv1 multiplies the replay store's f16 pixels (exact) by a two-piece f16 split of the "
                       "f32 weight / gradient (within one f32 ulp of it), conv2 forward / dW / dX (round 5; and conv3's dW / dX at rows of "
                       "16+ pixels) three-piece bf16 splits of both f32 operands with "
                       "the six largest piece products (the dropped three are at most half an f32 ulp of the product): measured as close to "
                       "the float64 oracle as the exact-product build (`control_exact_products`) and closer than the f32-input MFMA kernels "
                       "(`control`); DESIGN.md 4, 6"),
        "config": {"workload": "%s: %s pixel obs %dx%dx%d, batch=%d per GPU, replay %d rows/GPU in HBM (%s), "
                               "target soft-update every %d minibatches%s" % (
                                   args.workload, "DDPG" if kind == "ddpg" else "NAF (shared trunk, Momentum)", shape[0], shape[1], ch, B,
                                   replay_rows, args.replay_store, BATCHES_PER_STEP, ", --use-batch-norm" if args.use_batch_norm else ""),
                   "parallelism": parallelism,
                   "global_steps_per_sec": round(steps / elapsed, 3),
                   "per_rank_steps_per_sec": per_rank,
                   "dp_vs_fused_same_process": dp_vs_fused,
                   "data_parallel": dp_diag,
                   "conv_gflop_per_step": round(conv_flops_step / 1e9, 3),
                   "conv_gflop_per_step_as_the_reference_executes_it": (round(2.0 * B * (5 * F + 2 * Bk) / 1e9, 3) if kind == "ddpg" else None),
                   "mlp_gflop_per_step": mlp_gflop,
                   "conv_bound_frac_whole_step": round(bound_us_step / (1e3 * ms_per_step), 4),
                   "conv_bound_frac_basis": "sum over the conv launches of (algorithmic FLOPs / peak of the pipe the launch runs on) / measured "
                                            "step time; pipes: " + "; ".join("%s = %.1f TFLOP/s (%s)" % (k, v[0], v[1]) for k, v in PIPES.items())},
        "roofline": roofline,
        "layers": rows,
        "non_conv_us_per_step": round(other_us, 2),
        "conv_kernels_without_a_layers_row": unmapped_conv,
        "kernels": kernels,
    }
    side = rank == 0 and world == 1 and not args.quick
    if side and not args.no_cpu_baseline and kind == "ddpg":
        out["cpu_baseline"] = cpu_baseline_torch(shape, B)
        out["cpu_baseline_numpy"] = cpu_baseline_numpy(shape, B)
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    else:
        out["cpu_baseline"] = None
    if learner is not None:
        learner.close()
    agent.close()
    ctx.close()
    if side and args.workload == "cfg3" and not args.use_batch_norm:
        # the f32 twin of the headline: conv1 / conv2 on the f32-input MFMA kernels (ablation build of the library)
        c = sub_bench(["--steps", "100", "--warmup", "10", "--workload", args.workload],
                      env={"CARTPOLEPP_ABLATION": "1", "CPP_CONV_K16": "0", "CPP_CONV_B16": "0"})
        out["control"] = {"what": "same step, conv1 and conv2 on the f32-input MFMA kernels (v_mfma_f32_16x16x4_f32 only)",
                          "switches": "CARTPOLEPP_ABLATION=1 CPP_CONV_K16=0 CPP_CONV_B16=0 (libcartpolepp_hip_ablation.so)",
                          "value": c.get("value"), "unit": "steps/s", "ms_per_step": c.get("ms_per_step"),
                          "layers": [{k: r[k] for k in ("layer", "avg_launch_us", "pipe", "frac")} for r in c.get("layers", [])],
                          "error": c.get("error")}
        # the exact-product twin: three f16 pieces / nine bf16 products (what rounds 1-2 shipped)
        c = sub_bench(["--steps", "100", "--warmup", "10", "--workload", args.workload, "--precision", "exact"])
        out["control_exact_products"] = {
            "what": "same step, same library, cpp_ctx_set_precision(CPP_PRECISION_EXACT): three f16 pieces of conv1's f32 operand, all nine bf16 "
                    "piece products in conv2 (every product exact)", "switches": "bench.py --precision exact",
            "value": c.get("value"), "unit": "steps/s", "ms_per_step": c.get("ms_per_step"),
            "layers": [{k: r[k] for k in ("layer", "avg_launch_us", "pipe", "frac")} for r in c.get("layers", [])], "error": c.get("error")}
        extra = {}
        for wl, st in (("cfg2", 100), ("cfg4", 100), ("cfg5", 30)):
            e = sub_bench(["--steps", str(st), "--warmup", "10", "--workload", wl])
            extra[wl] = {"metric": e.get("metric"), "value": e.get("value"), "unit": "steps/s", "ms_per_step": e.get("ms_per_step"),
                         "steps": e.get("steps"), "workload": (e.get("config") or {}).get("workload"),
                         "roofline_frac": (e.get("roofline") or {}).get("frac"), "error": e.get("error")}
        e = sub_bench(["--steps", "100", "--warmup", "10", "--workload", args.workload, "--batches-per-step", "1"])
        extra["cfg3_target_update_every_minibatch"] = {
            "what": "the same workload with --batches-per-step 1: both target soft updates after EVERY minibatch (SURVEY 8d's second variant); one "
                    "train-step call, one hipGraph replay and one stand-alone sample pass per minibatch",
            "value": e.get("value"), "unit": "steps/s", "ms_per_step": e.get("ms_per_step"), "steps": e.get("steps"), "error": e.get("error")}
        out["extra"] = extra
    sys.stdout.flush()
    try:                                   # RCCL's banner sits in the C stdio buffer: flush it to the redirected fd first
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
