"""bench.py prints ONE JSON line with the contract's keys (the driver parses it): run the quick variant on the GPU and check the
line's shape.  (The full variant adds cpu_baseline / control / extra and takes minutes; profiles/r02_kernel_stats.md keeps one.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_quick_line_has_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "5", "--quick"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                               # exactly one line on stdout: library banners go to stderr
    d = json.loads(lines[0])
    assert d["metric"] == "DDPG training steps/sec, 64x64x18 pixel obs, batch=256" and d["unit"] == "steps/s"
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 5 and d["warmup_steps_run"] >= 200
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32-acc/f16x2,bf16x6" and d["data"] == "synthetic"      # (f32 accumulation; operand pieces of the default precision)
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-3          # N = 1: value = 1 / step time
    assert d["config"]["workload"].startswith("cfg3") and "model" not in d["config"]
    assert d["config"]["conv_gflop_per_step"] == pytest.approx(68.053, abs=1e-3)
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["kernel"] == "conv1_fwd_f16"
    assert rf["peak"] == pytest.approx(2500.0 / 2, abs=0.1) and 0.0 < rf["frac"] < 1.0             # two f16 pieces per weight (conv_k16.h)
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    assert rf["flops_per_launch"] == pytest.approx(2 * 256 * 4096 * 4500 * 4, rel=1e-4)          # four networks' conv1, algorithmic (GFLOP rounded to 3 places)
    assert rf["achieved"] == pytest.approx(rf["flops_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12, rel=1e-3)
    assert rf["traffic"] is None or rf["traffic"] > 1e8
    for row in d["layers"]:
        assert 0.0 < row["frac"] <= 1.0, row
    assert d["cpu_baseline"] is None                            # --quick


def test_bench_gpus_2_as_a_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment is the driver's form: it must start its N ranks itself
    (torch.distributed.run), and rank 0's ONE JSON line must say n_gpus = N with the whole-job rate.  On a one-GPU box the two ranks
    share GPU 0 through the gloo diagnostic backend (RCCL refuses two ranks on one device): plumbing, not a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--diag-backend", "gloo", "--steps", "10",
                        "--warmup", "5", "--quick"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "weak"
    per_rank = d["config"]["per_rank_steps_per_sec"]
    assert len(per_rank) == 2 and all(x > 0 for x in per_rank)
    assert d["value"] == pytest.approx(2 * d["config"]["global_steps_per_sec"], rel=1e-3)
    assert "dp2" in d["config"]["parallelism"]


def test_bench_force_dp_compares_the_two_graphs_in_one_process():
    """the data-parallel step (ncclAllReduce, norm and optimiser inside ONE hipGraph per outer step) against the single
    learner's fused graph at world size 1 -- alternating blocks in the SAME process (two processes on one chip differ by its clock state).
    The bar is the verdict's 0.98; measured 0.992-0.994 (profiles/r04_configs.md cfg3_dp1)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--quick", "--force-dp"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert "dp1" in d["config"]["parallelism"] and "RCCL" in d["config"]["parallelism"]
    ab = d["config"]["dp_vs_fused_same_process"]
    assert ab["blocks"] == 8 and len(ab["ratio_per_block_pair"]) == 8
    assert ab["ratio"] >= 0.98, ab
    assert d["value"] >= 0.97 * ab["dp_steps_per_sec"], (d["value"], ab)      # the timed region behind the barrier is not a slower one
    # what a first N > 1 run must report by itself -- the step's form per rank, replica bit-identity, the exposed all-reduce
    dp = d["config"]["data_parallel"]
    assert dp["paths"] == ["hipgraph"] and "hipgraph" in d["config"]["parallelism"], dp
    assert len(dp["per_rank"]) == 1 and dp["per_rank"][0]["rank"] == 0 and dp["per_rank"][0]["reason"] == "", dp
    assert dp["replicas_bit_identical"] is True and len(dp["per_rank"][0]["params_sha256_16"]) == 16, dp
    assert dp["exposed_allreduce_us_per_minibatch"] is not None and 0.5 < dp["exposed_allreduce_us_per_minibatch"] < 200.0, dp
    assert dp["allreduce_bytes"] == 4 * 236163, dp          # cfg3: actor 87 282 + critic 148 881 floats (0.94 MB)
