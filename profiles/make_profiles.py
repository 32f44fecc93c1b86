#!/usr/bin/env python
"""gpurun_out/{prof,pmc}_<round> rocpd databases (written by profiles/run_profiles.sh on the GPU box) ->
profiles/<round>_kernel_stats.md and profiles/<round>_pmc.json.

  python profiles/make_profiles.py r01

HBM bytes follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE in separate passes, both in KB,
FETCH_SIZE doubled for the gfx950 under-count of 16-byte coalesced reads (an upper bound for narrower reads);
values are means per dispatch of one kernel symbol.
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel symbol prefix -> short name used by bench.py
SHORT = [("void conv_fwd_kernel<18, 5, 4, 0, 0>", "conv1_fwd"), ("void conv_dw_kernel<18, 5, 4, 0>", "conv1_dw"),
         ("void conv_fwd_kernel<10, 5, 2, 2, 0>", "conv2_fwd"), ("void conv_fwd_kernel<10, 5, 2, 3, 1>", "conv2_dx"),
         ("void conv_dw_kernel<10, 5, 2, 2>", "conv2_dw"), ("void conv_fwd_kernel<10, 3, 1, 2, 0>", "conv3_fwd"),
         ("void conv_fwd_kernel<10, 3, 1, 3, 1>", "conv3_dx"), ("void conv_dw_kernel<10, 3, 1, 2>", "conv3_dw"),
         ("void gather_stats_kernel<__half>", "gather_stats"), ("gemm_batch_kernel", "gemm_batch"),
         # (ky,o)-column forward kernels: <CIN, KS, XT, IPW, IN_MODE>
         ("void conv_fwd_kyo_kernel<18, 5, 1, 1, 0,", "conv1_fwd"), ("void conv_fwd_kyo_kernel<10, 5, 1, 2, 2,", "conv2_fwd"),
         ("void conv_fwd_kyo_kernel<10, 3, 1, 4, 2,", "conv3_fwd"), ("void conv_dw_kyo_kernel<18, 5", "conv1_dw"),
         ("void conv_fwd_kyo_kernel<10, 5, 1, 2, 3,", "conv2_dx"), ("void conv_dw_kyo_kernel<10, 5", "conv2_dw"),
         # f16 pipes with f32-exact operands
         ("void conv_fwd_k16_kernel<18, 5", "conv1_fwd_f16"), ("void conv_dw16_kernel<18, 5", "conv1_dw_f16"),
         # round 5: conv1 forward as a row-streaming implicit GEMM, weights in registers (conv_rs16.h); its operand images as a launch of their own
         ("void conv_fwd_rs16_kernel<", "conv1_fwd_f16"), ("void conv1_image_kernel<", "conv1_image"), ("opt_apply_kernel", "clip_sgd"), ("conv_dw_reduce_kernel", "dw_reduce"),
         # bf16 pipes (conv2), whole-image conv3 forward, the fused heads, and the launches that carry two kernels
         ("void conv_fwd_k16_kernel<10, 5", "conv2_fwd"), ("void conv_dwb16_kernel<10, 5", "conv2_dw"), ("conv3_img_kernel", "conv3_fwd"),
         ("void ddpg_heads_kernel", "heads"), ("conv3_bwd_pair_kernel", "conv3_bwd"), ("conv2_bwd_pair_kernel", "conv2_bwd"), ("void conv2_bwd_pair_kernel", "conv2_bwd"),
         ("void reduce_gather_kernel<__half>", "reduce_gather"), ("conv1_dw_gather_kernel", "conv1_dw_gather"),
         # round 3: two networks per conv1-dW workgroup (conv_dw16.h NNET = 2)
         ("conv1_dw_pair_gather_kernel", "conv1_dw_gather"), ("void conv_dw16_pair_kernel<18, 5", "conv1_dw_f16"),
         # round 5: the wave-per-unit / row-streaming bodies
         ("conv_dw16_rs_kernel", "conv1_dw_f16"), ("void conv_dw16_rs_kernel<", "conv1_dw_f16"), ("void conv2_bwd_pair_rs_kernel", "conv2_bwd"), ("void conv3_bwd_pair_rs_kernel", "conv3_bwd"),
         ("void conv_dx_rs_kernel<5", "conv2_dx"), ("void conv_dx_rs_kernel<3", "conv3_dx"), ("void conv_dw_rs_kernel<5", "conv2_dw"), ("void conv_dw_rs_kernel<3", "conv3_dw"), ("void conv_fw_rs_kernel<3", "conv3_fwd")]


def short(name):
    for p, s in SHORT:
        if name.startswith(p):
            return s
    return None


def db_of(d, must=True):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*.db"), recursive=True)
    if not f and not must:
        return None
    assert f, "no rocpd database under gpurun_out/%s" % d
    return sqlite3.connect(f[0])


def other_configs(rnd):
    """profiles/<round>_configs.md: bench line + rocprofv3 kernel stats of every other configuration run_profiles.sh ran."""
    names = sorted(os.path.basename(p)[len("bench_%s_" % rnd):-len(".json")]
                   for p in glob.glob(os.path.join(ROOT, "gpurun_out", "bench_%s_*.json" % rnd)))
    if not names:
        return
    with open(os.path.join(ROOT, "profiles", "%s_configs.md" % rnd), "w") as f:
        f.write("# Round %s: the other configurations (profiles/run_profiles.sh), one MI355X\n\n" % rnd[1:])
        f.write("Every number quoted in README.md / DESIGN.md for a configuration other than cfg3 comes from this file.  Per\n"
                "configuration: the `bench.py --quick` JSON line (un-profiled run) and the `rocprofv3 --kernel-trace --stats` kernel\n"
                "table of the same command.\n\n")
        for name in names:
            line = open(os.path.join(ROOT, "gpurun_out", "bench_%s_%s.json" % (rnd, name))).read().strip()
            try:
                d = json.loads(line)
                f.write("## %s -- %s steps/s (%s ms/step)\n\n%s\n\n" % (name, d["value"], d["ms_per_step"], d["config"]["workload"]))
                f.write("parallelism: %s\n\n" % d["config"]["parallelism"])
                f.write("| layer | kernels | us / launch | pipe | fraction of the pipe's bound |\n|---|---|---|---|---|\n")
                for r in d.get("layers", []):
                    f.write("| %s | %s | %s | %s | %s |\n" % (r["layer"], ", ".join(r["kernels"]), r["avg_launch_us"], r["pipe"], r["frac"]))
                f.write("\n```\n%s\n```\n\n" % line)
            except Exception as e:      # noqa: BLE001
                f.write("## %s\n\n(no parsable bench line: %s)\n\n" % (name, e))
            con = db_of("prof_%s_%s" % (rnd, name), must=False)
            if con is not None:
                f.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
                for kname, calls, tot, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()[:14]:
                    f.write("| `%s` | %d | %.1f | %.3f | %.2f |\n" % (kname[:110], calls, tot, avg, pct))
                f.write("\n")


def counters(con):
    out = {}
    if isinstance(con, dict):      # agg.json of profiles/shrink_pmc.py
        rows = [(name, ctr, v[0], v[1]) for name, cs in con.items() for ctr, v in cs.items()]
    else:
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name")
    for name, ctr, n, mean in rows:
        s = short(name)
        if s:
            out.setdefault(s, {})[ctr] = (n, mean)
    return out


def pmc_of(name):
    agg = os.path.join(ROOT, "gpurun_out", name, "agg.json")
    if os.path.exists(agg):
        return json.load(open(agg))
    return db_of(name)


def main(rnd):
    con = db_of("prof_%s" % rnd)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    bench_line = open(os.path.join(ROOT, "gpurun_out", "bench_%s.json" % rnd)).read().strip()
    with open(os.path.join(ROOT, "profiles", "%s_kernel_stats.md" % rnd), "w") as f:
        f.write("# Round %s profile: bench.py (cfg3, 64x64x18, B=256), MI355X\n\n" % rnd[1:])
        f.write("Command (profiles/run_profiles.sh): `rocprofv3 --kernel-trace --stats -d gpurun_out/prof_%s -o k -- python bench.py "
                "--quick --steps 50 --warmup 10 --profile-steps 5`\n\n" % rnd)
        f.write("70 minibatch steps in total (warm-up, hipGraph-replayed timed steps, and the eager HIP-event pass); durations in\n"
                "microseconds, from the rocpd database's `top_kernels` view (profiles/make_profiles.py).  One `conv_fwd_rs16_kernel<18>` (round 5;\n"
                "rounds 2-4: `conv_fwd_k16_kernel<18,5,2,2>`) launch computes conv1 of all four networks of a minibatch (blockIdx.y = network); likewise conv2/conv3 forward;\n"
                "the dW / dX launches carry the actor and the critic together -- conv2's and conv3's dW and dX share one launch each\n"
                "(`conv2_bwd_pair_kernel`, `conv3_bwd_pair_kernel`; round 5: `..._rs_kernel`, both halves on the bf16 pipes' row-streaming bodies), and conv1's dW\n"
                "(round 5: `conv_dw16_rs_kernel`, one wave per (network, 32-pixel column) unit; a template on the channel count from round 6) shares its launch with the next minibatch's\n"
                "sample pass (`conv1_dw_gather_kernel`, from round 3 `conv1_dw_pair_gather_kernel`: one workgroup serves the actor AND the\n"
                "critic, and the sample pass copies the store's per-state sums instead of reading pixels; `conv_dw16_kernel` /\n"
                "`conv_dw16_pair_kernel` alone closes each 5-minibatch graph).\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            f.write("| `%s` | %d | %.1f | %.3f | %.2f |\n" % (name, calls, tot, avg, pct))
        f.write("\n## bench.py JSON line of the same build (un-profiled default run, same box)\n\n```\n%s\n```\n" % bench_line)
    sq, fe, wr = counters(pmc_of("pmc_%s_sq" % rnd)), counters(pmc_of("pmc_%s_fetch" % rnd)), counters(pmc_of("pmc_%s_write" % rnd))
    kernels = {}
    for k in sq:
        e = {"dispatches_sampled": sq[k]["SQ_WAVE_CYCLES"][0]}
        g = lambda c: sq[k].get(c, (0, 0.0))[1]
        e["mfma_busy_cycles_per_launch"] = g("SQ_VALU_MFMA_BUSY_CYCLES")
        e["sq_busy_cycles_per_launch"] = g("SQ_BUSY_CYCLES")
        e["wave_cycles_per_launch"] = g("SQ_WAVE_CYCLES")
        tot = g("SQ_ACTIVE_INST_ANY") + g("SQ_WAIT_ANY") + g("SQ_WAIT_INST_ANY")
        if tot > 0:
            e["wave_time_split"] = {"active_inst": round(g("SQ_ACTIVE_INST_ANY") / tot, 4), "wait_any": round(g("SQ_WAIT_ANY") / tot, 4),
                                    "wait_inst": round(g("SQ_WAIT_INST_ANY") / tot, 4)}
        if g("SQ_LDS_IDX_ACTIVE") > 0:
            e["lds_bank_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
        if k in fe and k in wr:
            e["FETCH_SIZE_KB_raw"] = fe[k]["FETCH_SIZE"][1]
            e["WRITE_SIZE_KB"] = wr[k]["WRITE_SIZE"][1]
            e["hbm_read_bytes_corrected"] = 2.0 * 1024.0 * e["FETCH_SIZE_KB_raw"]
            e["hbm_write_bytes"] = 1024.0 * e["WRITE_SIZE_KB"]
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
        kernels[k] = e
    blob = {"source": "profiles/run_profiles.sh: rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --quick --steps 10 --warmup 5 "
                      "--profile-steps 5 (three separate passes: SQ_*, FETCH_SIZE, WRITE_SIZE)",
            "note": "means per dispatch; FETCH_SIZE (KB) doubled per the gfx950 correction for 16-byte coalesced reads; conv*_fwd "
                    "launches carry four networks, conv*_dw / conv*_dx launches two",
            "kernels": kernels}
    with open(os.path.join(ROOT, "profiles", "%s_pmc.json" % rnd), "w") as f:
        json.dump(blob, f, indent=1)
    print(json.dumps({k: {"hbm": v.get("hbm_bytes_per_launch"), "split": v.get("wave_time_split")} for k, v in kernels.items()}, indent=1))
    other_configs(rnd)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
