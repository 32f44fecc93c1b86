#!/usr/bin/env python
"""A/B builds of one kernel translation unit, timed inside the whole step with bench.py.

  python profiles/ablate_kyo.py build conv_fwd_kyo_l1 NAME=-DFLAG[,-DFLAG2] ...   (here: hipcc cross-compiles)
  python profiles/ablate_kyo.py run KERNEL NAME ...                               (on the GPU box; KERNEL e.g. conv1_fwd)

Each variant is the shipped library with that one object file rebuilt with extra -D flags
(cartpoleplusplus_amd/lib/libcartpolepp_hip_ablate_<NAME>.so, git-ignored; loaded with CARTPOLEPP_ABLATION=ablate_<NAME>).  BASE = the shipped library.
"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cartpoleplusplus_amd", "lib")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]

def build(unit, specs):
    procs = []
    for spec in specs:
        name, flags = spec.split("=", 1)
        obj = "/tmp/abl_%s.o" % name
        procs.append((name, obj, subprocess.Popen(["hipcc"] + FLAGS + flags.split(",") + ["-c", os.path.join(ROOT, "cartpoleplusplus_amd/csrc/%s.hip" % unit), "-o", obj])))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        objs = [o for o in glob.glob(os.path.join(LIB, "obj", "*.o")) if os.path.basename(o) != unit + ".o" and not os.path.basename(o).startswith("abl_")]
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(LIB, "libcartpolepp_hip_ablate_%s.so" % name)] + objs + [obj, "-L/opt/rocm/lib", "-lrccl"])
        print("built", name)

def run(kernel, names):
    for name in ["BASE"] + names:
        env = dict(os.environ)
        if name != "BASE":
            env["CARTPOLEPP_ABLATION"] = "ablate_%s" % name
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--quick", "--steps", "100", "--warmup", "10"],
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        try:
            d = json.loads(out.strip().splitlines()[-1])
            print("%-28s %8.1f steps/s   %s %.4f ms/step" % (name, d["value"], kernel, d["kernels"][kernel]["ms_per_step"]))
        except Exception as e:
            print(name, "FAILED", e, out[-300:])

if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        run(sys.argv[2], sys.argv[3:])
