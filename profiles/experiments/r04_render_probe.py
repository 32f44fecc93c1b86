"""Round 4, first measurement: the fused cfg3 step on RENDER-like inputs (synthetic_env.RasterCartpole) against the f64 oracle,
per build (release two-piece / exact three-piece / all-f32-MFMA control), with and without a blind camera.
Run: python profiles/experiments/r04_render_probe.py [B] [rows]   (CARTPOLEPP_ABLATION selects the build)"""
import os, sys, json, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import fused_step_against_f64_oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 600
for fill in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("noise", "render", "render-blind", "render-glint")):
    try:
        rep = fused_step_against_f64_oracle((64, 64, 3, 2, 3), B, rows=rows, graph=True, fill=fill, f32_twin=True, report_only=True)
        rep = {k: v for k, v in rep.items() if k not in ("events",)}
        print("PROBE", os.environ.get("CARTPOLEPP_ABLATION", "release"), os.environ.get("CPP_CONV_K16", ""), fill, json.dumps(rep))
    except Exception as e:
        print("PROBE-FAIL", fill, repr(e)[:2000])
        traceback.print_exc()
