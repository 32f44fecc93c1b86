import sys
from tests.helpers import fused_step_against_f64_oracle
for seed in (6, 7, 8):
    rep = fused_step_against_f64_oracle((128, 128, 3, 2, 5), 96, rows=300, graph=True, seed=seed, report_only=True, f32_twin=True)
    print("REP", seed, {k: (round(v, 9) if isinstance(v, float) else v) for k, v in rep.items() if k in ("err_q", "err_td", "q_scale", "err_pool1", "err_pool2", "err_pool3", "rel_actor_grads", "rel_critic_grads", "f32_err_td", "f32_err_q")})
