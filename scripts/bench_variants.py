"""env-steps/s and per-kernel times for the plain robot and the ball-machine variant (GPU box)."""
import sys, time; sys.path.insert(0, ".")
import torch
from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG, BALL_ENV_CONFIG, REACH_ENV_CONFIG
for name, cfg in (("space", SPACE_ENV_CONFIG), ("space+ball machine", dict(SPACE_ENV_CONFIG, ball_machine_mode=True)),
                  ("ball", BALL_ENV_CONFIG), ("ball+ball machine", dict(BALL_ENV_CONFIG, ball_machine_mode=True)),
                  ("reaching task", REACH_ENV_CONFIG),
                  ("space, N=24 contact samples", dict(SPACE_ENV_CONFIG, collision_checks_per_step=24)),
                  ("space, N=1", dict(SPACE_ENV_CONFIG, collision_checks_per_step=1))):
    B = 65536
    cfg = dict(cfg); extra = {k: cfg.pop(k) for k in ("collision_checks_per_step",) if k in cfg}
    e = VecSafeMotionsEnv(B, seed=0, auto_reset=True, return_info=False, **extra, **cfg)
    e.reset(); act = torch.rand((B, 7), device="cuda") * 2 - 1
    for _ in range(10): e.step(act)
    e.synchronize(); e.set_profiling(True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): e.step(act)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = e.get_profile(); e.close()
    print("%-20s %.1fM env-steps/s" % (name, B * 100 / dt / 1e6), {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in pr.items()}, flush=True)
