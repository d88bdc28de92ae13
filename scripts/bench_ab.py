"""A/B of one scene under an environment switch of the library (GPU box): python scripts/bench_ab.py VAR [config]"""
import os, subprocess, sys
var = sys.argv[1]; cfg = sys.argv[2] if len(sys.argv) > 2 else "space"
code = r'''
import sys, time; sys.path.insert(0, ".")
import torch
from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG, BALL_ENV_CONFIG, REACH_ENV_CONFIG
cfg = {"space": SPACE_ENV_CONFIG, "bm": dict(SPACE_ENV_CONFIG, ball_machine_mode=True), "ball": BALL_ENV_CONFIG, "reach": REACH_ENV_CONFIG}[sys.argv[1]]
B = int(__import__('os').environ.get('SMR_BENCH_B', 65536))
e = VecSafeMotionsEnv(B, seed=0, auto_reset=True, return_info=False, **cfg)
e.reset(); act = torch.rand((B, 7), device="cuda") * 2 - 1
for _ in range(10): e.step(act)
e.synchronize(); e.set_profiling(True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): e.step(act)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
pr = e.get_profile(); e.close()
print("%.1fM env-steps/s" % (B * 200 / dt / 1e6), {k: round(v["ms"] / max(v["launches"], 1), 3) for k, v in pr.items()}, flush=True)
'''
for val in ("0", "1"):
    env = dict(os.environ, **{var: val})
    print(var + "=" + val, subprocess.run([sys.executable, "-c", code, cfg], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
