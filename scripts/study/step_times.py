"""Per-step wall time of the first steps after reset() (GPU box): where the driver's short run (5 warm-up + 20 timed steps) loses against the
steady state - every env ends its first episode at step 20 together.   python scripts/study/step_times.py [config] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
import safemotionsrisk_amd as smr  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "space"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cfg, kw, _, _ = bench._env_config(name)
kw = {k: v for k, v in kw.items() if k != "risk_networks"}
B = 65536
env = smr.VecSafeMotionsEnv(B, seed=0, auto_reset=True, **kw, **cfg)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = [torch.rand((B, 7), device="cuda", generator=g) * 2 - 1 for _ in range(8)]
env.synchronize()
out = []
for i in range(steps):
    t0 = time.perf_counter()
    o, r, d, info = env.step(acts[i % 8])
    env.synchronize()
    out.append((time.perf_counter() - t0) * 1e3)
    nd = int(d.sum().item())
    print("step %2d: %.3f ms, %5d envs done" % (i + 1, out[-1], nd), flush=True)
