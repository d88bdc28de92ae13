"""brake_check kernel: how many of a wavefront's sample steps evaluate, and how many lanes take part (GPU box; library built with
-DSMR_PROFILE_ROWS).     python scripts/study/brake_hist.py [CONFIG]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
import safemotionsrisk_amd as smr  # noqa: E402
from safemotionsrisk_amd import lib as _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "reach_brake"
B = 65536
cfg, kw, _, _ = bench._env_config(name)
kw = {k: v for k, v in kw.items() if k != "risk_networks"}
env = smr.VecSafeMotionsEnv(B, seed=0, auto_reset=True, **kw, **cfg)
L = _lib.load()
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = [torch.rand((B, 7), device="cuda", generator=g) * 2 - 1 for _ in range(8)]
for i in range(20):
    env.step(acts[i % 8])
env.synchronize()
out = (C.c_ulonglong * 64)()
assert L.smr_debug_job_hist(out, 1) == 0
steps = 20
for i in range(steps):
    env.step(acts[i % 8])
env.synchronize()
assert L.smr_debug_job_hist(out, 0) == 0
h = np.array(out[:], dtype=np.float64) / steps
waves, lanes, segs = h[51], h[52], h[53]
print("waves per step %.0f, lanes %.0f, segments per lane %.2f, wave-level sample steps per wave %.1f" % (waves, lanes, segs / max(lanes, 1), h[50] / max(waves, 1)))
tot_p = tot_l = 0
for r in range(24):
    p, l = h[2 * r], h[2 * r + 1]
    tot_p += p; tot_l += l
    if p:
        print("row %2d: evaluation passes per wave %.1f, lanes per pass %.1f, evaluations per lane %.2f" % (r, p / (waves / 24), l / p, l / (lanes / 24)))
print("all rows: passes per wave %.1f, lanes per pass %.1f, evaluations per lane %.2f" % (tot_p / waves, tot_l / max(tot_p, 1), tot_l / lanes))
