"""Iterations of the planet jobs against what the enumeration knows about them (GPU box; library built with -DSMR_PROFILE_ROWS):
rows = cutoff - key bound in key steps of 2.5 mm (end-of-step jobs), last row the contact items, columns = runs settled after 1, 2, 3-4, 5-8, 9+ iterations,
jobs with a long scan, jobs.     python scripts/study/job_hist.py CONFIG"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
import safemotionsrisk_amd as smr  # noqa: E402
from safemotionsrisk_amd import lib as _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "space"
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
h = np.array(out[:], dtype=np.float64).reshape(8, 8) / steps / B
print("per env and step; columns: its 1, 2, 3-4, 5-8, 9+ | long scan | jobs")
names = ["slack < 1 key step", "< 2", "< 4", "< 8", "< 16", "< 32", "more", "contact items"]
for r in range(8):
    print("%-20s" % names[r] + " ".join("%6.3f" % x for x in h[r, :7]))
