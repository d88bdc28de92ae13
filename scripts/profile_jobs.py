"""The schedule of the jobs kernel's wavefronts in one step (GPU box; library built with SMR_HIPCC_FLAGS=-DSMR_PROFILE_ROWS):
when they start, when their tables are staged, when they end, and what the last ones to end did.

    python scripts/profile_jobs.py CONFIG [ENVS]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
import safemotionsrisk_amd as smr  # noqa: E402
from safemotionsrisk_amd import lib as _lib  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "space"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    cfg, kw, _, _ = bench._env_config(name)
    kw = {k: v for k, v in kw.items() if k != "risk_networks"}
    env = smr.VecSafeMotionsEnv(B, seed=0, auto_reset=True, **kw, **cfg)
    L = _lib.load()
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = [torch.rand((B, 7), device="cuda", generator=g) * 2 - 1 for _ in range(8)]
    for i in range(30):
        env.step(acts[i % 8])
    env.synchronize()
    out = (C.c_ulonglong * (2 * 16384))()
    rec = (C.c_uint * (8 * 65536))()
    for rep in range(3):
        assert L.smr_debug_block_times(None, None, 1) == 0
        env.step(acts[rep])
        env.synchronize()
        assert L.smr_debug_block_times(out, rec, 0) == 0
        t = np.array(out[:], dtype=np.float64).reshape(-1, 2)[:2048] * 10e-3      # 100 MHz ticks -> microseconds
        r = np.array(rec[:], dtype=np.float64).reshape(-1, 8)[:2048]
        ok = t[:, 1] > 0
        t, r = t[ok], r[ok]
        t0 = t[:, 0].min()
        start, end, staged = t[:, 0] - t0, t[:, 1] - t0, t[:, 0] - t0 + r[:, 3] * 10e-3
        print("step %d: %d wavefronts, kernel span %.1f us; starts: median %.1f p90 %.1f max %.1f; staged: median %.1f max %.1f; ends: median %.1f p90 %.1f p99 %.1f max %.1f"
              % (rep, len(t), end.max(), np.median(start), np.percentile(start, 90), start.max(), np.median(staged), staged.max(),
                 np.median(end), np.percentile(end, 90), np.percentile(end, 99), end.max()))
        order = np.argsort(-end)[:12]
        for w in order:
            print("    wave %4d (block %3d): start %6.1f staged %6.1f end %6.1f us, refills %d, trips %d, longest run %d iterations"
                  % (w, w // 8, start[w], staged[w], end[w], r[w, 0], r[w, 1], r[w, 2]))
        for lo, hi in ((1, 1), (2, 2), (3, 9)):
            m = (r[:, 0] >= lo) & (r[:, 0] <= hi)
            if m.any():
                print("    wavefronts with %d..%d refills: %4d, end median %.1f max %.1f us, trips median %.0f" % (lo, hi, m.sum(), np.median(end[m]), end[m].max(), np.median(r[m, 1])))


if __name__ == "__main__":
    main()
