"""The schedule of the collide kernel's blocks in one step (GPU box; library built with SMR_HIPCC_FLAGS=-DSMR_PROFILE_ROWS): when the
blocks of each grid row start and end, how many run at once, and what the wavefronts of the slowest blocks spent their cycles on.

    python scripts/profile_blocks.py CONFIG      (SMR_SYNC_RESET=1 in the environment: without the reset kernels beside it)
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
    cfg, kw, _, _ = bench._env_config(name)
    kw = {k: v for k, v in kw.items() if k != "risk_networks"}
    B = 65536
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
    assert L.smr_debug_block_times(None, None, 1) == 0
    env.step(acts[0])
    env.synchronize()
    assert L.smr_debug_block_times(out, rec, 0) == 0
    t = np.array(out[:], dtype=np.float64).reshape(-1, 2)
    rec = np.array(rec[:], dtype=np.float64).reshape(-1, 8)
    s = env.scene.struct
    rows = s.n_bodies + s.n_self_pairs
    nbx = (B + 255) // 256
    t = t[:rows * nbx] * 10e-3          # 100 MHz ticks -> microseconds
    t -= t[:, 0].min()
    dur = t[:, 1] - t[:, 0]
    print("config %s: %d rows x %d blocks, kernel span %.1f us" % (name, rows, nbx, t[:, 1].max()))
    for r in range(rows):
        tr = t[r * nbx:(r + 1) * nbx]
        d = tr[:, 1] - tr[:, 0]
        print("  row %2d: starts %7.1f .. %7.1f us, ends %7.1f .. %7.1f us, block duration mean %6.1f median %6.1f p90 %6.1f max %6.1f us"
              % (r, tr[:, 0].min(), tr[:, 0].max(), tr[:, 1].min(), tr[:, 1].max(), d.mean(), np.median(d), np.percentile(d, 90), d.max()))
    edges = np.linspace(0, t[:, 1].max(), 17)
    for a, b in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (a + b)
        print("  t = %7.1f us: %5d blocks running" % (mid, int(((t[:, 0] <= mid) & (t[:, 1] > mid)).sum())))
    heavy = rec[:4 * 5 * nbx]            # the wavefronts of the first five rows
    print("  contact items per wavefront (first five rows): mean %.1f, p90 %.0f, max %.0f; of them past the bound test: mean %.1f, max %.0f"
          % (heavy[:, 5].mean(), np.percentile(heavy[:, 5], 90), heavy[:, 5].max(), heavy[:, 6].mean(), heavy[:, 6].max()))
    print("  per wavefront of those rows, mean us: static %.1f, keys %.1f, planet runs %.1f, items %.1f" % tuple(heavy[:, i].mean() * 16 / 2400.0 for i in (0, 1, 2, 4)))
    k = 16 / 2400.0                      # record unit (16 cycles) -> microseconds at 2.4 GHz
    print("  wavefront records (us at 2.4 GHz: static, planet keys, planet runs, contact items; trips of the planet-run loop):")
    order = np.argsort(-dur)
    for label, ids in (("slowest", order[:6]), ("median", order[len(order) // 2:len(order) // 2 + 3])):
        for b in ids:
            w = rec[4 * b:4 * b + 4]
            print("    %s block %5d (row %d) %6.1f us: " % (label, b, b // nbx, dur[b]) +
                  "  ".join("[%5.1f %5.1f %6.1f %6.1f; %3d]" % (x[0] * k, x[1] * k, x[2] * k, x[4] * k, x[3]) for x in w))


if __name__ == "__main__":
    main()
