"""Where the collide kernel's time goes, per grid row (GPU box; needs a library built with SMR_HIPCC_FLAGS=-DSMR_PROFILE_ROWS):
shader-clock cycles of the kernel's wavefronts summed per row = per self-collision pair / per body.

    SMR_HIPCC_FLAGS=-DSMR_PROFILE_ROWS python -c "from safemotionsrisk_amd import lib; lib.build(force=True)"   (here)
    python scripts/profile_rows.py CONFIG [steps]                                                               (GPU box)
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
    name = sys.argv[1] if len(sys.argv) > 1 else "space_bm"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    cfg, kw, _, _ = bench._env_config(name)
    kw = {k: v for k, v in kw.items() if k != "risk_networks"}
    nenv = int(os.environ.get('SMR_BENCH_B', '65536'))
    env = smr.VecSafeMotionsEnv(nenv, seed=0, auto_reset=True, **kw, **cfg)
    L = _lib.load()
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = [torch.rand((nenv, 7), device="cuda", generator=g) * 2 - 1 for _ in range(8)]
    for i in range(20):
        env.step(acts[i % 8])
    env.synchronize()
    out = (C.c_ulonglong * 128)()
    assert L.smr_debug_row_cycles(out, 1) == 0
    for i in range(steps):
        env.step(acts[i % 8])
    env.synchronize()
    assert L.smr_debug_row_cycles(out, 0) == 0
    cyc = np.array(out[:64], dtype=np.float64) / steps
    waves = np.array(out[64:], dtype=np.float64) / steps
    s = env.scene.struct
    n_pairs = s.n_self_pairs
    total = cyc[:32].sum()
    print("config %s: %.1f M wave-cycles per step over %d rows" % (name, total / 1e6, int((waves > 0).sum())))
    phases = {32: "reset: start position", 33: "reset: velocity / acceleration", 34: "reset: planet phase", 36: "staged spawn: launch",
              38: "target point", 48: "keys kernel: table queries", 49: "keys kernel: planet keys", 50: "keys kernel: upper bound of the best candidate",
              39: "jobs kernel: trips, inside the scanned side's support (per wavefront)", 40: "jobs kernel: trips, inside the hull climb (per wavefront)",
              41: "jobs kernel: trips, rest of gjk_iterate (per wavefront)", 42: "jobs kernel: lanes with a run, summed over the trips (per wavefront)",
              43: "jobs kernel: jobs settled after 1 iteration (per wavefront)", 44: "jobs kernel: ... after 2", 45: "jobs kernel: ... after 3-4", 46: "jobs kernel: ... after 5-8",
              47: "jobs kernel: ... after 9+", 35: "jobs kernel: lane-trips entered with a simplex of <= 1 points", 37: "jobs kernel: ... of 2 points", 53: "jobs kernel: ... of 3 points",
              51: "jobs kernel: refill atomic (part of setup)", 52: "jobs kernel: job word + state loads (part of setup, lanes of the first job in the wave)",
              54: "candidate poses: table tests (wavefront rounds: sum of cycles, number)", 55: "candidate poses: self-collision pairs (rounds that get there)", 56: "candidate poses: rounds per step (waves column)",
              57: "jobs kernel: refills per wavefront (count)",
              58: "jobs kernel: staging", 59: "jobs kernel: job setup (per wavefront)", 60: "jobs kernel: trips (per wavefront)", 61: "jobs kernel: trips per wavefront (count)",
              62: "jobs kernel: MAX iterations of one job", 63: "jobs kernel: MAX cycles of a wavefront"}
    nenv = int(os.environ.get("SMR_BENCH_B", "65536"))
    total = cyc[:32].sum()
    for r in range(64):
        if waves[r] == 0:
            continue
        if r in (62, 63):
            print("  %-32s %d" % (phases.get(r), int(out[r])))
            continue
        if r >= 32:
            print("  %-32s %8.2f M cycles  %8.0f cycles per wave, %6.0f waves per step" % (phases.get(r, "phase %d" % r), cyc[r] / 1e6, cyc[r] / waves[r], waves[r]))
            continue
        what = "body row %d" % r if r < s.n_bodies else "pair row %d" % (r - s.n_bodies)       # pairs: nearest partner first
        print("  row %2d %-18s %8.2f M cycles %5.1f%%  %6.0f cycles per wave" % (r, what, cyc[r] / 1e6, 100 * cyc[r] / total, cyc[r] / waves[r]))


if __name__ == "__main__":
    main()
