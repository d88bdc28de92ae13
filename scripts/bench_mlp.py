"""Stand-alone timing of the network kernel (smr_mlp_forward) on the GPU box: rows/s and TFLOP/s of the two networks of the
risk-aware step (risk network 30-256-128-1, backup policy 23-256-128-14), nothing else running on the card.

    python scripts/bench_mlp.py [rows]
"""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from safemotionsrisk_amd.policy import backup_policy, risk_network  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    rng = np.random.default_rng(1)

    def synth(sizes):
        ws = [(rng.standard_normal((a, b)) / np.sqrt(a)).astype(np.float32) for a, b in zip(sizes[:-1], sizes[1:])]
        return ws, [(0.1 * rng.standard_normal(b)).astype(np.float32) for b in sizes[1:]]

    for name, net, n_in in (("risk 30-256-128-1", risk_network(*synth([30, 256, 128, 1])), 30),
                            ("backup 23-256-128-14", backup_policy(*synth([23, 256, 128, 14])), 23),
                            ("square 256-256-256-256", backup_policy(*synth([256, 256, 256, 256])), 256)):
        x = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (rows, n_in)).astype(np.float32)).cuda()
        y = net(x)
        torch.cuda.synchronize()
        ref = net.reference(x.cpu().numpy())
        err = float(np.abs(y.cpu().numpy() - ref).max())
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(50):
            net(x)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 50
        sizes = net.sizes
        macs = sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))
        pad = [(sizes[0] + 15) // 16 * 16] + [(s + 63) // 64 * 64 for s in sizes[1:]]
        macs_pad = sum(a * b for a, b in zip(pad[:-1], pad[1:]))
        print("%-22s rows %d  %.1f us  %.1f M rows/s  %.2f TFLOP/s useful, %.2f TFLOP/s issued (157 peak f32 MFMA)  max err %.2e"
              % (name, rows, ms * 1e3, rows / ms / 1e3, 2 * macs * rows / ms / 1e9, 2 * macs_pad * rows / ms / 1e9, err))


if __name__ == "__main__":
    main()
