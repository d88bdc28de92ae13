"""Does the chip take two half batches side by side faster than one whole batch?  (GPU box)  Two handles of B / 2 envs on two streams,
stepped in the same loop iteration, against one handle of B envs; same scene, same contact cadence.
    python scripts/study/two_halves.py [CONFIG] [B] [STEPS]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
import safemotionsrisk_amd as smr

name = sys.argv[1] if len(sys.argv) > 1 else "space"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cfg, kw, _, _ = bench._env_config(name)
kw = {k: v for k, v in kw.items() if k != "risk_networks"}


def run(parts):
    n = B // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    envs = []
    for p in range(parts):
        with torch.cuda.stream(streams[p]):
            envs.append(smr.VecSafeMotionsEnv(n, seed=0, auto_reset=True, env_index_base=p * n, **kw, **cfg))
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = [[torch.rand((n, 7), device="cuda", generator=g) * 2 - 1 for _ in range(8)] for _ in range(parts)]
    torch.cuda.synchronize()
    for p in range(parts):
        with torch.cuda.stream(streams[p]):
            envs[p].reset()
    for i in range(30):
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                envs[p].step(acts[p][i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                envs[p].step(acts[p][i % 8])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for e in envs:
        e.synchronize(); e.close()
    return B * steps / el / 1e6, el / steps * 1e3


for parts in (1, 2, 4, 1, 2):
    v, ms = run(parts)
    print("%s: %d handle(s) of %d envs on %d stream(s): %.2f M env-steps/s, %.3f ms per step of the whole batch" % (name, parts, B // parts, parts, v, ms), flush=True)
