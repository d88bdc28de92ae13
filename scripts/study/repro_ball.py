import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import safemotionsrisk_amd as smr
B = 512
env = smr.VecSafeMotionsEnv(B, device=0, seed=1, auto_reset=bool(int(sys.argv[1])) if len(sys.argv) > 1 else True, **smr.BALL_ENV_CONFIG)
print("created", flush=True)
env.reset(); env.synchronize(); print("reset ok", flush=True)
for t in range(5):
    env.step(torch.zeros((B, 7), device="cuda")); env.synchronize(); print("step", t, "ok", flush=True)
