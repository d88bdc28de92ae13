import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
B=65536
e=VecSafeMotionsEnv(B, seed=0, auto_reset=True, return_info=False, **dict(SPACE_ENV_CONFIG, ball_machine_mode=True))
e.reset(); act=torch.rand((B,7),device='cuda')*2-1
for _ in range(45): e.step(act)
e.synchronize(); e.close()
