"""Where the step time of the ball-machine variant goes (GPU box)."""
import sys; sys.path.insert(0, '.')
import torch
import safemotionsrisk_amd.env as envmod
from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
_orig = envmod.build_scene
_patch = {}
def patched(*a, **k):
    b = _orig(*a, **k)
    for key, val in _patch.items():
        setattr(b.struct, key, val)
    return b
envmod.build_scene = patched
def run(name, **patch):
    _patch.clear(); _patch.update(patch)
    B = 65536
    e = VecSafeMotionsEnv(B, seed=0, auto_reset=True, return_info=False, **dict(SPACE_ENV_CONFIG, ball_machine_mode=True))
    e.reset(); act = torch.rand((B, 7), device='cuda') * 2 - 1
    for _ in range(5): e.step(act)
    e.synchronize(); e.set_profiling(True)
    for _ in range(40): e.step(act)
    pr = e.get_profile(); e.close()
    print(name, {k: round(v['ms'] / max(v['launches'], 1), 3) for k, v in pr.items()}, flush=True)
run('baseline')
run('no self pairs', n_self_pairs=0)
run('bm bodies out of static/moving masks', static_link_mask=0b0011111100, moving_link_mask=0b0011111000)
run('bm bodies out of all masks + no self', static_link_mask=0b0011111100, moving_link_mask=0b0011111000, contact_link_mask=0b0011111111, n_self_pairs=0)
