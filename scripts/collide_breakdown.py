"""Where the collide kernel's time goes: switch parts of the query set off through the scene masks (GPU box)."""
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
def run(name, N=None, **patch):
    _patch.clear(); _patch.update(patch)
    B = 65536
    e = VecSafeMotionsEnv(B, seed=0, auto_reset=True, return_info=False, collision_checks_per_step=N, **SPACE_ENV_CONFIG)
    e.reset(); act = torch.rand((B, 7), device='cuda') * 2 - 1
    for _ in range(5): e.step(act)
    e.synchronize(); e.set_profiling(True)
    for _ in range(40): e.step(act)
    pr = e.get_profile(); e.close()
    print(name, {k: round(v['ms'] / max(v['launches'], 1), 3) for k, v in pr.items()}, flush=True)
VARIANTS = {
    'baseline': {}, 'no static': dict(static_link_mask=0), 'no moving distance': dict(moving_link_mask=0),
    'no contact': dict(contact_link_mask=0), 'no moving, no contact': dict(moving_link_mask=0, contact_link_mask=0),
    'only contact': dict(moving_link_mask=0, static_link_mask=0), 'only static': dict(moving_link_mask=0, contact_link_mask=0),
    'nothing': dict(moving_link_mask=0, contact_link_mask=0, static_link_mask=0),
    'only moving': dict(static_link_mask=0, contact_link_mask=0),
    'N=1': dict(N=1), 'N=1 no contact': dict(N=1, contact_link_mask=0),
}
for name in (sys.argv[1:] or list(VARIANTS)):        # e.g. under rocprofv3: one variant per run
    run(name, **VARIANTS[name])
