import sys, time; sys.path.insert(0,'.')
import torch
from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
def run(name, **over):
    cfg=dict(SPACE_ENV_CONFIG, **over)
    B=65536
    e=VecSafeMotionsEnv(B, seed=0, auto_reset=True, return_info=False, **cfg)
    e.reset(); act=torch.rand((B,7),device='cuda')*2-1
    for _ in range(5): e.step(act)
    e.synchronize(); e.set_profiling(True)
    for _ in range(40): e.step(act)
    pr=e.get_profile(); e.close()
    print(name, {k:round(v['ms']/max(v['launches'],1),3) for k,v in pr.items()})
VARIANTS = {
    'baseline': {},
    'kin_prob=1 (no walks)': dict(collision_avoidance_kinematic_state_sampling_probability=1.0),
    'kin_prob=0 (all walks)': dict(collision_avoidance_kinematic_state_sampling_probability=0.0),
    'stay=1 (walk ends at once)': dict(collision_avoidance_kinematic_state_sampling_probability=0.0, collision_avoidance_stay_in_state_probability=1.0),
    'no planets': dict(planet_mode=False, terminate_on_collision_with_moving_obstacle=False),
    'no planets, kin=1': dict(planet_mode=False, terminate_on_collision_with_moving_obstacle=False, collision_avoidance_kinematic_state_sampling_probability=1.0),
}
for name in (sys.argv[1:] or list(VARIANTS)):
    run(name, **VARIANTS[name])
