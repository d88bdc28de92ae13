"""Long runs of the staged pipelines (GPU box): many steps at full size, synchronising now and then - smr_synchronize fails if a
reset was ever committed without its staged start state - and checking the state against the limits."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG, BALL_ENV_CONFIG, REACH_ENV_CONFIG
for name, cfg, B, T in (("space", SPACE_ENV_CONFIG, 65536, 3000), ("space+bm", dict(SPACE_ENV_CONFIG, ball_machine_mode=True), 65536, 1000),
                        ("ball", BALL_ENV_CONFIG, 65536, 2000), ("reach", REACH_ENV_CONFIG, 32768, 1500)):
    env = VecSafeMotionsEnv(B, seed=1, auto_reset=True, **cfg)
    env.reset()
    s = env.scene.struct
    lo, hi, V, A = (np.array(x) for x in (s.pos_lo, s.pos_hi, s.vel_max, s.acc_max))
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    dones = 0
    for t in range(T):
        act = torch.rand((B, 7), device="cuda", generator=gen) * 2 - 1
        obs, rew, done, info = env.step(act)
        if t % 250 == 249:
            env.synchronize()
            dones += int(done.sum())
            st = env.get_state()
            assert np.isfinite(obs.cpu().numpy()).all() and np.isfinite(rew.cpu().numpy()).all()
            assert np.abs(-1 + 2 * (st["q"] - lo) / (hi - lo)).max() <= 1.001 and (np.abs(st["v"]) / V).max() <= 1.001 and (np.abs(st["a"]) / A).max() <= 1.001
            assert (st["step_count"] <= s.episode_steps).all()
    env.synchronize(); env.close()
    print(name, "ok:", T, "steps x", B, "envs", flush=True)
