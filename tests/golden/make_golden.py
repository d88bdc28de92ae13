"""Generates tests/golden/space_step.npz: seeded inputs and the float64 oracle's outputs.

The reference itself cannot produce vectors (pybullet/klimits are not installable), so
these fixtures pin the oracle against accidental change and give the GPU tests a data-only target that
needs no oracle build on the box.  Run from the repo root:  python tests/golden/make_golden.py [--bm | --reach | --brake] [--checks N]
(--brake: the reaching task with the braking-trajectory safety layer -> reach_brake_step.npz; its start states are pushed towards the
table so that braking trajectories are executed inside the fixture)
(--checks N: N contact samples per step instead of the default 24 = the reference's cadence -> space_step_n<N>.npz)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.oracle import OracleEnv                                   # noqa: E402
from safemotionsrisk_amd.scene import REACH_ENV_CONFIG, SPACE_ENV_CONFIG, build_scene, reach_brake_env_config   # noqa: E402


def main(ball_machine=False, reach=False, checks=None, brake=False):
    """ball_machine: the same scene on robot_ball_machine.urdf (-> space_bm_step.npz; info column 1 = self distance);
    reach: the target-point reaching task (-> reach_step.npz; extra state: target [B][5], target_count)"""
    B, T = 512, 6
    scene = build_scene(reach_brake_env_config() if brake else REACH_ENV_CONFIG if reach else dict(SPACE_ENV_CONFIG, ball_machine_mode=ball_machine),
                        auto_reset=False, collision_checks_per_step=checks)
    env = OracleEnv(scene, B, seed=11, num_threads=4)
    env.reset()
    st0 = env.get_state()
    # spread the episode counters so that terminations at step 20 occur inside the fixture
    rng = np.random.default_rng(42)
    st0["step_count"] = rng.integers(0, 20, B).astype(np.int32)
    if reach:                               # every fourth target 7-9 cm from the target link point: reached within the fixture
        near = np.array([env.body_frames(st0["q"][e])[0][8] for e in range(0, B, 4)]) + np.array([0.0, 0.0, 0.1])
        off = rng.normal(size=near.shape)
        off *= (rng.uniform(0.07, 0.09, len(near)) / np.linalg.norm(off, axis=1))[:, None]
        st0["target"][::4, :3] = (near + off).astype(np.float32)
        st0["target"][::4, 3] = st0["target"][::4, 4] = np.linalg.norm(st0["target"][::4, :3] - near, axis=1)
        st0["target"] = st0["target"].astype(np.float32).astype(np.float64)
    if brake:                               # every third env: elbow bent down towards the table, moving on down
        st0["q"][::3, 1] = rng.uniform(1.2, 1.9, len(st0["q"][::3])) * rng.choice([-1, 1], len(st0["q"][::3]))
        st0["q"][::3, 3] = rng.uniform(-1.0, 1.0, len(st0["q"][::3]))
        st0["v"][::3, 1] = np.sign(st0["q"][::3, 1]) * rng.uniform(0.2, 1.2, len(st0["q"][::3]))
    env.set_state(**st0)
    out = dict(B=B, T=T, seed=11, checks=int(scene.struct.collision_checks_per_step), q0=st0["q"].astype(np.float32), v0=st0["v"].astype(np.float32),
               a0=st0["a"].astype(np.float32), planet_idx0=st0["planet_idx"], step_count0=st0["step_count"],
               moving_hit0=st0["moving_hit"], target0=st0["target"].astype(np.float32), target_count0=st0["target_count"])
    # the oracle steps from the float32-rounded state the GPU will be given
    env.set_state(q=out["q0"], v=out["v0"], a=out["a0"])
    actions = rng.uniform(-1, 1, (T, B, 7)).astype(np.float32)
    actions[:, ::7] = np.sign(actions[:, ::7])            # some bang-bang envs ride the limits
    obs, rew, done, reason, info, states, targets, tcounts = [], [], [], [], [], [], [], []
    for t in range(T):
        o, r, d, rs, inf = env.step(actions[t].astype(np.float64))
        obs.append(o); rew.append(r); done.append(d); reason.append(rs); info.append(inf)
        st = env.get_state()
        states.append(np.concatenate([st["q"], st["v"], st["a"]], axis=1))
        targets.append(st["target"]); tcounts.append(st["target_count"])
        # every step restarts from the float32 image of the oracle state: one-step parity, no drift
        env.set_state(q=st["q"].astype(np.float32), v=st["v"].astype(np.float32), a=st["a"].astype(np.float32),
                      step_count=np.where(d, 0, st["step_count"]), moving_hit=np.where(d, 0, st["moving_hit"]),
                      target=st["target"].astype(np.float32))
    out.update(target=np.array(targets), target_count=np.array(tcounts))
    out.update(actions=actions, obs=np.array(obs), reward=np.array(rew), done=np.array(done), reason=np.array(reason),
               info=np.array(info), state=np.array(states))
    # component vectors
    q = rng.uniform(-2.0, 2.0, (256, 7)).astype(np.float32)
    idx = rng.integers(0, 1200, (256, 2)).astype(np.int32)
    idx[:, 1] = (idx[:, 0] - 480) % 1200
    out.update(comp_q=q, comp_idx=idx,
               comp_dist=np.array([env.distances(q[i].astype(np.float64), idx[i]) for i in range(256)]),
               comp_fk=np.array([env.fk(q[i].astype(np.float64))[0] for i in range(256)]))
    s = scene.struct
    p = rng.uniform(np.array(s.pos_lo), np.array(s.pos_hi), (256, 7)).astype(np.float32)
    v = rng.uniform(-np.array(s.vel_max), np.array(s.vel_max), (256, 7)).astype(np.float32)
    a = rng.uniform(-np.array(s.acc_max), np.array(s.acc_max), (256, 7)).astype(np.float32)
    rr = [env.safe_acc_range(p[i].astype(np.float64), v[i].astype(np.float64), a[i].astype(np.float64)) for i in range(256)]
    out.update(comp_p=p, comp_v=v, comp_a=a, comp_lo=np.array([r[0] for r in rr]), comp_hi=np.array([r[1] for r in rr]),
               comp_code=np.array([r[2] for r in rr]))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                       "reach_brake_step.npz" if brake else "reach_step.npz" if reach else "space_bm_step.npz" if ball_machine else
                       "space_step.npz" if checks is None else "space_step_n%d.npz" % checks)
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes; done fraction", np.mean(out["done"]), "reasons", np.unique(out["reason"]))


if __name__ == "__main__":
    main(ball_machine="--bm" in sys.argv, reach="--reach" in sys.argv or "--brake" in sys.argv, brake="--brake" in sys.argv,
         checks=int(sys.argv[sys.argv.index("--checks") + 1]) if "--checks" in sys.argv else None)
