"""The one fixture that comes from the REFERENCE ITSELF: tests/golden/ref_trajectory.json, written by
tests/golden/make_ref_trajectory.py from safemotions/utils/trajectory_manager.py as imported in the build container.  It pins the
episode-length rule of row A16 (tm.py:87,187-192) - the scene builder's episode_steps, the oracle's and the kernels' length
termination - and nothing else; parity of everything that lives in pybullet / klimits stays unpinned."""
import json
import os

import numpy as np
import pytest

from safemotionsrisk_amd.scene import SPACE_ENV_CONFIG, build_scene

REF = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_trajectory.json")))
CASES = [(c["trajectory_time_step"], c["trajectory_duration"]) for c in REF["cases"]]
UP, FAR_IDX = np.zeros(7), [0, 720]        # straight-up arm, planets on the far side of their orbits: nothing else can end the episode


def _case(dt, duration):
    return next(c for c in REF["cases"] if c["trajectory_time_step"] == dt and c["trajectory_duration"] == duration)


def test_fixture_is_the_reference_rule():
    assert REF["termination_trajectory_length"] == 2 and len(REF["cases"]) == 10
    for c in REF["cases"]:
        n = c["trajectory_length"]
        assert c["finished"] == [i >= n - 1 for i in range(n + 2)] and c["reasons"] == [2] and c["first_finished_index"] == n - 1
    assert _case(0.1, 2.0)["trajectory_length"] == 21 and _case(0.1, 8.0)["trajectory_length"] == 81


@pytest.mark.parametrize("dt,duration", CASES)
def test_scene_builder_and_oracle_follow_it(built, dt, duration):
    from oracle.oracle import OracleEnv
    c = _case(dt, duration)
    scene = build_scene(dict(SPACE_ENV_CONFIG, trajectory_time_step=dt, trajectory_duration=duration), auto_reset=False)
    assert scene.struct.episode_steps == c["first_finished_index"] == c["trajectory_length"] - 1
    env = OracleEnv(scene, 1, seed=0)
    env.set_state(q=UP, v=0, a=0, planet_idx=FAR_IDX, step_count=0, moving_hit=0)
    for index in range(1, c["first_finished_index"] + 1):            # base.py:1044: the index tested is the incremented episode length
        env.set_state(planet_idx=FAR_IDX)
        _, _, done, reason, info = env.step(np.zeros((1, 7)))
        assert info[0, 7] == index and bool(done[0]) == c["finished"][index]
        assert reason[0] == (REF["termination_trajectory_length"] if c["finished"][index] else -1)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dt,duration", [(0.1, 2.0), (0.1, 3.0), (0.05, 1.04), (0.1, 0.25)])
def test_kernels_follow_it(built, dt, duration):
    import torch
    from safemotionsrisk_amd import VecSafeMotionsEnv
    c = _case(dt, duration)
    B = 64
    env = VecSafeMotionsEnv(B, device=0, seed=0, auto_reset=False, **dict(SPACE_ENV_CONFIG, trajectory_time_step=dt, trajectory_duration=duration))
    env.set_state(q=np.zeros((B, 7), np.float32), v=np.zeros((B, 7), np.float32), a=np.zeros((B, 7), np.float32),
                  planet_idx=np.tile(np.array(FAR_IDX, np.int32), (B, 1)), step_count=np.zeros(B, np.int32), moving_hit=np.zeros(B, np.int32))
    zero = torch.zeros((B, 7), device="cuda")
    for index in range(1, c["first_finished_index"] + 1):
        env.set_state(planet_idx=np.tile(np.array(FAR_IDX, np.int32), (B, 1)))
        _, _, done, info = env.step(zero)
        assert (info["values"][:, 7] == index).all() and (done.bool() == c["finished"][index]).all()
        assert (info["termination_reason"] == (REF["termination_trajectory_length"] if c["finished"][index] else -1)).all()
    env.close()
