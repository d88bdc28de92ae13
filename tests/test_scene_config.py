"""Host logic: env_config -> scene (A5 limits, masks, thresholds), validation of unsupported options,
drop-in constants.  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from safemotionsrisk_amd import scene as sc
from safemotionsrisk_amd.scene import SPACE_ENV_CONFIG, build_scene

REF_PARAMS = "/root/reference/safemotions/trained_networks/backup_networks/space/params.json"


def test_joint_limits_match_survey(space_scene):
    s = space_scene.struct
    assert np.allclose(list(s.pos_hi), [2.93206, 2.059395, 2.93206, 2.059395, 2.93206, 2.059395, 3.019326], atol=1e-6)
    assert np.allclose(list(s.pos_lo), -np.array(list(s.pos_hi)))
    assert np.allclose(list(s.vel_max), [1.7104, 1.7104, 1.7453, 2.2689, 2.4435, 3.1416, 3.1416], atol=1e-4)
    assert list(s.acc_max) == [15.0, 7.5, 10.0, 12.5, 15.0, 20.0, 20.0]
    assert np.allclose(list(s.jerk_max), [300, 150, 200, 250, 300, 400, 400])           # min(2 a / dt, 7500 ...)


def test_masks_thresholds_and_sizes(space_scene):
    s = space_scene.struct
    assert s.static_link_mask == 0b11111100        # iiwa_link_2..7: table observed AND closest_point_active
    assert s.moving_link_mask == 0b11111000        # iiwa_link_3..7 (scene.py:288-290)
    assert s.static_max_distance == pytest.approx(0.102) and s.moving_max_distance == pytest.approx(0.6)
    assert s.collision_checks_per_step == 24 and s.updates_per_step == 24 and s.episode_steps == 20
    assert s.obs_size == 23 and s.obs_num_planets == 1 and s.obs_planet_size_per_planet == 2
    assert s.n_pieces == 8 + 1 + 24 + 1 and s.n_static == 1
    assert [s.start_box[c][0] for c in range(3)] == [-1.1, -1.1, -0.1]
    assert s.contact_distance == pytest.approx(0.02)
    assert list(s.ee_offset) == [0.0, 0.0, 0.126]


def test_collision_checks_option():
    # default = the reference's contact cadence, one check per 1/240 s obstacle update (ctlp.py:2631-2637); collision_check_time
    # has no say in it (it only samples the braking-trajectory checker, ctlp.py:99-103,3167-3185)
    assert build_scene(SPACE_ENV_CONFIG).struct.collision_checks_per_step == 24
    assert build_scene(dict(SPACE_ENV_CONFIG, collision_check_time=None)).struct.collision_checks_per_step == 24
    assert build_scene(SPACE_ENV_CONFIG, collision_checks_per_step=3).struct.collision_checks_per_step == 3
    assert build_scene(SPACE_ENV_CONFIG, collision_checks_per_step=1).struct.collision_checks_per_step == 1
    with pytest.raises(ValueError):
        build_scene(SPACE_ENV_CONFIG, collision_checks_per_step=5)     # must divide 24


def test_obs_planet_size_variants():
    assert build_scene(dict(SPACE_ENV_CONFIG, obs_planet_size_per_planet=1)).obs_size == 22
    assert build_scene(dict(SPACE_ENV_CONFIG, obs_planet_size_per_planet=3)).obs_size == 24
    cfg = dict(SPACE_ENV_CONFIG, planet_two_time_shift=None, planet_two_period=4.0)
    b = build_scene(cfg)
    assert b.obs_size == 25 and b.struct.planet[1].coupled == 0 and b.struct.planet[1].n_orbit == 960


@pytest.mark.parametrize("key,value,exc", [
    ("punish_end_max_torque", True, NotImplementedError), ("punish_braking_trajectory_max_torque", True, NotImplementedError),
    ("check_braking_trajectory_collisions", True, NotImplementedError), ("use_real_robot", True, NotImplementedError),
    ("human_network_checkpoint", "human_network/checkpoint/checkpoint", NotImplementedError),
    ("risk_config_dir", "risk_networks/state_action/space", FileNotFoundError),      # (no such directory here: the reference raises the same)
    ("robot_scene", 9, NotImplementedError), ("obstacle_scene", 3, NotImplementedError),
    ("collision_avoidance_mode", False, NotImplementedError), ("max_resampling_attempts", 3, ValueError),
    ("closest_point_safety_distance", 0.2, ValueError),
])
def test_unsupported_options_fail_loudly(key, value, exc):
    with pytest.raises(exc):
        build_scene(dict(SPACE_ENV_CONFIG, **{key: value}))


def test_reaching_task_scene_fields():
    """target points: the target link point is observed absolutely and relatively (2 x 3 values,
    behind the 21 joint values), episodes start at rest outside collision_avoidance_mode, and the start pose keeps the
    larger clearances of ctlp.py:1468-1477 (safety distance + 0.09 m to obstacles, + 0.04 m to the robot itself)."""
    from safemotionsrisk_amd.scene import REACH_ENV_CONFIG
    b = build_scene(REACH_ENV_CONFIG)
    s = b.struct
    assert s.use_target_points == 1 and s.collision_avoidance_mode == 0 and b.obs_size == 29
    assert s.obs_target_pos == 1 and s.obs_target_rel == 1 and s.target_radius == pytest.approx(0.065)
    assert all(hi > lo for lo, hi in zip(s.target_rel_min, s.target_rel_max))
    with pytest.raises(NotImplementedError):
        build_scene(dict(REACH_ENV_CONFIG, use_target_points=False))       # collision_avoidance_mode=False needs a task
    with pytest.raises(NotImplementedError):
        build_scene(dict(REACH_ENV_CONFIG, target_point_sequence=1))       # alternating targets: multi-robot scenes only
    assert build_scene(dict(SPACE_ENV_CONFIG, use_target_points=True)).obs_size == 23       # observation flags default to off
    assert build_scene(dict(SPACE_ENV_CONFIG, use_target_points=True, obs_add_target_point_relative_pos=True)).obs_size == 23 + 3


def test_missing_experiment_name():
    cfg = dict(SPACE_ENV_CONFIG)
    del cfg["experiment_name"]
    with pytest.raises(TypeError):
        build_scene(cfg)


@pytest.mark.skipif(not os.path.exists(REF_PARAMS), reason="reference checkout not mounted")
def test_readme_config_equals_shipped_params_json():
    """SPACE_ENV_CONFIG = the shipped backup_networks/space env_config, except ball_machine_mode and
    the target_link_offset that goes with it (fixture: SURVEY section 8c)."""
    with open(REF_PARAMS) as f:
        ref = json.load(f)["env_config"]
    readme_only = set()
    for k, v in SPACE_ENV_CONFIG.items():
        if k in ("experiment_name",):
            continue
        if k not in ref:            # flags of the README command that are newer than the shipped checkpoint
            readme_only.add(k)
            continue
        assert ref[k] == v or (isinstance(v, list) and list(ref[k]) == list(v)), (k, ref[k], v)
    assert readme_only <= {"collision_avoidance_kinematic_state_sampling_probability"}
    assert ref["ball_machine_mode"] is True
    # the shipped env_config is accepted UNCHANGED: robot_ball_machine.urdf = 10 collision bodies, 14 self-collision pairs,
    # target link "ball_machine" with a zero offset (scene.py:188-199)
    b = build_scene(ref)
    assert b.struct.n_bodies == 10 and b.struct.n_self_pairs == 14 and b.struct.ee_body == 8 and list(b.struct.ee_offset) == [0, 0, 0]
    assert b.obs_size == 23 and b.struct.static_link_mask == 0b1111111100 and b.struct.moving_link_mask == 0b1111111000


def test_ball_machine_scene_fields():
    """ball_machine_mode adds the two convex pieces of link "ball_machine" as bodies 8 and 9 (fixed on iiwa_link_7) and the
    base adapter's cylinder as a world piece that only takes part in self-collision pairs (ctlp.py:1438-1446)."""
    plain, bm = build_scene(SPACE_ENV_CONFIG), build_scene(dict(SPACE_ENV_CONFIG, ball_machine_mode=True))
    assert plain.struct.n_bodies == 8 and plain.struct.n_self_pairs == 0 and plain.struct.ee_body == 7
    s = bm.struct
    assert s.n_bodies == 10 and [s.body_parent[b] for b in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 7, 7]
    assert list(s.body_xyz[8]) == [0.0, 0.175, 0.1645] and list(s.body_xyz[9]) == [0.0, 0.175, 0.1645]
    pairs = [(s.self_pair[i][0], s.self_pair[i][1]) for i in range(s.n_self_pairs)]
    adapter = -(pairs[0][1] + 1)
    assert pairs == [(b, p) for b in (8, 9) for p in [-(adapter + 1), 0, 1, 2, 3, 4, 5]]
    pc = bm.pieces[adapter]
    assert pc.kind == 1 and abs(pc.cyl_radius - 0.159) < 1e-12 and abs(pc.cyl_half_length - 0.009) < 1e-12
    assert list(pc.center) == [-0.2, 0.0, 0.01]
    assert s.contact_link_mask == 0b1111111111 and bm.obs_size == plain.obs_size


@pytest.mark.skipif(not os.path.exists("/root/reference/safemotions/description/urdf/robot.urdf"),
                    reason="reference checkout not mounted")
def test_committed_asset_tables_match_reference_assets():
    from safemotionsrisk_amd.assets import compile_assets
    fresh = compile_assets("/root/reference")
    committed = np.load(sc._DATA)
    assert set(fresh) == set(committed.files)
    for k in fresh:
        assert np.array_equal(np.asarray(fresh[k]), committed[k]), k


def test_termination_constants_and_spaces():
    from safemotionsrisk_amd.env import VecSafeMotionsEnv as E
    assert (E.TERMINATION_UNSET, E.TERMINATION_JOINT_LIMITS, E.TERMINATION_TRAJECTORY_LENGTH, E.TERMINATION_SELF_COLLISION,
            E.TERMINATION_COLLISION_WITH_STATIC_OBSTACLE, E.TERMINATION_COLLISION_WITH_MOVING_OBSTACLE) == (-1, 1, 2, 3, 4, 5)


# ---- the env_config keys of risk-aware stepping (train.py:168-309 -> safe_motions_base.py:170-180,502-585) ----

def _risk_tree(tmp_path, action_size=7, backup_over=None, observation_size=23):
    """a trained_networks directory with ONE risk network description and the params.json of its backup policy (plain JSON, as shipped)"""
    root = tmp_path / "trained_networks"
    (root / "risk_networks" / "state_action" / "space").mkdir(parents=True)
    (root / "backup_networks" / "space" / "checkpoint").mkdir(parents=True)
    (root / "risk_networks" / "state_action" / "space" / "risk_config.json").write_text(json.dumps(
        dict(action_size=action_size, checkpoint="backup_networks/space/checkpoint/checkpoint", observation_size=observation_size)))
    backup = dict(SPACE_ENV_CONFIG, ball_machine_mode=True, **(backup_over or {}))
    (root / "backup_networks" / "space" / "params.json").write_text(json.dumps(dict(env_config=backup)))
    return str(root)


def test_risk_env_config_keys(tmp_path):
    from safemotionsrisk_amd.scene import REACH_ENV_CONFIG, resolve_risk_config
    root = _risk_tree(tmp_path)
    task = dict(REACH_ENV_CONFIG, ball_machine_mode=False, planet_one_period=9.0, risk_config_dir="risk_networks/state_action/space", risk_threshold=0.065,
                risk_state_config=0, risk_state_deterministic_backup_trajectory=False, risk_check_initial_backup_trajectory=False,
                risk_state_initial_backup_trajectory_steps=20, risk_store_ground_truth=False, risk_ground_truth_episodes_per_file=None,
                risk_ignore_estimation_probability=0.0, risk_state_backup_trajectory_steps=None)
    b = build_scene(task, trained_networks_dir=root)
    assert b.risk == dict(observation_size=23, action_size=7, checkpoint="backup_networks/space/checkpoint/checkpoint", risk_threshold=0.065,
                          risk_state_config=0)
    # the scene the backup policy was trained in wins (safe_motions_base.py:525-545): ball machine mounted, its planet period
    assert b.struct.n_bodies == 10 and b.env_config["planet_one_period"] == SPACE_ENV_CONFIG["planet_one_period"]
    assert build_scene(dict(task, risk_threshold=None), trained_networks_dir=root).risk["risk_threshold"] == 0.05      # base.py:222-223
    # a dict instead of a directory (the constructor's risk_config argument)
    b2 = build_scene(dict({k: v for k, v in task.items() if k != "risk_config_dir"},
                          risk_config=dict(action_size=None, checkpoint="backup_networks/space/checkpoint/checkpoint", observation_size=23),
                          risk_state_config=1), trained_networks_dir=root)
    assert b2.risk["action_size"] is None and b2.risk["risk_state_config"] == 1
    with pytest.raises(ValueError):                          # safe_motions_base.py:548-558
        build_scene(dict(task, trajectory_time_step=0.05), trained_networks_dir=root)
    with pytest.raises(FileNotFoundError):
        build_scene(task, trained_networks_dir=str(tmp_path / "nowhere"))
    with pytest.raises(NotImplementedError):                 # the initial-backup-trajectory loop (observations.py:155-185) is out of scope
        build_scene(dict(task, risk_check_initial_backup_trajectory=True), trained_networks_dir=root)
    # the modes that step a background copy of the env, the ground truth and the ignore probability (safe_motions_base.py:1518-1577,1605-1655)
    sim = dict({k: v for k, v in task.items() if k != "risk_config_dir"}, risk_state_config=4, risk_store_ground_truth=True, risk_ignore_estimation_probability=0.35,
               risk_state_deterministic_backup_trajectory=True, risk_ground_truth_episodes_per_file=5,
               risk_config=dict(action_size=None, checkpoint="backup_networks/space/checkpoint/checkpoint", observation_size=None))     # evaluate.py:588-594
    b4 = build_scene(sim, trained_networks_dir=root)
    assert b4.risk == dict(observation_size=None, action_size=None, checkpoint="backup_networks/space/checkpoint/checkpoint", risk_threshold=0.065, risk_state_config=4,
                           backup_trajectory_steps=20, store_ground_truth=True, ignore_probability=0.35, ground_truth_episodes_per_file=5,
                           deterministic_backup_trajectory=True)                         # 20 = the backup policy's 2.0 s episodes / 0.1 s, base.py:573-576
    assert build_scene(dict(sim, risk_state_config=3, risk_state_backup_trajectory_steps=7, risk_store_ground_truth=False), trained_networks_dir=root).risk["backup_trajectory_steps"] == 7
    with pytest.raises(ValueError):
        build_scene(dict(sim, risk_state_config=5), trained_networks_dir=root)
    with pytest.raises(ValueError):
        build_scene(dict(sim, risk_ignore_estimation_probability=1.5), trained_networks_dir=root)
    with pytest.warns(UserWarning):                          # safe_motions_base.py:560-568
        resolve_risk_config(dict(task, closest_point_safety_distance=0.02), root)
    # without a risk network the risk keys are inert, as in the reference
    assert build_scene(dict(REACH_ENV_CONFIG, risk_threshold=0.1, risk_state_config=0)).risk is None


TASK_PARAMS = "/root/reference/safemotions/trained_networks/task_networks/reaching_task/space/state_action/params.json"


@pytest.mark.skipif(not os.path.exists(TASK_PARAMS), reason="reference checkout not mounted")
def test_shipped_task_params_json_builds():
    """trained_networks/task_networks/reaching_task/space/state_action/params.json as shipped, minus risk_check_initial_backup_trajectory
    (the initial-backup-trajectory loop of obs.py:155-185 is out of scope): only risk_config.json and the backup
    policy's params.json are read - no checkpoint, no SavedModel."""
    with open(TASK_PARAMS) as f:
        cfg = json.load(f)["env_config"]
    assert cfg["risk_check_initial_backup_trajectory"] is True
    b = build_scene(dict(cfg, risk_check_initial_backup_trajectory=False), trained_networks_dir="/root/reference/safemotions/trained_networks")
    assert b.risk == dict(observation_size=23, action_size=7, checkpoint="backup_networks/space/checkpoint/checkpoint", risk_threshold=0.065, risk_state_config=0)
    assert b.struct.use_target_points and b.struct.n_bodies == 10 and b.obs_size - 6 == 23


def test_more_than_24_contact_samples_per_step_are_rejected():
    """trajectory_time_step = 0.2 s has 48 obstacle updates per step; the kernels carry the sample index in job words and
    item-queue entries sized for at most 24 samples (csrc/smr_device.h: kMaxChecksPerStep): the default N = 48 must be
    refused loudly - it used to alias the end-of-step jobs' sample index - and a divisor <= 24 is the explicit choice."""
    from safemotionsrisk_amd import SPACE_ENV_CONFIG, build_scene
    from safemotionsrisk_amd.scene import MAX_COLLISION_CHECKS_PER_STEP
    cfg = dict(SPACE_ENV_CONFIG, trajectory_time_step=0.2, trajectory_duration=4.0)
    with pytest.raises(ValueError, match="collision_checks_per_step"):
        build_scene(cfg)
    b = build_scene(cfg, collision_checks_per_step=24)
    assert b.struct.updates_per_step == 48 and b.struct.collision_checks_per_step == 24 == MAX_COLLISION_CHECKS_PER_STEP
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "safemotionsrisk_amd", "csrc", "smr_device.h")).read()
    assert "constexpr int kMaxChecksPerStep = %d;" % MAX_COLLISION_CHECKS_PER_STEP in src
