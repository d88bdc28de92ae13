"""Host-side scene construction: reference ``env_config`` keys -> ``smr_scene_t`` (include/smr_scene.h).

Mirrors what the reference does once per process in its constructors:

* joint limits, ``scene.py:20-22,350-371,441-455`` (A5)
* observed-link bookkeeping, ``ctlp.py:579-586,2419-2434``, ``scene.py:226-228,288-292``
* planet orbit tables, ``ctlp.py:4403-4437`` (A9)
* reward / termination / start-sampling constants, ``rewards.py:40-93,409-427``, ``ctlp.py:172-183,1468-1477``

The geometry comes from ``data/iiwa7_assets.npz`` (see assets.py).  Unsupported configurations raise
``NotImplementedError`` / ``ValueError`` instead of silently running something else.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

SCENE_VERSION = 8
NJ = 7
NLINK = 8
MAX_BODIES = 10
MAX_SELF_PAIRS = 16
MAX_PLANETS = 2
MAX_STATIC = 4

SIM_TIME_STEP = 1.0 / 240.0                      # base.py:34
JOINT_LIMITS_SAFETY_BUFFER_IIWA = 0.035          # scene.py:20
MAX_ACCELERATION_IIWA = [15.0, 7.5, 10.0, 12.5, 15.0, 20.0, 20.0]   # scene.py:21
MAX_JERK_IIWA = [7500, 3750, 5000, 6250, 7500, 10000, 10000]         # scene.py:22
BULLET_CONTACT_BREAKING_THRESHOLD = 0.02         # Bullet gContactBreakingThreshold

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "iiwa7_assets.npz")


PIECE_HULL, PIECE_CYLINDER_Z = 0, 1


class Piece(C.Structure):
    _fields_ = [("vert_begin", C.c_int32), ("vert_count", C.c_int32), ("center", C.c_double * 3),
                ("radius", C.c_double), ("margin", C.c_double), ("kind", C.c_int32), ("_pad", C.c_int32),
                ("cyl_radius", C.c_double), ("cyl_half_length", C.c_double)]


class Planet(C.Structure):
    _fields_ = [("piece_begin", C.c_int32), ("piece_count", C.c_int32), ("n_orbit", C.c_int32),
                ("index_shift", C.c_int32), ("coupled", C.c_int32), ("_pad", C.c_int32),
                ("orbit_pose", C.POINTER(C.c_double)), ("orbit_local", C.POINTER(C.c_double)),
                ("radius_xy", C.c_double * 2)]


_D3 = C.c_double * 3
_D7 = C.c_double * NJ


class Scene(C.Structure):
    _fields_ = [
        ("version", C.c_int32), ("obs_size", C.c_int32),
        ("dt", C.c_double), ("episode_steps", C.c_int32), ("updates_per_step", C.c_int32),
        ("collision_checks_per_step", C.c_int32), ("auto_reset", C.c_int32),
        ("pos_lo", _D7), ("pos_hi", _D7), ("vel_max", _D7), ("acc_max", _D7), ("jerk_max", _D7),
        ("base_xyz", _D3), ("joint_xyz", _D3 * NJ), ("joint_axis", _D3 * NJ), ("ee_offset", _D3),
        ("verts", C.POINTER(C.c_double)), ("pieces", C.POINTER(Piece)),
        ("adj_offsets", C.POINTER(C.c_int32)), ("adj", C.POINTER(C.c_int32)),
        ("n_verts", C.c_int32), ("n_pieces", C.c_int32), ("n_bodies", C.c_int32), ("link_piece", C.c_int32 * MAX_BODIES),
        ("body_parent", C.c_int32 * MAX_BODIES), ("ee_body", C.c_int32),
        ("body_xyz", _D3 * MAX_BODIES), ("body_rot", (C.c_double * 9) * MAX_BODIES),
        ("n_self_pairs", C.c_int32), ("self_pair", (C.c_int32 * 2) * MAX_SELF_PAIRS),
        ("n_static", C.c_int32), ("static_piece", C.c_int32 * MAX_STATIC), ("static_link_mask", C.c_uint32),
        ("static_max_distance", C.c_double),
        ("moving_link_mask", C.c_uint32), ("contact_link_mask", C.c_uint32),
        ("moving_max_distance", C.c_double), ("contact_distance", C.c_double),
        ("n_planets", C.c_int32), ("obs_planet_size_per_planet", C.c_int32), ("obs_num_planets", C.c_int32),
        ("_pad1", C.c_int32),
        ("planet", Planet * MAX_PLANETS),
        ("planet_obs_pos_min", _D3), ("planet_obs_pos_max", _D3),
        ("use_ball", C.c_int32), ("ball_piece", C.c_int32), ("ball_radius", C.c_double), ("ball_speed", C.c_double),
        ("ball_sphere_center", _D3), ("ball_sphere_radius", C.c_double), ("ball_sphere_height", C.c_double * 2),
        ("ball_sphere_angle", C.c_double * 2), ("ball_high_angle_prob", C.c_double),
        ("ball_check_invalid_target", C.c_int32), ("ball_random_initial_position", C.c_int32),
        ("ball_invalid_min", _D3), ("ball_invalid_max", _D3), ("ball_final_min", _D3), ("ball_final_max", _D3),
        ("ball_obs_pos_min", _D3), ("ball_obs_pos_max", _D3), ("ball_obs_vel_min", _D3), ("ball_obs_vel_max", _D3),
        ("ball_active_xy_radius", C.c_double), ("ball_target_box", (C.c_double * 2) * 3),
        ("ball_target_min_dist_static", C.c_double), ("ball_target_min_dist_self", C.c_double), ("plane_z", C.c_double), ("table_half_xy", C.c_double * 2),
        ("use_target_points", C.c_int32), ("obs_target_pos", C.c_int32), ("obs_target_rel", C.c_int32), ("target_normalize", C.c_int32),
        ("target_min_dist_static", C.c_double), ("target_min_dist_self", C.c_double),
        ("target_radius", C.c_double), ("target_bonus", C.c_double), ("target_factor", C.c_double),
        ("target_rel_min", _D3), ("target_rel_max", _D3),
        ("collision_avoidance_mode", C.c_int32), ("_pad3", C.c_int32),
        ("punish_action", C.c_int32), ("terminate_on_self_collision", C.c_int32),
        ("terminate_on_static_collision", C.c_int32), ("terminate_on_moving_collision", C.c_int32),
        ("action_punishment_min_threshold", C.c_double), ("action_max_punishment", C.c_double),
        ("self_max_reward", C.c_double), ("self_max_reward_distance", C.c_double),
        ("static_max_reward", C.c_double), ("static_max_reward_distance", C.c_double),
        ("moving_max_reward", C.c_double), ("moving_max_reward_distance", C.c_double),
        ("low_acc_max_reward", C.c_double), ("low_acc_threshold", C.c_double),
        ("low_vel_max_reward", C.c_double), ("low_vel_threshold", C.c_double),
        ("termination_bonus", C.c_double), ("early_termination_punishment", C.c_double),
        ("collision_distance", C.c_double),
        ("check_braking_trajectory_collisions", C.c_int32), ("brake_checks_per_step", C.c_int32), ("brake_timeout_steps", C.c_int32),
        ("_pad4", C.c_int32), ("brake_safety_distance", C.c_double), ("brake_acc_max", _D7), ("brake_jerk_max", _D7),
        ("brake_stop_velocity", C.c_double), ("brake_stop_acceleration", C.c_double),
        ("punish_adaptation", C.c_int32), ("punish_end_min_distance", C.c_int32), ("punish_braking_trajectory_min_distance", C.c_int32),
        ("_pad5", C.c_int32), ("adaptation_max_punishment", C.c_double),
        ("end_min_distance_max_punishment", C.c_double), ("end_min_distance_max_threshold", C.c_double),
        ("braking_trajectory_max_punishment", C.c_double), ("braking_trajectory_min_distance_max_threshold", C.c_double),
        ("start_box", (C.c_double * 2) * 3), ("start_min_dist_static", C.c_double),
        ("start_min_dist_self", C.c_double),
        ("kinematic_state_sampling_mode", C.c_int32), ("_pad2", C.c_int32),
        ("kinematic_state_sampling_probability", C.c_double), ("stay_in_state_probability", C.c_double),
    ]


# --------------------------------------------------------------------------------------------
# env_config handling
# --------------------------------------------------------------------------------------------

#: README "Training of the backup policy / Space environment" command as an env_config (README.md:74,
#: train.py:168-309); identical to trained_networks/backup_networks/space/params.json except
#: ball_machine_mode (False here = the README command and BASELINE configs[2]; True, the shipped value, is supported too).
SPACE_ENV_CONFIG = dict(
    experiment_name="backup_space", acc_limit_factor=1.0, action_max_punishment=0.4,
    action_punishment_min_threshold=0.95, closest_point_safety_distance=0.01,
    collision_avoidance_episode_early_termination_punishment=-15.0,
    collision_avoidance_episode_termination_bonus=15.0,
    collision_avoidance_kinematic_state_sampling_mode=True,
    collision_avoidance_kinematic_state_sampling_probability=0.7,
    collision_avoidance_low_acceleration_max_reward=0.0, collision_avoidance_low_acceleration_threshold=1.0,
    collision_avoidance_low_velocity_max_reward=0.0, collision_avoidance_low_velocity_threshold=1.0,
    collision_avoidance_mode=True, collision_avoidance_moving_obstacles_max_reward_distance=0.6,
    collision_avoidance_moving_obstacles_max_reward=3.0, collision_avoidance_self_collision_max_reward_distance=0.05,
    collision_avoidance_self_collision_max_reward=1.0, collision_avoidance_static_obstacles_max_reward_distance=0.1,
    collision_avoidance_static_obstacles_max_reward=1.0, collision_avoidance_stay_in_state_probability=0.3,
    collision_check_time=0.033, episodes_per_simulation_reset=4000, jerk_limit_factor=1.0,
    obs_planet_size_per_planet=2, obstacle_scene=5, planet_mode=True,
    planet_one_center=[-0.1, 0.0, 0.8], planet_one_euler_angles=[0.35, 0, 0], planet_one_period=5.0,
    planet_one_radius_xy=[0.65, 0.8], planet_two_center=[-0.1, 0, 0.8], planet_two_euler_angles=[-0.35, 0, 0],
    planet_two_radius_xy=[0.75, 0.8], planet_two_time_shift=-2.0, pos_limit_factor=1.0, punish_action=True,
    robot_scene=0, solver_iterations=50, starting_point_cartesian_range_scene=1,
    terminate_on_collision_with_moving_obstacle=True, terminate_on_collision_with_static_obstacle=True,
    terminate_on_self_collision=True, trajectory_duration=2.0, trajectory_time_step=0.1,
    use_controller_target_velocities=True, vel_limit_factor=1.0,
)

#: README ball-environment command as env_config (README.md "Ball environment").
BALL_ENV_CONFIG = dict(
    {k: v for k, v in SPACE_ENV_CONFIG.items() if not k.startswith("planet_") and k != "obs_planet_size_per_planet"},
    experiment_name="backup_ball", moving_object_sphere_center=[0, 0, 0.5], moving_object_sphere_radius=2.5,
    moving_object_sphere_height_min_max=[-0.5, 0.5], moving_object_sphere_angle_min_max=[0, 6.2831],
    moving_object_speed_meter_per_second=6.0, moving_object_check_invalid_target_link_point_positions=True,
    moving_object_random_initial_position=True, use_moving_objects=True,
)

#: ... and with the braking-trajectory safety layer of the README's reaching-task commands (README.md:95
#: --check_braking_trajectory_collisions --closest_point_safety_distance=0.01 --collision_check_time=0.033)
def reach_brake_env_config():
    return dict(REACH_ENV_CONFIG, experiment_name="reaching_task_space_braking", check_braking_trajectory_collisions=True)


#: ... with the reward terms that read the braking layer's results (rewards.py:315-334,357-375), thresholds and weights as
#: trained_networks/human_network/params.json sets them (the one shipped config that switches the layer on)
def reach_brake_punish_env_config():
    return dict(reach_brake_env_config(), experiment_name="reaching_task_space_braking_punish", punish_adaptation=True, adaptation_max_punishment=1.0,
                punish_end_min_distance=True, end_min_distance_max_punishment=0.5, end_min_distance_max_threshold=0.05,
                punish_braking_trajectory_min_distance=True, braking_trajectory_min_distance_max_threshold=0.05,
                braking_trajectory_max_punishment=1.0)


#: the target-point reaching task in the space scene: trained_networks/task_networks/reaching_task/space/state_action/
#: params.json without its risk network (risk_config_dir) - SafeMotionsEnv with
#: TargetPointReachingReward, ball machine mounted, 8 s episodes
REACH_ENV_CONFIG = dict(
    SPACE_ENV_CONFIG, experiment_name="reaching_task_space", ball_machine_mode=True, collision_avoidance_mode=False,
    collision_avoidance_kinematic_state_sampling_mode=False, collision_avoidance_kinematic_state_sampling_probability=1.0,
    collision_avoidance_episode_early_termination_punishment=0.0, collision_avoidance_episode_termination_bonus=0.0,
    collision_avoidance_low_acceleration_max_reward=1.0, collision_avoidance_low_acceleration_threshold=0.1,
    collision_avoidance_low_velocity_max_reward=1.0, collision_avoidance_low_velocity_threshold=0.1,
    collision_avoidance_moving_obstacles_max_reward=0.0, collision_avoidance_moving_obstacles_max_reward_distance=0.3,
    collision_avoidance_self_collision_max_reward=0.0, collision_avoidance_static_obstacles_max_reward=0.0,
    use_target_points=True, target_point_cartesian_range_scene=0, target_point_relative_pos_scene=0, target_point_radius=0.065,
    target_point_sequence=0, target_point_reached_reward_bonus=5.0, obs_add_target_point_pos=True,
    obs_add_target_point_relative_pos=True, normalize_reward_to_initial_target_point_distance=True,
    target_point_reward_factor=1.0, trajectory_duration=8.0,
)

# constructor defaults of the reference for the keys this path reads (base.py:85-207, rewards.py:26-54,409-417,
# obs.py:48, actions.py:29-44)
_DEFAULTS = dict(
    pos_limit_factor=1, vel_limit_factor=1, acc_limit_factor=1, jerk_limit_factor=1,
    trajectory_duration=8.0, trajectory_time_step=0.1, robot_scene=0, obstacle_scene=0,
    closest_point_safety_distance=0.1, starting_point_cartesian_range_scene=0, collision_check_time=None,
    use_moving_objects=False, planet_mode=False, planet_one_center=None, planet_one_radius_xy=None,
    planet_one_euler_angles=None, planet_one_period=None, planet_two_center=None, planet_two_radius_xy=None,
    planet_two_euler_angles=None, planet_two_period=None, planet_two_time_shift=None,
    ball_machine_mode=False, terminate_on_self_collision=False, terminate_on_collision_with_static_obstacle=False,
    terminate_on_collision_with_moving_obstacle=False, collision_avoidance_mode=False,
    collision_avoidance_kinematic_state_sampling_mode=False,
    collision_avoidance_kinematic_state_sampling_probability=1.0, collision_avoidance_stay_in_state_probability=0.3,
    collision_avoidance_new_state_sample_time_range=None, target_link_offset=None, target_link_name=None,
    obs_planet_size_per_planet=1, punish_action=False, action_punishment_min_threshold=0.9,
    action_max_punishment=1.0, collision_avoidance_self_collision_max_reward=0.0,
    collision_avoidance_self_collision_max_reward_distance=0.05, collision_avoidance_static_obstacles_max_reward=0.0,
    collision_avoidance_static_obstacles_max_reward_distance=0.1, collision_avoidance_moving_obstacles_max_reward=0.0,
    collision_avoidance_moving_obstacles_max_reward_distance=0.30,
    collision_avoidance_low_acceleration_max_reward=1.0, collision_avoidance_low_acceleration_threshold=0.1,
    collision_avoidance_low_velocity_max_reward=1.0, collision_avoidance_low_velocity_threshold=0.1,
    collision_avoidance_episode_termination_bonus=0, collision_avoidance_episode_early_termination_punishment=-0,
    normalize_reward_to_frequency=False, action_mapping_factor=1.0, action_preprocessing_function=None,
    moving_object_sequence=0, moving_object_sphere_center=None, moving_object_sphere_radius=None,
    moving_object_sphere_height_min_max=None, moving_object_sphere_angle_min_max=None,
    moving_object_speed_meter_per_second=1.0, moving_object_aim_at_current_robot_position=False,
    moving_object_check_invalid_target_link_point_positions=False, moving_object_random_initial_position=False,
    target_point_cartesian_range_scene=0, use_target_points=False, target_point_relative_pos_scene=0,
    target_point_radius=0.05, target_point_sequence=0, target_point_reached_reward_bonus=0.0,
    target_point_use_actual_position=False, obs_add_target_point_pos=False, obs_add_target_point_relative_pos=False,
    normalize_reward_to_initial_target_point_distance=False, target_point_reward_factor=1.0,
    check_braking_trajectory_collisions=False, acc_limit_factor_braking=1.0, jerk_limit_factor_braking=1.0,
    punish_adaptation=False, adaptation_max_punishment=1.0, punish_end_min_distance=False,                     # rewards.py:219-246
    end_min_distance_max_punishment=1.0, end_min_distance_max_threshold=None, punish_braking_trajectory_min_distance=False,
    braking_trajectory_min_distance_max_threshold=None, braking_trajectory_max_punishment=1.0,
)

# features of the reference outside the hot-path scope ("out of scope"/"next")
_UNSUPPORTED_TRUE = ("use_real_robot", "punish_end_max_torque", "punish_braking_trajectory_max_torque",
                     "target_point_use_actual_position",
                     "check_braking_trajectory_torque_limits", "use_gui", "render_video",
                     "risk_check_initial_backup_trajectory", "static_robot",
                     "moving_object_aim_at_current_robot_position", "normalize_reward_to_frequency",
                     "obstacle_use_computed_actual_values", "always_use_collision_avoidance_starting_point_sampling")
_UNSUPPORTED_NOT_NONE = ("human_network_checkpoint",
                         "collision_avoidance_new_state_sample_time_range", "action_preprocessing_function",
                         "target_link_name")


# ---- risk-aware stepping: the env_config keys (train.py:168-309 -> safe_motions_base.py:170-180,222-223,502-585) ----
#: parameters of the env that the BACKUP policy's env_config overrides (safe_motions_base.py:525-538): the risk network and the
#: backup policy were trained in that scene
RISK_OVERWRITE_PARAMETERS = ("planet_mode", "planet_one_center", "planet_one_radius_xy", "planet_one_euler_angles", "planet_one_period",
                             "planet_two_center", "planet_two_radius_xy", "planet_two_euler_angles", "planet_two_period",
                             "planet_two_time_shift", "use_moving_objects", "moving_object_sequence", "moving_object_area_center",
                             "moving_object_area_width_height", "moving_object_sphere_center", "moving_object_sphere_radius",
                             "moving_object_sphere_height_min_max", "moving_object_sphere_angle_min_max",
                             "moving_object_speed_meter_per_second", "moving_object_aim_at_current_robot_position",
                             "moving_object_check_invalid_target_link_point_positions", "moving_object_active_number_single",
                             "ball_machine_mode", "human_network_checkpoint", "human_network_use_full_observation")
RISK_ERROR_PARAMETERS = ("robot_scene", "trajectory_time_step", "acc_limit_factor", "jerk_limit_factor", "pos_limit_factor", "vel_limit_factor")
RISK_WARNING_PARAMETERS = ("obstacle_scene", "closest_point_safety_distance", "starting_point_cartesian_range_scene",
                           "use_controller_target_velocities")
RISK_CHECK_CURRENT_STATE, RISK_CHECK_NEXT_STATE_KINEMATIC_FORECASTING, RISK_CHECK_NEXT_STATE_FULL_FORECASTING = 0, 1, 2   # base.py:73-77
RISK_CHECK_NEXT_STATE_SIMULATE_NEXT_STEP, RISK_CHECK_NEXT_STATE_SIMULATE_NEXT_STEP_AND_BACKUP_TRAJECTORY = 3, 4             # base.py:78-83
RISK_GROUND_TRUTH_EPISODES_PER_FILE = 100                                                                                    # base.py:38


def resolve_risk_config(config, trained_networks_dir=None):
    """``risk_config`` / ``risk_config_dir`` of an env_config -> (env_config with the backup policy's scene parameters put in, risk
    description or None).  Follows SafeMotionsBase._load_risk_config (safe_motions_base.py:502-585): ``risk_config.json`` (plain
    JSON: ``observation_size``, ``action_size`` - null for a network that rates states -, ``checkpoint``) is read from the directory
    (as given, else under ``trained_networks_dir``, the reference's ``safemotions/trained_networks``); the ``params.json`` beside
    the backup policy's checkpoint directory supplies the scene the two networks were trained in, overrides RISK_OVERWRITE_PARAMETERS,
    must agree on RISK_ERROR_PARAMETERS and warns about RISK_WARNING_PARAMETERS.
    ONLY these two JSON files are opened.  The checkpoint itself (a pickle) and the risk network's SavedModel are never read: the
    env built from such a config refuses to step until ``set_risk_networks`` has been given the weights as arrays
    (policy.Mlp.from_npz)."""
    import json
    import warnings
    config = dict(config)
    risk = config.pop("risk_config", None)
    risk_dir = config.pop("risk_config_dir", None)
    if risk is None and risk_dir is None:
        for k in ("risk_state_config", "risk_threshold"):        # without a risk network the reference ignores them too
            config.pop(k, None)
        return config, None
    base = trained_networks_dir or os.environ.get("SMR_TRAINED_NETWORKS_DIR")
    if risk is None:
        if not os.path.isdir(risk_dir) and base:
            risk_dir = os.path.join(base, risk_dir)
        path = os.path.join(risk_dir, "risk_config.json")
        if not os.path.isfile(path):
            raise FileNotFoundError("Could not find config file {}.".format(path))                       # base.py:510-511
        with open(path, "r") as f:
            risk = json.load(f)
    risk = dict(risk)
    checkpoint = risk["checkpoint"]
    if not os.path.isfile(checkpoint) and base:
        checkpoint = os.path.join(base, checkpoint)
    params = os.path.join(os.path.dirname(os.path.dirname(checkpoint)), "params.json")
    if not os.path.isfile(params):
        raise FileNotFoundError("Could not find checkpoint params file {}.".format(params))             # base.py:520-522
    with open(params, "r") as f:
        backup_env_config = json.load(f)["env_config"]
    for k in RISK_OVERWRITE_PARAMETERS:
        if k in backup_env_config:
            config[k] = backup_env_config[k]
    for k in RISK_ERROR_PARAMETERS:
        if k in backup_env_config and _cfg(config, k) != backup_env_config[k]:
            raise ValueError("The {} parameter of the risk network does not match with the environment (risk: {} / env: {}).".format(
                k, backup_env_config[k], _cfg(config, k)))                                               # base.py:548-558
    for k in RISK_WARNING_PARAMETERS:
        if k in backup_env_config and k in config and config[k] != backup_env_config[k]:
            warnings.warn("The {} parameter of the risk network does not match with the environment (risk: {} / env: {}).".format(
                k, backup_env_config[k], config[k]))
    state_config = config.pop("risk_state_config", None)
    state_config = RISK_CHECK_CURRENT_STATE if state_config is None else int(state_config)
    if not RISK_CHECK_CURRENT_STATE <= state_config <= RISK_CHECK_NEXT_STATE_SIMULATE_NEXT_STEP_AND_BACKUP_TRAJECTORY:
        raise ValueError("risk_state_config=%d (safe_motions_base.py:73-83 knows 0 .. 4)" % state_config)
    if risk.get("action_size") is not None and int(risk["action_size"]) != NJ:
        raise ValueError("The action size of the risk network does not match with num_manip_joints")    # base.py:611-614
    threshold = config.pop("risk_threshold", None)
    if config.get("risk_use_backup_agent_for_initial_backup_trajectory_only") not in (None, False):
        raise NotImplementedError("env_config['risk_use_backup_agent_for_initial_backup_trajectory_only'] belongs to the initial-backup-trajectory loop "
                                  "(observations.py:155-185), outside the batched hot path")
    config.pop("risk_use_backup_agent_for_initial_backup_trajectory_only", None)
    # The background copy of the env (safe_motions_base.py:1518-1577,1605-1655): its length, the ground truth it labels, the probability with
    # which a risky rating is ignored.  risk_state_deterministic_backup_trajectory: the copy here always sees the balls / target points
    # the env itself will see (every draw is keyed by seed, env and counter) - the reference's True; with False it draws independent ones.
    steps = config.pop("risk_state_backup_trajectory_steps", None)
    if steps is None:                                        # "trajectory steps as during the training of the backup agent", base.py:573-576
        steps = int(round(float(backup_env_config["trajectory_duration"]) / float(_cfg(config, "trajectory_time_step"))))
    ignore_p = config.pop("risk_ignore_estimation_probability", None)
    ignore_p = 0.0 if ignore_p is None else float(ignore_p)
    if not 0.0 <= ignore_p <= 1.0:
        raise ValueError("risk_ignore_estimation_probability=%r" % ignore_p)
    per_file = config.pop("risk_ground_truth_episodes_per_file", None)
    out = dict(observation_size=None if risk.get("observation_size") is None else int(risk["observation_size"]), action_size=risk.get("action_size"),
               checkpoint=risk["checkpoint"], risk_threshold=0.05 if threshold is None else float(threshold), risk_state_config=state_config)   # base.py:222-223
    extra = dict(backup_trajectory_steps=int(steps), store_ground_truth=bool(config.pop("risk_store_ground_truth", False)), ignore_probability=ignore_p,
                 ground_truth_episodes_per_file=RISK_GROUND_TRUTH_EPISODES_PER_FILE if per_file is None else int(per_file),
                 deterministic_backup_trajectory=bool(config.pop("risk_state_deterministic_backup_trajectory", False)))
    config.pop("risk_state_initial_backup_trajectory_steps", None)      # only read by the initial-backup-trajectory loop (out of scope)
    uses_copy = state_config >= RISK_CHECK_NEXT_STATE_SIMULATE_NEXT_STEP and risk.get("action_size") is None
    if uses_copy or extra["store_ground_truth"] or ignore_p != 0.0:
        out.update(extra)                                    # (kept out of the description of a plain network config: it is compared as a whole)
    return config, out


def _cfg(config, key):
    """config[key], falling back to the reference constructor's default when absent or None."""
    value = config.get(key)
    return _DEFAULTS.get(key) if value is None else value


def validate_env_config(config):
    """Reject reference options the batched path does not implement (loudly, never silently)."""
    if "experiment_name" not in config:
        raise TypeError("__init__() missing 1 required positional argument: 'experiment_name'")  # base.py:86
    for k in _UNSUPPORTED_TRUE:
        if config.get(k):
            raise NotImplementedError("env_config[%r]=%r is outside the batched hot path" % (k, config[k]))
    for k in _UNSUPPORTED_NOT_NONE:
        if config.get(k) is not None:
            raise NotImplementedError("env_config[%r]=%r is outside the batched hot path" % (k, config[k]))
    if not config.get("collision_avoidance_mode", False) and not config.get("use_target_points", False):
        raise NotImplementedError("SafeMotionsEnv (collision_avoidance_mode=False) without use_target_points: its reward "
                                  "would be the action punishment alone")
    if config.get("use_target_points") and _cfg(config, "target_point_sequence") != 0:
        raise NotImplementedError("target_point_sequence != 0 (one robot: TARGET_POINT_SIMULTANEOUS)")
    if _cfg(config, "robot_scene") != 0:
        raise NotImplementedError("robot_scene=%r: only the single iiwa7 (robot_scene=0)" % config["robot_scene"])
    if _cfg(config, "obstacle_scene") not in (0, 5):
        raise NotImplementedError("obstacle_scene=%r: only 0 (none) and 5 (table mesh)" % config["obstacle_scene"])
    if _cfg(config, "action_mapping_factor") != 1.0:
        raise NotImplementedError("action_mapping_factor != 1.0")
    if _cfg(config, "moving_object_sequence") != 0:
        raise NotImplementedError("moving_object_sequence != 0")
    if config.get("planet_mode") and config.get("use_moving_objects"):
        raise NotImplementedError("planet_mode together with use_moving_objects")
    if config.get("check_braking_trajectory_collisions") and config.get("collision_avoidance_mode"):
        # with the braking method the collision-avoidance start-state sampling re-draws velocities / accelerations at the SAME start
        # position until its braking trajectory is collision free (ctlp.py:1499-1545); positions are accepted from 1 mm off the
        # table (ctlp.py:1468-1472) while braking needs closest_point_safety_distance, so that loop need not end.  The reference
        # only documents the braking method with the reaching task (README.md:95: collision_avoidance_mode off, start at rest)
        raise NotImplementedError("check_braking_trajectory_collisions together with collision_avoidance_mode: the reference's "
                                  "start-state sampling (ctlp.py:1499-1545) does not terminate for start positions closer to the "
                                  "table than closest_point_safety_distance; use it with the target-point task (SafeMotionsEnv)")
    if config.get("max_resampling_attempts", 0):
        # base.py:1663 unpacks 4 values from a 2-tuple: the reference itself raises here
        raise ValueError("max_resampling_attempts > 0 is broken in the reference (base.py:1663)")


# --------------------------------------------------------------------------------------------
# A9: planet orbit tables
# --------------------------------------------------------------------------------------------

def quaternion_from_euler(rpy):
    """pybullet.getQuaternionFromEuler: q = qz(yaw) * qy(pitch) * qx(roll), xyzw."""
    r, p, y = [0.5 * float(a) for a in rpy]
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


def quaternion_multiply(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def quaternion_to_matrix(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def planet_orbit_tables(center, radius_xy, euler_angles, period, update_time_step, rotations_per_period,
                        default_orn, orbit_interpolation_steps=1000):
    """Per-update-step pose of a planet: equal arc length per update (ctlp.py:4397-4433).

    Returns (pose[n,7] = global xyz + quaternion xyzw, local_xy[n,2]).
    """
    center = np.asarray(center, dtype=np.float64)
    direction = 1.0
    if period < 0:
        period, direction = -period, -1.0
    theta = np.linspace(0.0, direction * 2 * np.pi, orbit_interpolation_steps)
    local = np.stack([radius_xy[0] * np.cos(theta), radius_xy[1] * np.sin(theta), np.zeros_like(theta)], axis=1)
    seg = np.linalg.norm(np.diff(local, axis=0), axis=1)
    cum = np.concatenate(([0.0], np.cumsum(seg)))
    t = np.arange(0, period, update_time_step)
    idx = np.searchsorted(cum, t / period * cum[-1])
    orn_shift = quaternion_from_euler(euler_angles)
    rot_shift = quaternion_to_matrix(orn_shift)
    pos = center + local[idx] @ rot_shift.T
    n = len(t)
    quat = np.tile(np.asarray(default_orn, dtype=np.float64), (n, 1))
    if rotations_per_period:
        z_rot = direction * t / period * 2 * np.pi * rotations_per_period
        for i in range(n):
            q = quaternion_from_euler([0.0, 0.0, z_rot[i]])
            if tuple(default_orn) != (0, 0, 0, 1):
                q = quaternion_multiply(default_orn, q)
            if tuple(orn_shift) != (0, 0, 0, 1):
                q = quaternion_multiply(orn_shift, q)
            quat[i] = q
    return np.ascontiguousarray(np.concatenate([pos, quat], axis=1)), np.ascontiguousarray(local[idx, :2])


# --------------------------------------------------------------------------------------------
# scene builder
# --------------------------------------------------------------------------------------------

def _bounding_sphere(v):
    c = 0.5 * (v.min(axis=0) + v.max(axis=0))
    return c, float(np.linalg.norm(v - c, axis=1).max())


def ball_target_height_time(initial_height, initial_z_speed, target_height, g=9.81):
    """Ball.get_target_height_time (ctlp.py:4346-4361), continuous time only."""
    sq = initial_z_speed ** 2 + 2 * g * (initial_height - target_height)
    if sq < 0:
        return float("nan")
    t = (initial_z_speed + np.sqrt(sq)) / g
    return t if t > 0 else float("nan")


def _build_ball(config, s, assets, add_piece):
    """Ball scene constants: scene.py:226-257,600, ctlp.py:288-348,706-710,1723-1954."""
    center = config.get("moving_object_sphere_center")
    if center is None:
        raise NotImplementedError("use_moving_objects without moving_object_sphere_center (planar release area)")
    radius = float(config.get("moving_object_sphere_radius") or 5.0)                     # scene.py:247-249
    h = config.get("moving_object_sphere_height_min_max") or [-0.1 * radius, 0.1 * radius]
    ang = config.get("moving_object_sphere_angle_min_max") or [0.0, 2 * np.pi]
    speed = float(_cfg(config, "moving_object_speed_meter_per_second"))
    ball_radius = 0.038                                                                  # scene.py:600
    s.use_ball = 1
    s.ball_radius, s.ball_speed, s.ball_sphere_radius = ball_radius, speed, radius
    s.ball_piece = add_piece(assets["ball_verts"] * (ball_radius / 0.12), _piece_adj(assets, "ball", 0))               # ctlp.py:4043-4044 globalScaling
    s.moving_link_mask = _link_mask(["iiwa_link_%d" % i for i in range(2, 8)])          # scene.py:226-228
    for k in range(NLINK, s.n_bodies):                                                   # + the ball machine, scene.py:229-230
        s.moving_link_mask |= 1 << k
    for c in range(3):
        s.ball_sphere_center[c] = center[c]
    s.ball_sphere_height[0], s.ball_sphere_height[1] = h
    s.ball_sphere_angle[0], s.ball_sphere_angle[1] = ang
    s.ball_high_angle_prob = 1.0                                                         # scene.py:76
    s.ball_check_invalid_target = int(bool(_cfg(config, "moving_object_check_invalid_target_link_point_positions")))
    s.ball_random_initial_position = int(bool(_cfg(config, "moving_object_random_initial_position")))
    for c, (a, b) in enumerate([[-0.4, 0.0], [-0.2, 0.2], [0.0, 0.5]]):                  # ctlp.py:706-710
        s.ball_invalid_min[c], s.ball_invalid_max[c] = a, b
    final = [[-1.0, 1.0], [-1.3, 1.3], [-0.3, 1.5]]                                      # ctlp.py:304-308
    for c in range(3):
        s.ball_final_min[c], s.ball_final_max[c] = final[c]
    # observation ranges, ctlp.py:309-348
    max_initial_height = center[2] + h[1]
    t_up = speed / 9.81
    max_height = max_initial_height + speed * t_up - 0.5 * 9.81 * t_up ** 2
    pos_rng = [[-(center[0] + radius), center[0] + radius], [-(center[1] + radius), center[1] + radius],
               [final[2][0], max_height]]
    zero_t = ball_target_height_time(max_initial_height, -speed, ball_radius)
    min_speed = -speed - 9.81 * zero_t
    vel_rng = [[-speed, speed], [-speed, speed], [min_speed, -min_speed]]
    for c in range(3):
        s.ball_obs_pos_min[c], s.ball_obs_pos_max[c] = pos_rng[c]
        s.ball_obs_vel_min[c], s.ball_obs_vel_max[c] = vel_rng[c]
    s.ball_active_xy_radius = 1.25                                                       # ctlp.py:2830
    _set_target_box(config, s)
    s.ball_target_min_dist_static = float(_cfg(config, "closest_point_safety_distance")) + 0.09   # ctlp.py:1725-1726
    s.ball_target_min_dist_self = float(_cfg(config, "closest_point_safety_distance"))            # ctlp.py:2357


def _set_target_box(config, s):
    """_target_point_cartesian_range, ctlp.py:191-229 (ball targets and target points)"""
    tbox = {0: [[-0.6, 0.6], [-0.8, 0.8], [0.1, 1]], 1: [[-0.6, 0.6], [-0.3, 0.3], [0.1, 1]],
            2: [[-0.4, 0.4], [-0.4, 0.4], [0.1, 1]], 3: [[-0.2, 0.8], [-0.9, 0.9], [0.3, 1.45]],
            4: [[-0.8, 0.8], [-0.9, 0.9], [0.3, 1.45]], 5: [[-0.1, 0.8], [-0.9, 0.9], [0.3, 1.45]],
            6: [[-0.1, 0.8], [-1.0, 1.0], [0.15, 1.65]], 7: [[-0.1, 0.75], [-0.8, 0.8], [1.0, 1.65]],
            8: [[-1.0, 1.0], [-1.0, 1.0], [-0.7, 1.3]], 10: [[-0.1, 0.6], [-0.8, 0.8], [0.1, 1]]
            }.get(_cfg(config, "target_point_cartesian_range_scene"), [[-0.6, 0.6], [-0.8, 0.8], [0.1, 1]])  # ctlp.py:191-229
    for c in range(3):
        s.ball_target_box[c][0], s.ball_target_box[c][1] = tbox[c]


def _piece_adj(assets, family, i):
    """(offsets relative to the piece, neighbour ids) of piece i of an asset family (assets.hull_adjacency)."""
    vo = assets[family + "_offsets"]
    ao = assets[family + "_adj_offsets"][vo[i]:vo[i + 1] + 1]
    return (ao - ao[0]).astype(np.int32), assets[family + "_adj"][ao[0]:ao[-1]].astype(np.int32)


class SceneBundle:
    """Owns the numpy buffers the ``smr_scene_t`` points into."""

    def __init__(self):
        self.struct = Scene()
        self._keep = []
        self.pieces = []
        self.verts = None
        self.env_config = None
        self.risk = None

    def pointer(self):
        return C.byref(self.struct)

    @property
    def obs_size(self):
        return int(self.struct.obs_size)


def joint_limits(config, assets):
    """A5: scene.py:350-371 (URDF +- safety buffer) and :441-455 (factors, jerk = min(2 a/dt, max_jerk))."""
    dt = float(_cfg(config, "trajectory_time_step"))
    pos_lo = (assets["robot_lower"] + JOINT_LIMITS_SAFETY_BUFFER_IIWA) * _cfg(config, "pos_limit_factor")
    pos_hi = (assets["robot_upper"] - JOINT_LIMITS_SAFETY_BUFFER_IIWA) * _cfg(config, "pos_limit_factor")
    vel = assets["robot_velocity"] * _cfg(config, "vel_limit_factor")
    acc = np.array(MAX_ACCELERATION_IIWA) * _cfg(config, "acc_limit_factor")
    jerk = np.array([min(2 * acc[i] / dt, MAX_JERK_IIWA[i]) * _cfg(config, "jerk_limit_factor") for i in range(NJ)])
    return pos_lo, pos_hi, vel, acc, jerk


def _link_mask(names):
    mask = 0
    for n in names:
        mask |= 1 << int(n.rsplit("_", 1)[1])
    return mask


#: largest N the kernels take (csrc/smr_device.h: kMaxChecksPerStep; smr_create rejects more)
MAX_COLLISION_CHECKS_PER_STEP = 24


def build_scene(env_config, collision_checks_per_step=None, contact_distance=BULLET_CONTACT_BREAKING_THRESHOLD,
                auto_reset=True, assets_path=_DATA, trained_networks_dir=None):
    """Build the scene description for ``Env(**env_config)`` of the reference (train.py:620).

    collision_checks_per_step: N samples of the moving-obstacle contact latch per step.  Default: the reference's
    cadence - one contact check after every 1/240 s obstacle update, i.e. ``updates_per_step`` = 24 per step
    (ctlp.py:2631-2637 ``planet_updates_per_check = 1``, ctlp.py:2825-2858 for the ball).  A smaller N (a divisor
    of 24; 1 = end of step only) is an explicit opt-in that trades fidelity for nothing much: the kernels skip
    intermediate samples by a conservative sweep bound, so N = 24 costs about what N = 1 does.
    ``collision_check_time`` does NOT drive N: in the reference it only sets the sampling of the braking-trajectory
    checker (ctlp.py:99-103,3167-3185), which is where this library uses it too (``check_braking_trajectory_collisions``).
    """
    config, risk = resolve_risk_config(env_config, trained_networks_dir)
    validate_env_config(config)
    assets = np.load(assets_path)
    b = SceneBundle()
    b.env_config = config
    b.risk = risk                                    # what the env_config's risk keys asked for (resolve_risk_config), or None
    s = b.struct
    s.version = SCENE_VERSION

    dt = float(_cfg(config, "trajectory_time_step"))
    s.dt = dt
    s.episode_steps = int(round(float(_cfg(config, "trajectory_duration")) / dt))      # tm.py:87,189
    s.updates_per_step = int(round(dt / SIM_TIME_STEP))                                 # base.py:755
    if collision_checks_per_step is None:
        collision_checks_per_step = s.updates_per_step                                  # ctlp.py:2631-2637,2825-2858
    if collision_checks_per_step < 1 or s.updates_per_step % collision_checks_per_step:
        raise ValueError("collision_checks_per_step=%d must divide the %d obstacle updates per step"
                         % (collision_checks_per_step, s.updates_per_step))
    if collision_checks_per_step > MAX_COLLISION_CHECKS_PER_STEP:
        # the kernels pack the sample index into job words and item-queue entries and hold N - 1 items per lane (csrc/smr_device.h:
        # kMaxChecksPerStep); a trajectory_time_step above 0.1 s has more than 24 obstacle updates per step
        raise ValueError("collision_checks_per_step=%d is not supported (at most %d): with trajectory_time_step=%g there are %d obstacle updates "
                         "per step - pass collision_checks_per_step=<a divisor of %d that is <= %d>"
                         % (collision_checks_per_step, MAX_COLLISION_CHECKS_PER_STEP, dt, s.updates_per_step, s.updates_per_step,
                            MAX_COLLISION_CHECKS_PER_STEP))
    s.collision_checks_per_step = int(collision_checks_per_step)
    s.auto_reset = int(bool(auto_reset))

    pos_lo, pos_hi, vel, acc, jerk = joint_limits(config, assets)
    for j in range(NJ):
        s.pos_lo[j], s.pos_hi[j], s.vel_max[j], s.acc_max[j], s.jerk_max[j] = pos_lo[j], pos_hi[j], vel[j], acc[j], jerk[j]
        for c in range(3):
            s.joint_xyz[j][c] = assets["robot_joint_xyz"][j][c]
            s.joint_axis[j][c] = assets["robot_joint_axis"][j][c]
    offset = _cfg(config, "target_link_offset")
    if offset is None:                                                                   # scene.py:196-208
        if config.get("ball_machine_mode"):
            offset = [0, 0, 0.10] if config.get("use_target_points") else [0, 0, 0]
        else:
            offset = [0, 0, 0.126]
    for c in range(3):
        s.base_xyz[c] = assets["robot_base_xyz"][c]
        s.ee_offset[c] = offset[c]

    # ---- geometry: pieces ----
    margin = float(assets["urdf_collision_margin"])
    vert_blocks, pieces = [], []

    adj_blocks = []

    def add_piece(v, adj):
        begin = sum(len(x) for x in vert_blocks)
        vert_blocks.append(np.asarray(v, dtype=np.float64))
        adj_blocks.append(adj)
        c, r = _bounding_sphere(vert_blocks[-1])
        pieces.append([begin, len(v), c, r, margin, PIECE_HULL, 0.0, 0.0])
        return len(pieces) - 1

    # corner i of a box = sign bits (x, y, z) of i: the box edges join corners that differ in one bit
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    box_adj = (np.arange(0, 25, 3, dtype=np.int32), np.array([i ^ b for i in range(8) for b in (1, 2, 4)], dtype=np.int32))

    lo = assets["link_offsets"]
    for k in range(NLINK):
        s.link_piece[k] = add_piece(assets["link_verts"][lo[k]:lo[k + 1]], _piece_adj(assets, "link", k))
        s.body_parent[k] = k
    for k in range(MAX_BODIES):
        for c in range(9):
            s.body_rot[k][c] = 1.0 if c in (0, 4, 8) else 0.0
    s.n_bodies = NLINK
    s.ee_body = NLINK - 1                                                                # iiwa_link_7, scene.py:191-192
    s.n_self_pairs = 0
    ball_machine = bool(config.get("ball_machine_mode"))
    body_names = ["iiwa_link_%d" % i for i in range(NLINK)]
    if ball_machine:
        # robot_ball_machine.urdf: link "ball_machine" (a compound of convex pieces, one body each) fixed on iiwa_link_7;
        # it is the target link (scene.py:188-189), closest_point_active (ctlp.py:585-586) and pairs with the base
        # adapter and iiwa_link_0..5 for self-collisions (ctlp.py:1438-1446)
        bo = assets["bm_offsets"]
        # base adapter: Bullet cylinder of radius r, length L has the implicit core (r - margin, L/2 - margin) plus the margin
        rad, length = [float(x) for x in assets["adapter_radius_length"]]
        adapter = add_piece(assets["adapter_center"] + corners * np.array([rad, rad, 0.5 * length]), box_adj)
        pieces[adapter][5:] = [PIECE_CYLINDER_Z, rad - margin, 0.5 * length - margin]
        pieces[adapter][2] = np.asarray(assets["adapter_center"], dtype=np.float64)
        for i in range(len(bo) - 1):
            k = s.n_bodies
            if k >= MAX_BODIES:
                raise ValueError("too many ball machine pieces")
            s.link_piece[k] = add_piece(assets["bm_verts"][bo[i]:bo[i + 1]], _piece_adj(assets, "bm", i))
            s.body_parent[k] = NLINK - 1
            for c in range(3):
                s.body_xyz[k][c] = assets["bm_joint_xyz"][c]
            for c in range(9):
                s.body_rot[k][c] = assets["bm_joint_rot"].reshape(9)[c]
            for partner in [-(adapter + 1)] + list(range(0, 6)):
                s.self_pair[s.n_self_pairs][0], s.self_pair[s.n_self_pairs][1] = k, partner
                s.n_self_pairs += 1
            body_names.append("ball_machine")
            s.n_bodies += 1
        s.ee_body = NLINK                                                                # both pieces share the link frame

    def body_mask(names):
        return sum(1 << k for k, n in enumerate(body_names) if n in names)

    closest_point_active = ["iiwa_link_%d" % i for i in range(2, 8)]                    # ctlp.py:579-581
    if ball_machine:
        closest_point_active.append("ball_machine")                                      # ctlp.py:585-586
    s.n_static = 0
    s.static_link_mask = 0
    if _cfg(config, "obstacle_scene") == 5:
        # table: Bullet box of half extents h has the implicit core h - margin plus the margin
        half = 0.5 * assets["table_box_size"] - margin
        s.static_piece[0] = add_piece(assets["table_box_center"] + corners * half, box_adj)
        s.n_static = 1
        observed = ["iiwa_link_%d" % i for i in range(1, 8)] + ["ball_machine"]         # ctlp.py:2419-2434
        s.static_link_mask = body_mask([n for n in observed if n in closest_point_active])
        s.table_half_xy[0], s.table_half_xy[1] = 0.6, 0.8                               # ctlp.py:1901-1902
    s.plane_z = -0.94                                                                    # scene.py:172

    # _closest_point_maximum_relevant_distance, ctlp.py:354-366 with rewards.py:81-93
    safety = float(_cfg(config, "closest_point_safety_distance"))
    rmax = -1.0
    if _cfg(config, "collision_avoidance_self_collision_max_reward") != 0:
        rmax = float(_cfg(config, "collision_avoidance_self_collision_max_reward_distance"))
    if _cfg(config, "collision_avoidance_static_obstacles_max_reward") != 0 and \
            float(_cfg(config, "collision_avoidance_static_obstacles_max_reward_distance")) > rmax:
        rmax = float(_cfg(config, "collision_avoidance_static_obstacles_max_reward_distance"))
    ca_mode_ = bool(_cfg(config, "collision_avoidance_mode"))
    for flag, key in (("punish_braking_trajectory_min_distance", "braking_trajectory_min_distance_max_threshold"),
                      ("punish_end_min_distance", "end_min_distance_max_threshold")):         # TargetPointReachingReward, rewards.py:270-300
        if not ca_mode_ and _cfg(config, flag):
            if _cfg(config, key) is None:
                raise ValueError("%s requires %s to be specified" % (flag, key))               # rewards.py:275-282
            rmax = max(rmax, float(_cfg(config, key)))
    if rmax == -1.0:
        s.static_max_distance = safety + 0.002
    else:
        if rmax <= safety:
            raise ValueError("reward_maximum_relevant_distance {} needs to be greater than "
                             "closest_point_safety_distance {}".format(rmax, safety))        # ctlp.py:361-364
        s.static_max_distance = rmax + 0.002

    # ---- moving obstacles ----
    moving_reward = float(_cfg(config, "collision_avoidance_moving_obstacles_max_reward"))
    s.moving_max_distance = 0.001 if moving_reward == 0 else \
        float(_cfg(config, "collision_avoidance_moving_obstacles_max_reward_distance"))  # rewards.py:142-145
    s.contact_distance = float(contact_distance)
    s.contact_link_mask = (1 << s.n_bodies) - 1
    s.n_planets = 0
    s.obs_planet_size_per_planet = int(_cfg(config, "obs_planet_size_per_planet"))
    s.obs_num_planets = 0
    s.use_ball = 0
    obs_size = 3 * NJ                                                                    # obs.py:54,108
    if config.get("planet_mode"):
        s.moving_link_mask = body_mask(["iiwa_link_%d" % i for i in range(3, 8)] + ["ball_machine"])   # scene.py:288-292
        update_time_step = dt / s.updates_per_step                                       # ctlp.py:97
        one_center = config.get("planet_one_center")
        two_center = config.get("planet_two_center")
        one_period = _cfg(config, "planet_one_period") or 5.0                            # scene.py:267-268
        specs = []
        if one_center is not None:                                                       # ctlp.py:726-750
            specs.append(dict(center=one_center,
                              radius_xy=config.get("planet_one_radius_xy") or [0.65, 0.8],
                              euler=config.get("planet_one_euler_angles") if config.get("planet_one_euler_angles")
                              is not None else [0.35, 0, 0],
                              period=one_period, rotations=1, default_orn=quaternion_from_euler([0, 0, -np.pi / 2]),
                              verts=assets["iss_verts"] * 0.6, offsets=assets["iss_offsets"], family="iss", time_shift=None))
        if two_center is not None:                                                       # ctlp.py:751-776
            time_shift = config.get("planet_two_time_shift") if one_center is not None else None  # scene.py:276
            period2 = one_period if time_shift is not None else (config.get("planet_two_period") or 5.0)
            specs.append(dict(center=two_center,
                              radius_xy=config.get("planet_two_radius_xy") or [0.75, 0.8],
                              euler=config.get("planet_two_euler_angles") if config.get("planet_two_euler_angles")
                              is not None else [-0.35, 0, 0],
                              period=period2, rotations=4, default_orn=np.array([0.0, 0.0, 0.0, 1.0]),
                              verts=assets["asteroid_verts"], offsets=assets["asteroid_offsets"], family="asteroid",
                              time_shift=time_shift))
        if len(specs) > MAX_PLANETS:
            raise ValueError("too many planets")
        for p, sp in enumerate(specs):
            pose, local = planet_orbit_tables(sp["center"], sp["radius_xy"], sp["euler"], sp["period"],
                                              update_time_step, sp["rotations"], sp["default_orn"])
            b._keep += [pose, local]
            pl = s.planet[p]
            first = None
            for i in range(len(sp["offsets"]) - 1):
                idx = add_piece(sp["verts"][sp["offsets"][i]:sp["offsets"][i + 1]], _piece_adj(assets, sp["family"], i))
                first = idx if first is None else first
            pl.piece_begin, pl.piece_count = first, len(sp["offsets"]) - 1
            pl.n_orbit = len(pose)
            pl.coupled = 0
            pl.index_shift = 0
            if sp["time_shift"]:                                                         # ctlp.py:778-779 (truthy)
                pl.index_shift = int(pl.n_orbit * sp["time_shift"] / abs(sp["period"]))  # ctlp.py:4435-4437
                pl.coupled = 1
            elif sp["time_shift"] is not None:
                # time_shift == 0: the reference computes index_shift 0 but never couples the planet -> its
                # reset leaves current_time_step_index None (ctlp.py:4473) and update() raises
                raise ValueError("planet_two_time_shift == 0 crashes the reference (ctlp.py:4473,4504)")
            pl.orbit_pose = pose.ctypes.data_as(C.POINTER(C.c_double))
            pl.orbit_local = local.ctypes.data_as(C.POINTER(C.c_double))
            pl.radius_xy[0], pl.radius_xy[1] = sp["radius_xy"][0], sp["radius_xy"][1]
        s.n_planets = len(specs)
        if s.n_planets == 2 and s.planet[1].coupled and s.planet[0].n_orbit != s.planet[1].n_orbit:
            raise ValueError("coupled planets need equal periods")
        s.obs_num_planets = s.n_planets                                                  # obs.py:93-98
        if s.n_planets and config.get("planet_two_time_shift") is not None:
            s.obs_num_planets = 1
        if s.obs_planet_size_per_planet not in (1, 2, 3):
            raise ValueError("obs_planet_size_per_planet must be 1, 2 or 3")
        obs_size += s.obs_num_planets * s.obs_planet_size_per_planet
        for c, (a, bb) in enumerate([[-0.9, 0.7], [-0.85, 0.85], [0.45, 1.2]]):          # scene.py:283-284
            s.planet_obs_pos_min[c], s.planet_obs_pos_max[c] = a, bb
    elif config.get("use_moving_objects"):
        _build_ball(config, s, assets, add_piece)
        obs_size += 6                                                                    # obs.py:79-91 (one robot, sequence 0)
    else:
        s.moving_link_mask = 0

    # ---- target-point reaching task (A14) ----
    ca_mode = bool(config.get("collision_avoidance_mode", False))
    s.collision_avoidance_mode = int(ca_mode)
    s.use_target_points = int(bool(config.get("use_target_points")))
    if s.use_target_points:
        safety_d = float(_cfg(config, "closest_point_safety_distance"))
        _set_target_box(config, s)
        s.obs_target_pos = int(bool(_cfg(config, "obs_add_target_point_pos")))
        s.obs_target_rel = int(bool(_cfg(config, "obs_add_target_point_relative_pos")))
        s.target_normalize = int(bool(_cfg(config, "normalize_reward_to_initial_target_point_distance")))
        s.target_min_dist_static = safety_d + 0.09                                        # ctlp.py:1661-1664
        s.target_min_dist_self = safety_d                                                 # ctlp.py:1665-1666,2220
        s.target_radius = float(_cfg(config, "target_point_radius"))
        s.target_bonus = float(_cfg(config, "target_point_reached_reward_bonus"))
        s.target_factor = float(_cfg(config, "target_point_reward_factor"))
        if _cfg(config, "target_point_relative_pos_scene") != 0:
            raise NotImplementedError("target_point_relative_pos_scene != 0")
        for c, (a, bb) in enumerate(zip([-1.6, -2, -1.5], [1.6, 2, 1.5])):               # ctlp.py:184
            s.target_rel_min[c], s.target_rel_max[c] = a, bb
        obs_size += 3 * (s.obs_target_pos + s.obs_target_rel)                             # obs.py:66-75

    s.obs_size = obs_size

    # ---- reward / termination ----
    s.punish_action = int(bool(_cfg(config, "punish_action")))
    s.terminate_on_self_collision = int(bool(_cfg(config, "terminate_on_self_collision")))
    s.terminate_on_static_collision = int(bool(_cfg(config, "terminate_on_collision_with_static_obstacle")))
    s.terminate_on_moving_collision = int(bool(_cfg(config, "terminate_on_collision_with_moving_obstacle")))
    s.action_punishment_min_threshold = float(_cfg(config, "action_punishment_min_threshold"))
    s.action_max_punishment = float(_cfg(config, "action_max_punishment"))
    s.self_max_reward = float(_cfg(config, "collision_avoidance_self_collision_max_reward"))
    s.self_max_reward_distance = float(_cfg(config, "collision_avoidance_self_collision_max_reward_distance"))
    s.static_max_reward = float(_cfg(config, "collision_avoidance_static_obstacles_max_reward"))
    s.static_max_reward_distance = float(_cfg(config, "collision_avoidance_static_obstacles_max_reward_distance"))
    s.moving_max_reward = moving_reward
    s.moving_max_reward_distance = float(_cfg(config, "collision_avoidance_moving_obstacles_max_reward_distance"))
    s.low_acc_max_reward = float(_cfg(config, "collision_avoidance_low_acceleration_max_reward"))
    s.low_acc_threshold = float(_cfg(config, "collision_avoidance_low_acceleration_threshold"))
    s.low_vel_max_reward = float(_cfg(config, "collision_avoidance_low_velocity_max_reward"))
    s.low_vel_threshold = float(_cfg(config, "collision_avoidance_low_velocity_threshold"))
    s.termination_bonus = float(_cfg(config, "collision_avoidance_episode_termination_bonus"))
    s.early_termination_punishment = float(_cfg(config, "collision_avoidance_episode_early_termination_punishment"))
    s.collision_distance = 0.001                                                         # rewards.py:115,119,153

    # ---- braking-trajectory safety layer (btg.py, ctlp.py:3026-3207) ----
    s.check_braking_trajectory_collisions = int(bool(_cfg(config, "check_braking_trajectory_collisions")))
    cct = _cfg(config, "collision_check_time")
    s.brake_checks_per_step = int(round(max(1, dt / (0.05 if cct is None else cct))))      # ctlp.py:99-103
    s.brake_timeout_steps = int(np.floor(2.0 / dt + 1e-9)) + 1                             # braking_timeout=2.0, ctlp.py:3468-3473
    s.brake_safety_distance = float(_cfg(config, "closest_point_safety_distance"))
    for j in range(NJ):                                                                    # actions.py:76-82
        ab = float(_cfg(config, "acc_limit_factor_braking")) * acc[j]
        s.brake_acc_max[j] = ab
        s.brake_jerk_max[j] = min(2 * float(_cfg(config, "jerk_limit_factor_braking")) * ab / dt, jerk[j])
    s.brake_stop_velocity, s.brake_stop_acceleration = 0.01, 0.01                          # btg.py:45,75-77
    if s.check_braking_trajectory_collisions and s.static_max_distance < s.brake_safety_distance:
        raise ValueError("closest_point_safety_distance beyond the distance the closest-point queries report")
    # the reward terms that read the braking layer's results: TargetPointReachingReward only (rewards.py:229-262)
    reach_reward = not bool(_cfg(config, "collision_avoidance_mode"))
    s.punish_adaptation = int(reach_reward and bool(_cfg(config, "punish_adaptation")))
    s.punish_end_min_distance = int(reach_reward and bool(_cfg(config, "punish_end_min_distance")))
    s.punish_braking_trajectory_min_distance = int(reach_reward and bool(_cfg(config, "punish_braking_trajectory_min_distance")))
    s.adaptation_max_punishment = float(_cfg(config, "adaptation_max_punishment"))
    s.end_min_distance_max_punishment = float(_cfg(config, "end_min_distance_max_punishment"))
    s.braking_trajectory_max_punishment = float(_cfg(config, "braking_trajectory_max_punishment"))
    s.end_min_distance_max_threshold = float(_cfg(config, "end_min_distance_max_threshold") or 0.0)
    s.braking_trajectory_min_distance_max_threshold = float(_cfg(config, "braking_trajectory_min_distance_max_threshold") or 0.0)
    for flag, thr in ((s.punish_end_min_distance, s.end_min_distance_max_threshold),
                      (s.punish_braking_trajectory_min_distance, s.braking_trajectory_min_distance_max_threshold)):
        if flag and thr <= s.brake_safety_distance:
            raise ValueError("the distance threshold of a braking-layer punishment must exceed closest_point_safety_distance")

    # ---- reset ----
    box = {0: [[-0.6, 0.6], [-0.8, 0.8], [0.1, 1]], 1: [[-1.1, 1.1], [-1.1, 1.1], [-0.1, 1.5]],
           2: [[-0.1, 0.6], [-0.8, 0.8], [-0.1, 1.2]]}.get(_cfg(config, "starting_point_cartesian_range_scene"),
                                                             [[-0.6, 0.6], [-0.8, 0.8], [0.1, 1]])  # ctlp.py:172-183
    for c in range(3):
        s.start_box[c][0], s.start_box[c][1] = box[c]
    if ca_mode:
        s.start_min_dist_static = 0.001                                                  # ctlp.py:1468-1472
        s.start_min_dist_self = 0.001
    else:
        s.start_min_dist_static = float(_cfg(config, "closest_point_safety_distance")) + 0.09    # ctlp.py:1473-1477
        s.start_min_dist_self = float(_cfg(config, "closest_point_safety_distance")) + 0.04
    s.kinematic_state_sampling_mode = int(bool(_cfg(config, "collision_avoidance_kinematic_state_sampling_mode")))
    s.kinematic_state_sampling_probability = float(_cfg(config, "collision_avoidance_kinematic_state_sampling_probability"))
    s.stay_in_state_probability = float(_cfg(config, "collision_avoidance_stay_in_state_probability"))

    # ---- finalise geometry buffers ----
    b.verts = np.ascontiguousarray(np.concatenate(vert_blocks), dtype=np.float64)
    arr = (Piece * len(pieces))()
    for i, (begin, count, c, r, m, kind, cyl_r, cyl_h) in enumerate(pieces):
        arr[i].vert_begin, arr[i].vert_count, arr[i].radius, arr[i].margin = begin, count, r, m
        arr[i].kind, arr[i].cyl_radius, arr[i].cyl_half_length = kind, cyl_r, cyl_h
        for k in range(3):
            arr[i].center[k] = c[k]
    b.pieces = arr
    s.verts = b.verts.ctypes.data_as(C.POINTER(C.c_double))
    s.pieces = C.cast(arr, C.POINTER(Piece))
    s.n_verts = len(b.verts)
    # hull adjacency (CSR over all vertices, neighbour ids local to the piece): the support search of the HIP
    # library climbs along it instead of scanning the hull
    offs, ids = [np.zeros(1, dtype=np.int32)], []
    for o, a in adj_blocks:
        offs.append(np.asarray(o[1:], dtype=np.int32) + offs[-1][-1])
        ids.append(np.asarray(a, dtype=np.int32))
    b.adj_offsets = np.ascontiguousarray(np.concatenate(offs), dtype=np.int32)
    b.adj = np.ascontiguousarray(np.concatenate(ids), dtype=np.int32)
    assert len(b.adj_offsets) == s.n_verts + 1
    s.adj_offsets = b.adj_offsets.ctypes.data_as(C.POINTER(C.c_int32))
    s.adj = b.adj.ctypes.data_as(C.POINTER(C.c_int32))
    s.n_pieces = len(pieces)
    return b
