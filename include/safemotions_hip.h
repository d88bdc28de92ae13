/*
 * safemotions_hip.h - C ABI of libsmr_hip.so: the batched, MI355X-native replacement for the
 * per-RL-step hot path of translearn/safeMotionsRisk.
 *
 * The reference has no native boundary on this path: its boundary is the Python gym surface
 *     Env(**env_config)                         train.py:620, base.py:85-207
 *     obs = env.reset()                         observations.py:144-187  -> base.py:913-1022
 *     obs, reward, done, info = env.step(a)     base.py:1043-1227
 * one environment per process.  This library is what a maintainer binds underneath that surface to
 * run B environments per call on one GPU (one environment per wavefront lane); the ctypes stub is
 * safemotionsrisk_amd/lib.py and the gym-style shim is safemotionsrisk_amd/env.py.
 *
 * Conventions
 *  - every function returns 0 on success and a negative SMR_E_* code on failure; nothing throws
 *    across the ABI; smr_last_error() returns the message of the last failure on that handle
 *    (handle == NULL: the last smr_create failure of the calling thread).
 *  - "dev" pointers are device pointers on the handle's GPU, owned by the caller (e.g. torch tensors);
 *    "host" pointers are ordinary host memory.  The library owns all persistent env state.
 *  - all kernels of a handle are enqueued on the hipStream_t given to smr_create (NULL = the legacy
 *    default stream); calls return without synchronising unless documented otherwise.
 *  - a handle is not re-entrant: one call at a time per handle.
 */
#ifndef SAFEMOTIONS_HIP_H
#define SAFEMOTIONS_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "smr_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SMR_ABI_VERSION 10

#define SMR_OK 0
#define SMR_E_INVALID (-1)      /* bad argument / unsupported scene            */
#define SMR_E_HIP (-2)          /* a HIP runtime call failed                   */
#define SMR_E_NO_DEVICE (-3)    /* no usable GPU (the library has no CPU path) */
#define SMR_E_SAMPLING (-4)     /* a rejection-sampling loop of reset / ball launch / target point ran out of attempts (the
                                   reference raises ValueError there, ctlp.py:2127-2136); detected on the device, reported by the
                                   next synchronising call (smr_synchronize, smr_get_state, smr_set_state, smr_get_profile) */

/* columns of the optional per-step info matrix (float32 [B][SMR_INFO_N]); the Python shim turns
 * them into the reference's info dict (base.py:1350-1386, rewards.py:490-498) */
#define SMR_INFO_D_STATIC 0           /* minimum_distance_to_static_obstacles after the 0.001 clamp, rewards.py:115-117 */
#define SMR_INFO_D_SELF 1             /* minimum_distance_self_collision,          rewards.py:119-121: exact below the range the
                                         reference reads - max(self-collision max_reward_distance, 0.001), rewards.py:119-126 -, at or
                                         beyond it the cut-off value ("no closest point", ctlp.py:354-366) */
#define SMR_INFO_D_MOVING 2           /* minimum_distance_to_moving_obstacles,     rewards.py:147-155 */
#define SMR_INFO_ACTION_PUNISHMENT 3  /* rewards.py:164-169 */
#define SMR_INFO_R_SELF 4             /* self_collision_reward,                     rewards.py:123-126 */
#define SMR_INFO_R_STATIC 5           /* static_obstacles_collision_reward,         rewards.py:128-131 */
#define SMR_INFO_R_MOVING 6           /* moving_obstacles_collision_reward,         rewards.py:157-159 */
#define SMR_INFO_EPISODE_LENGTH 7     /* _episode_length of the step just taken,    base.py:1044,1374 */
#define SMR_INFO_BRAKED 8             /* 1: the braking trajectory was executed instead of the action (adaptation_punishment,
                                         ctlp.py:3060,3077-3078); 0 without check_braking_trajectory_collisions */
#define SMR_INFO_TARGET_REACHED 9     /* 1: the target point was reached in this step, ctlp.py:2787-2821 */
#define SMR_INFO_BALL_EVENT 10        /* final state of a ball retired in this step: 1 hit_robot, 2 hit_obstacle, 3 missed_robot
                                         (ctlp.py:2839-2861); 0 none */
#define SMR_INFO_RISKY 11             /* 1: the risk network rated the action risky and the backup policy's action was executed
                                         (risky_action_rate, actions.py:303-340); 0 without risk networks */
#define SMR_INFO_END_MIN_DISTANCE_PUNISHMENT 12   /* end_min_distance_punishment in [0, 1], rewards.py:322-332; 0 without punish_end_min_distance */
#define SMR_INFO_BRAKING_TRAJECTORY_PUNISHMENT 13 /* braking_trajectory_min_distance_punishment in [0, 1], ctlp.py:2393-2415 */
#define SMR_INFO_START_ATTEMPTS 14    /* _starting_point_sampling_attempts of the env's running episode: candidates drawn until its start
                                         position was collision free (ctlp.py:1480-1488) -> obstacles_starting_point_sampling_attempts at
                                         the end of the episode (ctlp.py:1154) */
#define SMR_INFO_N 15

/* kernels timed by smr_get_profile */
#define SMR_K_PLAN 0      /* A2/A4: safe acceleration range + action mapping            */
#define SMR_K_COLLIDE 1   /* A3/A6/A7/A8/A11: cubic samples, FK, GJK distances, contacts     */
#define SMR_K_FINISH 2    /* A1/A12/A13/A15/A16: state update, reward, termination, obs  */
#define SMR_K_RESET 3     /* A17/A9: start-state and planet-phase sampling               */
#define SMR_K_SPAWN 4     /* A10: a new ball for the envs whose ball was retired         */
#define SMR_K_BRAKE 5     /* braking-trajectory safety layer (adapt_action, ctlp.py:3055-3153) */
#define SMR_K_RISK 6      /* risk network + backup policy (actions.py:303-340, base.py:1498-1502,1598-1603) */
/* the kernels of the collide stage where it runs as pair jobs (planet scenes): their times are part of SMR_K_COLLIDE's as well */
#define SMR_K_KEYS 7      /* (env, body) lanes: frames, table bounds / shortcut, candidate keys, sweep of the contact samples */
#define SMR_K_ENUMERATE 8 /* the pairs / contact samples that can matter -> job list */
#define SMR_K_JOBS 9      /* one GJK run per job */
#define SMR_K_COUNT 10

typedef struct smr_env smr_env_t;   /* opaque */

/* Host-side snapshot of the persistent env state (get/set: parity injection and checkpointing; the
 * reference's counterpart is the attribute copy of base.py:1801-1891).  Any pointer may be NULL. */
typedef struct smr_host_state {
    float *q;              /* [B][7] joint position setpoint (_start_position, base.py:1179)     */
    float *v;              /* [B][7] joint velocity setpoint (_start_velocity, base.py:1180)     */
    float *a;              /* [B][7] joint acceleration      (_start_acceleration, base.py:1181) */
    int32_t *planet_idx;   /* [B][2] Planet._current_time_step_index, ctlp.py:4504               */
    int32_t *step_count;   /* [B]    _episode_length, base.py:1044                               */
    int32_t *moving_hit;   /* [B]    latched Planet._collision_detected, ctlp.py:4570-4579       */
    uint32_t *reset_count; /* [B]    resets of the env so far = Philox tag of its next reset            */
    float *ball;           /* [B][10] ball launch point, launch velocity, initial euler angles, spin rate
                              (Ball._base_pos, _initial_speed_vector, _initial_orn_euler, _angular_velocity_vector[1],
                              ctlp.py:4027-4059) */
    int32_t *ball_i;       /* [B][5] _position_update_step_counter, _max_time_update_step_counter, obstacle_hit_time,
                              active flag, balls spawned so far (ctlp.py:4076,4101,1903-1915) */
    float *target;         /* [B][5] target point: position, distance at the last observation, distance when sampled
                              (_target_point_pos_list, _last_target_point_distance_list, _initial_target_point_distance_list,
                              ctlp.py:236-245) */
    uint32_t *target_count;/* [B]    target points sampled so far = Philox tag of the next one */
} smr_host_state_t;

int smr_abi_version(void);
size_t smr_scene_sizeof(void);          /* sizeof(smr_scene_t) as compiled into the library */

/* Env(**env_config) x B.  device: HIP device ordinal.  stream: hipStream_t or NULL.
 * env_index_base: the handle owns the envs [env_index_base, env_index_base + num_envs) of a larger, global batch (one handle per
 * GPU).  Everything an env draws is a pure function of (seed, GLOBAL env index, its counters): the Philox key
 * is the seed, the env word of the counter is env_index_base + e.  Shards therefore form a partition of the global batch - the
 * concatenation of their outputs is bit-identical to one handle over the whole batch, whatever the world size.  0 for a single
 * handle.  (The reference seeds numpy's global generator per process, base.py:1704-1710.) */
int smr_create(const smr_scene_t *scene, int32_t num_envs, uint32_t env_index_base, int32_t device, uint64_t seed, void *stream,
               smr_env_t **out);
int smr_destroy(smr_env_t *env);
const char *smr_last_error(const smr_env_t *env);

int32_t smr_num_envs(const smr_env_t *env);
int32_t smr_obs_size(const smr_env_t *env);

/* env.reset() for the envs whose mask byte is non-zero (mask_dev == NULL: all).  obs_dev float32
 * [B][obs_size] receives the observation of the reset envs (other rows untouched); may be NULL. */
int smr_reset(smr_env_t *env, const uint8_t *mask_dev, float *obs_dev);

/* env.step(action) for all B envs.  actions_dev float32 [B][7] in [-1,1] (base.py:1051);
 * obs_dev float32 [B][obs_size]; reward_dev float32 [B]; done_dev uint8 [B];
 * reason_dev int8 [B] = SMR_TERMINATION_* (-1 while running); info_dev float32 [B][SMR_INFO_N] or NULL.
 * With scene.auto_reset a finished env is re-sampled before the call returns and obs_dev holds the
 * first observation of its next episode (done/reward/reason/info describe the finished step). */
int smr_step(smr_env_t *env, const float *actions_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
             int8_t *reason_dev, float *info_dev);

/* current observation of every env (obs.py:313-351) without stepping */
int smr_observe(smr_env_t *env, float *obs_dev);

int smr_get_state(smr_env_t *env, const smr_host_state_t *host_out);   /* synchronises */
int smr_set_state(smr_env_t *env, const smr_host_state_t *host_in);    /* synchronises */
int smr_synchronize(smr_env_t *env);

/* Per-kernel HIP-event timing (events are recorded on the handle's stream around the launches of a step while
 * enabled; smr_get_profile synchronises, returns accumulated milliseconds and launch counts per
 * SMR_K_* kernel since the last call and clears them).  enable: 0 off, 1 every step, n > 1 every n-th step -
 * every pair of events is a pair of barrier packets in the queue, 65 us per step at 65 536 envs when every
 * launch of every step is timed. */
int smr_set_profiling(smr_env_t *env, int32_t enable);
int smr_get_profile(smr_env_t *env, double ms_out[SMR_K_COUNT], int64_t launches_out[SMR_K_COUNT]);
/* Episodes that ended inside smr_step and were restarted there (auto_reset) since smr_create: what the reference's _episode_counter
 * (base.py:913-916, one increment per reset()) has counted on top of the explicit smr_reset calls.  Synchronises the stream. */
int smr_episode_count(smr_env_t *env, int64_t *auto_resets_out);
/* log_obstacle_data (ctlp.py:1279-1282,2639-2660: LinkData.closest_point_distance_set of the static obstacles): static_dev float32
 * [n_bodies][B] <- the distance of every body to the static obstacles at the END of the last smr_step, as the step computed it: exact
 * below the query range (closest_point_maximum_relevant_distance, 0.102 m), the range itself beyond it - the reference asks with
 * distance = 10 m and after every 1/240 s update; bodies the table cannot reach report the range.  Enqueued on the stream. */
int smr_link_distances(smr_env_t *env, float *static_dev);

/* ---- fully connected networks on the GPU (the backup policy model/keras_fcnet_last_layer_activation.py and the
 * risk network train_risk_network.py:273-282).  Weights come as plain host arrays - a Keras Dense kernel [in][out] row major and
 * its bias per layer; nothing is deserialised here (the reference's checkpoints are pickle files / SavedModels: a maintainer
 * exports the arrays once). ---- */
#define SMR_ACT_LINEAR 0
#define SMR_ACT_TANH 1
#define SMR_ACT_SWISH 2
#define SMR_ACT_RELU 3
#define SMR_ACT_SIGMOID 4
#define SMR_ACT_SELU 5
#define SMR_MLP_MAX_LAYERS 4
#define SMR_MLP_MAX_WIDTH 256

typedef struct smr_mlp smr_mlp_t;   /* opaque */

/* sizes [n_layers + 1]: input width, the layers' output widths (each <= SMR_MLP_MAX_WIDTH); activations [n_layers] = SMR_ACT_*.
 * On failure returns a negative SMR_E_* code and leaves the message in smr_last_error(NULL). */
int smr_mlp_create(int32_t n_layers, const int32_t *sizes, const float *const *weights_host, const float *const *biases_host,
                   const int32_t *activations, int32_t device, smr_mlp_t **out);
int smr_mlp_destroy(smr_mlp_t *mlp);
/* y_dev [n_rows][ldy] <- network(x_dev [n_rows][ldx]); device pointers, enqueued on `stream` (hipStream_t or NULL) */
int smr_mlp_forward(smr_mlp_t *mlp, const float *x_dev, int32_t ldx, float *y_dev, int32_t ldy, int32_t n_rows, void *stream);

/* Risk-aware stepping (actions.py:303-340 with a state-action risk network, base.py:1498-1502,1598-1603): before the action is
 * planned, risk = risk_network([risk observation, action]); where risk >= threshold the deterministic action of the backup
 * policy for the risk observation (the first 7 outputs) replaces it.  The risk observation is the observation without the
 * target-point columns (obs.py:419-431).  risk: input obs_size - target columns + 7, one output; backup: input obs_size - target
 * columns, >= 7 outputs.  Pass NULL networks to switch it off (risk alone may be NULL for SMR_RISK_SIMULATE_BACKUP_TRAJECTORY, which
 * rates by simulation only).  The handles must outlive the env's use of them. */
int smr_set_risk_networks(smr_env_t *env, smr_mlp_t *risk, smr_mlp_t *backup, float risk_threshold);
/* A risk network WITHOUT an action input (risk_config.json "action_size": null; its input is the risk observation alone) rates a
 * state; risk_state_config (base.py:73-83,174) says which: */
#define SMR_RISK_STATE_ACTION (-1)    /* (not a risk_state_config: the mode in force when the network takes the action) */
#define SMR_RISK_CURRENT_STATE 0      /* RISK_CHECK_CURRENT_STATE: the current risk observation, base.py:1504-1506 */
#define SMR_RISK_NEXT_KINEMATIC 1     /* RISK_CHECK_NEXT_STATE_KINEMATIC_FORECASTING: the next step's kinematic columns for the requested
                                         action, the rest as now (base.py:1507-1517, obs.py:192-231) */
#define SMR_RISK_NEXT_FULL 2          /* RISK_CHECK_NEXT_STATE_FULL_FORECASTING: the planets / the ball one step on as well */
#define SMR_RISK_SIMULATE_NEXT_STEP 3  /* RISK_CHECK_NEXT_STATE_SIMULATE_NEXT_STEP: a background copy of the env is stepped with the action; a
                                         collision there is risk 1, else the network rates the copy's risk observation (base.py:1520-1556) */
#define SMR_RISK_SIMULATE_BACKUP_TRAJECTORY 4 /* RISK_CHECK_NEXT_STATE_SIMULATE_NEXT_STEP_AND_BACKUP_TRAJECTORY: ... then `backup_steps` actions of
                                         the backup policy; risk 1 if the copy ends in a collision, else 0 - no risk network (base.py:1529-1577) */
/* (3 and 4 need the background copy: smr_set_risk_rollout) */
int smr_set_risk_state_config(smr_env_t *env, int32_t risk_state_config);
/* The background copy of the env - the reference's "backup client" (switch_to_backup_client / switch_back_to_main_client,
 * base.py:1801-1842; its uses base.py:1518-1577 and 1605-1650).  `shadow` is a second handle of the SAME scene with auto_reset = 0 and
 * episode_steps = backup_steps + 1 (the copy's trajectory_length = backup_steps + 2, base.py:1819), created by the caller with the same
 * num_envs, seed, env_index_base, device and stream; smr_step(env) then, on the device and before the action is planned:
 *   copies the envs' state into the shadow (step counter 0), steps the shadow with the requested action and - mode 4, or ground truth -
 *   `backup_steps` more times with the backup policy's action for the shadow's risk observation, and notes per env whether the copy
 *   ended in a collision (a termination other than the trajectory length); the env itself is not touched.
 * Every random draw of the library is keyed by (seed, env, counter), so the copy sees the balls and target points the env itself will
 * see: the reference's risk_state_deterministic_backup_trajectory = True; with False the reference draws independent ones.
 * ignore_probability: risk_ignore_estimation_probability, base.py:1652-1655 - a risky rating is dropped with this probability (a draw
 *   keyed by the env's reset and step counters).
 * Ground truth (risk_store_ground_truth, base.py:1567-1586,1605-1650; NULL: off), written by every smr_step, rows float32:
 *   gt_state_action_dev [B][R + 7 + 3]: risk observation before the step | requested action | risk label | risk prediction | valid
 *       (mode 4: the reference's "state_action_risk" table; modes -1 .. 3: its single table - state-risk networks of modes 1 - 3 carry
 *       the copy's risk observation after the first simulated step instead, valid only if that step did not end the copy's episode)
 *   gt_state_dev [B][R + 2] (mode 4 only, "state_risk"): the copy's risk observation after the first simulated step | risk label | valid
 * R = obs_size - target-point columns.  shadow = NULL switches the copy off (modes 3 / 4 then fail at smr_step). */
int smr_set_risk_rollout(smr_env_t *env, smr_env_t *shadow, int32_t backup_steps, float ignore_probability,
                         float *gt_state_action_dev, float *gt_state_dev);

/* ---- component entry points (parity tests): each runs the same device functions the step kernels
 * use on n host-provided inputs and copies the results back; all synchronise ---- */
/* A6: link frames of iiwa_link_0..7; pos_out [n][8][3], rot_out [n][8][9] row-major */
int smr_test_fk(smr_env_t *env, const float *q_host, int32_t n, float *pos_out, float *rot_out);
/* A4: per joint [lo, hi] and violation code; p/v/a/lo/hi [n][7], code [n][7] */
int smr_test_safe_acc_range(smr_env_t *env, const float *p_host, const float *v_host, const float *a_host,
                            int32_t n, float *lo_out, float *hi_out, int32_t *code_out);
/* A7/A8/A11: out [n][4] = d_static, d_self, d_moving (no latch), contact flag; planet_idx [n][2] */
int smr_test_distances(smr_env_t *env, const float *q_host, const int32_t *planet_idx_host, int32_t n,
                       float *out);
/* ball scene: the same four values against a ball given by ball_host [n][10] (layout of smr_host_state_t.ball)
 * after k_host[n] updates */
int smr_test_ball_distances(smr_env_t *env, const float *q_host, const float *ball_host, const int32_t *k_host,
                            int32_t n, float *out);

/* self-collision pairs: the table lower bounds the step's pruning uses (csrc/smr_device.h "Self pairs: bound tables") for n joint
 * vectors; out [n][SMR_MAX_SELF_PAIRS], -3e38 for a pair without a table.  Each must not exceed the pair's distance (ctlp.py:3345-3372). */
int smr_test_self_bounds(smr_env_t *env, const float *q_host, int32_t n, float *out);
/* braking layer: what the LAST smr_step planned - acc_out [brake_timeout_steps + 1][7][B] the braking accelerations of the segments
 * behind the planned step ([0]: the one that replaces a refused action, btg.py:44-93), steps_out [B] braking segments until
 * standstill (-1: braking timeout), braked_out [B] the decision (ctlp.py:3026-3053).  Any pointer may be NULL.  Synchronises. */
int smr_test_brake_plan(smr_env_t *env, float *acc_out, int32_t *steps_out, int32_t *braked_out);

#ifdef __cplusplus
}
#endif
#endif /* SAFEMOTIONS_HIP_H */
