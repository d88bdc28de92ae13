/*
 * smr_scene.h - plain-C description of one SafeMotions scene (INPUT DATA layout, no code).
 *
 * The host (safemotionsrisk_amd/scene.py) fills this struct from the reference's env_config keys
 * (train.py:168-309, trained_networks/backup_networks/<scene>/params.json) and from the geometry tables
 * compiled out of the reference's URDF/mesh assets (safemotionsrisk_amd/assets.py).  It is the one
 * argument of smr_create() (include/safemotions_hip.h).  All numbers are float64 here; the HIP
 * library converts to float32 device tables when the handle is created.
 *
 * Each field cites the reference line that defines its meaning.  Short names:
 *   base.py = safemotions/envs/safe_motions_base.py      ctlp.py = robot_scene/collision_torque_limit_prevention.py
 *   scene.py = robot_scene/robot_scene_base.py           tm.py = utils/trajectory_manager.py
 *   rewards.py / obs.py / actions.py = envs/decorators/<name>.py
 */
#ifndef SMR_SCENE_H
#define SMR_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMR_SCENE_VERSION 8
#define SMR_NJ 7          /* manipulator joints (iiwa_joint_1..7, robot.urdf)            */
#define SMR_NLINK 8       /* chain links iiwa_link_0..7; link k moves with joints 1..k      */
#define SMR_MAX_BODIES 10 /* collision bodies = the 8 chain links + convex pieces rigidly mounted on one of them
                             (ball machine on iiwa_link_7, robot_ball_machine.urdf); one convex piece per body */
#define SMR_MAX_SELF_PAIRS 16
#define SMR_MAX_PLANETS 2
#define SMR_MAX_STATIC 4

/* termination reasons, base.py:63-70 */
#define SMR_TERMINATION_UNSET (-1)
#define SMR_TERMINATION_JOINT_LIMITS 1
#define SMR_TERMINATION_TRAJECTORY_LENGTH 2
#define SMR_TERMINATION_SELF_COLLISION 3
#define SMR_TERMINATION_COLLISION_WITH_STATIC_OBSTACLE 4
#define SMR_TERMINATION_COLLISION_WITH_MOVING_OBSTACLE 5

/* One convex piece = convex hull of verts[vert_begin .. vert_begin+vert_count) inflated by `margin`
 * (Bullet: btConvexHullShape + collision margin).  center/radius: a bounding sphere of the
 * un-inflated hull in the same frame as the vertices, used only to cull far pairs. */
#define SMR_PIECE_HULL 0
#define SMR_PIECE_CYLINDER_Z 1   /* Bullet btCylinderShapeZ: axis z through `center`, core radius / half length below; its
                                    vertices are the 8 corners of its bounding box (bounds only, not the support) */
typedef struct smr_piece {
    int32_t vert_begin;
    int32_t vert_count;
    double center[3];
    double radius;
    double margin;
    int32_t kind;             /* SMR_PIECE_* */
    int32_t _pad;
    double cyl_radius;        /* core (margin already subtracted) */
    double cyl_half_length;
} smr_piece_t;

typedef struct smr_planet {
    int32_t piece_begin;      /* pieces[piece_begin .. +piece_count) in the planet's body frame   */
    int32_t piece_count;
    int32_t n_orbit;          /* Planet.num_time_steps, ctlp.py:4412-4416,4631 (1200)               */
    int32_t index_shift;      /* ctlp.py:4435-4437; 0 and coupled=0 for an independent planet        */
    int32_t coupled;          /* 1: index follows planet 0 (+index_shift), ctlp.py:4481-4485          */
    int32_t _pad;
    const double *orbit_pose; /* n_orbit x 7: global pos xyz, orientation quat xyzw, ctlp.py:4415-4433 */
    const double *orbit_local;/* n_orbit x 2: orbit-plane x,y, ctlp.py:4416 (obs.py:280-288)           */
    double radius_xy[2];      /* obs.py:283-288                                                       */
} smr_planet_t;

typedef struct smr_scene {
    int32_t version;                    /* SMR_SCENE_VERSION */
    int32_t obs_size;                   /* obs.py:108-110 */

    /* ---- time ---- */
    double dt;                          /* trajectory_time_step, base.py:110 */
    int32_t episode_steps;              /* trajectory_length - 1 = round(T/dt), tm.py:87,189 */
    int32_t updates_per_step;           /* round(dt / (1/240)) = 24, base.py:755 */
    int32_t collision_checks_per_step;  /* N samples of the moving-obstacle contact test per step,
                                           must divide updates_per_step (24 = ctlp.py:2632-2637
                                           cadence, 3 = ctlp.py:103 cadence, 1 = end of step only) */
    int32_t auto_reset;                 /* re-sample an env inside step() when it is done */

    /* ---- kinematic limits (A5), scene.py:20-22,350-371,441-455 ---- */
    double pos_lo[SMR_NJ], pos_hi[SMR_NJ];
    double vel_max[SMR_NJ], acc_max[SMR_NJ], jerk_max[SMR_NJ];

    /* ---- kinematic chain (A6), description/urdf/robot.urdf ---- */
    double base_xyz[3];                 /* world -> iiwa_link_0 */
    double joint_xyz[SMR_NJ][3];        /* parent link frame -> joint k origin (all rpy = 0) */
    double joint_axis[SMR_NJ][3];
    double ee_offset[3];                /* target_link_offset in iiwa_link_7, scene.py:196-208 */

    /* ---- geometry ---- */
    const double *verts;                /* n_verts x 3 */
    const smr_piece_t *pieces;
    const int32_t *adj_offsets;         /* n_verts + 1: CSR over all vertices of the hull adjacency (edges of the
                                           triangulated convex hull of each piece, safemotionsrisk_amd/assets.py) */
    const int32_t *adj;                 /* neighbour vertex ids, LOCAL to the vertex's piece; a piece has <= 256 vertices */
    int32_t n_verts;
    int32_t n_pieces;
    int32_t n_bodies;                   /* 8, or 8 + pieces of the ball machine (ball_machine_mode, scene.py:171) */
    int32_t link_piece[SMR_MAX_BODIES]; /* piece (body frame) of body k; bodies 0..7 = iiwa_link_k */
    int32_t body_parent[SMR_MAX_BODIES];/* chain link the body is rigidly attached to (k itself for k < 8) */
    int32_t ee_body;                    /* body whose frame carries ee_offset (target link, scene.py:185-208) */
    double body_xyz[SMR_MAX_BODIES][3]; /* parent link frame -> body frame (identity for k < 8) */
    double body_rot[SMR_MAX_BODIES][9];
    /* self-collision pairs (A7 self), ctlp.py:1398-1460,3345-3372: (body, partner); partner >= 0: body index,
     * partner < 0: world-frame piece -(partner + 1) (the base adapter's cylinder) */
    int32_t n_self_pairs;
    int32_t self_pair[SMR_MAX_SELF_PAIRS][2];

    /* static obstacles (A7): pieces given in the world frame */
    int32_t n_static;
    int32_t static_piece[SMR_MAX_STATIC];
    uint32_t static_link_mask;          /* bit k: iiwa_link_k is observed by the table and
                                           closest_point_active, ctlp.py:579-581,2419-2434,3297-3298 */
    double static_max_distance;         /* _closest_point_maximum_relevant_distance, ctlp.py:354-366 */

    /* moving obstacles (A8-A11) */
    uint32_t moving_link_mask;          /* scene.py:226-228 (ball) / :288-290 (planets) */
    uint32_t contact_link_mask;         /* links whose contact latches a hit, ctlp.py:4188,4572 */
    double moving_max_distance;         /* rewards.py:142-145 */
    double contact_distance;            /* separation below which Bullet reports a contact point
                                           (contact breaking threshold, 0.02 m) */
    int32_t n_planets;
    int32_t obs_planet_size_per_planet; /* obs.py:48,275-292 */
    int32_t obs_num_planets;            /* obs.py:93-98 */
    int32_t _pad1;
    smr_planet_t planet[SMR_MAX_PLANETS];
    double planet_obs_pos_min[3], planet_obs_pos_max[3];   /* scene.py:283-284 */

    /* ball (A10), ctlp.py:1723-1954,4001-4361 */
    int32_t use_ball;
    int32_t ball_piece;                 /* body-frame piece, already scaled by radius/0.12 */
    double ball_radius;                 /* scene.py:600 */
    double ball_speed;                  /* moving_object_speed_meter_per_second */
    double ball_sphere_center[3];       /* moving_object_sphere_center */
    double ball_sphere_radius;
    double ball_sphere_height[2];
    double ball_sphere_angle[2];
    double ball_high_angle_prob;        /* scene.py:76 */
    int32_t ball_check_invalid_target;  /* moving_object_check_invalid_target_link_point_positions */
    int32_t ball_random_initial_position;
    double ball_invalid_min[3], ball_invalid_max[3];       /* ctlp.py:706-710 */
    double ball_final_min[3], ball_final_max[3];           /* ctlp.py:304-308 */
    double ball_obs_pos_min[3], ball_obs_pos_max[3];       /* ctlp.py:309-333 */
    double ball_obs_vel_min[3], ball_obs_vel_max[3];       /* ctlp.py:343-348 */
    double ball_active_xy_radius;       /* 1.25, ctlp.py:2830 */
    double ball_target_box[3][2];       /* _target_point_cartesian_range, ctlp.py:191-229 */
    double ball_target_min_dist_static; /* ctlp.py:1726 */
    double ball_target_min_dist_self;   /* closest_point_safety_distance, ctlp.py:2357 */
    double plane_z;                     /* scene.py:172 */
    double table_half_xy[2];            /* ctlp.py:1901-1902 */

    /* ---- target-point reaching task (A14), ctlp.py:1658-1721,2210-2350,2787-2821, rewards.py:217-396, obs.py:313-335 ----
     * target points are sampled in ball_target_box (= _target_point_cartesian_range) */
    int32_t use_target_points;
    int32_t obs_target_pos, obs_target_rel;      /* obs_add_target_point_pos / _relative_pos */
    int32_t target_normalize;                    /* normalize_reward_to_initial_target_point_distance */
    double target_min_dist_static;               /* safety + 0.09, ctlp.py:1661-1664 */
    double target_min_dist_self;                 /* safety, ctlp.py:1665-1666,2220 */
    double target_radius, target_bonus, target_factor;
    double target_rel_min[3], target_rel_max[3]; /* _target_point_relative_pos_min_max, ctlp.py:184 */

    /* ---- reward (A12/A13), rewards.py ---- */
    int32_t collision_avoidance_mode;   /* 1: CollisionAvoidanceReward + collision-avoidance start states (rewards.py:399-502,
                                           ctlp.py:1490-1654); 0: TargetPointReachingReward, start at rest (rewards.py:217-396) */
    int32_t _pad3;
    int32_t punish_action;
    int32_t terminate_on_self_collision;
    int32_t terminate_on_static_collision;
    int32_t terminate_on_moving_collision;
    double action_punishment_min_threshold, action_max_punishment;
    double self_max_reward, self_max_reward_distance;
    double static_max_reward, static_max_reward_distance;
    double moving_max_reward, moving_max_reward_distance;
    double low_acc_max_reward, low_acc_threshold;
    double low_vel_max_reward, low_vel_threshold;
    double termination_bonus, early_termination_punishment;
    double collision_distance;          /* 0.001, rewards.py:115,119,153 */

    /* ---- braking-trajectory safety layer: btg.py:44-93, ctlp.py:3026-3207,3468-3506, adapt_action :3055-3153 ----
     * An action is executed only if the step it commands, followed by braking to a standstill, keeps every observed link at least
     * brake_safety_distance from the static obstacles and the self-collision partners at brake_checks_per_step samples per step;
     * otherwise the braking acceleration replaces it. */
    int32_t check_braking_trajectory_collisions;   /* base.py:122 */
    int32_t brake_checks_per_step;      /* _collision_checks_per_time_step = round(max(1, dt / collision_check_time)), ctlp.py:99-103 */
    int32_t brake_timeout_steps;        /* first number of braking steps m with m dt > braking_timeout = 2.0 s, ctlp.py:3468-3473 */
    int32_t _pad4;
    double brake_safety_distance;       /* closest_point_safety_distance, ctlp.py:3307,3364 */
    double brake_acc_max[SMR_NJ];       /* acc_limit_factor_braking x max accelerations, actions.py:76-77 */
    double brake_jerk_max[SMR_NJ];      /* actions.py:78-82 */
    double brake_stop_velocity;         /* 0.01: below it (and brake_stop_acceleration) a joint counts as stopped, btg.py:45,75-77 */
    double brake_stop_acceleration;
    /* the reward terms of TargetPointReachingReward that read the braking layer's results (rewards.py:229-262,315-334,357-375);
     * each punishment in [0, 1] is also an info column */
    int32_t punish_adaptation;          /* - adaptation_max_punishment where the braking trajectory was executed instead of the action
                                           (adaptation_punishment, ctlp.py:3060,3077-3078; rewards.py:319-320,356) */
    int32_t punish_end_min_distance;    /* - end_min_distance_max_punishment x quadratic(end_min_distance): the minimum distance (static
                                           obstacles and self-collision partners, ctlp.py:3282-3374) at the END of the executed step
                                           (the first entry of the valid braking trajectory's min_distance list, ctlp.py:3099-3102,3194;
                                           rewards.py:322-332) */
    int32_t punish_braking_trajectory_min_distance;   /* - braking_trajectory_max_punishment x quadratic(the smallest minimum distance
                                           over the samples of this step's braking-trajectory check, 1 below the safety distance)
                                           (get_braking_trajectory_punishment, ctlp.py:2393-2415; rewards.py:361-375) */
    int32_t _pad5;
    double adaptation_max_punishment;
    double end_min_distance_max_punishment, end_min_distance_max_threshold;
    double braking_trajectory_max_punishment, braking_trajectory_min_distance_max_threshold;

    /* ---- reset (A17), ctlp.py:1461-1656,1980-2138 ---- */
    double start_box[3][2];             /* ctlp.py:172-183 */
    double start_min_dist_static;       /* ctlp.py:1468-1477 */
    double start_min_dist_self;
    int32_t kinematic_state_sampling_mode;
    int32_t _pad2;
    double kinematic_state_sampling_probability;
    double stay_in_state_probability;
} smr_scene_t;

#ifdef __cplusplus
}
#endif
#endif /* SMR_SCENE_H */
