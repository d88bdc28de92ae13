// smr_device.h - device-side tables and device functions of libsmr_hip.so (gfx950 / CDNA4 only).
//
// One environment per wavefront lane.  Everything that is shared by the whole batch (limits, chain,
// hull vertices, piece table, planet orbit tables) lives in read-only device memory and is indexed
// with wave-uniform indices, so the compiler fetches it with scalar loads (s_load_dwordx4/x8) into
// SGPRs and the inner loops are pure VALU on per-lane registers; per-env state is SoA in HBM.
//
// Reference short names as in include/smr_scene.h.
#pragma once

#ifndef SMR_HOST_EMUL          // tests/host_emul compiles these device functions for the host (debugging only)
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "../../include/safemotions_hip.h"

#ifndef SMR_CNT
#define SMR_CNT(i)            // work counters of the host emulation build (tests/host_emul)
#define SMR_CNT_SLOT(i)
#endif
#ifndef SMR_DEV_COUNT         // cycle / count slots of the profiling build (smr_hip.hip, SMR_PROFILE_ROWS)
#define SMR_DEV_CLOCK() 0ull
#define SMR_DEV_COUNT(i, n)
#define SMR_DEV_MAX(i, n)
#define SMR_WAVE_REC(f, n)
#endif

namespace smr {

// Batch-shared read-only tables are read through the constant address space: with a wave-uniform
// index the compiler then emits s_load_* (scalar cache -> SGPRs) instead of per-lane flat loads.
// (A pointer fetched from a struct is a generic pointer, and LLVM treats generic loads as divergent.)
#ifndef SMR_HOST_EMUL
#define SMR_CONST __attribute__((address_space(4)))
#define SMR_GLOBAL __attribute__((address_space(1)))
#endif

// ---------------------------------------------------------------------------------------------
// tables (float32 images of smr_scene_t, built by smr_create)
// ---------------------------------------------------------------------------------------------
struct DPiece {
    int begin, count;
    float cx, cy, cz, radius;   // bounding sphere of the hull, piece frame
    float margin;
    float hx, hy, hz;           // half extents of the hull's AABB about (cx,cy,cz), piece frame
    float in[3][3];             // three points INSIDE the hull (means of vertex subsets: all / upper / lower half along
                                // the longest box axis): any pair of interior points bounds the hull distance from above
    int ids_begin, ids_count;   // the piece's neighbour-id groups in DScene::ids4 (hill-climbing support)
    int kind;                   // SMR_PIECE_HULL / SMR_PIECE_CYLINDER_Z (axis z through (cx,cy,cz), core cyl_r x cyl_h) /
                                // kPieceAlignedBox (found by the table builder: a hull that is an axis-aligned box)
    float cyl_r, cyl_h;
    float bx0, bx1, by0, by1, btop;   // kPieceAlignedBox: footprint [bx0,bx1] x [by0,by1] and the z of the top face (piece frame)
    float cap_a[3], cap_b[3], cap_r;  // bounding capsule of the hull (segment a-b, radius), piece frame: links and the planets'
                                      // pieces are elongated, so capsule-capsule bounds cull twice the pairs spheres do
};
constexpr int kPieceDwords = 36;
constexpr int kPieceAlignedBox = 100;
constexpr int kMaxPlanetPieces = 28;       // lane-autonomous planet queries: one candidate key byte per piece in 7 registers
constexpr int kMaxPlanetVerts = 512;       // ... and the LDS budget for the planets' vertices
constexpr int kPlanetDescDwords = 8;
constexpr float kKeyStep = 0.0025f;        // candidate key = floor(lower bound / kKeyStep), 255 = done / no candidate
constexpr int kStartBins = 8;                                // cube-map resolution of the climb start table
constexpr int kStartCells = 6 * kStartBins * kStartBins;     // entries per piece
// the reset-side kernels (reset, spawn, target: smr_hip.hip) climb over the hulls of ALL bodies, a different one per lane: one LDS
// pool per block with every body's tables (body_pool_tables, smr_tables.h), kBodyPoolHead dwords of offsets per body in front
constexpr float kSelfTabStep = 4.0e-6f;                      // self-pair bound tables: uint16 entries in units of 4 um (0 .. 0.262 m)
constexpr float kSelfTabMax = 0.262f;
constexpr int kBodyPoolHead = 4;                             // dword offsets into the pool: vertices, id groups, start map, DPiece
constexpr int kBodyPoolDwords = 15360;                       // 60 KB: two 256-lane blocks per CU

struct DPlanet {
    int piece_begin, piece_count, n_orbit, index_shift, coupled;
    float radius_x, radius_y;
    float bx, by, bz, bradius;  // bounding sphere of the whole planet (all pieces, margins included), planet frame
    const float *orbit_pose;    // [n_orbit][8]: pos xyz, quat xyzw, pad
    const float *orbit_local;   // [n_orbit][2]
};

struct DScene {
    int B, obs_size, episode_steps, updates_per_step, checks_per_step, auto_reset;
    float dt;
    float pos_lo[SMR_NJ], pos_hi[SMR_NJ], vel_max[SMR_NJ], acc_max[SMR_NJ], jerk_max[SMR_NJ];
    float pos_lo_res[SMR_NJ], pos_hi_res[SMR_NJ], vel_max_res[SMR_NJ];   // float64 limit minus its float32 image
    float base[3], joint_xyz[SMR_NJ][3], joint_axis[SMR_NJ][3], ee_offset[3];
    const float4 *verts;
    const DPiece *pieces;
    const uint32_t *span;      // per vertex (indexed like verts): first neighbour-id group relative to the piece | groups << 16
    const uint32_t *ids4;      // groups of 4 neighbour ids, one byte each, local to the piece
    const uint8_t *start;      // kStartCells per piece: support vertex of each cube-map cell centre
    int n_bodies;                         // 8 chain links + pieces rigidly mounted on one (ball machine on iiwa_link_7)
    int link_piece[SMR_MAX_BODIES];
    int body_parent[SMR_MAX_BODIES], ee_body;
    float body_xyz[SMR_MAX_BODIES][3], body_rot[SMR_MAX_BODIES][9];   // parent link frame -> body frame
    int n_self_pairs, self_pair[SMR_MAX_SELF_PAIRS][2];               // (body, partner body | -(world piece + 1))
    // the reset-side kernels deal the pairs to the lanes of a group by PARTNER: lane g (= chain link g) tests the pairs whose partner is
    // link g - its own frame and piece record are at hand -, pairs with a world-fixed partner go to the lanes with the fewest
    int pair_lane[8][4], pair_lane_max;                                // pair indices (-1: none), and the longest list
    const uint32_t *body_pool;            // image of the reset-side kernels' LDS pool (body_pool_tables)
    int body_pool_dwords;
    // pair-job path of the collide stage ("Pair jobs" below): on / off, and the LDS pool of its first / jobs kernels - like body_pool,
    // with the hull tables of the bodies that take part in moving / contact / static queries only
    int planets_jobs, job_pool_dwords;
    int ball_jobs, ball_jobs_vn;          // ball scenes (round 4): the ball's GJK runs are jobs too; its padded vertex count in the job kernels' LDS
    const uint32_t *job_pool;
    // self-collision pairs ("Self pairs" below): jobs_on - the job lists exist (pair jobs and / or self-pair jobs); self_cut - the
    // distance range the reward and the termination read (rewards.py:119-126): a self distance at or above it is "none"; the bound
    // tables of the pairs whose relative pose depends on the last two or three joints only
    int jobs_on;
    float self_cut;
    int self_tab_nd[SMR_MAX_SELF_PAIRS];        // 0: the pair has no table; else its axes are the joints 7 - nd .. 6
    int self_tab_off[SMR_MAX_SELF_PAIRS];       // first entry in self_tab
    int self_tab_n[SMR_MAX_SELF_PAIRS][3];      // cells per axis
    float self_tab_h[SMR_MAX_SELF_PAIRS][3], self_tab_ih[SMR_MAX_SELF_PAIRS][3];   // cell width and its inverse
    float self_tab_lever[SMR_MAX_SELF_PAIRS][3];  // no point of the body moves farther than lever x |dq| relative to the partner link
    const uint16_t *self_tab;                   // lower bound of the pair's distance at the cell centre, units of kSelfTabStep
    int self_cyl_slot[SMR_MAX_SELF_PAIRS];      // pairs against a world-fixed cylinder: slot of its parameters in the job kernels' LDS (-1: none)
    int n_static, static_piece[SMR_MAX_STATIC];
    unsigned static_mask;
    float static_max;
    unsigned moving_mask, contact_mask;
    float moving_max, contact_dist;
    int n_planets, obs_planet_size, obs_num_planets;
    DPlanet planet[SMR_MAX_PLANETS];
    // all planet pieces [planet_p0, +planet_pn) and their vertex slots [planet_v0, +planet_vn): what the step kernel stages
    // in LDS for the lane-autonomous planet queries (planets_auto = 0: does not fit, wave-lockstep queries)
    int planets_auto, planet_p0, planet_pn, planet_vn;   // planet_vn: vertex slots of auto_verts (4 floats each)
    // behind them in the same tables: the static piece (the table), descriptor auto_static (-1: not there), so that its rare GJK runs
    // join the lane-autonomous loop of the planet runs; auto_pn / auto_vn: descriptors / vertex slots staged in all
    int auto_static, auto_pn, auto_vn;
    int body_order[SMR_MAX_BODIES];   // dispatch order of the bodies in the collide grid (longest-running first)
    int pair_order[SMR_MAX_SELF_PAIRS];   // ... and of the self-collision pairs behind them
    // conservative advancement over the intermediate contact samples (contact sweep, below): no point of body b moves faster than
    // sum_j |qdot_j| sweep_R[b][j]; no point of planet p moves farther than planet_step[p] per orbit index; the ball stays inside
    // the sphere of radius ball_bound_radius (margin included) about its frame origin
    float sweep_R[SMR_MAX_BODIES][SMR_NJ];
    float ee_R[SMR_NJ];               // the same for the target link point (A14: its reach test at the 24 updates of a step, finish_kernel)
    float planet_step[SMR_MAX_PLANETS];
    float ball_bound_radius;
    const float *auto_verts;   // the planet pieces' vertices, every piece padded to a multiple of 4 with its last vertex
    const float *auto_desc;    // kPlanetDescDwords per piece: first slot, padded count, margin, planet (0 / 1), centre xyz
    float planet_obs_min[3], planet_obs_max[3];
    // ball (A10)
    int use_ball, ball_piece, ball_check_invalid, ball_random_initial;
    float ball_radius, ball_active_xy_radius;
    float ball_obs_pos_min[3], ball_obs_pos_max[3], ball_obs_vel_min[3], ball_obs_vel_max[3];
    float ball_target_box[3][2], ball_target_min_static, ball_target_min_self;
    double ball_speed, ball_center[3], ball_sphere_radius, ball_height[2], ball_angle[2], ball_high_prob;   // spawn math runs in float64
    double ball_invalid_min[3], ball_invalid_max[3], ball_final_min[3], ball_final_max[3], plane_z, table_half_xy[2];
    double dt64;
    // target-point reaching task (A14)
    int use_target, obs_target_pos, obs_target_rel, obs_target_size, target_normalize, ca_mode;
    float target_min_static, target_min_self, target_radius, target_bonus, target_factor;
    float target_rel_min[3], target_rel_max[3];
    int punish_action, term_self, term_static, term_moving;
    float action_thr, action_max_pun;
    float self_max_reward, self_max_dist, static_max_reward, static_max_dist, moving_max_reward, moving_max_dist;
    float low_acc_reward, low_acc_thr, low_vel_reward, low_vel_thr;
    float term_bonus, early_term_pun, collision_dist;
    // braking-trajectory safety layer (btg.py, ctlp.py:3026-3207)
    int check_braking, brake_checks, brake_timeout_steps;
    float brake_safety, brake_acc_max[SMR_NJ], brake_jerk_max[SMR_NJ], brake_stop_v, brake_stop_a;
    // the reward terms that read the braking layer's results (rewards.py:315-334,357-375; smr_scene.h)
    int punish_adaptation, punish_end_min, punish_bt_min;
    float adaptation_max_pun, end_min_max_pun, end_min_thr, bt_max_pun, bt_thr;
    float start_box[3][2];
    float start_min_static, start_min_self;
    int kin_mode;
    float kin_prob, stay_prob;
    unsigned seed_lo, seed_hi;
    unsigned env_base;          // index of this handle's env 0 in the global batch: Philox counter word of env e = env_base + e, so that
                                // the shards of a batch draw what one handle over the whole batch draws (smr_create, env_index_base)
};

// per-env state, struct of arrays ([component][B]) so that lane e of a wave reads consecutive words
struct DState {
    float *q, *v, *a;          // [7][B]
    int *planet_idx;           // [2][B]
    int *step_count;           // [B]
    int *moving_hit;           // [B]
    unsigned *reset_count;     // [B]  number of resets of the env = Philox tag of its next reset
    float *a1;                 // [7][B]   plan -> collide/finish
    // the safe acceleration range of the NEXT step (round 4): a function of the state the step ends in, not of the next action - range_kernel
    // computes it beside the distance kernels as soon as the step is planned, the staged start states bring theirs along, and the next
    // step's plan kernel only maps the action into it.  range_valid[e] == 0: plan computes it itself (injected state, unstaged reset)
    float *range_lo, *range_hi;   // [7][B]
    int *range_valid;             // [B]
    float *next_lo, *next_hi;     // [2][7][B] the ranges of the staged start states
    float *res;                // [n_bodies][B] collide -> finish: contact of body l at any of the N contact samples of the step
    float *dist;               // [2][8][B] collide -> finish: d_static, d_moving of link l at the end-of-step sample
    int *braked;               // [B] 1: the braking acceleration replaced this step's action (adapt_action, ctlp.py:3055-3153)
    int *risky;                // [B] 1: the risk network rated this step's action risky, the backup policy's action was executed
    float *brake_acc;          // [brake_timeout_steps + 1][7][B] end acceleration of braking segment m >= 1 behind the planned step
                               // ([0]: the braking acceleration at the current state, what adapt_action puts in place of the action)
    int *brake_steps;          // [B] braking segments until the robot stands (-1: not within the braking timeout)
    unsigned *bt_min;          // [B] float bits of the smallest distance brake_check_kernel's rows found along this step's braking trajectory
                               //     (_braking_trajectory_minimum_distance, ctlp.py:3157,3185-3186; atomicMin, plan_kernel starts it at +inf)
    float *lb_moving;          // [n_bodies][B] lower bound of the body's distance to every moving piece at the end of the last step
                               // (< 0: none - new episode, injected state); the next step's lanes rule out their first samples with it
    int *done_list;            // [B]
    int *done_count;           // [1]
    float *ball;               // [10][B] launch point p0, launch velocity v0, initial euler angles e0, spin rate
    int *ball_i;               // [5][B]  update counter k, k_max (passed the robot), k_hit (table/ground), active, spawn counter
    int *spawn_list;           // [B]  envs whose ball was retired in this step
    int *spawn_count;          // [1]
    float *tp;                 // [5][B] target point: position, distance at the last observation, distance when sampled
    unsigned *tp_count;        // [B] target points sampled so far = Philox tag of the next one
    int *tp_list;              // [B] envs that need a new target point (reached, or reset)
    int *tp_list_count;        // [1]
    // staged target points (scenes with staged start states): the NEXT target of every env - a pure function of (seed, env,
    // targets drawn so far) - is prepared on the side stream; finish installs it when a target is reached / an env restarts
    float *next_tp;            // [2][3][B]  two staged points per env (target counts c and c + 1, slot = count & 1), like the start states
    unsigned *next_tp_tag;     // [2][B] the target count the staged point was drawn for
    int *tp_refill_list;       // [B] envs whose staged point was consumed in this step
    unsigned *tp_refill_tags;  // parallel: the count to draw for
    int *tp_refill_count;      // this step's counter; two slots alternate like refill_count
    int *tp_refill_zero;
    // staged balls (ball scenes with staged start states): the launch of every env's NEXT ball - a pure function of (seed, env,
    // balls spawned so far) - is prepared on the side stream; spawn_kernel installs it (and picks the contact-free initial update)
    float *next_ball;          // [2][10][B] p0, v0, e0, omega; two staged launches per env (spawn counts c and c + 1, slot = count & 1)
    int *next_ball_i;          // [2][2][B] kmax, khit
    unsigned *next_ball_tag;   // [2][B] the spawn count the staged launch was drawn for
    unsigned *spawn_tags;      // parallel to spawn_list: the count to draw for when the entry is re-staged
    // staged start states (scenes without a ball): the start state of every env's NEXT episode, prepared off the
    // critical path on a second stream and committed by the finish kernel when the env is done
    // TWO staged start states per env, slot = tag & 1 (tags reset_count and reset_count + 1): the refill of a consumed slot then has
    // two step periods before the env can need it - it never sits on the critical path, whatever the main path shrinks to
    float *next_q, *next_v, *next_a;   // [2][7][B]
    int *next_planet_idx;              // [2][2][B]
    int *next_hit;                     // [2][B]
    unsigned *next_attempts;           // [2][B] candidates drawn for the staged start position
    unsigned *start_attempts;          // [B] ... for the start position of the running episode (_starting_point_sampling_attempts, ctlp.py:1488)
    unsigned *staged_tag;              // [2][B]  reset_count value the staged state belongs to (~0u: none)
    unsigned *refill_tags;             // parallel to refill_list: the tag to prepare (the env may restart again before the refill runs)
    int *refill_list;                  // [B]  envs whose staged state was consumed
    int *refill_count;                 // the counter of THIS step's list: two slots alternate, finish clears the other one
    int *refill_zero;                  // ... (refill_zero, or null) so that no memset sits between the step's kernels
    int *stale_commits;                // [1]  commits that found no matching staged state (must stay 0)
    int *error_flags;                  // [1]  kSampling* bits: a bounded rejection-sampling loop ran out of attempts (the reference raises
                                       //      ValueError there, ctlp.py:2127-2136); reported by the next synchronising call
    // pair jobs (planet scenes, "Pair jobs" below): the step's GJK runs as independent (env, body, piece, sample) jobs in global lists
    uint32_t *keyw;                    // [7][n_bodies][B] candidate key words of (env, body) at the end-of-step sample: keys -> first kernel
    unsigned *env_best;                // [B] smallest (key << 16 | body << 8 | piece) over the env's bodies (atomicMin; ~0u: no candidate)
    unsigned *env_min;                 // [B] float bits of the env's moving-obstacle distance: atomicMin over its jobs (non-negative floats
                                       //     order like their bits); plan_kernel starts it at moving_max + 0.002
    unsigned *env_dstatic;             // [B] ... of the static distance found by deferred table runs (starts at static_max)
    float *frames;                     // [n_bodies][B][12] frame of (body, env) at t = dt: keys kernel -> the end-of-step jobs (job_setup)
    unsigned *lb1;                     // [n_bodies][B] float bits of a lower bound of (env, body)'s BEST candidate's distance (low 5 bits: its piece; the
                                       //     bound rounded down to make room): the first GJK iteration of that pair, made by the keys kernel anyway
    unsigned *env_dself;               // [B] ... of the self-collision distance: atomicMin over the env's self-pair jobs (starts at static_max)
    unsigned long long *items;         // intermediate contact samples still to be tested: job words (env, body, k) of kind kJobItem
    unsigned long long *jobs;          // GJK runs to make: job words (env, body, piece, k, kind)
    int *item_count, *job_count;       // [1] each, cleared by plan_kernel
    int *job_next;                     // [1] the next job of the list nobody has taken yet (jobs_kernel), cleared by plan_kernel
    int job_cap;                       // capacity of jobs[]
    // Two regions of jobs[] (round 4): [0, job_split) the jobs whose scanned side is short (ISS pieces up to 24 padded vertices, the table's
    // 8 corners) and the self-collision pairs (appended first, by self_bounds_kernel: they sit together at the front), [job_split, job_cap)
    // the jobs with a long scan (asteroid: 64 vertices, the station's largest piece: 56).  A wavefront scans as long as its longest lane:
    // mixed, nearly every trip ran the asteroid's 8 rounds of 8 vertices for lanes that needed 1 to 3.
    int *job_count_big;                // [1] jobs in the second region, cleared by plan_kernel
    int job_split;
    unsigned big_piece_mask;           // planet pieces (bit = index into the planet descriptors) whose padded vertex list is longer than kSmallScan
    int *overflow_env;                 // [B] 1: a job of this env found the list full - overflow_kernel evaluates ALL its queries itself
    unsigned long long *episodes;      // [1] episodes that ended inside step() and were restarted there (auto_reset): smr_episode_count
    float *defer;                      // [defer_cap][kDeferWords] runs that left their lane unsettled ("deferred runs"): jobs_kernel -> jobs_coop_kernel
    int *defer_count;                  // [1] cleared by plan_kernel
    int defer_cap;
};

enum { kSamplingStart = 1, kSamplingPlanetPhase = 2, kSamplingBallLaunch = 4, kSamplingBallTarget = 8, kSamplingTargetPoint = 16,
       kSamplingBallInitialStep = 32, kSamplingKinematicState = 64 };
#ifndef SMR_HOST_EMUL
__device__ __forceinline__ void sampling_exhausted(const DState &st, int bit) { atomicOr(st.error_flags, bit); }
#endif

// ---------------------------------------------------------------------------------------------
// small vector algebra
// ---------------------------------------------------------------------------------------------
struct F3 { float x, y, z; };
typedef float V4 __attribute__((vector_size(16)));       // a plain 16-byte vector: loads / stores through address-space pointers (dwordx4)
// One vertex = 4 floats at a 16-byte boundary (tables, LDS pools and LDS arrays are laid out that way): read as ONE 128-bit access.  Through a
// plain `const float *` the compiler cannot know the alignment and splits the read into ds_read2_b32 / ds_read_b32 pairs - twice the LDS
// instructions, each with its own bank conflicts (seen in the ISA of the jobs kernel: 47 dword reads, no ds_read_b128, round 4).
__device__ __forceinline__ V4 load_v4(const float *p) { return *(const V4 *)p; }
#ifndef SMR_HOST_EMUL
__device__ __forceinline__ V4 load_v4(const SMR_GLOBAL float *p) { return *(const SMR_GLOBAL V4 *)p; }
#endif
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ F3 neg(F3 a) { return {-a.x, -a.y, -a.z}; }

struct Xf { float r[9]; F3 t; };   // x_world = R x_body + t, R row-major

__device__ __forceinline__ F3 rot(const Xf &X, F3 v) {
    return {fmaf(X.r[2], v.z, fmaf(X.r[1], v.y, X.r[0] * v.x)), fmaf(X.r[5], v.z, fmaf(X.r[4], v.y, X.r[3] * v.x)),
            fmaf(X.r[8], v.z, fmaf(X.r[7], v.y, X.r[6] * v.x))};
}
__device__ __forceinline__ F3 rot_t(const Xf &X, F3 v) {
    return {fmaf(X.r[6], v.z, fmaf(X.r[3], v.y, X.r[0] * v.x)), fmaf(X.r[7], v.z, fmaf(X.r[4], v.y, X.r[1] * v.x)),
            fmaf(X.r[8], v.z, fmaf(X.r[5], v.y, X.r[2] * v.x))};
}
__device__ __forceinline__ F3 apply(const Xf &X, F3 v) { return rot(X, v) + X.t; }
// T^-1 o X: the frame X seen from the frame T (rigid transforms)
__device__ __forceinline__ Xf inv_compose(const Xf &T, const Xf &X) {
    Xf R;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R.r[3 * i + j] = fmaf(T.r[6 + i], X.r[6 + j], fmaf(T.r[3 + i], X.r[3 + j], T.r[i] * X.r[j]));
    R.t = rot_t(T, F3{X.t.x - T.t.x, X.t.y - T.t.y, X.t.z - T.t.z});
    return R;
}

// ---------------------------------------------------------------------------------------------
// RNG: Philox4x32-10.  Stateless addressing: draw d of (env, stream, tag) is word d&3 of
// philox(counter = (d>>2, env, stream, tag), key = seed); tag = the env's reset counter.  Streams are
// independent, so the lanes of a group can evaluate candidates of one stream in parallel and still pick
// "the first valid one" exactly like a sequential loop.  24-bit uniforms are exact in float32.
// (Reference: numpy's global RNG, base.py:1704-1710.)
// ---------------------------------------------------------------------------------------------
enum { S_POS = 0, S_JOINT = 1, S_WALK = 8, S_PLANET = 9, S_MODE = 10, S_BALL_RELEASE = 11, S_BALL_POSE = 12, S_BALL_MISC = 13, S_TARGET = 14 };

struct Rng {
    unsigned k0, k1, env, tag;
    __device__ __forceinline__ unsigned u32(unsigned stream, unsigned draw) const {
        unsigned c0 = draw >> 2, c1 = env, c2 = stream, c3 = tag, a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            unsigned n0 = hi1 ^ c1 ^ a, n2 = hi0 ^ c3 ^ b;
            c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        unsigned sel = draw & 3u;
        return sel == 0 ? c0 : sel == 1 ? c1 : sel == 2 ? c2 : c3;
    }
    __device__ __forceinline__ float uniform01(unsigned stream, unsigned draw) const { return (float)(u32(stream, draw) >> 8) * (1.0f / 16777216.0f); }
    __device__ __forceinline__ float uniform(unsigned stream, unsigned draw, float lo, float hi) const { return fmaf(uniform01(stream, draw), hi - lo, lo); }
    __device__ __forceinline__ double uniform01d(unsigned stream, unsigned draw) const { return (double)(u32(stream, draw) >> 8) * (1.0 / 16777216.0); }
    __device__ __forceinline__ double uniformd(unsigned stream, unsigned draw, double lo, double hi) const { return lo + uniform01d(stream, draw) * (hi - lo); }
    __device__ __forceinline__ int randint(unsigned stream, unsigned draw, int n) const {
        return (int)(((unsigned long long)(u32(stream, draw) >> 8) * (unsigned long long)n) >> 24);
    }
};

// ---------------------------------------------------------------------------------------------
// A3: constant-jerk segment (actions.py:468-487)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float interp_pos(float p0, float v0, float a0, float a1, float t, float dt) {
    return p0 + v0 * t + 0.5f * a0 * t * t + (1.0f / 6.0f) * ((a1 - a0) / dt) * t * t * t;
}
__device__ __forceinline__ float interp_vel(float v0, float a0, float a1, float t, float dt) {
    return v0 + a0 * t + 0.5f * ((a1 - a0) / dt) * t * t;
}

// ---------------------------------------------------------------------------------------------
// A4: safe acceleration range (klimits call sites actions.py:97-106,206-211; Kiemel & Kroeger 2021).
// Velocity limits in closed form, position limits by bisection over a velocity-aware braking
// profile; see DESIGN.md "safe acceleration range".  Accelerations are piecewise linear with knots
// every dt, |a| <= A, knot-to-knot change <= D = J dt.
// ---------------------------------------------------------------------------------------------
struct JointLim {
    float plo, phi, V, A, J, dt, plo_res, phi_res, V_res;   // *_res: float64 limit - float32 image
    // 1 / dt and 1 / J for the inner loops of the range computation (an IEEE division is ten instructions; the braking profiles
    // below run seven of them per knot otherwise)
    float idt, iJ;
    __device__ __forceinline__ void derive() { idt = 1.0f / dt; iJ = 1.0f / J; }
};

constexpr float kNoBound = 3.0e38f;


// largest end acceleration a1 (unclamped) that keeps v <= Vlim forever; -kNoBound if none exists.
// c = head room when the segment ends with a1 = 0.  For a1 > 0 the profile that lowers a by D per knot
// peaks at v1 + a1^2/(2J) (full-jerk ramp) or v1 + a1^2 dt/(2(a1+A)) when the floor -A cuts the ramp
// short (only if D > A); for a1 < 0 with a0 > 0 the peak lies inside the first segment.
// The limit is passed as (V, tol) so that the head room (V - v0) + tol keeps full float32 precision when the
// joint rides on the limit (V - v0 is exact for v0 near V).
__device__ __forceinline__ float vel_upper_a1max(float v0, float a0, float V, float tol, const JointLim &L) {
    const float dt = L.dt, D = L.J * dt;
    const float head = (V - v0) + (tol + L.V_res);
    const float c = head - 0.5f * a0 * dt;
    if (c >= 0.f) {
        float x = L.J * (-0.5f * dt + __builtin_amdgcn_sqrtf(0.25f * dt * dt + 2.0f * c * L.iJ));      // 1-ulp square roots: the tolerances are 1e-5 V
        if (D > L.A) {
            float K = 2.0f * c * L.idt;
            float xa = ((K - L.A) + __builtin_amdgcn_sqrtf((L.A - K) * (L.A - K) + 8.0f * K * L.A)) * 0.25f;
            if (xa < D - L.A) x = xa;
        }
        return x;
    }
    if (a0 > 0.f && head > 0.f) return a0 - a0 * a0 * dt * (0.5f * __builtin_amdgcn_rcpf(head));
    return -kNoBound;
}

// displacement of the constant-jerk segment after time t (interp_pos without the start position)
__device__ __forceinline__ float seg_disp(float v0, float a0, float a1, float t, float dt) {
    return v0 * t + 0.5f * a0 * t * t + (1.0f / 6.0f) * ((a1 - a0) / dt) * t * t * t;
}

// the same with 1 / dt given, for the knot loops
__device__ __forceinline__ float seg_disp_i(float v0, float a0, float a1, float t, float idt) {
    return v0 * t + 0.5f * a0 * t * t + (1.0f / 6.0f) * ((a1 - a0) * idt) * t * t * t;
}
__device__ __forceinline__ float seg_dmax_i(float v0, float a0, float a1, float dt, float idt) {
    float m = seg_disp_i(v0, a0, a1, dt, idt);
    float j = (a1 - a0) * idt;
    float disc = a0 * a0 - 2.0f * j * v0;
    if (disc >= 0.f) {
        float sq = __builtin_amdgcn_sqrtf(disc);
        float qq = -(a0 + (a0 >= 0.f ? sq : -sq));
        float r0 = qq * __builtin_amdgcn_rcpf(j), r1 = 2.0f * v0 * __builtin_amdgcn_rcpf(qq);   // inf/nan when a divisor is 0: rejected below
        if (r0 > 0.f && r0 < dt) m = fmaxf(m, seg_disp_i(v0, a0, a1, r0, idt));
        if (r1 > 0.f && r1 < dt) m = fmaxf(m, seg_disp_i(v0, a0, a1, r1, idt));
    }
    return m;
}

// largest displacement reached inside (0, dt] of a segment
__device__ __forceinline__ float seg_dmax(float v0, float a0, float a1, float dt) {
    float m = seg_disp(v0, a0, a1, dt, dt);
    float j = (a1 - a0) / dt;
    float disc = a0 * a0 - 2.0f * j * v0;
    if (disc >= 0.f) {
        float sq = sqrtf(disc);
        float qq = -(a0 + (a0 >= 0.f ? sq : -sq));
        float r0 = qq / j, r1 = 2.0f * v0 / qq;      // inf/nan when a divisor is 0: rejected by the range test
        if (r0 > 0.f && r0 < dt) m = fmaxf(m, seg_disp(v0, a0, a1, r0, dt));
        if (r1 > 0.f && r1 < dt) m = fmaxf(m, seg_disp(v0, a0, a1, r1, dt));
    }
    return m;
}

// Margin (gap - largest displacement) of the profile "after a0 -> a1 always take the lowest acceleration the
// lower velocity limit still allows", followed until the joint moves down again.  >= 0: the position limit
// holds.  Works on displacements relative to the current position (gap = limit - p0), so float32 resolves the
// few-micrometre margins that decide the bound near a limit.  Continuous and decreasing in a1.
__device__ inline float pos_margin(float gap, float v0, float a0, float a1, float Vtol, const JointLim &L) {
    SMR_CNT(0);
    float dmax = seg_dmax_i(v0, a0, a1, L.dt, L.idt);
    float d = seg_disp_i(v0, a0, a1, L.dt, L.idt);
    float v = v0 + 0.5f * (a0 + a1) * L.dt, a = a1;
    const float D = L.J * L.dt;
    for (int it = 0; it < 4096; ++it) {
        if (v <= 0.f && a <= 0.f) break;
        SMR_CNT(1);
        if (dmax > gap + 1.0f) break;                           // hopeless (also bounds the loop for absurd inputs)
        float floor_v = -vel_upper_a1max(-v, -a, L.V, Vtol, L);
        float an = fmaxf(fmaxf(a - D, -L.A), fminf(floor_v, fminf(a + D, L.A)));
        dmax = fmaxf(dmax, d + seg_dmax_i(v, a, an, L.dt, L.idt));
        d += seg_disp_i(v, a, an, L.dt, L.idt);
        v += 0.5f * (a + an) * L.dt;
        a = an;
    }
    return gap - dmax;
}

__device__ __forceinline__ bool pos_upper_ok(float gap, float v0, float a0, float a1, float Vtol, const JointLim &L) {
    return pos_margin(gap, v0, a0, a1, Vtol, L) >= 0.f;
}

// Largest a1 in [lo, hi] with pos_margin >= 0, where pos_margin(hi) < 0 is known and the margin decreases with
// a1; `infeasible` when not even lo qualifies.  Cooperative k-section: the lanes of a SEGMENT of W adjacent lanes
// (W = 64: the wavefront; 8: one reset group; 1: a lone lane = plain bisection, the host emulation) share the work of
// the segment's n open brackets - about one lane in ten needs one, the rest would idle through its ~25 bisection
// steps.  Each bracket gets k = W/n helpers (a power of two) that evaluate k points of it per round, so it shrinks
// (k+1)-fold per round: 4 rounds with 64 helpers, 8 with 8.  The answer is the largest evaluated feasible point, so
// pos_margin(result) >= 0 exactly as with bisection (the float64 oracle bisects), and the two agree to the bracket
// tolerance 2e-7 A.  All W lanes of a segment must call this together.
template <int W>
__device__ inline float coop_pos_upper_end(bool need, float gap, float v, float a, float lo, float hi, float Vtol, const JointLim &L,
                                           bool &infeasible) {
    const int lane = W > 1 ? (int)__lane_id() : 0;
    const int seg_base = W >= 64 ? 0 : (lane & ~(W - 1));
    const int r = lane - seg_base;
    const unsigned long long segmask = W >= 64 ? ~0ull : ((1ull << W) - 1ull);
    const unsigned long long needs = (__ballot(need) >> seg_base) & segmask;
    const int n = __popcll(needs);
    infeasible = false;
    if (n == 0) return lo;                                        // uniform over the segment
    int k = 1;
    while (2 * k * n <= W) k *= 2;
    const unsigned long long kmask = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
    const int t = r / k, sub = r % k;                             // this lane evaluates point `sub` of bracket t
    const bool helper = t < n;
    unsigned long long m = needs;
    for (int i = 0; i < t && m; ++i) m &= m - 1;                  // owner of bracket t = t-th set bit of needs
    const int owner = seg_base + ((helper && m) ? __ffsll((long long)m) - 1 : 0);
    const float tg = __shfl(gap, owner), tv = __shfl(v, owner), ta = __shfl(a, owner), tvtol = __shfl(Vtol, owner);
    float tlo = __shfl(lo, owner), thi = __shfl(hi, owner);
    JointLim TL = L;
    TL.V = __shfl(L.V, owner); TL.A = __shfl(L.A, owner); TL.J = __shfl(L.J, owner); TL.V_res = __shfl(L.V_res, owner);
    TL.iJ = __shfl(L.iJ, owner);
    const float xtol = 2e-7f * TL.A;
    bool tinf = false;
    for (int round = 0; round < 48; ++round) {
        const bool live = helper && !tinf && (round == 0 || thi - tlo > xtol);
        if (((__ballot(live) >> seg_base) & segmask) == 0ull) break;          // uniform over the segment
        // round 0 includes lo itself (its feasibility is not known yet): points lo + w s/k; later rounds the k interior points
        const float w = thi - tlo, den = round == 0 ? 1.0f / (float)k : 1.0f / (float)(k + 1);
        const int first = round == 0 ? 0 : 1;
        const float x = tlo + w * ((float)(sub + first) * den);
        const bool feas = live && pos_margin(tg, tv, ta, x, tvtol, TL) >= 0.f;
        const unsigned long long fb = (__ballot(feas) >> seg_base) & segmask;
        const unsigned long long bits = (fb >> (t * k)) & kmask;
        const int c = bits == kmask ? k : __ffsll((long long)~bits) - 1;       // leading feasible points
        if (live) {
            // evaluated points have indices first .. k-1+first; index 0 is tlo and index k (k+1 after round 0) is thi
            if (round == 0 && c == 0) tinf = true;                                   // lo itself violates the limit
            else {
                const float nlo = tlo + w * ((float)(c + first - 1) * den);          // last feasible point (bitwise the x of its helper)
                const float nhi = c < k ? tlo + w * ((float)(c + first) * den) : thi;
                if (c > 0) tlo = nlo;
                thi = nhi;
            }
        }
    }
    // hand the result to the owners: bracket of the owner with rank q among the needs sits in lanes q k .. q k + k - 1
    const int rank = __popcll(needs & ((1ull << r) - 1ull));
    const int src = seg_base + ((need ? rank : 0) * k);
    const float res = __shfl(tlo, src);
    const int rinf = __shfl((int)tinf, src);
    infeasible = need && rinf != 0;
    return need ? res : lo;
}

// W: cooperation width (see coop_pos_upper_end); `want`: this lane's result is used (a lane that only tags along,
// e.g. the eighth lane of a reset group, opens no bracket).  All W lanes of a segment must call this together.
template <int W>
__device__ inline void safe_acc_range_joint(const JointLim &L, float p, float v, float a, float &lo, float &hi, int &code, bool want = true) {
    const float D = L.J * L.dt;
    float blo = fmaxf(-L.A, a - D), bhi = fminf(L.A, a + D);
    if (blo > bhi) blo = bhi = fminf(fmaxf(a, -L.A), L.A);
    // design tolerance for the range, wider report tolerance for the violation code (rounding hysteresis)
    const float vtol = 1e-5f * L.V, vrep = 3e-5f * L.V;
    const float ptol = 1e-6f * (L.phi - L.plo), prep = 4e-6f * (L.phi - L.plo);
    code = 0;
    float uv = vel_upper_a1max(v, a, L.V, vtol, L), lv = -vel_upper_a1max(-v, -a, L.V, vtol, L);
    if (uv < blo) { if (vel_upper_a1max(v, a, L.V, vrep, L) < blo) code |= 1; uv = blo; }
    if (lv > bhi) { if (-vel_upper_a1max(-v, -a, L.V, vrep, L) > bhi) code |= 1; lv = bhi; }
    float vlo = fmaxf(blo, lv), vhi = fminf(bhi, uv);
    if (vlo > vhi) { code |= 4; vlo = vhi = 0.5f * (vlo + vhi); }
    // exact for p near the limit (Sterbenz) + the part of the float64 limit that float32 cannot hold
    const float gap_hi = (L.phi - p) + L.phi_res, gap_lo = (p - L.plo) - L.plo_res;
    // a bracket is open on a side when the far end of [vlo, vhi] violates that position limit (lower side: mirrored)
    const bool need_up = want && pos_margin(gap_hi + ptol, v, a, vhi, vtol, L) < 0.f;
    const bool need_dn = want && pos_margin(gap_lo + ptol, -v, -a, -vlo, vtol, L) < 0.f;
    float up = vhi, dn = -vlo;                                    // dn: mirrored lower end
    bool inf_up = false, inf_dn = false;
    {   // first pass: every lane's first open side; second pass: the lanes with both sides open (rare)
        const bool first_up = need_up, any_need = need_up || need_dn;
        bool inf1;                                                // ONE call site: the segment's lanes must arrive together
        const float r1 = coop_pos_upper_end<W>(any_need, (first_up ? gap_hi : gap_lo) + ptol, first_up ? v : -v, first_up ? a : -a,
                                               first_up ? vlo : -vhi, first_up ? vhi : -vlo, vtol, L, inf1);
        if (first_up) { up = r1; inf_up = inf1; } else if (need_dn) { dn = r1; inf_dn = inf1; }
        bool inf2;
        const float r2 = coop_pos_upper_end<W>(need_up && need_dn, gap_lo + ptol, -v, -a, -vhi, -vlo, vtol, L, inf2);
        if (need_up && need_dn) { dn = r2; inf_dn = inf2; }
    }
    if (inf_up && !pos_upper_ok(gap_hi + prep, v, a, vlo, vtol, L)) code |= 2;
    if (inf_dn && !pos_upper_ok(gap_lo + prep, -v, -a, -vhi, vtol, L)) code |= 2;
    hi = up;
    lo = -dn;
    if (lo > hi) {
        code |= 4;
        float mid = 0.5f * (lo + hi);
        lo = hi = fminf(fmaxf(mid, vlo), vhi);
    }
}

// Only the violation code of safe_acc_range_joint (0: the state is feasible) - what the kinematic start-state sampling asks for on
// every one of its velocity / acceleration tries (ctlp.py:1499-1545).  The code needs no range end: whether a position bracket is
// open (the margin at the far end of the velocity interval), and whether it is infeasible (the margin at its near end - the point
// the root finder evaluates first), are single margin evaluations.  Only a lane with BOTH brackets open needs the ends themselves
// (lo > hi), and gets them from the full routine.  Same arithmetic, same code.  All W lanes of a segment must call together.
template <int W>
__device__ inline int safe_state_code(const JointLim &L, float p, float v, float a, bool want = true) {
    const float D = L.J * L.dt;
    float blo = fmaxf(-L.A, a - D), bhi = fminf(L.A, a + D);
    if (blo > bhi) blo = bhi = fminf(fmaxf(a, -L.A), L.A);
    const float vtol = 1e-5f * L.V, vrep = 3e-5f * L.V;
    const float ptol = 1e-6f * (L.phi - L.plo), prep = 4e-6f * (L.phi - L.plo);
    int code = 0;
    float uv = vel_upper_a1max(v, a, L.V, vtol, L), lv = -vel_upper_a1max(-v, -a, L.V, vtol, L);
    if (uv < blo) { if (vel_upper_a1max(v, a, L.V, vrep, L) < blo) code |= 1; uv = blo; }
    if (lv > bhi) { if (-vel_upper_a1max(-v, -a, L.V, vrep, L) > bhi) code |= 1; lv = bhi; }
    float vlo = fmaxf(blo, lv), vhi = fminf(bhi, uv);
    if (vlo > vhi) { code |= 4; vlo = vhi = 0.5f * (vlo + vhi); }
    const float gap_hi = (L.phi - p) + L.phi_res, gap_lo = (p - L.plo) - L.plo_res;
    const bool need_up = want && pos_margin(gap_hi + ptol, v, a, vhi, vtol, L) < 0.f;
    const bool need_dn = want && pos_margin(gap_lo + ptol, -v, -a, -vlo, vtol, L) < 0.f;
    const bool both = need_up && need_dn;
    if (__any(both)) {                                    // rare: the ends decide
        float lo, hi;
        int full;
        safe_acc_range_joint<W>(L, p, v, a, lo, hi, full, both);
        if (both) return full;
    }
    if (need_up && pos_margin(gap_hi + ptol, v, a, vlo, vtol, L) < 0.f && !pos_upper_ok(gap_hi + prep, v, a, vlo, vtol, L)) code |= 2;
    if (need_dn && pos_margin(gap_lo + ptol, -v, -a, -vhi, vtol, L) < 0.f && !pos_upper_ok(gap_lo + prep, -v, -a, -vhi, vtol, L)) code |= 2;
    return code;
}

__device__ __forceinline__ JointLim joint_lim(const DScene &S, int j) {
    JointLim L = {S.pos_lo[j], S.pos_hi[j], S.vel_max[j], S.acc_max[j], S.jerk_max[j], S.dt, S.pos_lo_res[j], S.pos_hi_res[j], S.vel_max_res[j], 0.f, 0.f};
    L.derive();
    return L;
}

// ---------------------------------------------------------------------------------------------
// Braking acceleration of one joint (BrakingTrajectoryGenerator.get_braking_acceleration, btg.py:44-78; klimits' braking mode
// restated as in the CPU restatement under oracle/: PARITY UNPINNED).  The lowest end acceleration a1 inside the jerk / acceleration box
// from which the acceleration can be taken back to zero, by at most D = J dt per knot, without the velocity crossing zero
// (mirrored for v < 0): the joint reaches (v, a) = (0, 0) at a knot and stays there.
// ---------------------------------------------------------------------------------------------
__device__ inline float braking_acceleration_joint(const DScene &S, int j, float v, float a) {
    if (fabsf(v) < S.brake_stop_v && fabsf(a) < S.brake_stop_a) return 0.f;        // btg.py:75-77
    const float A = S.brake_acc_max[j], dt = S.dt, D = S.brake_jerk_max[j] * dt;
    const float sgn = v < 0.f ? -1.f : 1.f, vm = sgn * v, am = sgn * a;
    float blo = fmaxf(-A, am - D), bhi = fminf(A, am + D);
    if (blo > bhi) blo = bhi = fminf(fmaxf(am, -A), A);
    const float tol = 1e-5f * S.vel_max[j];
    const float c = vm + tol + 0.5f * am * dt;                                     // velocity after the step if a1 = 0
    float a1;
    if (c >= 0.f) {                                                                // n whole knots of D, then the rest, back to zero
        a1 = -A;
        for (int n = 0; n < 64; ++n) {
            const float x = (-c / dt - 0.5f * (float)(n * (n + 1)) * D) / (float)(n + 1);
            if (x >= -(float)(n + 1) * D - 1e-6f) { a1 = fminf(x, 0.f); break; }
        }
    } else if (am < 0.f && vm + tol > 0.f) a1 = am + am * am * dt / (2.0f * (vm + tol));   // the dip inside the step stays at zero
    else a1 = bhi;
    return sgn * fmaxf(blo, fminf(a1, bhi));
}

// Does the clip of braking acceleration b into the safe range of (p, v, a) (btg.py:80-93) change it?  No when b lies inside the
// velocity-admissible interval and both position margins hold for it (the two tests the range itself is made of); the position
// margins are only evaluated where a limit is within a generous stopping distance.
__device__ inline bool braking_clip_needed(const JointLim &L, float p, float v, float a, float b) {
    const float D = L.J * L.dt, eps = 1e-6f * L.A;
    float blo = fmaxf(-L.A, a - D), bhi = fminf(L.A, a + D);
    if (blo > bhi) blo = bhi = fminf(fmaxf(a, -L.A), L.A);
    const float vtol = 1e-5f * L.V, ptol = 1e-6f * (L.phi - L.plo);
    const float vhi = fminf(bhi, fmaxf(blo, vel_upper_a1max(v, a, L.V, vtol, L))), vlo = fmaxf(blo, fminf(bhi, -vel_upper_a1max(-v, -a, L.V, vtol, L)));
    if (b < vlo - eps || b > vhi + eps) return true;
    const float vm = fabsf(v) + L.A * L.dt;
    const float reach = 2.0f * (vm * (2.0f * L.dt + L.A * L.iJ) + 0.5f * vm * vm / L.A);
    const float gap_hi = (L.phi - p) + L.phi_res, gap_lo = (p - L.plo) - L.plo_res;
    if (gap_hi < reach && pos_margin(gap_hi + ptol, v, a, b, vtol, L) < 0.f) return true;
    if (gap_lo < reach && pos_margin(gap_lo + ptol, -v, -a, -b, vtol, L) < 0.f) return true;
    return false;
}

// ---------------------------------------------------------------------------------------------
// A6: forward kinematics, one joint at a time (robot.urdf chain; ctlp.py:5163-5195)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ Xf fk_base(const DScene &S) {
    Xf T;
    T.r[0] = 1.f; T.r[1] = 0.f; T.r[2] = 0.f; T.r[3] = 0.f; T.r[4] = 1.f; T.r[5] = 0.f; T.r[6] = 0.f; T.r[7] = 0.f; T.r[8] = 1.f;
    T.t = {S.base[0], S.base[1], S.base[2]};
    return T;
}

// T <- T * Trans(joint_xyz[k]) * Rot(joint_axis[k], qk): frame of iiwa_link_(k+1)
// ... with the joint's sine and cosine given (a lane that builds several frames of one configuration computes them once)
__device__ __forceinline__ void fk_advance_sc(const DScene &S, int k, float s, float c, Xf &T) {
    T.t = apply(T, F3{S.joint_xyz[k][0], S.joint_xyz[k][1], S.joint_xyz[k][2]});
    float ax = S.joint_axis[k][0], ay = S.joint_axis[k][1], az = S.joint_axis[k][2];
    float C = 1.f - c;
    float j0 = c + ax * ax * C, j1 = ax * ay * C - az * s, j2 = ax * az * C + ay * s;
    float j3 = ay * ax * C + az * s, j4 = c + ay * ay * C, j5 = ay * az * C - ax * s;
    float j6 = az * ax * C - ay * s, j7 = az * ay * C + ax * s, j8 = c + az * az * C;
    float n[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float r0 = T.r[3 * i], r1 = T.r[3 * i + 1], r2 = T.r[3 * i + 2];
        n[3 * i] = r0 * j0 + r1 * j3 + r2 * j6;
        n[3 * i + 1] = r0 * j1 + r1 * j4 + r2 * j7;
        n[3 * i + 2] = r0 * j2 + r1 * j5 + r2 * j8;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) T.r[i] = n[i];
}
__device__ __forceinline__ void fk_advance(const DScene &S, int k, float qk, Xf &T) {
    float s, c;
    sincosf(qk, &s, &c);
    fk_advance_sc(S, k, s, c, T);
}

// the same with the hardware sine / cosine (v_sin_f32 / v_cos_f32, absolute error ~1e-6): for BOUNDS only - link frames that feed
// a distance or a flag always come from fk_advance
__device__ __forceinline__ void fk_advance_fast(const DScene &S, int k, float qk, Xf &T) {
    T.t = apply(T, F3{S.joint_xyz[k][0], S.joint_xyz[k][1], S.joint_xyz[k][2]});
    float ax = S.joint_axis[k][0], ay = S.joint_axis[k][1], az = S.joint_axis[k][2];
#ifdef SMR_HOST_EMUL
    float s = std::sin(qk), c = std::cos(qk);
#else
    float s = __sinf(qk), c = __cosf(qk);
#endif
    float C = 1.f - c;
    float j0 = c + ax * ax * C, j1 = ax * ay * C - az * s, j2 = ax * az * C + ay * s;
    float j3 = ay * ax * C + az * s, j4 = c + ay * ay * C, j5 = ay * az * C - ax * s;
    float j6 = az * ax * C - ay * s, j7 = az * ay * C + ax * s, j8 = c + az * az * C;
    float n[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float r0 = T.r[3 * i], r1 = T.r[3 * i + 1], r2 = T.r[3 * i + 2];
        n[3 * i] = r0 * j0 + r1 * j3 + r2 * j6;
        n[3 * i + 1] = r0 * j1 + r1 * j4 + r2 * j7;
        n[3 * i + 2] = r0 * j2 + r1 * j5 + r2 * j8;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) T.r[i] = n[i];
}

__device__ __forceinline__ Xf quat_pose(const SMR_GLOBAL float *row) {   // row: pos xyz, quat xyzw
    float x = row[3], y = row[4], z = row[5], w = row[6];
    float n = x * x + y * y + z * z + w * w;
    float s = n > 0.f ? 2.0f / n : 0.f;
    Xf X;
    X.r[0] = 1.f - s * (y * y + z * z); X.r[1] = s * (x * y - z * w);       X.r[2] = s * (x * z + y * w);
    X.r[3] = s * (x * y + z * w);       X.r[4] = 1.f - s * (x * x + z * z); X.r[5] = s * (y * z - x * w);
    X.r[6] = s * (x * z - y * w);       X.r[7] = s * (y * z + x * w);       X.r[8] = 1.f - s * (x * x + y * y);
    X.t = {row[0], row[1], row[2]};
    return X;
}

// ---------------------------------------------------------------------------------------------
// GJK distance between two convex vertex sets (the algorithm behind pybullet.getClosestPoints for
// convex hulls).  Support mapping: a linear scan with the vertex in SGPRs (uniform index -> scalar
// load) and the search direction per lane.  Sub-simplex selection: closest-point-on-segment /
// triangle / tetrahedron region tests (Ericson, Real-Time Collision Detection, 5.1).
// ---------------------------------------------------------------------------------------------
template <bool IDENTITY>
__device__ __forceinline__ F3 support(const SMR_CONST float *verts, int begin, int count, const Xf &X, F3 d) {
    F3 dl = IDENTITY ? d : rot_t(X, d);
    float best = -3.0e38f;
    int bi = 0;
    begin = __builtin_amdgcn_readfirstlane(begin);      // wave-uniform by construction (piece index is uniform)
    count = __builtin_amdgcn_readfirstlane(count);
    const SMR_CONST float *base = verts + 4 * (size_t)begin;   // one vertex = 4 floats (x, y, z, 0)
#pragma unroll 4
    for (int i = 0; i < count; ++i) {
        float px = base[4 * i], py = base[4 * i + 1], pz = base[4 * i + 2];
        float s = fmaf(pz, dl.z, fmaf(py, dl.y, px * dl.x));
        bool gt = s > best;
        best = gt ? s : best;
        bi = gt ? i : bi;
    }
    F3 pl = {base[4 * bi], base[4 * bi + 1], base[4 * bi + 2]};   // per-lane gather of the winner
    return IDENTITY ? pl : apply(X, pl);
}

// ---- support of the robot-link hull (the B side of every query) -------------------------------
// ScanHull: the exhaustive scan above.  ClimbHull: hill climbing along the hull's edge graph - from a start vertex
// move to the neighbour with the largest dot product until none is larger; on a convex polytope such a vertex
// maximises the dot product over the whole hull.  Start: the previous support vertex of the same GJK run, or for
// its first iteration the entry of a small cube map of precomputed supports.  ~20-40 vertex reads per call
// instead of 250, per lane (the link hull lives in LDS in the step kernel, in global memory elsewhere).
struct ScanHull {
    const SMR_CONST float *verts; int begin, count;
    __device__ __forceinline__ F3 support(F3 dl, int &, bool) const {
        Xf unused;
        return smr::support<true>(verts, begin, count, unused, dl);
    }
};

__device__ __forceinline__ int climb_start_cell(F3 d) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int m; float a, b, s;
    if (ax >= ay && ax >= az) { m = d.x < 0.f ? 1 : 0; s = ax; a = d.y; b = d.z; }
    else if (ay >= az) { m = d.y < 0.f ? 3 : 2; s = ay; a = d.x; b = d.z; }
    else { m = d.z < 0.f ? 5 : 4; s = az; a = d.x; b = d.y; }
    const float k = 0.5f * (float)kStartBins / fmaxf(s, 1e-30f);
    int ia = (int)((a + s) * k), ib = (int)((b + s) * k);
    ia = ia < 0 ? 0 : (ia > kStartBins - 1 ? kStartBins - 1 : ia);
    ib = ib < 0 ? 0 : (ib > kStartBins - 1 ? kStartBins - 1 : ib);
    return (m * kStartBins + ia) * kStartBins + ib;
}

template <class FP, class UP, class BP>
struct ClimbHull {
    FP v;        // vertices of the piece, 4 floats each: x, y, z and - as bits - the vertex's span into ids4 (first group |
                 // groups << 16): it arrives with the coordinates, one dependent read less per step of the climb
    UP ids4;     // the piece's id groups
    BP start;    // the piece's start map
    __device__ __forceinline__ F3 support(F3 dl, int &cur, bool first) const {
        if (first) cur = (int)start[climb_start_cell(dl)];
        const V4 c0 = load_v4(v + 4 * cur);
        F3 pv = {c0[0], c0[1], c0[2]};
        uint32_t sp = (uint32_t)__float_as_int(c0[3]);
        float best = fmaf(pv.z, dl.z, fmaf(pv.y, dl.y, pv.x * dl.x));
        for (int guard = 0; guard < 256; ++guard) {
            const int g0 = (int)(sp & 0xffffu), g1 = g0 + (int)(sp >> 16);
            int nb = cur;
            // The first two id groups (a hull vertex has six neighbours on average) are fetched together and their eight vertices are
            // all in flight at once: two dependent LDS round trips per step of the climb instead of four - the step is a latency chain.
            // (The second word is read even when the vertex has one group - it lies inside the tables - but then not used.)
            const uint32_t fa = ids4[g0], fb_raw = ids4[g0 + 1];
            const uint32_t fb = g1 - g0 > 1 ? fb_raw : fa;
            int id8[8];
            V4 q8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                id8[k] = (int)(((k < 4 ? fa : fb) >> (8 * (k & 3))) & 255u);
                q8[k] = load_v4(v + 4 * id8[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float sq = fmaf(q8[k][2], dl.z, fmaf(q8[k][1], dl.y, q8[k][0] * dl.x));
                if (sq > best) { best = sq; nb = id8[k]; pv = F3{q8[k][0], q8[k][1], q8[k][2]}; sp = (uint32_t)__float_as_int(q8[k][3]); }
            }
            for (int g = g0 + 2; g < g1; ++g) {
                const uint32_t four = ids4[g];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int id = (int)((four >> (8 * k)) & 255u);
                    const V4 q = load_v4(v + 4 * id);
                    const float sq = fmaf(q[2], dl.z, fmaf(q[1], dl.y, q[0] * dl.x));
                    if (sq > best) { best = sq; nb = id; pv = F3{q[0], q[1], q[2]}; sp = (uint32_t)__float_as_int(q[3]); }
                }
            }
            if (nb == cur) break;
            cur = nb;
        }
        return pv;
    }
};
using GlobalHull = ClimbHull<const SMR_GLOBAL float *, const SMR_GLOBAL uint32_t *, const SMR_GLOBAL uint8_t *>;
using LocalHull = ClimbHull<const float *, const uint32_t *, const uint8_t *>;      // LDS copy (collide kernel)

template <class WP>
__device__ __forceinline__ DPiece piece_from_words(WP w) {
    static_assert(sizeof(DPiece) == 4 * kPieceDwords, "DPiece is read as dwords");
    DPiece P;
    P.begin = w[0]; P.count = w[1];
    P.cx = __int_as_float(w[2]); P.cy = __int_as_float(w[3]); P.cz = __int_as_float(w[4]); P.radius = __int_as_float(w[5]);
    P.margin = __int_as_float(w[6]); P.hx = __int_as_float(w[7]); P.hy = __int_as_float(w[8]); P.hz = __int_as_float(w[9]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) P.in[i][c] = __int_as_float(w[10 + 3 * i + c]);
    P.ids_begin = w[19]; P.ids_count = w[20];
    P.kind = w[21]; P.cyl_r = __int_as_float(w[22]); P.cyl_h = __int_as_float(w[23]);
    P.bx0 = __int_as_float(w[24]); P.bx1 = __int_as_float(w[25]); P.by0 = __int_as_float(w[26]); P.by1 = __int_as_float(w[27]);
    P.btop = __int_as_float(w[28]);
#pragma unroll
    for (int c = 0; c < 3; ++c) { P.cap_a[c] = __int_as_float(w[29 + c]); P.cap_b[c] = __int_as_float(w[32 + c]); }
    P.cap_r = __int_as_float(w[35]);
    return P;
}
// idx is wave uniform: scalar loads
__device__ __forceinline__ DPiece load_piece(const DPiece *pieces, int idx) {
    return piece_from_words((const SMR_CONST int *)pieces + kPieceDwords * (size_t)__builtin_amdgcn_readfirstlane(idx));
}
// idx may differ between lanes (reset / spawn: lane g of a group works link g)
__device__ __forceinline__ DPiece load_piece_lane(const DPiece *pieces, int idx) {
    return piece_from_words((const SMR_GLOBAL int *)pieces + kPieceDwords * (size_t)idx);
}

// the link hull of piece index pi (already loaded as P) read from the global tables
__device__ __forceinline__ GlobalHull global_hull(const DScene &S, int pi, const DPiece &P) {
    return GlobalHull{(const SMR_GLOBAL float *)S.verts + 4 * (size_t)P.begin,
                      (const SMR_GLOBAL uint32_t *)S.ids4 + P.ids_begin, (const SMR_GLOBAL uint8_t *)S.start + kStartCells * (size_t)pi};
}

// the hull of body b in a block's LDS copy of DScene::body_pool
struct BodyPool {
    const uint32_t *pool;
    __device__ __forceinline__ LocalHull hull(int b) const {
        const uint32_t *h = pool + kBodyPoolHead * b;
        return LocalHull{(const float *)pool + h[0], pool + h[1], (const uint8_t *)(pool + h[2])};
    }
    __device__ __forceinline__ DPiece piece(int b) const { return piece_from_words((const int *)(pool + pool[kBodyPoolHead * b + 3])); }
};

// quotient to 1 ulp of the reciprocal (v_rcp_f32 + a multiplication instead of the ten instructions of an IEEE division): for the
// barycentric parameters of the sub-simplex solves - a relative error of 1e-7 in a parameter moves the closest point by less than
// 1e-7 of an edge length
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// closest point of segment (a,b) to the origin; mask: bit0 a, bit1 b support it
__device__ __forceinline__ F3 seg_closest(F3 a, F3 b, int &mask) {
    F3 e = b - a;
    float t = -dot(a, e), den = dot(e, e);
    if (t <= 0.f) { mask = 1; return a; }
    if (t >= den) { mask = 2; return b; }
    mask = 3;
    return a + fast_div(t, den) * e;
}

// Closest point of triangle (a,b,c) to the origin; mask: bit0 a, bit1 b, bit2 c support it.
// Barycentric signs come from triple products with the normal n = (b-a)x(c-a) (signed sub-areas), not from
// the squared-dot-product form of Ericson 5.1.5: the Minkowski points of a metre-sized obstacle (the table)
// form needle triangles whose vertices are ~1 m from the origin while the answer is centimetres, and the
// squared form then loses 3-4 digits in float32.
__device__ __forceinline__ F3 tri_closest(F3 a, F3 b, F3 c, int &mask) {
    F3 e0 = b - a, e1 = c - a;
    F3 n = cross(e0, e1);
    float nn = dot(n, n);
    float la = dot(n, cross(b, c)), lb = dot(n, cross(c, a)), lc = dot(n, cross(a, b));   // la + lb + lc = nn
    if (nn > 0.f && la >= 0.f && lb >= 0.f && lc >= 0.f) {
        mask = 7;
        return fast_div(dot(a, n), nn) * n;               // foot of the perpendicular
    }
    // outside (or degenerate): the closest point lies on an edge opposite to a negative coordinate
    float best = 3.0e38f;
    F3 bv = a;
    mask = 1;
    const bool degenerate = !(nn > 0.f);
    if (lc < 0.f || degenerate) { int m; F3 p = seg_closest(a, b, m); float d = dot(p, p); if (d < best) { best = d; bv = p; mask = m; } }
    if (lb < 0.f || degenerate) { int m; F3 p = seg_closest(a, c, m); float d = dot(p, p); if (d < best) { best = d; bv = p; mask = (m & 1) | ((m & 2) << 1); } }
    if (la < 0.f || degenerate) { int m; F3 p = seg_closest(b, c, m); float d = dot(p, p); if (d < best) { best = d; bv = p; mask = m << 1; } }
    return bv;
}

__device__ __forceinline__ F3 sel3(bool c, F3 a, F3 b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }   // never `c ? a : b` on objects
__device__ __forceinline__ F3 pick_lowest(int m, F3 a, F3 b, F3 c, F3 d) { return sel3(m & 1, a, sel3(m & 2, b, sel3(m & 4, c, d))); }

// Closest point of the simplex {w0..w(n-1)} to the origin; the simplex is reduced to the vertices
// supporting it.  Returns true when the origin is enclosed (n == 4 and inside).
__device__ inline bool simplex_solve(F3 &w0, F3 &w1, F3 &w2, F3 &w3, int &n, F3 &v) {
    int mask;
    if (n == 1) { v = w0; return false; }
    if (n == 2) {
        v = seg_closest(w0, w1, mask);
    } else if (n == 3) {
        v = tri_closest(w0, w1, w2, mask);
    } else {
        // faces (vertex bits): abc|d, acd|b, adb|c, bdc|a
        float best = 3.0e38f;
        mask = 15;
        F3 bv = {0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            F3 p = f == 3 ? w1 : w0;
            F3 q = f == 0 ? w1 : f == 1 ? w2 : w3;
            F3 r = f == 0 ? w2 : f == 1 ? w3 : f == 2 ? w1 : w2;
            F3 o = f == 0 ? w3 : f == 1 ? w1 : f == 2 ? w2 : w0;
            F3 nrm = cross(q - p, r - p);
            F3 op = o - p;
            float sp = -dot(p, nrm), so = dot(op, nrm);
            // a (nearly) flat tetrahedron has no reliable inside: treat the face as a candidate
            bool outside = (sp * so < 0.f) || (so * so <= 1e-10f * dot(nrm, nrm) * dot(op, op));
            if (outside) {
                int m3;
                F3 pt = tri_closest(p, q, r, m3);
                float d2 = dot(pt, pt);
                if (d2 < best) {
                    best = d2;
                    bv = pt;
                    int bp = f == 3 ? 2 : 1, bq = f == 0 ? 2 : f == 1 ? 4 : 8, br = f == 0 ? 4 : f == 1 ? 8 : f == 2 ? 2 : 4;
                    mask = ((m3 & 1) ? bp : 0) | ((m3 & 2) ? bq : 0) | ((m3 & 4) ? br : 0);
                }
            }
        }
        if (mask == 15) return true;
        v = bv;
    }
    int m = mask;
    F3 n0 = pick_lowest(m, w0, w1, w2, w3);
    m &= m - 1;
    F3 n1 = pick_lowest(m, w0, w1, w2, w3);
    m &= m - 1;
    F3 n2 = pick_lowest(m, w0, w1, w2, w3);
    w0 = n0; w1 = n1; w2 = n2;
    n = __popc(mask);
    return false;
}

// GJK distance between convex sets A and B given by their world-space support mappings sa(dir, hint, first) and
// sb(dir, hint, first) (hint: a per-run integer the mapping may use as a warm start; first: first call of the run),
// margins excluded, started from the point v of A - B.  Returns a lower bound > cutoff as soon as one is proven, and
// the current upper bound as soon as it drops below stop_below (callers only test it against a threshold then).
// warm (trajectory walks that query the same pair again and again): in - when non-zero - the start direction instead of v (any
// direction will do: before the first simplex point is in, v only serves as the direction of the first support call), out the
// vector the run ended with, i.e. the direction of the closest points
// lower (the builder of the self-pair bound tables): the largest support-plane bound v.w / |v| seen - a rigorous lower bound of the
// distance however the run ends (the value returned on a stall is an upper bound)
template <class SA, class SB>
__device__ inline float gjk_core(const SA &sa, const SB &sb, F3 v, float cutoff, float stop_below, F3 *warm = nullptr, float *lower = nullptr) {
    F3 w0 = {0, 0, 0}, w1 = w0, w2 = w0, w3 = w0;
    int n = 0, hint_a = 0, hint_b = 0, stalls = 0;
    if (warm && dot(*warm, *warm) > 0.f) v = *warm;
    float vv = dot(v, v);
    const float cutoff2 = cutoff * cutoff, stop2 = stop_below > 0.f ? stop_below * stop_below : 0.f;
    SMR_CNT(2);
    for (int it = 0; it < 32; ++it) {
        SMR_CNT(3);
        SMR_CNT(4);
        F3 w = sa(neg(v), hint_a, it == 0) - sb(v, hint_b, it == 0);
        float vw = dot(v, w);
        if (lower && vw > 0.f) *lower = fmaxf(*lower, vw / sqrtf(vv));
#ifdef SMR_HOST_EMUL_TRACE
        printf("  it %d n %d vv %.9g vw %.9g v=(%.7g %.7g %.7g) w=(%.7g %.7g %.7g)\n", it, n, vv, vw, v.x, v.y, v.z, w.x, w.y, w.z);
#endif
        if (vw > 0.f && vw * vw > cutoff2 * vv) { if (warm) *warm = v; return (vw / sqrtf(vv)); }     // separating plane farther than cutoff
        if (n > 0 && (vv - vw) <= 1e-6f * vv) break;                         // converged
        // the support point is already a simplex vertex: v is optimal up to rounding (in exact arithmetic
        // v.w == v.v then; in float32 the test above can miss it and a duplicate vertex would make the
        // sub-simplex degenerate)
        {
            F3 e0 = w - w0, e1 = w - w1, e2 = w - w2;
            const float tiny = 1e-12f * fmaxf(dot(w, w), 1e-6f);
            bool dup = (n > 0 && dot(e0, e0) <= tiny) || (n > 1 && dot(e1, e1) <= tiny) || (n > 2 && dot(e2, e2) <= tiny);
            if (dup) break;
        }
        if (n == 0) w0 = w; else if (n == 1) w1 = w; else if (n == 2) w2 = w; else w3 = w;
        ++n;
        F3 vn;
        if (simplex_solve(w0, w1, w2, w3, n, vn)) return 0.f;                // origin inside the simplex
        float vvn = dot(vn, vn);
#ifdef SMR_HOST_EMUL_TRACE
        printf("  solved: n %d vvn %.9g vn=(%.7g %.7g %.7g)\n", n, vvn, vn.x, vn.y, vn.z);
#endif
        if (!(vvn == vvn)) break;                                            // degenerate sub-simplex: keep the last valid v
        // No measurable progress.  Worse than rounding explains: keep the last v.  Equal within rounding: the new
        // vertex helped by less than float32 resolves (it is far from the current feature); the slightly turned
        // direction may still find the vertex that does, so go on - a few times.
        if (it > 0 && vvn >= vv) {
            if (vvn > vv * (1.0f + 4e-7f) || ++stalls > 3) break;
        }
        v = vn;
        vv = vvn;
        if (vv <= stop2) break;
        if (n == 4) return 0.f;
    }
    if (warm) *warm = v;
    return sqrtf(vv);
}

// One GJK run as resumable state: gjk_iterate() advances it by one iteration, so that the lanes of a wavefront can each
// work on their OWN pair and move on to their next pair as soon as it is settled (planet_queries_auto).
struct GjkRun {
    F3 w0, w1, w2, w3, v;
    float vv, cutoff2, stop2;
    int n, it, stalls, hint_a, hint_b;
#ifdef SMR_PROFILE_TRIP
    unsigned long long pc_sa = 0, pc_sb = 0, pc_rest = 0;      // wavefront cycles inside the two support calls / the rest of gjk_iterate
#endif
    __device__ __forceinline__ void start(F3 v0, float cutoff, float stop_below) {
        w0 = {0, 0, 0}; w1 = w0; w2 = w0; w3 = w0;
        v = v0; vv = dot(v0, v0);
        cutoff2 = cutoff * cutoff; stop2 = stop_below > 0.f ? stop_below * stop_below : 0.f;
        n = 0; it = 0; stalls = 0; hint_a = 0; hint_b = 0;
        SMR_CNT(2);
    }
};

// sa(dir, hint, first) / sb(dir, hint, first): world-space support points of A and B.  Returns true when the run is
// settled; dist as gjk_distance returns it (a lower bound > cutoff as soon as one is proven, the current upper bound as
// soon as it drops below stop_below, 0 when the hulls intersect, else the distance).  Same tests in the same order as the
// loop of gjk_distance.
template <class SA, class SB>
__device__ __forceinline__ bool gjk_iterate(GjkRun &g, const SA &sa, const SB &sb, float &dist) {
    SMR_CNT(3);
    SMR_CNT(4);
    const bool first = g.it == 0;
#ifdef SMR_PROFILE_TRIP
    const unsigned long long pt0 = SMR_DEV_CLOCK();
    const F3 wa = sa(neg(g.v), g.hint_a, first);
    const unsigned long long pt1 = SMR_DEV_CLOCK();
    const F3 wb = sb(g.v, g.hint_b, first);
    const unsigned long long pt2 = SMR_DEV_CLOCK();
    g.pc_sa += pt1 - pt0; g.pc_sb += pt2 - pt1;
    struct RestClock { GjkRun &g; unsigned long long t; __device__ ~RestClock() { g.pc_rest += SMR_DEV_CLOCK() - t; } } rest_clock_{g, pt2};
    const F3 w = wa - wb;
#else
    const F3 w = sa(neg(g.v), g.hint_a, first) - sb(g.v, g.hint_b, first);
#endif
    const float vw = dot(g.v, w);
    if (vw > 0.f && vw * vw > g.cutoff2 * g.vv) { dist = vw / sqrtf(g.vv); return true; }
    bool settled = g.n > 0 && (g.vv - vw) <= 1e-6f * g.vv;
    if (!settled) {
        const F3 e0 = w - g.w0, e1 = w - g.w1, e2 = w - g.w2;
        const float tiny = 1e-12f * fmaxf(dot(w, w), 1e-6f);
        settled = (g.n > 0 && dot(e0, e0) <= tiny) || (g.n > 1 && dot(e1, e1) <= tiny) || (g.n > 2 && dot(e2, e2) <= tiny);
    }
    if (!settled) {
        {   // selects, not an if / else-if chain: hipcc 7.2 dropped the x component of `g.w3 = w` in that form when this
            // function was inlined into a per-lane trip loop (seen in an ISA-level trace, gfx950 -O3)
            // ... and component by component: a select between two F3 OBJECTS is lowered to a select between their addresses,
            // which pins the whole run state in scratch memory (seen in the ISA: ~70 scratch accesses per trip)
            const int slot = g.n;
            g.w0 = sel3(slot == 0, w, g.w0); g.w1 = sel3(slot == 1, w, g.w1); g.w2 = sel3(slot == 2, w, g.w2); g.w3 = sel3(slot >= 3, w, g.w3);
        }
        ++g.n;
        F3 vn;
        if (simplex_solve(g.w0, g.w1, g.w2, g.w3, g.n, vn)) { dist = 0.f; return true; }
        const float vvn = dot(vn, vn);
        if (!(vvn == vvn)) settled = true;
        else if (g.it > 0 && vvn >= g.vv && (vvn > g.vv * (1.0f + 4e-7f) || ++g.stalls > 3)) settled = true;
        else {
            g.v = vn;
            g.vv = vvn;
            if (g.vv <= g.stop2) settled = true;
            else if (g.n == 4) { dist = 0.f; return true; }
        }
    }
    if (settled || ++g.it >= 32) { dist = sqrtf(g.vv); return true; }
    return false;
}

// squared distance between segments p1-q1 and p2-q2 (clamped closest-point parameters; degenerate segments allowed)
__device__ __forceinline__ float seg_seg_dist2(F3 p1, F3 q1, F3 p2, F3 q2) {
    const F3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const float a = fmaxf(dot(d1, d1), 1e-12f), e = fmaxf(dot(d2, d2), 1e-12f), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2);
    const float den = a * e - b * b, ra = __builtin_amdgcn_rcpf(a);
    // reciprocals to 1 ulp: any parameters in [0, 1] give points ON the segments, the error of the distance stays at
    // rounding level (callers shave 1e-5 relative + 1e-6 off before they use it as a lower bound)
    float s = den > 1e-9f * a * e ? fminf(fmaxf((b * f - c * e) * __builtin_amdgcn_rcpf(den), 0.f), 1.f) : 0.f;
    float t = b * s + f;
    if (t < 0.f) { t = 0.f; s = fminf(fmaxf(-c * ra, 0.f), 1.f); }
    else if (t > e) { t = 1.f; s = fminf(fmaxf((b - c) * ra, 0.f), 1.f); }
    else t = t * __builtin_amdgcn_rcpf(e);
    const F3 diff = {r.x + d1.x * s - d2.x * t, r.y + d1.y * s - d2.y * t, r.z + d1.z * s - d2.z * t};
    return dot(diff, diff);
}

// A = hull piece scanned with scalar loads (wave-uniform piece) under XA (IDENTITY: XA is the identity), B = link hull hb under XB.
// The same loop as gjk_core, written out: routed through gjk_core's callables the step kernel was 10 % slower (measured).
template <bool IDENTITY, class HB>
__device__ inline float gjk_distance(const SMR_CONST float *verts, const DPiece &A, const Xf &XA, const HB &hb,
                                     const Xf &XB, F3 v, float cutoff, float stop_below, F3 *warm = nullptr) {
    F3 w0 = {0, 0, 0}, w1 = w0, w2 = w0, w3 = w0;
    int n = 0, hint = 0, stalls = 0;
    if (warm && dot(*warm, *warm) > 0.f) v = *warm;      // see gjk_core
    float vv = dot(v, v);
    const float cutoff2 = cutoff * cutoff, stop2 = stop_below > 0.f ? stop_below * stop_below : 0.f;
    SMR_CNT(2);
    for (int it = 0; it < 32; ++it) {
        SMR_CNT(3);
        SMR_CNT(4);
        F3 w = support<IDENTITY>(verts, A.begin, A.count, XA, neg(v)) - apply(XB, hb.support(rot_t(XB, v), hint, it == 0));
        float vw = dot(v, w);
        if (vw > 0.f && vw * vw > cutoff2 * vv) { if (warm) *warm = v; return (vw / sqrtf(vv)); }
        if (n > 0 && (vv - vw) <= 1e-6f * vv) break;
        {
            F3 e0 = w - w0, e1 = w - w1, e2 = w - w2;
            const float tiny = 1e-12f * fmaxf(dot(w, w), 1e-6f);
            bool dup = (n > 0 && dot(e0, e0) <= tiny) || (n > 1 && dot(e1, e1) <= tiny) || (n > 2 && dot(e2, e2) <= tiny);
            if (dup) break;
        }
        if (n == 0) w0 = w; else if (n == 1) w1 = w; else if (n == 2) w2 = w; else w3 = w;
        ++n;
        F3 vn;
        if (simplex_solve(w0, w1, w2, w3, n, vn)) return 0.f;
        float vvn = dot(vn, vn);
        if (!(vvn == vvn)) break;
        if (it > 0 && vvn >= vv) {
            if (vvn > vv * (1.0f + 4e-7f) || ++stalls > 3) break;
        }
        v = vn;
        vv = vvn;
        if (vv <= stop2) break;
        if (n == 4) return 0.f;
    }
    if (warm) *warm = v;
    return sqrtf(vv);
}

// signed distance (margins subtracted) between obstacle piece pa and link piece pb, or any value
// > cutoff when they are farther apart than cutoff (ctlp.py:2915-2931: no closest point returned)
// defer: when given, the GJK run is not made - *defer is set and 3e38 returned (every bound and shortcut before it still applies)
template <bool IDENTITY, class HB>
__device__ inline float piece_distance(const DScene &S, int pa, const Xf &XA, const DPiece &B, const HB &hb, const Xf &XB,
                                       float cutoff, float stop_below, F3 *warm = nullptr, bool *defer = nullptr) {
    const DPiece A = load_piece(S.pieces, pa);
    const float margins = A.margin + B.margin;
    F3 cb = apply(XB, F3{B.cx, B.cy, B.cz});
    F3 ca = IDENTITY ? F3{A.cx, A.cy, A.cz} : apply(XA, F3{A.cx, A.cy, A.cz});
    F3 d = ca - cb;
    float lb;
    if (IDENTITY) {   // sphere of the link against the obstacle's axis-aligned box
        float ex = fmaxf(fabsf(d.x) - A.hx, 0.f), ey = fmaxf(fabsf(d.y) - A.hy, 0.f), ez = fmaxf(fabsf(d.z) - A.hz, 0.f);
        lb = sqrtf(ex * ex + ey * ey + ez * ez) - B.radius - margins;
    } else {
        lb = sqrtf(dot(d, d)) - A.radius - B.radius - margins;
    }
    if (lb > cutoff) return lb;
    {   // bounding capsules: the link's against the obstacle piece's, or against the plane of the table top
        const F3 b0 = apply(XB, F3{B.cap_a[0], B.cap_a[1], B.cap_a[2]}), b1 = apply(XB, F3{B.cap_b[0], B.cap_b[1], B.cap_b[2]});
        if (!IDENTITY) {
            const F3 a0 = apply(XA, F3{A.cap_a[0], A.cap_a[1], A.cap_a[2]}), a1 = apply(XA, F3{A.cap_b[0], A.cap_b[1], A.cap_b[2]});
            lb = sqrtf(seg_seg_dist2(a0, a1, b0, b1)) * 0.99999f - 1e-6f - A.cap_r - B.cap_r - margins;
            if (lb > cutoff) return lb;
        } else if (A.kind == kPieceAlignedBox) {
            lb = fminf(b0.z, b1.z) - B.cap_r - A.btop - margins - 1e-6f;
            if (lb > cutoff) return lb;
        }
    }
    if (IDENTITY && A.kind == kPieceAlignedBox) {
        // The table: when the lowest point of the link hull lies above the box's top face, inside its footprint, the
        // distance is that point's height above the face - any other hull point is at least as high, so at least as far
        // from the box.  One support call instead of a GJK run; anything else (over the edge, below the top) falls through.
        int hint = 0;
        const F3 low = apply(XB, hb.support(rot_t(XB, F3{0.f, 0.f, -1.f}), hint, true));
        if (low.x >= A.bx0 && low.x <= A.bx1 && low.y >= A.by0 && low.y <= A.by1 && low.z >= A.btop) return (low.z - A.btop) - margins;
    }
    if (defer) { *defer = true; return 3.0e38f; }
    return gjk_distance<IDENTITY>((const SMR_CONST float *)S.verts, A, XA, hb, XB, d, cutoff + margins, stop_below + margins, warm) - margins;
}

__device__ __forceinline__ Xf planet_pose(const DScene &S, int p, int idx) {
    return quat_pose((const SMR_GLOBAL float *)S.planet[p].orbit_pose + 8 * (size_t)idx);
}

// ---------------------------------------------------------------------------------------------
// One collision sample: FK of q and, per link, the table / planet queries it takes part in.
//   want_dist   : compute d_static (A7) and d_moving (A8) - the end-of-step sample
//   want_contact: compute the Bullet-contact flag between planets and robot links (A11)
// ---------------------------------------------------------------------------------------------
struct Sample { float d_static, d_moving, d_self; bool contact; };

// frame of body b (>= 8: rigidly mounted on iiwa_link_7, whose frame is T7)
__device__ __forceinline__ Xf mounted_frame(const DScene &S, int b, const Xf &T7) {
    Xf X;
    X.t = apply(T7, F3{S.body_xyz[b][0], S.body_xyz[b][1], S.body_xyz[b][2]});
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            X.r[3 * i + j] = T7.r[3 * i] * S.body_rot[b][j] + T7.r[3 * i + 1] * S.body_rot[b][3 + j] + T7.r[3 * i + 2] * S.body_rot[b][6 + j];
    return X;
}

// frame of body b for joint vector q
__device__ inline Xf body_frame(const DScene &S, const float q[SMR_NJ], int b) {
    const int link = b < SMR_NLINK ? b : SMR_NLINK - 1;
    Xf T = fk_base(S);
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j)
        if (j < link) fk_advance(S, j, q[j], T);
    return b < SMR_NLINK ? T : mounted_frame(S, b, T);
}

// sines and cosines of a configuration's joint angles: the reset-side kernels build many frames of one candidate
struct Trig { float sn[SMR_NJ], cs[SMR_NJ]; float q_last[3]; };   // q_last: the angles of joints 4, 5, 6 (axes of the self-pair bound tables)
__device__ inline Xf body_frame_sc(const DScene &S, const Trig &P, int b) {
    const int link = b < SMR_NLINK ? b : SMR_NLINK - 1;
    Xf T = fk_base(S);
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j)
        if (j < link) fk_advance_sc(S, j, P.sn[j], P.cs[j], T);
    return b < SMR_NLINK ? T : mounted_frame(S, b, T);
}

// ---------------------------------------------------------------------------------------------
// Self pairs: bound tables.  The relative pose of a body mounted on iiwa_link_7 and its partner link k depends on the joints
// k .. 6 only: two angles for iiwa_link_5, three for iiwa_link_4 - the pairs that are near in most configurations (the ball machine
// sits 6 - 8 cm from iiwa_link_5 half of the time).  For those pairs smr_create tabulates, with the device GJK itself
// (self_table_kernel), a rigorous LOWER bound of the distance at every cell centre of a grid over these angles (the largest
// support-plane bound of the run, rounded down to 4 um).  The distance is 1-Lipschitz in the displacement of the body's points
// relative to the partner, and joint j moves them by at most lever_j |dq_j|, so  d(q) >= tab[cell(q)] - sum_j lever_j |q_j - centre_j|.
// Every caller that only asks "closer than x?" - the start-state / target / ball-target sampling (ctlp.py:2170-2208), the braking
// trajectory checker (ctlp.py:3345-3372), the pruning of the step's pair jobs - is answered by one 2-byte read where the bound clears
// x; the GJK run is kept for the configurations it cannot clear.  The oracle makes the literal queries, so the parity tests check
// the argument, and tests/test_gpu_parity.py compares the bounds with the oracle's distances directly.  A bound <= 0 carries no
// information (overlapping hulls are tabulated as 0).
// ---------------------------------------------------------------------------------------------
// q4, q5, q6: the angles of the last three joints.  i may differ between the lanes.
__device__ inline float self_tab_bound(const DScene &S, int i, float q4, float q5, float q6) {
    const int nd = S.self_tab_nd[i];
    if (nd == 0) return -kNoBound;
    int idx = 0;
    float slack = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d >= nd) continue;
        const int j = SMR_NJ - nd + d;
        const float qj = (j == 4) ? q4 : (j == 5 ? q5 : q6);
        const int n = S.self_tab_n[i][d];
        int c = (int)floorf((qj - S.pos_lo[j]) * S.self_tab_ih[i][d]);
        c = c < 0 ? 0 : (c > n - 1 ? n - 1 : c);
        const float centre = S.pos_lo[j] + ((float)c + 0.5f) * S.self_tab_h[i][d];
        slack += fabsf(qj - centre) * S.self_tab_lever[i][d];
        idx = idx * n + c;
    }
    const float t = (float)((const SMR_GLOBAL uint16_t *)S.self_tab)[S.self_tab_off[i] + idx] * kSelfTabStep;
    return t - slack * 1.0001f - 2e-5f;
}

// A7 self: distance of self-collision pair i = (body a with hull (LP, hull) under T, partner): the partner is another
// chain link (its hull climbed from the global tables, per lane) or the base adapter's cylinder in the world frame
// (btCylinderShapeZ::localGetSupportingVertexWithoutMargin).  Margins subtracted; > cutoff means "farther than cutoff".
// LANE: the pair index differs between lanes (reset groups deal the pairs to their lanes), so nothing is wave uniform.
// bodies / trig (the reset-side kernels): the partner link's hull comes from that LDS pool instead of the global tables, its frame
// from the configuration's sines and cosines.
// TABLE: consult the pair's bound table (above) before the GJK run; lower: see gjk_core (the table builder)
template <bool LANE = false, bool TABLE = true, class HB>
__device__ inline float self_pair_distance(const DScene &S, int i, const float q[SMR_NJ], const DPiece &LP, const HB &hull,
                                           const Xf &T, float cutoff, float stop_below, const BodyPool *bodies = nullptr, const Trig *trig = nullptr,
                                           F3 *warm = nullptr, float *lower = nullptr, const Xf *partner_frame = nullptr) {
    // partner_frame (the reset-side kernels: the lane that tests the pair IS the partner link's lane and has its frame already)
    const int partner = S.self_pair[i][1];
    const int pa = partner >= 0 ? S.link_piece[partner] : -(partner + 1);
    // (the partner link's record: from the LDS pool where there is one - a per-lane read of 36 dwords from global memory otherwise)
    const DPiece A = (LANE && bodies && partner >= 0) ? bodies->piece(partner) : LANE ? load_piece_lane(S.pieces, pa) : load_piece(S.pieces, pa);
    const float margins = A.margin + LP.margin;
    const F3 cb = apply(T, F3{LP.cx, LP.cy, LP.cz});
    const auto sb = [&](F3 d, int &hint, bool first) { return apply(T, hull.support(rot_t(T, d), hint, first)); };
    if (partner >= 0) {
        if (TABLE) {                                         // two bytes that settle nine in ten of the near pairs: before any frame
            const float lbt = trig ? self_tab_bound(S, i, trig->q_last[0], trig->q_last[1], trig->q_last[2]) : self_tab_bound(S, i, q[4], q[5], q[6]);
            if (lbt > cutoff) return lbt;
        }
        const Xf XA = partner_frame ? *partner_frame : trig ? body_frame_sc(S, *trig, partner) : body_frame(S, q, partner);
        const F3 d = apply(XA, F3{A.cx, A.cy, A.cz}) - cb;
        float lb = sqrtf(dot(d, d)) - A.radius - LP.radius - margins;
        if (lb > cutoff) return lb;
        {   // bounding capsules (both bodies are elongated: the spheres above overlap for every neighbour of a link)
            const F3 a0 = apply(XA, F3{A.cap_a[0], A.cap_a[1], A.cap_a[2]}), a1 = apply(XA, F3{A.cap_b[0], A.cap_b[1], A.cap_b[2]});
            const F3 b0 = apply(T, F3{LP.cap_a[0], LP.cap_a[1], LP.cap_a[2]}), b1 = apply(T, F3{LP.cap_b[0], LP.cap_b[1], LP.cap_b[2]});
            lb = sqrtf(seg_seg_dist2(a0, a1, b0, b1)) * 0.99999f - 1e-6f - A.cap_r - LP.cap_r - margins;
            if (lb > cutoff) return lb;
        }
        if (bodies) {
            const LocalHull ha = bodies->hull(partner);
            const auto sa = [&](F3 dd, int &hint, bool first) { return apply(XA, ha.support(rot_t(XA, dd), hint, first)); };
            return gjk_core(sa, sb, d, cutoff + margins, stop_below + margins, warm) - margins;
        }
        const GlobalHull ha = global_hull(S, pa, A);
        const auto sa = [&](F3 dd, int &hint, bool first) { return apply(XA, ha.support(rot_t(XA, dd), hint, first)); };
        const float dist = gjk_core(sa, sb, d, cutoff + margins, stop_below + margins, warm, lower) - margins;
        if (lower) *lower -= margins;
        return dist;
    }
    const F3 d = F3{A.cx, A.cy, A.cz} - cb;
    const float lb = sqrtf(dot(d, d)) - A.radius - LP.radius - margins;
    if (lb > cutoff) return lb;
    const auto sa = [&](F3 dd, int &, bool) {
        const float sxy = sqrtf(dd.x * dd.x + dd.y * dd.y);
        F3 p = sxy > 0.f ? F3{A.cx + A.cyl_r * dd.x / sxy, A.cy + A.cyl_r * dd.y / sxy, A.cz} : F3{A.cx + A.cyl_r, A.cy, A.cz};
        p.z += dd.z < 0.f ? -A.cyl_h : A.cyl_h;
        return p;
    };
    return gjk_core(sa, sb, d, cutoff + margins, stop_below + margins, warm) - margins;
}

__device__ inline Sample collide_sample(const DScene &S, const float q[SMR_NJ], const int idx[SMR_MAX_PLANETS],
                                        bool want_dist, bool want_contact) {
    Sample out;
    out.d_static = S.static_max;
    out.d_self = S.static_max;
    out.d_moving = S.moving_max + 0.002f;          // ctlp.py:3261
    out.contact = false;
    Xf XP[SMR_MAX_PLANETS];
#pragma unroll
    for (int p = 0; p < SMR_MAX_PLANETS; ++p)
        if (p < S.n_planets) XP[p] = planet_pose(S, p, idx[p]);
    // once a contact is certain the moving distance is 0 anyway (ctlp.py:3229-3234)
    const float stop_moving = want_contact ? S.contact_dist : S.collision_dist;
    Xf Tc = fk_base(S);
    Xf I = Tc;   // unused placeholder for the identity branch
    for (int l = 0; l < S.n_bodies; ++l) {
        if (l > 0 && l < SMR_NLINK) fk_advance(S, l - 1, q[l - 1], Tc);
        const Xf T = l < SMR_NLINK ? Tc : mounted_frame(S, l, Tc);
        const int pl = S.link_piece[l];
        const DPiece LP = load_piece(S.pieces, pl);
        const GlobalHull hull = global_hull(S, pl, LP);
        if (want_dist && ((S.static_mask >> l) & 1u)) {
            for (int s = 0; s < S.n_static; ++s) {
                float d = piece_distance<true>(S, S.static_piece[s], I, LP, hull, T, out.d_static, S.collision_dist);
                if (d <= S.static_max && d < out.d_static) out.d_static = d;
            }
        }
        if (want_dist)
            for (int i = 0; i < S.n_self_pairs; ++i) {
                if (S.self_pair[i][0] != l) continue;
                float d = self_pair_distance(S, i, q, LP, hull, T, out.d_self, S.collision_dist);
                if (d <= S.static_max && d < out.d_self) out.d_self = d;
            }
        const bool mov_link = (S.moving_mask >> l) & 1u, con_link = (S.contact_mask >> l) & 1u;
        if (!(mov_link || con_link)) continue;
        const F3 lc = apply(T, F3{LP.cx, LP.cy, LP.cz});
        for (int p = 0; p < S.n_planets; ++p) {
            const int pb = S.planet[p].piece_begin, pc = S.planet[p].piece_count;
            // whole-planet bounding sphere against the link's: skips all of its pieces at once
            const F3 pcen = apply(XP[p], F3{S.planet[p].bx, S.planet[p].by, S.planet[p].bz}) - lc;
            const float plb = sqrtf(dot(pcen, pcen)) - S.planet[p].bradius - LP.radius - LP.margin;
            for (int c = 0; c < pc; ++c) {
                const bool mov = want_dist && mov_link;
                const bool con = want_contact && con_link && !out.contact;
                if (!(mov || con)) continue;
                if (plb > (mov ? fmaxf(fminf(out.d_moving, S.moving_max), con ? S.contact_dist : 0.f) : S.contact_dist)) continue;
                float cutoff = mov ? fminf(out.d_moving, S.moving_max) : S.contact_dist;
                if (con) cutoff = fmaxf(cutoff, S.contact_dist);
                float d = piece_distance<false>(S, pb + c, XP[p], LP, hull, T, cutoff, stop_moving);
                if (mov && d <= S.moving_max && d < out.d_moving) out.d_moving = d;
                if (con && d <= S.contact_dist) out.contact = true;
            }
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// A10: ball.  Pose after k updates (Ball.update, ctlp.py:4103-4126): ballistic position, orientation from
// euler angles e0 + t (0, omega, 0) (pybullet.getQuaternionFromEuler).
// ---------------------------------------------------------------------------------------------
struct BallState { float p0[3], v0[3], e0[3], omega; };

__device__ __forceinline__ BallState load_ball(const DState &st, int B, int e) {
    BallState b;
#pragma unroll
    for (int c = 0; c < 3; ++c) { b.p0[c] = st.ball[c * B + e]; b.v0[c] = st.ball[(3 + c) * B + e]; b.e0[c] = st.ball[(6 + c) * B + e]; }
    b.omega = st.ball[9 * B + e];
    return b;
}

__device__ inline Xf ball_pose(const DScene &S, const BallState &b, int k, float vel[3]) {
    const float t = (float)k * (S.dt / (float)S.updates_per_step);
    Xf X;
    X.t = {b.p0[0] + b.v0[0] * t, b.p0[1] + b.v0[1] * t, b.p0[2] + b.v0[2] * t - 0.5f * 9.81f * t * t};
    float sr, cr, sp, cp, sy, cy;
    sincosf(0.5f * b.e0[0], &sr, &cr);
    sincosf(0.5f * (b.e0[1] + t * b.omega), &sp, &cp);
    sincosf(0.5f * b.e0[2], &sy, &cy);
    float row[7] = {0.f, 0.f, 0.f, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
    float x = row[3], y = row[4], z = row[5], w = row[6];
    X.r[0] = 1.f - 2.f * (y * y + z * z); X.r[1] = 2.f * (x * y - z * w);       X.r[2] = 2.f * (x * z + y * w);
    X.r[3] = 2.f * (x * y + z * w);       X.r[4] = 1.f - 2.f * (x * x + z * z); X.r[5] = 2.f * (y * z - x * w);
    X.r[6] = 2.f * (x * z - y * w);       X.r[7] = 2.f * (y * z + x * w);       X.r[8] = 1.f - 2.f * (x * x + y * y);
    if (vel) { vel[0] = b.v0[0]; vel[1] = b.v0[1]; vel[2] = b.v0[2] - 9.81f * t; }
    return X;
}

// first update u in 1..updates_per_step at which the ball is retired for having passed the robot or reached
// the table / ground (ctlp.py:2836-2848, 4196-4211); updates_per_step + 1 if it survives the step
__device__ __forceinline__ int ball_retire_update(const DScene &S, int k, int kmax, int khit) {
    int u1 = kmax - k + 1, u2 = khit - k;              // k + u > kmax  |  k + u >= khit
    int u = u1 < u2 ? u1 : u2;
    if (u < 1) u = 1;
    return u > S.updates_per_step ? S.updates_per_step + 1 : u;
}

// The same queries for ONE robot link (the step path runs one lane per (env, sample, link): 8x the
// wavefronts of collide_sample, which is what keeps the SIMDs occupied while lanes wait on each other).
// T is the link frame; every loop bound and piece index is wave uniform.
template <class HB>
__device__ inline Sample collide_link(const DScene &S, int l, const DPiece &LP, const HB &hull, const Xf &T,
                                      const int idx[SMR_MAX_PLANETS], const Xf *XB, bool want_static, bool want_moving,
                                      bool want_contact, const float *q_self = nullptr, bool skip_planets = false, float moving_cut = -1.f,
                                      bool *static_defer = nullptr, int *ball_defer = nullptr) {
    // ball_defer (ball scenes in the job form): the GJK run against the ball is not made - *ball_defer = 1 | mov << 1 | con << 2 when the
    // bounds could not settle it (the job kernels make it), 0 otherwise
    // static_defer (scenes with auto_static): a GJK run against the static piece is not made here - the caller hands it to planet_gjk
    // moving_cut: moving distances above it are "no closest point" (default: the scene's moving_max; the contact sweep asks for less)
    const float moving_max = moving_cut >= 0.f ? moving_cut : S.moving_max;
    Sample out;
    out.d_static = S.static_max;
    out.d_self = S.static_max;
    out.d_moving = moving_max + 0.002f;
    out.contact = false;
    if (want_static && ((S.static_mask >> l) & 1u)) {
        for (int s = 0; s < S.n_static; ++s) {
            float d = piece_distance<true>(S, S.static_piece[s], T, LP, hull, T, out.d_static, S.collision_dist, nullptr, static_defer);
            if (d <= S.static_max && d < out.d_static) out.d_static = d;
        }
    }
    if (q_self)                                         // end-of-step sample: the self-collision pairs of this body
        for (int i = 0; i < S.n_self_pairs; ++i) {
            if (S.self_pair[i][0] != l) continue;
            float d = self_pair_distance(S, i, q_self, LP, hull, T, out.d_self, S.collision_dist);
            if (d <= S.static_max && d < out.d_self) out.d_self = d;
        }
    if (skip_planets) return out;                       // the caller runs planet_queries_auto instead
    // an explicit moving_cut (contact sweep) asks for the distance whatever the reward's link set says
    const bool mov = want_moving && (((S.moving_mask >> l) & 1u) || moving_cut >= 0.f);
    const bool con_link = (S.contact_mask >> l) & 1u;
    if (!(mov || (want_contact && con_link))) return out;
    const float stop_moving = want_contact ? S.contact_dist : S.collision_dist;
    if (S.use_ball) {                                   // one moving piece: the ball, pose XB
        const bool con = want_contact && con_link;
        float cutoff = mov ? moving_max : S.contact_dist;
        if (con) cutoff = fmaxf(cutoff, S.contact_dist);
        bool pend = false;
        float d = piece_distance<false>(S, S.ball_piece, *XB, LP, hull, T, cutoff, stop_moving, nullptr, ball_defer ? &pend : nullptr);
        if (ball_defer) *ball_defer = pend ? (1 | ((int)mov << 1) | ((int)con << 2)) : 0;
        if (pend) return out;
        if (mov && d <= moving_max && d < out.d_moving) out.d_moving = d;
        if (con && d <= S.contact_dist) out.contact = true;
        return out;
    }
    const F3 lc = apply(T, F3{LP.cx, LP.cy, LP.cz});
    // Upper bound on the minimum distance from interior points (3 of the link x 1 per piece): pairs that cannot
    // beat it are culled by their bounding spheres or leave GJK after one or two iterations.  The pair that
    // attains the minimum is never cut: its distance is <= the bound.
    float cut_moving = moving_max;
    if (mov) {
        const F3 l0 = apply(T, F3{LP.in[0][0], LP.in[0][1], LP.in[0][2]}), l1 = apply(T, F3{LP.in[1][0], LP.in[1][1], LP.in[1][2]}),
                 l2 = apply(T, F3{LP.in[2][0], LP.in[2][1], LP.in[2][2]});
        float ub2 = 3.0e38f, mar = 0.f;
        for (int p = 0; p < S.n_planets; ++p) {
            const Xf XP = planet_pose(S, p, idx[p]);
            const int pb = S.planet[p].piece_begin, pc = S.planet[p].piece_count;
            for (int c = 0; c < pc; ++c) {
                const DPiece PP = load_piece(S.pieces, pb + c);
                const F3 m = apply(XP, F3{PP.in[0][0], PP.in[0][1], PP.in[0][2]});
                const F3 e0 = m - l0, e1 = m - l1, e2 = m - l2;
                ub2 = fminf(ub2, fminf(dot(e0, e0), fminf(dot(e1, e1), dot(e2, e2))));
                mar = PP.margin;
            }
        }
        cut_moving = fminf(cut_moving, (sqrtf(ub2) - LP.margin - mar) * 1.00001f + 1e-6f);
        cut_moving = fmaxf(cut_moving, S.contact_dist);        // never below the contact threshold tested with it
    }
    for (int p = 0; p < S.n_planets; ++p) {
        const Xf XP = planet_pose(S, p, idx[p]);
        const int pb = S.planet[p].piece_begin, pc = S.planet[p].piece_count;
        // whole-planet bounding sphere against the link's: skips all of its pieces at once
        const F3 pcen = apply(XP, F3{S.planet[p].bx, S.planet[p].by, S.planet[p].bz}) - lc;
        const float plb = sqrtf(dot(pcen, pcen)) - S.planet[p].bradius - LP.radius - LP.margin;
        for (int c = 0; c < pc; ++c) {
            const bool con = want_contact && con_link && !out.contact;
            if (!(mov || con)) continue;
            float cutoff = mov ? fminf(out.d_moving, cut_moving) : S.contact_dist;
            if (con) cutoff = fmaxf(cutoff, S.contact_dist);
            if (plb > cutoff) continue;
            SMR_CNT_SLOT(pb + c);
            float d = piece_distance<false>(S, pb + c, XP, LP, hull, T, cutoff, stop_moving);
            if (mov && d <= moving_max && d < out.d_moving) out.d_moving = d;
            if (con && d <= S.contact_dist) out.contact = true;
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// Lane-autonomous planet queries.  In the lockstep loop of collide_link a wavefront visits a piece as soon as ONE of its
// 64 lanes cannot rule it out (13 of 25 pieces per link, ~10 lanes busy each, as long as the slowest of them).  Here every
// lane keeps its own candidates - a byte key per piece from the capsule-capsule lower bound - and takes them best first:
// the piece with the smallest bound usually attains the minimum, after which the running minimum rules out nearly all
// others (1.7 GJK runs per lane instead of 2.5).  One GJK iteration per trip of the wave's loop on the lane's CURRENT
// pair, the next candidate as soon as a pair is settled: a wavefront takes as many trips as its busiest lane has iterations.
// The lanes' pieces differ, so the obstacle side is scanned per lane from an LDS copy of the planets' vertices.
// ---------------------------------------------------------------------------------------------
__device__ inline void stage_planets(const DScene &S, int tid, int nthreads, float *pv, float *pd) {
    const SMR_GLOBAL float *gv = (const SMR_GLOBAL float *)S.auto_verts, *gd = (const SMR_GLOBAL float *)S.auto_desc;
    for (int i = tid; i < 4 * S.auto_vn; i += nthreads) pv[i] = gv[i];
    for (int i = tid; i < kPlanetDescDwords * S.auto_pn; i += nthreads) pd[i] = gd[i];
}

// pass 1 (lockstep, scalar piece loads): the key of every piece for the lanes with `live` set; pieces whose lower bound exceeds
// cutoff_max get key 255.  Whole-planet and piece bounding spheres first; a wavefront none of whose lanes passes them skips the
// capsule arithmetic.  The lane's 28 key bytes live in LDS (keys[w * kstride], w < 7): registers are what limits the collide kernel.
// Returns a lower bound of the distance (margins subtracted) between the body and all planet pieces - the smallest of the bounds
// the keys come from, beyond cutoff_max too (the contact sweep jumps by it).
__device__ inline float planet_keys(const DScene &S, uint32_t *keys, int kstride, const DPiece &LP, const Xf &T, const int idx[SMR_MAX_PLANETS],
                                    bool live, float cutoff_max) {
    const F3 lc = apply(T, F3{LP.cx, LP.cy, LP.cz});
    float lb_min = 3.0e38f;
#pragma unroll
    for (int w = 0; w < 7; ++w) keys[w * kstride] = 0xFFFFFFFFu;
    if (__any(live)) {
        const F3 l0 = apply(T, F3{LP.cap_a[0], LP.cap_a[1], LP.cap_a[2]}), l1 = apply(T, F3{LP.cap_b[0], LP.cap_b[1], LP.cap_b[2]});
        const int second_planet = S.n_planets > 1 ? 1 : 0;
        const Xf X0 = planet_pose(S, 0, idx[0]);
        const Xf X1 = planet_pose(S, second_planet, idx[second_planet]);
        const F3 e0 = apply(X0, F3{S.planet[0].bx, S.planet[0].by, S.planet[0].bz}) - lc,
                 e1 = apply(X1, F3{S.planet[second_planet].bx, S.planet[second_planet].by, S.planet[second_planet].bz}) - lc;
        const float lbw0 = __builtin_amdgcn_sqrtf(dot(e0, e0)) * 0.99999f - 1e-6f - S.planet[0].bradius - LP.radius - LP.margin;
        const float lbw1 = __builtin_amdgcn_sqrtf(dot(e1, e1)) * 0.99999f - 1e-6f - S.planet[second_planet].bradius - LP.radius - LP.margin;
        const bool near0 = live && lbw0 <= cutoff_max;
        const bool near1 = live && lbw1 <= cutoff_max;
        if (!near0) lb_min = fminf(lb_min, lbw0);
        if (S.n_planets > 1 && !near1) lb_min = fminf(lb_min, lbw1);
#pragma unroll 1
        for (int w = 0; w < 7; ++w) {
            uint32_t word = 0xFFFFFFFFu;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = 4 * w + b;
                uint32_t key = 255u;
                if (i < S.planet_pn) {
                    const DPiece PP = load_piece(S.pieces, S.planet_p0 + i);
                    const bool second = S.n_planets > 1 && S.planet_p0 + i >= S.planet[1].piece_begin;
                    if (second ? near1 : near0) {
                        const float margins = PP.margin + LP.margin;
                        const F3 cen = {PP.cx, PP.cy, PP.cz};
                        const F3 dc = (second ? apply(X1, cen) : apply(X0, cen)) - lc;
                        const float lb_s = __builtin_amdgcn_sqrtf(dot(dc, dc)) * 0.99999f - 1e-6f - PP.radius - LP.radius - margins;
                        float lb = lb_s;
                        if (lb_s <= cutoff_max) {
                            const F3 ca = {PP.cap_a[0], PP.cap_a[1], PP.cap_a[2]}, cb = {PP.cap_b[0], PP.cap_b[1], PP.cap_b[2]};
                            const F3 a0 = second ? apply(X1, ca) : apply(X0, ca), a1 = second ? apply(X1, cb) : apply(X0, cb);
                            const float lb_c = __builtin_amdgcn_sqrtf(seg_seg_dist2(a0, a1, l0, l1)) * 0.99999f - 1e-6f - PP.cap_r - LP.cap_r - margins;
                            lb = fmaxf(lb_s, lb_c);
                            if (lb <= cutoff_max) key = (uint32_t)fminf(fmaxf(lb * (1.0f / kKeyStep), 0.f), 254.f);
                        }
                        lb_min = fminf(lb_min, lb);
                    }
                }
                word = (word >> 8) | (key << 24);
            }
            keys[w * kstride] = word;
        }
    }
    return live ? lb_min : 3.0e38f;
}

// pass 2: lane-autonomous GJK over the lane's candidate keys, best first (see above).  mov: the minimum distance up to moving_cut
// goes to out.d_moving; want_contact: out.contact.
template <class HB>
__device__ inline void planet_gjk(const DScene &S, const float *pv, const float *pd, uint32_t *keys, int kstride,
                                  const DPiece &LP, const HB &hull, const Xf &T, const int idx[SMR_MAX_PLANETS], bool mov,
                                  bool want_contact, float moving_max, Sample &out, bool static_pending = false) {
    // static_pending: the lane's GJK run against the static piece (descriptor S.auto_static; collide_link left it out) is made here,
    // first - one lane in a hundred has one, and as a phase of its own it held up every second wavefront for its 7 iterations
    const float stop_moving = want_contact ? S.contact_dist : S.collision_dist;
    bool finished = !(mov || want_contact);
    bool cur_static = false;
    // pass 2: lane-autonomous GJK, best candidate first
    GjkRun g;
    g.start(F3{1.f, 0.f, 0.f}, 0.f, 0.f);     // defined state for the lanes that never activate
    bool active = false, con_cur = false;
    Xf XA = T;
    const float *av = pv;
    int ac = 1;
    float margins = 0.f;
    // pass 2 works in the LINK frame (XA = the obstacle piece's frame seen from it): the link-side support needs no
    // transform and the link's world frame T leaves the loop's registers
    const F3 lcl = {LP.cx, LP.cy, LP.cz};
    // (the run against the static piece stays in the WORLD frame - the metre-sized table loses three digits when its corners are
    // carried into the link frame -, so there the link side is the one that is transformed)
    const auto sb = [&](F3 d, int &hint, bool first) {
        const F3 p = hull.support(cur_static ? rot_t(T, d) : d, hint, first);
        return cur_static ? apply(T, p) : p;
    };
    const auto sa = [&](F3 d, int &, bool) {
        const F3 dl = rot_t(XA, d);
        float best = -3.0e38f;
        F3 bp = {0.f, 0.f, 0.f};
        int bi = 0;
        for (int i = 0; i < ac; i += 4) {                   // ac is a multiple of 4 (padded with copies of the last vertex)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const V4 q = load_v4(av + 4 * (i + u));
                const float sq = fmaf(q[2], dl.z, fmaf(q[1], dl.y, q[0] * dl.x));
                const bool gt = sq > best;
                best = gt ? sq : best;
                bi = gt ? i + u : bi;
            }
        }
        bp = {av[4 * bi], av[4 * bi + 1], av[4 * bi + 2]};
        return apply(XA, bp);
    };
    int prof_trips = 0;
    for (int trip = 0; trip < 33 * kMaxPlanetPieces; ++trip) {
        ++prof_trips;
        if (!active && static_pending) {
            static_pending = false;
            const float *d = pd + kPlanetDescDwords * S.auto_static;
            av = pv + 4 * __float_as_int(d[0]);
            ac = __float_as_int(d[1]);
            margins = d[2] + LP.margin;
#pragma unroll
            for (int i = 0; i < 9; ++i) XA.r[i] = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
            XA.t = F3{0.f, 0.f, 0.f};
            g.start(F3{d[4], d[5], d[6]} - apply(T, lcl), out.d_static + margins, S.collision_dist + margins);
            active = true; cur_static = true;
        }
        if (!active && !finished) {
            con_cur = want_contact && !out.contact;
            float cutoff = mov ? fminf(out.d_moving, moving_max) : S.contact_dist;
            if (con_cur) cutoff = fmaxf(cutoff, S.contact_dist);
            uint32_t best = 0xFFFFFFFFu, kw[7];
#pragma unroll
            for (int w = 0; w < 7; ++w) kw[w] = keys[w * kstride];
#pragma unroll
            for (int w = 0; w < 7; ++w)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t cand = (((kw[w] >> (8 * b)) & 0xFFu) << 5) | (uint32_t)(4 * w + b);
                    best = cand < best ? cand : best;
                }
            const int key = (int)(best >> 5), c = (int)(best & 31u);
            if (!(mov || con_cur) || key == 255 || (float)key * kKeyStep > cutoff) finished = true;
            else {
                keys[(c >> 2) * kstride] |= 0xFFu << (8 * (c & 3));
                const float *d = pd + kPlanetDescDwords * c;
                av = pv + 4 * __float_as_int(d[0]);
                ac = __float_as_int(d[1]);
                margins = d[2] + LP.margin;
                const int p = __float_as_int(d[3]);
                XA = inv_compose(T, quat_pose((const SMR_GLOBAL float *)(p ? S.planet[1].orbit_pose : S.planet[0].orbit_pose) + 8 * (size_t)(p ? idx[1] : idx[0])));
                SMR_CNT_SLOT(S.planet_p0 + c);
                g.start(apply(XA, F3{d[4], d[5], d[6]}) - lcl, cutoff + margins, stop_moving + margins);
                active = true;
            }
        }
        if (!__any(active)) break;
        if (active) {
            float dist;
            if (gjk_iterate(g, sa, sb, dist)) {
                const float d = dist - margins;
                if (cur_static) { if (d <= S.static_max && d < out.d_static) out.d_static = d; cur_static = false; }
                else {
                    if (mov && d <= moving_max && d < out.d_moving) out.d_moving = d;
                    if (con_cur && d <= S.contact_dist) out.contact = true;
                }
                active = false;
            }
        }
    }
    SMR_WAVE_REC(3, prof_trips); (void)prof_trips;
}


template <class HB>
__device__ inline void planet_queries_auto(const DScene &S, const float *pv, const float *pd, uint32_t *keys, int kstride,
                                           const DPiece &LP, const HB &hull, const Xf &T, const int idx[SMR_MAX_PLANETS], bool mov,
                                           bool want_contact, Sample &out) {
    float cutoff_max = mov ? S.moving_max : S.contact_dist;
    if (want_contact) cutoff_max = fmaxf(cutoff_max, S.contact_dist);
    planet_keys(S, keys, kstride, LP, T, idx, mov || want_contact, cutoff_max);
    planet_gjk(S, pv, pd, keys, kstride, LP, hull, T, idx, mov, want_contact, S.moving_max, out);
}

// ---------------------------------------------------------------------------------------------
// A11, the intermediate contact samples k = 1 .. N-1 of a step (the reference tests contact after every one of its 24 obstacle
// updates per step, ctlp.py:2631-2637,2849-2861).  The end-of-step lane of a body has, as a by-product of its distance queries, a
// rigorous LOWER bound of the distance between the body and every moving piece at t = dt.  Between two samples no point of the
// body moves farther than (dt / N) sum_j max|qdot_j| sweep_R[b][j] and no point of a moving obstacle farther than its per-update
// bound, so the samples N-1, N-2, .. N-m are contact free while m x that motion stays below bound - contact_dist; the bound the
// previous step left behind covers the first samples 1 .. m' the same way.  Only the samples in between become work items
// (lane, sample) in an LDS queue of the wavefront, which its lanes then work off together, 64 at a time - dense rounds however
// unevenly the items are spread over the lanes.  The flags are those of testing all N samples (which is what the float64 oracle
// does); most (env, body) lanes need no item at all, so N = 24 costs little more than N = 1.
// ---------------------------------------------------------------------------------------------
// largest |velocity| of a constant-jerk segment (v0, a0 -> a1 over dt), rounded up
__device__ __forceinline__ float cubic_speed_bound(float v0, float a0, float a1, float dt) {
    const float v1 = v0 + 0.5f * (a0 + a1) * dt;
    float m = fmaxf(fabsf(v0), fabsf(v1));
    if (a0 * a1 < 0.f) m = fmaxf(m, fabsf(v0 + 0.5f * a0 * (a0 * dt / (a0 - a1))));   // the acceleration changes sign inside the step
    return m * 1.00001f + 1e-7f;
}

// number of samples after the current one that a lower bound lb (> contact_dist) rules out, motion per sample `step` (> 0)
__device__ __forceinline__ int sweep_skip(float lb, float contact_dist, float step) {
    const float x = (lb - contact_dist - 1e-6f) / (step * 1.0001f + 1e-7f);      // i <= x  =>  i step < lb - contact_dist
    return x > 0.f ? (int)fminf(x, 1.0e6f) : 0;
}

// Lower bound of the distance (margins subtracted) between the body (piece LP under T) and every piece of the planets at orbit
// indices idx.  A bound above `enough` is returned as soon as it is known (whole-planet spheres, then piece spheres); only pieces
// that stay below it get the capsule-capsule bound.  Piece loads are wave uniform (scalar).
__device__ inline float planets_lower_bound(const DScene &S, const DPiece &LP, const Xf &T, const int idx[SMR_MAX_PLANETS], float enough) {
    const F3 lc = apply(T, F3{LP.cx, LP.cy, LP.cz});
    const F3 l0 = apply(T, F3{LP.cap_a[0], LP.cap_a[1], LP.cap_a[2]}), l1 = apply(T, F3{LP.cap_b[0], LP.cap_b[1], LP.cap_b[2]});
    float lb = 3.0e38f;
    for (int p = 0; p < S.n_planets; ++p) {
        const Xf X = planet_pose(S, p, idx[p]);
        const F3 e = apply(X, F3{S.planet[p].bx, S.planet[p].by, S.planet[p].bz}) - lc;
        const float lbw = sqrtf(dot(e, e)) * 0.99999f - 1e-6f - S.planet[p].bradius - LP.radius - LP.margin;
        float lbp = lbw;
        if (__any(lbw <= enough)) {                           // some lane of the wavefront is near this planet: its pieces
            float m = 3.0e38f;
            const int pb = S.planet[p].piece_begin, pc = S.planet[p].piece_count;
            for (int c = 0; c < pc; ++c) {
                const DPiece PP = load_piece(S.pieces, pb + c);
                const float margins = PP.margin + LP.margin;
                const F3 dc = apply(X, F3{PP.cx, PP.cy, PP.cz}) - lc;
                float b = sqrtf(dot(dc, dc)) * 0.99999f - 1e-6f - PP.radius - LP.radius - margins;
                if (b <= enough && lbw <= enough) {
                    const F3 a0 = apply(X, F3{PP.cap_a[0], PP.cap_a[1], PP.cap_a[2]}), a1 = apply(X, F3{PP.cap_b[0], PP.cap_b[1], PP.cap_b[2]});
                    b = fmaxf(b, sqrtf(seg_seg_dist2(a0, a1, l0, l1)) * 0.99999f - 1e-6f - PP.cap_r - LP.cap_r - margins);
                }
                m = fminf(m, b);
            }
            if (lbw <= enough) lbp = fmaxf(lbw, m);
        }
        lb = fminf(lb, lbp);
    }
    return lb;
}

// ... and against the ball at pose XB: the ball piece lies inside the sphere of radius ball_bound_radius about its frame origin
__device__ inline float ball_lower_bound(const DScene &S, const DPiece &LP, const Xf &T, const Xf &XB) {
    const F3 lc = apply(T, F3{LP.cx, LP.cy, LP.cz});
    const F3 l0 = apply(T, F3{LP.cap_a[0], LP.cap_a[1], LP.cap_a[2]}), l1 = apply(T, F3{LP.cap_b[0], LP.cap_b[1], LP.cap_b[2]});
    const F3 e = XB.t - lc;
    const float lb_s = sqrtf(dot(e, e)) * 0.99999f - 1e-6f - LP.radius;
    const float lb_c = sqrtf(seg_seg_dist2(XB.t, XB.t, l0, l1)) * 0.99999f - 1e-6f - LP.cap_r;
    return fmaxf(lb_s, lb_c) - S.ball_bound_radius - LP.margin;
}

// One lane of the collide kernel: body l of one env at the end-of-step sample (table / planet / ball distances of the reward and
// the contact test at t = dt), and how many of the intermediate contact samples 1 .. N-1 still need a test (k_need, see above).
// The state is read through strided pointers (the kernel: [component][B] arrays; the host emulation: plain arrays).
struct LaneInput {
    const float *q, *v, *a, *a1; int stride;           // joint j at [j * stride]
    int planet_idx[SMR_MAX_PLANETS], moving_hit;
    BallState ball; int ball_k0, ball_kmax, ball_khit, ball_active;
    float lb_start;                                     // lower bound of the moving-piece distance at t = 0 (the previous step's lb_end; < 0: none)
};
// k_first .. k_need: the intermediate samples that still need a test (empty when k_first > k_need); lb_end: the bound at t = dt
struct LaneResult { bool contact; float d_static, d_moving; int k_first, k_need; float lb_end; bool static_pending;
                    unsigned best; float ub;      // KEYS_ONLY: the best candidate (key << 5 | piece; ~0u: none) and an upper bound of ITS distance
                    Xf frame;                     // KEYS_ONLY: the body's frame at t = dt, handed on to the jobs of this (env, body)
                    float lb1;                    // KEYS_ONLY: lower bound of the best candidate's distance from the same support pair (0: none)
                    int ball_pending; };          // ball jobs: the end-of-step run against the ball was not made (bit 0: asked for, 1: distance, 2: contact)

// frame of body l at time t of the step's cubic
template <bool SELF, bool FAST = false>
__device__ __forceinline__ Xf lane_frame(const DScene &S, int l, const LaneInput &in, float t) {
    const int chain = l < SMR_NLINK ? l : SMR_NLINK - 1;
    Xf T = fk_base(S);
    for (int j = 0; j < chain; ++j) {                   // FK up to this body's link; joint values straight from the cubic
        const float qj = interp_pos(in.q[j * in.stride], in.v[j * in.stride], in.a[j * in.stride], in.a1[j * in.stride], t, S.dt);
        if (FAST) fk_advance_fast(S, j, qj, T); else fk_advance(S, j, qj, T);
    }
    if (SELF && l >= SMR_NLINK) T = mounted_frame(S, l, T);
    return T;
}

// KEYS_ONLY (the keys kernel of the pair-job path): no GJK run against a moving piece is made - the lane leaves its candidate keys (and
// the pending table run, static_pending) to the job kernels, and plans its intermediate samples from the key bounds alone
// the same for a body index that differs between the lanes (job kernels): the 28 state words are loaded up front, all in flight at once
// (in the loop above a lane-dependent trip count keeps the compiler from hoisting them: seven dependent round trips to HBM), and
// the chain is walked with the joint loop unrolled under the lane's predicate
template <bool FAST>
__device__ inline Xf lane_frame_any(const DScene &S, int b, const LaneInput &in, float t) {
    float qs[SMR_NJ];
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) qs[j] = interp_pos(in.q[j * in.stride], in.v[j * in.stride], in.a[j * in.stride], in.a1[j * in.stride], t, S.dt);
    const int link = b < SMR_NLINK ? b : SMR_NLINK - 1;
    Xf T = fk_base(S);
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j)
        if (j < link) { if (FAST) fk_advance_fast(S, j, qs[j], T); else fk_advance(S, j, qs[j], T); }
    return b < SMR_NLINK ? T : mounted_frame(S, b, T);
}

template <bool SELF, bool AUTO, bool KEYS_ONLY = false, class HB>
__device__ inline LaneResult collide_lane(const DScene &S, int l, const DPiece &LP, const HB &hull, const LaneInput &in,
                                          const float *lds_pv, const float *lds_pd, uint32_t *keys, int kstride) {
    LaneResult out = {false, S.static_max, S.moving_max + 0.002f, 1, 0, -1.f, false, 0xFFFFFFFFu, 3.0e38f};
    out.lb1 = 0.f; out.ball_pending = 0;
    const int N = S.checks_per_step, upd = S.updates_per_step / N;
    const bool con_link = (S.contact_mask >> l) & 1u;
    bool want_contact, sweep_contact, moving_alive = true;
    int k_last = N - 1;                                 // last intermediate sample that is tested at all
    if (!AUTO && S.use_ball) {           // (the lane-autonomous planet queries exist in planet scenes only)
        // the ball is tested at a sample only while it is in flight (ctlp.py:2836-2861): not yet retired by the
        // update counters, and inside the 1.25 m cylinder around the robot
        const int u_end = ball_retire_update(S, in.ball_k0, in.ball_kmax, in.ball_khit);
        moving_alive = in.ball_active && (N * upd < u_end);       // at the end-of-step sample
        want_contact = moving_alive && con_link;
        sweep_contact = in.ball_active != 0 && con_link;
        const int kl = (u_end - 1) / upd;                          // samples with k upd < u_end
        k_last = kl < k_last ? kl : k_last;
    } else {
        want_contact = sweep_contact = S.term_moving && S.n_planets > 0 && in.moving_hit == 0 && con_link;
    }
    if (k_last < 1) sweep_contact = false;
    const bool want_static = (S.static_mask >> l) & 1u;
    const bool want_moving = moving_alive && ((S.moving_mask >> l) & 1u);
    if (!(want_static || want_moving || want_contact || sweep_contact)) {
        if (KEYS_ONLY)
#pragma unroll
            for (int w = 0; w < 7; ++w) keys[w * kstride] = 0xFFFFFFFFu;
        return out;
    }
    const Xf T = lane_frame<SELF>(S, l, in, S.dt);
    if (KEYS_ONLY) out.frame = T;
    float lb = -1.f;                                    // lower bound of the distance to every moving piece at t = dt (< 0: none)
    Sample r;
    if (!AUTO && S.use_ball) {           // (the lane-autonomous planet queries exist in planet scenes only)
        const Xf XB = ball_pose(S, in.ball, in.ball_k0 + N * upd, nullptr);
        const bool near = XB.t.x * XB.t.x + XB.t.y * XB.t.y < S.ball_active_xy_radius * S.ball_active_xy_radius;
        r = collide_link(S, l, LP, hull, T, nullptr, &XB, want_static, want_moving, want_contact && near, nullptr, false, -1.f, nullptr,
                         S.ball_jobs ? &out.ball_pending : nullptr);
        if (moving_alive && sweep_contact) lb = ball_lower_bound(S, LP, T, XB);
    } else {
        int idx[SMR_MAX_PLANETS] = {0, 0};
#pragma unroll
        for (int p = 0; p < SMR_MAX_PLANETS; ++p)
            if (p < S.n_planets) idx[p] = (in.planet_idx[p] + N * upd) % S.planet[p].n_orbit;
        const unsigned long long c0 = SMR_DEV_CLOCK();
        bool static_pending = false;
        r = collide_link(S, l, LP, hull, T, idx, nullptr, want_static, want_moving, want_contact, nullptr, AUTO, -1.f,
                         (AUTO && S.auto_static >= 0) ? &static_pending : nullptr);
        if (AUTO) {
            float cutoff_max = want_moving ? S.moving_max : S.contact_dist;
            if (want_contact) cutoff_max = fmaxf(cutoff_max, S.contact_dist);
            const unsigned long long c1 = SMR_DEV_CLOCK();
            lb = planet_keys(S, keys, kstride, LP, T, idx, want_moving || want_contact, cutoff_max);
            const unsigned long long c2 = SMR_DEV_CLOCK();
            if (KEYS_ONLY) {
                out.static_pending = static_pending;
                // The best candidate (smallest key) and an UPPER bound of its distance from ONE pair of support points - the vector GJK's
                // first iteration would produce, a point of the Minkowski difference.  The minimum of these bounds over the env's bodies
                // prunes the env's pairs nearly as well as the exact distance of its best pair (4.6 pairs per env survive instead of
                // 3.8, uniformly drawn states) and costs no kernel of its own.
                uint32_t best = 0xFFFFFFFFu;
#pragma unroll
                for (int w = 0; w < 7; ++w) {
                    const uint32_t kw = keys[w * kstride];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t cand = (((kw >> (8 * b)) & 0xFFu) << 5) | (uint32_t)(4 * w + b);
                        best = cand < best ? cand : best;
                    }
                }
                if ((best >> 5) != 255u && want_moving) {
                    const int c = (int)(best & 31u);
                    const float *d = lds_pd + kPlanetDescDwords * c;
                    const float *av = lds_pv + 4 * __float_as_int(d[0]);
                    const int ac = __float_as_int(d[1]), p = __float_as_int(d[3]);
                    const Xf XA = inv_compose(T, quat_pose((const SMR_GLOBAL float *)(p ? S.planet[1].orbit_pose : S.planet[0].orbit_pose) + 8 * (size_t)(p ? idx[1] : idx[0])));
                    const F3 v = apply(XA, F3{d[4], d[5], d[6]}) - F3{LP.cx, LP.cy, LP.cz};      // piece centre - body centre, body frame
                    const F3 dl = rot_t(XA, neg(v));
                    float bs = -3.0e38f;
                    int bi = 0;
                    for (int i = 0; i < ac; i += 4)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const V4 q = load_v4(av + 4 * (i + u));
                            const float sq = fmaf(q[2], dl.z, fmaf(q[1], dl.y, q[0] * dl.x));
                            const bool gt = sq > bs;
                            bs = gt ? sq : bs;
                            bi = gt ? i + u : bi;
                        }
                    int hint = 0;
                    const F3 w = apply(XA, F3{av[4 * bi], av[4 * bi + 1], av[4 * bi + 2]}) - hull.support(v, hint, true);
                    out.ub = (sqrtf(dot(w, w)) - (d[2] + LP.margin)) * 1.00001f + 1e-6f;
                    // ... and the LOWER bound the same pair gives: it is the first iteration of the pair's GJK run (same start vector, same
                    // supports), whose separating-plane test v.w / |v| - margins > cutoff settles 56 % of all jobs (round 4 profile).  The
                    // enumeration knows the cutoff - the env's upper bound - and drops the candidate there instead of making it a job.
                    const float vw = dot(v, w);
                    if (vw > 0.f) out.lb1 = (vw / sqrtf(dot(v, v)) - (d[2] + LP.margin)) * 0.99999f - 1e-6f;
                }
                out.best = (best >> 5) != 255u ? best : 0xFFFFFFFFu;
            } else planet_gjk(S, lds_pv, lds_pd, keys, kstride, LP, hull, T, idx, want_moving, want_contact, S.moving_max, r, static_pending);
            SMR_DEV_COUNT(16, c1 - c0); SMR_DEV_COUNT(17, c2 - c1); SMR_DEV_COUNT(18, SMR_DEV_CLOCK() - c2);
            SMR_WAVE_REC(0, (c1 - c0) >> 4); SMR_WAVE_REC(1, (c2 - c1) >> 4); SMR_WAVE_REC(2, (SMR_DEV_CLOCK() - c2) >> 4);
            if (!(want_moving || want_contact)) lb = -1.f;
        } else if (sweep_contact && N > 1) {
            lb = planets_lower_bound(S, LP, T, idx, S.moving_max);
        }
    }
    out.contact = r.contact; out.d_static = r.d_static; out.d_moving = r.d_moving;
    if (N > 1 && sweep_contact && !r.contact) {
        // motion between two samples: the body's points and the moving pieces
        const int chain = l < SMR_NLINK ? l : SMR_NLINK - 1;
        float step = 0.f;
        for (int j = 0; j < chain; ++j) step += cubic_speed_bound(in.v[j * in.stride], in.a[j * in.stride], in.a1[j * in.stride], S.dt) * S.sweep_R[l][j];
        step *= S.dt / (float)N;
        if (!AUTO && S.use_ball) {           // (the lane-autonomous planet queries exist in planet scenes only)
            const float udt = S.dt / (float)S.updates_per_step, t0 = (float)in.ball_k0 * udt, t1 = t0 + S.dt;
            const float vz = fmaxf(fabsf(in.ball.v0[2] - 9.81f * t0), fabsf(in.ball.v0[2] - 9.81f * t1));
            step += sqrtf(in.ball.v0[0] * in.ball.v0[0] + in.ball.v0[1] * in.ball.v0[1] + vz * vz) * 1.0001f * udt * (float)upd;
        } else {
            float ps = 0.f;
            for (int p = 0; p < S.n_planets; ++p) ps = fmaxf(ps, S.planet_step[p]);
            step += ps * (float)upd;
        }
        const int safe = lb > S.contact_dist ? sweep_skip(lb, S.contact_dist, step) : 0;      // samples N-1 .. N-safe are contact free
        const int kn = N - 1 - safe;
        out.k_need = kn < k_last ? (kn > 0 ? kn : 0) : k_last;
        // ... and from the other end: the samples 1 .. m are covered by the bound at t = 0, the end-of-step bound of the previous step
        out.k_first = 1 + (in.lb_start > S.contact_dist ? sweep_skip(in.lb_start, S.contact_dist, step) : 0);
    }
    out.lb_end = r.contact ? -1.f : lb;
    return out;
}

// One work item: the contact test of body l at the intermediate sample k (1 <= k < N).  BOUND_ONLY: only whether the sample needs
// the real test at all - a lower bound of the distance to every moving piece that does not clear the contact distance (the wavefront
// gathers these and runs the GJK queries on them alone, in dense rounds of their own).
template <bool SELF, bool AUTO, bool BOUND_ONLY, class HB>
__device__ inline bool collide_item(const DScene &S, int l, const DPiece &LP, const HB &hull, const LaneInput &in, int k, bool live,
                                    const float *lds_pv, const float *lds_pd, uint32_t *keys, int kstride) {
    const int N = S.checks_per_step, upd = S.updates_per_step / N;
    const Xf T = lane_frame<SELF, BOUND_ONLY>(S, l, in, S.dt * (float)k / (float)N);      // bounds: hardware sine / cosine, covered by kBoundSlack
    constexpr float kBoundSlack = 1e-4f;
    if (!AUTO && S.use_ball) {
        const Xf XB = ball_pose(S, in.ball, in.ball_k0 + k * upd, nullptr);
        const bool near = XB.t.x * XB.t.x + XB.t.y * XB.t.y < S.ball_active_xy_radius * S.ball_active_xy_radius;
        if (BOUND_ONLY) return live && near && ball_lower_bound(S, LP, T, XB) <= S.contact_dist + kBoundSlack;
        return collide_link(S, l, LP, hull, T, nullptr, &XB, false, false, live && near).contact;
    }
    int idx[SMR_MAX_PLANETS] = {0, 0};
#pragma unroll
    for (int p = 0; p < SMR_MAX_PLANETS; ++p)
        if (p < S.n_planets) idx[p] = (in.planet_idx[p] + k * upd) % S.planet[p].n_orbit;
    if (BOUND_ONLY) {
        const float lb = AUTO ? planet_keys(S, keys, kstride, LP, T, idx, live, S.contact_dist) : planets_lower_bound(S, LP, T, idx, S.contact_dist);
        return live && lb <= S.contact_dist + kKeyStep + kBoundSlack;   // a candidate key (floor of the bound / kKeyStep) may exist
    }
    Sample r = collide_link(S, l, LP, hull, T, idx, nullptr, false, false, live, nullptr, AUTO);
    if (AUTO) planet_queries_auto(S, lds_pv, lds_pd, keys, kstride, LP, hull, T, idx, false, live, r);
    return r.contact;
}

// ---------------------------------------------------------------------------------------------
// Pair jobs (planet scenes).  In the lane-per-(env, body) loop above a wavefront runs as long as its busiest lane - a body next to
// the station works through ten pieces one after the other while 63 lanes wait - and every body of an env prunes with its OWN
// running minimum although only the minimum over the bodies is asked for (finish_kernel; ctlp.py:3263-3278).  Here a step's GJK runs
// are independent jobs (env, body, piece, sample) in a global list, one lane each:
//   keys kernel   (env, body) lanes: frame at t = dt, table bounds / shortcut, candidate keys of the 25 planet pieces -> keyw, the env's
//                 best (smallest) key by atomicMin -> env_best, the intermediate contact samples that survive the sweep bounds -> items
//   first kernel  one lane per env: the GJK run of the env's best pair.  Its distance d1 bounds the env's minimum from above, so only
//                 the pairs whose key bound does not exceed d1 can matter: they become jobs (2.9 per env on uniformly drawn states,
//                 8.7 GJK iterations per env in all instead of 19).  One lane per item: the bound test at its sample; the pieces
//                 that do not clear the contact distance become jobs.
//   jobs kernel   one lane per job, cutoff = the env's running minimum at that time; results by atomicMin (moving / static distance)
//                 and plain stores of 1 (contact flags) - both independent of the order the jobs run in, so the step stays
//                 deterministic and equal to the lane form up to rounding.
// The critical path of the step is one GJK run, not the longest lane's list of runs.
// ---------------------------------------------------------------------------------------------
// kJobSelf: self-collision pair `piece` of the env at the end of the step (the job word's piece field holds the pair index)
// kJobBallEnd / kJobBallItem (ball scenes): the ball at the end of the step / at contact sample k; the piece field carries mov | con << 1
enum { kJobPlanetEnd = 0, kJobItem = 1, kJobStaticEnd = 2, kJobSelf = 3, kJobBallEnd = 4, kJobBallItem = 5 };
constexpr int kSmallScan = 24;        // padded vertices of a "short" scanned side (three rounds of eight)
// job word: env 0..31 | body 32..35 | piece 36..41 | sample k 42..49 | kind 50..52.  k runs to checks_per_step (the end-of-step jobs carry
// k = N), which check_scene bounds by kMaxChecksPerStep - the lane form's item queue packs (lane << 5 | k) and holds N - 1 items per lane.
constexpr int kMaxChecksPerStep = 24;
constexpr int kJobBodyBits = 4, kJobPieceBits = 6, kJobKBits = 8, kJobKindBits = 3;
constexpr int kJobBodyShift = 32, kJobPieceShift = kJobBodyShift + kJobBodyBits, kJobKShift = kJobPieceShift + kJobPieceBits, kJobKindShift = kJobKShift + kJobKBits;
static_assert(kMaxChecksPerStep < (1 << kJobKBits) && kMaxChecksPerStep < 32, "sample index does not fit the job word / the item queue entry");
static_assert(SMR_MAX_BODIES <= (1 << kJobBodyBits) && SMR_MAX_SELF_PAIRS <= (1 << kJobPieceBits) && kJobKindShift + kJobKindBits <= 64, "job word fields");
__device__ __forceinline__ unsigned long long job_word(int env, int body, int piece, int k, int kind) {
    return (unsigned long long)(unsigned)env | ((unsigned long long)body << kJobBodyShift) | ((unsigned long long)piece << kJobPieceShift) |
           ((unsigned long long)k << kJobKShift) | ((unsigned long long)kind << kJobKindShift);
}
struct Job { int env, body, piece, k, kind; };
__device__ __forceinline__ Job job_unpack(unsigned long long w) {
    return {(int)(unsigned)(w & 0xFFFFFFFFull), (int)((w >> kJobBodyShift) & ((1ull << kJobBodyBits) - 1)), (int)((w >> kJobPieceShift) & ((1ull << kJobPieceBits) - 1)),
            (int)((w >> kJobKShift) & ((1ull << kJobKBits) - 1)), (int)((w >> kJobKindShift) & ((1ull << kJobKindBits) - 1))};
}
__device__ __forceinline__ unsigned dist_bits(float d) { return (unsigned)__float_as_int(fmaxf(d, 0.f)); }   // below 1 mm only "d < 0.001" is defined

// per-lane candidate pieces of the planets for body (LP under T) at orbit indices idx: emit(c) for every piece c (index into the planet
// descriptors) whose lower bound - the larger of sphere-sphere and capsule-capsule, as planet_keys - does not exceed cutoff.
// LP / T differ between the lanes; the piece loop is wave uniform (scalar piece loads).
template <class EMIT>
__device__ inline void planet_candidates(const DScene &S, const DPiece &LP, const Xf &T, const int idx[SMR_MAX_PLANETS], bool live, float cutoff,
                                         const EMIT &emit) {
    const F3 lc = apply(T, F3{LP.cx, LP.cy, LP.cz});
    const F3 l0 = apply(T, F3{LP.cap_a[0], LP.cap_a[1], LP.cap_a[2]}), l1 = apply(T, F3{LP.cap_b[0], LP.cap_b[1], LP.cap_b[2]});
    for (int p = 0; p < S.n_planets; ++p) {
        const Xf X = planet_pose(S, p, idx[p]);
        const F3 e = apply(X, F3{S.planet[p].bx, S.planet[p].by, S.planet[p].bz}) - lc;
        const float lbw = __builtin_amdgcn_sqrtf(dot(e, e)) * 0.99999f - 1e-6f - S.planet[p].bradius - LP.radius - LP.margin;
        const bool near = live && lbw <= cutoff;
        if (!__any(near)) continue;
        const int pb = S.planet[p].piece_begin, pc = S.planet[p].piece_count;
        for (int c = 0; c < pc; ++c) {
            const DPiece PP = load_piece(S.pieces, pb + c);
            const float margins = PP.margin + LP.margin;
            const F3 dc = apply(X, F3{PP.cx, PP.cy, PP.cz}) - lc;
            const float lb_s = __builtin_amdgcn_sqrtf(dot(dc, dc)) * 0.99999f - 1e-6f - PP.radius - LP.radius - margins;
            if (near && lb_s <= cutoff) {
                const F3 a0 = apply(X, F3{PP.cap_a[0], PP.cap_a[1], PP.cap_a[2]}), a1 = apply(X, F3{PP.cap_b[0], PP.cap_b[1], PP.cap_b[2]});
                const float lb_c = __builtin_amdgcn_sqrtf(seg_seg_dist2(a0, a1, l0, l1)) * 0.99999f - 1e-6f - PP.cap_r - LP.cap_r - margins;
                if (lb_c <= cutoff) emit(pb + c - S.planet_p0);
            }
        }
    }
}

// One lane's GJK run of a job: set up from the job word, advanced by one iteration per trip of the wavefront's loop (gjk_iterate: the
// same tests in the same order as planet_gjk, in the same frames - planet pieces in the LINK frame, the table in the world frame).
struct JobLane {
    GjkRun g;
    Xf XA, T;                    // planet piece seen from the body frame | the body's world frame (table runs)
    LocalHull hull;
    const float *av; int ac;     // the scanned side: padded vertex list in LDS (ac == 0: a world-fixed cylinder, av = centre xyz, radius, half length)
    float margins;
    bool is_static;
    int hull_body;               // whose hull `hull` is (deferred runs name it)
};

// Lower bound of the distance of self-collision pair i (wave uniform) at joint vector q from its bound table, bounding spheres and
// bounding capsules - no hull is touched.  Frames from the hardware sine / cosine: callers allow kBoundSlack.  One pass down the
// chain gives both frames, the partner link's is a prefix of the body's.
// enough: a bound above it is returned as soon as it is known (the table first: two bytes, and it settles nine in ten of the near pairs)
constexpr float kBoundSlack = 1e-4f;
__device__ inline float self_pair_lower_bound(const DScene &S, int i, const float q[SMR_NJ], float enough) {
    const int a = S.self_pair[i][0], partner = S.self_pair[i][1];
    float lb = partner >= 0 ? self_tab_bound(S, i, q[4], q[5], q[6]) + kBoundSlack : -kNoBound;      // (exact angles: no slack needed there)
    if (lb > enough) return lb;
    const DPiece LP = load_piece(S.pieces, S.link_piece[a]);
    const DPiece A = load_piece(S.pieces, partner >= 0 ? S.link_piece[partner] : -(partner + 1));
    const float margins = A.margin + LP.margin;
    const int link_a = a < SMR_NLINK ? a : SMR_NLINK - 1;
    Xf Ta = fk_base(S), Tp = Ta;
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) {
        if (j == partner) Tp = Ta;
        if (j < link_a) fk_advance_fast(S, j, q[j], Ta);
    }
    if (partner >= link_a) {                                 // (not the case for bodies mounted on the last link)
        Tp = fk_base(S);
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) if (j < partner) fk_advance_fast(S, j, q[j], Tp);
    }
    if (a >= SMR_NLINK) Ta = mounted_frame(S, a, Ta);
    const F3 cb = apply(Ta, F3{LP.cx, LP.cy, LP.cz});
    if (partner < 0) {
        const F3 d = F3{A.cx, A.cy, A.cz} - cb;
        return fmaxf(lb, __builtin_amdgcn_sqrtf(dot(d, d)) * 0.99999f - 1e-6f - A.radius - LP.radius - margins);
    }
    const F3 d = apply(Tp, F3{A.cx, A.cy, A.cz}) - cb;
    lb = fmaxf(lb, __builtin_amdgcn_sqrtf(dot(d, d)) * 0.99999f - 1e-6f - A.radius - LP.radius - margins);
    if (lb > enough) return lb;
    const F3 a0 = apply(Tp, F3{A.cap_a[0], A.cap_a[1], A.cap_a[2]}), a1 = apply(Tp, F3{A.cap_b[0], A.cap_b[1], A.cap_b[2]});
    const F3 b0 = apply(Ta, F3{LP.cap_a[0], LP.cap_a[1], LP.cap_a[2]}), b1 = apply(Ta, F3{LP.cap_b[0], LP.cap_b[1], LP.cap_b[2]});
    return fmaxf(lb, __builtin_amdgcn_sqrtf(seg_seg_dist2(a0, a1, b0, b1)) * 0.99999f - 1e-6f - A.cap_r - LP.cap_r - margins);
}

// the GJK run of self-collision pair `pair` at joint vector qs
__device__ inline void self_job_setup(const DScene &S, const BodyPool &BP, const float *cyl, const float qs[SMR_NJ], int pair, float cutoff,
                                      float stop_below, JobLane &J) {
        // pair (a, partner): the partner link's hull is climbed in ITS frame, the body's vertices are scanned under XA = partner^-1 o body
        // (both frames from one pass down the chain: the partner link's is a prefix of the body's); against the world-fixed cylinder
        // the body's hull is climbed under its world frame, as in a table run
        const int a = S.self_pair[pair][0], partner = S.self_pair[pair][1];
        const int link_a = a < SMR_NLINK ? a : SMR_NLINK - 1;
        Xf Ta = fk_base(S), Tp = Ta;
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) {
            if (j == partner) Tp = Ta;
            if (j < link_a) fk_advance(S, j, qs[j], Ta);
        }
        if (partner >= link_a) Tp = body_frame(S, qs, partner);
        if (a >= SMR_NLINK) Ta = mounted_frame(S, a, Ta);
        const DPiece LA = BP.piece(a);
        if (partner >= 0) {
            const DPiece LP = BP.piece(partner);
            J.hull = BP.hull(partner); J.hull_body = partner;
            J.XA = inv_compose(Tp, Ta);
            J.T = Tp;
            J.av = (const float *)BP.pool + BP.pool[kBodyPoolHead * a];
            J.ac = (LA.count + 7) & ~7;                       // the pool pads every vertex list to a multiple of 8 (body_pool_tables)
            J.margins = LA.margin + LP.margin;
            J.is_static = false;
            J.g.start(apply(J.XA, F3{LA.cx, LA.cy, LA.cz}) - F3{LP.cx, LP.cy, LP.cz}, cutoff + J.margins, stop_below + J.margins);
        } else {
            const float *c = cyl + 8 * S.self_cyl_slot[pair];
            J.hull = BP.hull(a); J.hull_body = a;
            J.T = Ta;
#pragma unroll
            for (int i = 0; i < 9; ++i) J.XA.r[i] = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
            J.XA.t = F3{0.f, 0.f, 0.f};
            J.av = c; J.ac = 0;
            J.margins = LA.margin + c[5];
            J.is_static = true;
            J.g.start(F3{c[0], c[1], c[2]} - apply(Ta, F3{LA.cx, LA.cy, LA.cz}), cutoff + J.margins, stop_below + J.margins);
        }
}

// cyl: the LDS copy of the world-fixed cylinders' parameters (8 floats per slot: centre xyz, radius, half length; stage_job_tables)
// frames: the bodies' frames at t = dt the keys kernel left behind ([body][env][12]: rotation rows, translation) - an end-of-step
// job reads its body's frame instead of the 28 state words, seven sines / cosines and the chain (measured: the setups took 44 % of
// the jobs kernel); the jobs of intermediate samples and of the self-collision pairs compute theirs
template <bool SELF>
__device__ inline void job_setup(const DScene &S, const BodyPool &BP, const float *pv, const float *pd, const float *cyl, const LaneInput &in,
                                 const float *frames, const Job &jb, float cutoff, float stop_below, JobLane &J) {
    const int N = S.checks_per_step, upd = S.updates_per_step / N;
    const float t = jb.k >= N ? S.dt : S.dt * (float)jb.k / (float)N;
    if (SELF && jb.kind == kJobSelf) {
        float qs[SMR_NJ];
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) qs[j] = interp_pos(in.q[j * in.stride], in.v[j * in.stride], in.a[j * in.stride], in.a1[j * in.stride], t, S.dt);
        self_job_setup(S, BP, cyl, qs, jb.piece, cutoff, stop_below, J);
        return;
    }
    if (jb.kind == kJobBallEnd || jb.kind == kJobBallItem) {
        // the ball (ONE piece; its padded vertices lead the planets' slot of the tables) seen from the body frame at sample k
        J.T = lane_frame_any<false>(S, jb.body, in, t);
        const DPiece LPb = BP.piece(jb.body);
        const DPiece BPc = load_piece_lane(S.pieces, S.ball_piece);
        J.hull = BP.hull(jb.body); J.hull_body = jb.body;
        J.av = pv; J.ac = S.ball_jobs_vn;
        J.margins = BPc.margin + LPb.margin;
        J.is_static = false;
        J.XA = inv_compose(J.T, ball_pose(S, in.ball, in.ball_k0 + jb.k * upd, nullptr));
        J.g.start(apply(J.XA, F3{BPc.cx, BPc.cy, BPc.cz}) - F3{LPb.cx, LPb.cy, LPb.cz}, cutoff + J.margins, stop_below + J.margins);
        return;
    }
    if (jb.kind == kJobItem) J.T = lane_frame_any<false>(S, jb.body, in, t);
    else {
        const SMR_GLOBAL V4 *f = (const SMR_GLOBAL V4 *)frames + 3 * ((size_t)jb.body * S.B + jb.env);
        const V4 f0 = f[0], f1 = f[1], f2 = f[2];
        J.T.r[0] = f0[0]; J.T.r[1] = f0[1]; J.T.r[2] = f0[2]; J.T.r[3] = f0[3]; J.T.r[4] = f1[0]; J.T.r[5] = f1[1]; J.T.r[6] = f1[2]; J.T.r[7] = f1[3];
        J.T.r[8] = f2[0]; J.T.t = F3{f2[1], f2[2], f2[3]};
    }
    const DPiece LP = BP.piece(jb.body);
    J.hull = BP.hull(jb.body); J.hull_body = jb.body;
    const float *d = pd + kPlanetDescDwords * jb.piece;
    J.av = pv + 4 * __float_as_int(d[0]);
    J.ac = __float_as_int(d[1]);
    J.margins = d[2] + LP.margin;
    J.is_static = jb.kind == kJobStaticEnd;
    const F3 lcl = {LP.cx, LP.cy, LP.cz};
    if (J.is_static) {
#pragma unroll
        for (int i = 0; i < 9; ++i) J.XA.r[i] = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
        J.XA.t = F3{0.f, 0.f, 0.f};
        J.g.start(F3{d[4], d[5], d[6]} - apply(J.T, lcl), cutoff + J.margins, stop_below + J.margins);
    } else {
        const int p = __float_as_int(d[3]);
        const int i0 = (in.planet_idx[0] + jb.k * upd) % S.planet[0].n_orbit;
        const int i1 = S.n_planets > 1 ? (in.planet_idx[1] + jb.k * upd) % S.planet[1].n_orbit : 0;
        J.XA = inv_compose(J.T, quat_pose((const SMR_GLOBAL float *)(p ? S.planet[1].orbit_pose : S.planet[0].orbit_pose) + 8 * (size_t)(p ? i1 : i0)));
        J.g.start(apply(J.XA, F3{d[4], d[5], d[6]}) - lcl, cutoff + J.margins, stop_below + J.margins);
    }
}

// one iteration; true when the run is settled, dist = distance with the margins subtracted
template <bool SELF>
__device__ __forceinline__ bool job_trip(JobLane &J, float &dist) {
    const auto sb = [&](F3 d, int &hint, bool first) {
        const F3 p = J.hull.support(J.is_static ? rot_t(J.T, d) : d, hint, first);
        return J.is_static ? apply(J.T, p) : p;
    };
    const auto sa = [&](F3 d, int &, bool) {
        if (SELF && J.ac == 0) {                           // btCylinderShapeZ::localGetSupportingVertexWithoutMargin, world frame
            const float sxy = sqrtf(d.x * d.x + d.y * d.y), r = J.av[3];
            F3 p = sxy > 0.f ? F3{J.av[0] + r * d.x / sxy, J.av[1] + r * d.y / sxy, J.av[2]} : F3{J.av[0] + r, J.av[1], J.av[2]};
            p.z += d.z < 0.f ? -J.av[4] : J.av[4];
            return p;
        }
        const F3 dl = rot_t(J.XA, d);
        float best = -3.0e38f;
        int bi = 0;
        for (int i = 0; i < J.ac; i += 8) {                // padded to multiples of 8: eight vertex reads in flight per round trip
            float sq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const V4 q = load_v4(J.av + 4 * (i + u));
                sq[u] = fmaf(q[2], dl.z, fmaf(q[1], dl.y, q[0] * dl.x));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool gt = sq[u] > best;
                best = gt ? sq[u] : best;
                bi = gt ? i + u : bi;
            }
        }
        const V4 pw = load_v4(J.av + 4 * bi);
        return apply(J.XA, F3{pw[0], pw[1], pw[2]});
    };
    float d;
    if (!gjk_iterate(J.g, sa, sb, d)) return false;
    dist = d - J.margins;
    return true;
}

// ---- deferred runs (round 4): the long runs leave the lanes ---------------------------------------
// A run that has not settled after `defer_after` iterations in its lane is written out as a record and finished by jobs_coop_kernel,
// where a GROUP OF 16 LANES (one DPP row) works one run: every lane tests a sixteenth of each side's vertices (no hill climb: 252 vertices
// are sixteen per lane, read at consecutive addresses), two row reductions give the support points, and the sub-simplex solve runs on
// values the 16 lanes hold alike - the loops have fixed trip counts and every lane has work, so sixteen wavefronts per CU hide each
// other's latencies, which the lane form's 15 000-cycle trips of 64 unrelated runs never could.  The lanes that drew short runs no
// longer wait for the one in twenty that needs 6 to 18 iterations: the lane form's span is bounded by `defer_after` trips per hand-out.  Which form computes an iteration is a function of the run alone (its iteration count), not of what
// its neighbours in the wavefront do: the step stays deterministic.
// Record: kDeferWords dwords, 16-byte aligned:
//   0..11 frame (scanned side seen from the hull's frame; world-frame runs: the hull body's world frame)   12..14 v   15 cutoff^2   16 stop^2
//   17 margins   18 the env's minimum when the job was drawn   19 meta (run_meta)   20..21 job word   22..33 simplex   34 n | it << 8 | stalls << 16
constexpr int kDeferWords = 36;
__device__ __forceinline__ unsigned run_meta(int av_off, int ac, int hull_body, bool is_static, bool mov, bool con) {
    return (unsigned)av_off | ((unsigned)ac << 16) | ((unsigned)hull_body << 25) | ((unsigned)is_static << 29) | ((unsigned)mov << 30) | ((unsigned)con << 31);
}
#ifndef SMR_HOST_EMUL
__device__ inline void defer_write(SMR_GLOBAL V4 *rec, const JobLane &J, const float *lds0, unsigned long long word, bool mov, bool con, float seen) {
    const Xf &X = J.is_static ? J.T : J.XA;
    const GjkRun &g = J.g;
    rec[0] = V4{X.r[0], X.r[1], X.r[2], X.r[3]}; rec[1] = V4{X.r[4], X.r[5], X.r[6], X.r[7]}; rec[2] = V4{X.r[8], X.t.x, X.t.y, X.t.z};
    rec[3] = V4{g.v.x, g.v.y, g.v.z, g.cutoff2};
    rec[4] = V4{g.stop2, J.margins, seen, __int_as_float((int)run_meta((int)(J.av - lds0), J.ac, J.hull_body, J.is_static, mov, con))};
    rec[5] = V4{__int_as_float((int)(unsigned)(word & 0xFFFFFFFFull)), __int_as_float((int)(unsigned)(word >> 32)), g.w0.x, g.w0.y};
    rec[6] = V4{g.w0.z, g.w1.x, g.w1.y, g.w1.z};
    rec[7] = V4{g.w2.x, g.w2.y, g.w2.z, g.w3.x};
    rec[8] = V4{g.w3.y, g.w3.z, __int_as_float(g.n | (g.it << 8) | (g.stalls << 16)), 0.f};
}

// arg max over the 16 lanes of a DPP row (a quarter wavefront): every lane brings its best (value, index) and gets the index of the largest
// value back, ties to the lowest index.  Four rotations each for the maximum and for the smallest index that attains it: v_max_f32 /
// v_min_i32 with a DPP operand, no LDS crossbar, ~8 dependent VALU instructions.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false)); }
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int row_argmax(float best, int bi) {
    float m = best;
    m = fmaxf(m, dpp_f<0x128>(m)); m = fmaxf(m, dpp_f<0x124>(m)); m = fmaxf(m, dpp_f<0x122>(m)); m = fmaxf(m, dpp_f<0x121>(m));     // row_ror:8, 4, 2, 1
    int k = best == m ? bi : 0x7FFFFFFF;
    k = min(k, dpp_i<0x128>(k)); k = min(k, dpp_i<0x124>(k)); k = min(k, dpp_i<0x122>(k)); k = min(k, dpp_i<0x121>(k));
    return k;
}

// One iteration of a deferred run, worked by a GROUP of 16 lanes (gjk_iterate's tests in gjk_iterate's order on group-uniform values):
// every lane tests a sixteenth of each side's vertices, two row reductions give the support points, the sub-simplex solve runs on
// values all 16 lanes hold alike.  Returns true when the run has settled; dist as gjk_iterate reports it (margins not subtracted).
template <bool SELF>
__device__ __forceinline__ bool coop_iterate(GjkRun &g, const Xf &X, bool is_static, const float *av, int ac, const float *hv, int hcount, float &dist) {
    const int l16 = (int)__lane_id() & 15;
    const F3 nv = neg(g.v);
    F3 pa;
    if (SELF && ac == 0) {                               // btCylinderShapeZ::localGetSupportingVertexWithoutMargin, world frame (job_trip)
        const float sxy = sqrtf(nv.x * nv.x + nv.y * nv.y), r = av[3];
        pa = sxy > 0.f ? F3{av[0] + r * nv.x / sxy, av[1] + r * nv.y / sxy, av[2]} : F3{av[0] + r, av[1], av[2]};
        pa.z += nv.z < 0.f ? -av[4] : av[4];
    } else {
        const F3 dl = is_static ? nv : rot_t(X, nv);
        float best = -3.0e38f;
        int bi = 0;
        for (int i = l16; i < ac; i += 16) {
            const V4 q = load_v4(av + 4 * i);
            const float sq = fmaf(q[2], dl.z, fmaf(q[1], dl.y, q[0] * dl.x));
            if (sq > best) { best = sq; bi = i; }
        }
        const int ia = row_argmax(best, bi);
        const V4 pw = load_v4(av + 4 * ia);
        const F3 p = {pw[0], pw[1], pw[2]};
        pa = is_static ? p : apply(X, p);
    }
    F3 pb;
    {
        const F3 dl = is_static ? rot_t(X, g.v) : g.v;
        float best = -3.0e38f;
        int bi = 0;
        for (int i = l16; i < hcount; i += 16) {
            const V4 q = load_v4(hv + 4 * i);
            const float sq = fmaf(q[2], dl.z, fmaf(q[1], dl.y, q[0] * dl.x));
            if (sq > best) { best = sq; bi = i; }
        }
        const int ib = row_argmax(best, bi);
        const V4 pw = load_v4(hv + 4 * ib);
        const F3 p = {pw[0], pw[1], pw[2]};
        pb = is_static ? apply(X, p) : p;
    }
    const F3 w = pa - pb;
    const float vw = dot(g.v, w);
    if (vw > 0.f && vw * vw > g.cutoff2 * g.vv) { dist = vw / sqrtf(g.vv); return true; }
    bool settled = g.n > 0 && (g.vv - vw) <= 1e-6f * g.vv;
    if (!settled) {
        const F3 e0 = w - g.w0, e1 = w - g.w1, e2 = w - g.w2;
        const float tiny = 1e-12f * fmaxf(dot(w, w), 1e-6f);
        settled = (g.n > 0 && dot(e0, e0) <= tiny) || (g.n > 1 && dot(e1, e1) <= tiny) || (g.n > 2 && dot(e2, e2) <= tiny);
    }
    if (!settled) {
        const int slot = g.n;
        g.w0 = sel3(slot == 0, w, g.w0); g.w1 = sel3(slot == 1, w, g.w1); g.w2 = sel3(slot == 2, w, g.w2); g.w3 = sel3(slot >= 3, w, g.w3);
        ++g.n;
        F3 vn;
        if (simplex_solve(g.w0, g.w1, g.w2, g.w3, g.n, vn)) { dist = 0.f; return true; }
        const float vvn = dot(vn, vn);
        if (!(vvn == vvn)) settled = true;
        else if (g.it > 0 && vvn >= g.vv && (vvn > g.vv * (1.0f + 4e-7f) || ++g.stalls > 3)) settled = true;
        else {
            g.v = vn;
            g.vv = vvn;
            if (g.vv <= g.stop2) settled = true;
            else if (g.n == 4) { dist = 0.f; return true; }
        }
    }
    if (settled || ++g.it >= 32) { dist = sqrtf(g.vv); return true; }
    return false;
}
#endif

// A15: planet part of the observation (obs.py:262-292)
__device__ inline void write_planet_observation(const DScene &S, const int idx[SMR_MAX_PLANETS], float *row) {
    int o = 0;
    for (int p = 0; p < S.obs_num_planets; ++p) {
        const int i = idx[p];
        if (S.obs_planet_size == 1) {
            row[o++] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (float)i / (float)S.planet[p].n_orbit));
        } else if (S.obs_planet_size == 2) {
            const SMR_GLOBAL float *loc = (const SMR_GLOBAL float *)S.planet[p].orbit_local + 2 * (size_t)i;
            float rx = 1.05f * S.planet[p].radius_x, ry = 1.05f * S.planet[p].radius_y;
            row[o++] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (loc[0] + rx) / (2.f * rx)));
            row[o++] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (loc[1] + ry) / (2.f * ry)));
        } else {
            const SMR_GLOBAL float *pose = (const SMR_GLOBAL float *)S.planet[p].orbit_pose + 8 * (size_t)i;
            for (int c = 0; c < 3; ++c)
                row[o++] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (pose[c] - S.planet_obs_min[c]) / (S.planet_obs_max[c] - S.planet_obs_min[c])));
        }
    }
}

// A15: ball part of the observation (get_moving_object_observation, ctlp.py:2364-2377)
__device__ inline void write_ball_observation(const DScene &S, const BallState &b, int k, float *row) {
    float vel[3];
    Xf X = ball_pose(S, b, k, vel);
    float pos[3] = {X.t.x, X.t.y, X.t.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        row[c] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (pos[c] - S.ball_obs_pos_min[c]) / (S.ball_obs_pos_max[c] - S.ball_obs_pos_min[c])));
        row[3 + c] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (vel[c] - S.ball_obs_vel_min[c]) / (S.ball_obs_vel_max[c] - S.ball_obs_vel_min[c])));
    }
}

// A15: observation row (obs.py:313-351), clipped to [-1, 1]
__device__ inline void write_observation(const DScene &S, const float q[SMR_NJ], const float v[SMR_NJ],
                                         const float a[SMR_NJ], const int idx[SMR_MAX_PLANETS], float *row) {
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) row[j] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (q[j] - S.pos_lo[j]) / (S.pos_hi[j] - S.pos_lo[j])));
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) row[SMR_NJ + j] = fminf(1.f, fmaxf(-1.f, v[j] / S.vel_max[j]));
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) row[2 * SMR_NJ + j] = fminf(1.f, fmaxf(-1.f, a[j] / S.acc_max[j]));
    write_planet_observation(S, idx, row + 3 * SMR_NJ + S.obs_target_size);
}

// A15: target-point part of the observation (obs.py:313-335, ctlp.py:2240-2285): position in the target box, position
// relative to the target link point ee
__device__ inline void write_target_observation(const DScene &S, const float tp[3], F3 ee, float *row) {
    int o = 3 * SMR_NJ;
    if (S.obs_target_pos)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            row[o++] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (tp[c] - S.ball_target_box[c][0]) / (S.ball_target_box[c][1] - S.ball_target_box[c][0])));
    if (S.obs_target_rel) {
        const float rel[3] = {tp[0] - ee.x, tp[1] - ee.y, tp[2] - ee.z};
#pragma unroll
        for (int c = 0; c < 3; ++c)
            row[o++] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (rel[c] - S.target_rel_min[c]) / (S.target_rel_max[c] - S.target_rel_min[c])));
    }
}

// target link point (scene.py:185-208) of joint vector q
__device__ inline F3 ee_point(const DScene &S, const float q[SMR_NJ]) {
    return apply(body_frame(S, q, S.ee_body), F3{S.ee_offset[0], S.ee_offset[1], S.ee_offset[2]});
}

}  // namespace smr
