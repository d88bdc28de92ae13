// smr_hip.hip - kernels and C ABI of libsmr_hip.so (include/safemotions_hip.h).  gfx950 only.
//
// step() = three launches on the handle's stream (+ one for the envs that finished):
//   plan    (B x 7 lanes)  A2/A4  safe acceleration range of the current state, action -> end acceleration
//   collide (B x 2 x 8 lanes)  A3/A6/A7/A8/A11  (end-of-step sample | contact sweep, link l): FK + GJK distances of the reward and
//                          the contact test at the end of the step; the N-1 intermediate contact samples of a link by
//                          conservative advancement in one lane
//   finish  (B lanes)      A1/A12/A13/A15/A16  latch, state update, reward, termination, observation,
//                          wave-aggregated append of finished envs to the reset list
//   reset   (B lanes, only list entries work)  A17/A9  start state + planet phase by rejection sampling
// The B x 2 x 8 grid of `collide` puts every link on its own lane: 8-16x the wavefronts of a one-lane-per-env launch.  The GJK loops are divergent and wait on scalar loads; only many resident
// wavefronts per SIMD keep the 1024 SIMDs of the chip busy at B = 65 536.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifdef SMR_PROFILE_ROWS
// profiling build (scripts/profile_rows.py, profile_blocks.py): cycle / count slots the device code adds to - [i]: sum, [64 + i]:
// number of additions; slots 0..31: wavefront cycles per grid row of the collide kernel, 32..63: phases and counts (the first
// active lane of a wavefront adds) - and a record per wavefront of the collide kernel (g_wave_rec)
__device__ unsigned long long g_row_cycles[128];
__device__ unsigned long long g_job_hist[8 * 8];      // scripts/study/job_hist.py: [slack between cutoff and key bound][iterations 1, 2, 3-4, 5-8, 9+ | long scans | jobs]; scripts/study/brake_hist.py: [2 row] passes, lanes; 50.. walk counts
__device__ unsigned g_wave_rec[8 * 65536];
#define SMR_DEV_CLOCK() __builtin_readcyclecounter()
#define SMR_DEV_COUNT(i, n) do { if ((int)__lane_id() == __ffsll((long long)__ballot(true)) - 1) { atomicAdd(&g_row_cycles[32 + (i)], (unsigned long long)(n)); atomicAdd(&g_row_cycles[96 + (i)], 1ull); } } while (0)
#define SMR_DEV_MAX(i, n) do { if ((int)__lane_id() == __ffsll((long long)__ballot(true)) - 1) { atomicMax(&g_row_cycles[32 + (i)], (unsigned long long)(n)); atomicAdd(&g_row_cycles[96 + (i)], 1ull); } } while (0)
// field f (0..7) of this wavefront's record
#define SMR_WAVE_REC(f, n) do { const unsigned w_ = ((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)); \
                                if (w_ < 65536u && (int)__lane_id() == __ffsll((long long)__ballot(true)) - 1) g_wave_rec[8 * w_ + (f)] += (unsigned)(n); } while (0)
#endif
#include "smr_device.h"
#include "smr_tables.h"
#include "smr_mlp.h"

namespace smr {

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// one lane per (env, joint); the lanes of a wavefront share the root finding of its open position-limit brackets
// use_ranges: the ranges range_kernel / the staged start states left behind are taken where they are valid (DState::range_valid)
__global__ __launch_bounds__(256) void plan_kernel(const DScene *__restrict__ Sp, DState st, const float *__restrict__ actions, int use_ranges) {
    const DScene &S = *Sp;
    const int B = S.B;
    const int e0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e0 < B;                          // the wavefront searches its open brackets together: no lane leaves early
    const int e = live ? e0 : B - 1;
    const int j = blockIdx.y;
    float lo = 0.f, hi = 0.f;
    const bool have = use_ranges && st.range_valid[e] != 0;
    if (have) { lo = st.range_lo[j * B + e]; hi = st.range_hi[j * B + e]; }
    if (__any(live && !have)) {
        float lo2, hi2;
        int code;
        safe_acc_range_joint<64>(joint_lim(S, j), st.q[j * B + e], st.v[j * B + e], st.a[j * B + e], lo2, hi2, code, live && !have);
        if (!have) { lo = lo2; hi = hi2; }
    }
    if (!live) return;
    if (j == 0 && e == 0) {                            // this step's list counters (a memset each on the stream was a 5 us kernel of its own: 2 % of a ball-scene step)
        if (st.done_count) *st.done_count = 0;
        if (st.spawn_count) *st.spawn_count = 0;
        if (st.tp_list_count) *st.tp_list_count = 0;
    }
    if (j == 0 && S.check_braking) { st.braked[e] = 0; st.bt_min[e] = 0x7F800000u; }   // set by brake_plan_kernel / brake_check_kernel
    if (j == 0 && S.jobs_on) {                         // the accumulators and lists of this step's jobs (self_bounds / keys / enumerate / jobs kernels)
        if (S.planets_jobs) {
            st.env_best[e] = 0xFFFFFFFFu;
            st.env_min[e] = dist_bits(S.moving_max + 0.002f);
            st.env_dstatic[e] = dist_bits(S.static_max);
        }
        if (S.ball_jobs) st.env_min[e] = dist_bits(S.moving_max + 0.002f);
        st.env_dself[e] = dist_bits(S.static_max);
        st.overflow_env[e] = 0;
        if (e == 0) { *st.item_count = 0; *st.job_count = 0; *st.job_count_big = 0; *st.job_next = 0; *st.defer_count = 0; }
    }
    float u = actions[(size_t)e * SMR_NJ + j];
    st.a1[j * B + e] = lo + 0.5f * (u + 1.0f) * (hi - lo);      // actions.py:496-498 (klimits.denormalize)
}

// The next step's safe acceleration ranges, from the state the planned step ends in (finish_kernel's arithmetic: base.py:1179-1181): the grid
// of the plan kernel, on a stream of its own beside the distance kernels.  An env that ends its episode in this step gets the range of
// its staged start state instead (finish_kernel / commit_staged), or none (range_valid = 0).
// (one wavefront: joint j of the 64 envs from e0 on; its lanes share the root finding)
__device__ __forceinline__ void range_wave(const DScene &S, const DState &st, int j, int e0) {
    const int B = S.B;
    const bool live = e0 < B;
    const int e = live ? e0 : B - 1;
    const float q0 = st.q[j * B + e], v0 = st.v[j * B + e], a0 = st.a[j * B + e], a1 = st.a1[j * B + e];
    float lo, hi;
    int code;
    safe_acc_range_joint<64>(joint_lim(S, j), interp_pos(q0, v0, a0, a1, S.dt, S.dt), interp_vel(v0, a0, a1, S.dt, S.dt), a1, lo, hi, code, live);
    if (!live) return;
    st.range_lo[j * B + e] = lo; st.range_hi[j * B + e] = hi;
    if (j == 0) st.range_valid[e] = 1;
}
__global__ __launch_bounds__(256) void range_kernel(const DScene *__restrict__ Sp, DState st) {
    range_wave(*Sp, st, (int)blockIdx.y, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}
// ... as the last `range_blocks` blocks of a jobs kernel launch (that kernel is a chain of latencies on a tenth of the issue slots: the
// ranges ride along, without the two stream dependencies a kernel of their own costs): wavefront w = joint w / waves_per_joint
// ... and of the staged start states reset_kernel<true> just wrote (same list, same tags; on its stream, behind it - inside that kernel the
// call cost its 256-register instantiations a quarter to two thirds of their speed in spills): 8 lanes per entry, lane g < 7 = joint g
// direct: the unstaged reset path - the live state's range (reset_kernel<false> wrote q, v, a), so that every start state gets its first range the same way
__global__ __launch_bounds__(256) void staged_range_kernel(const DScene *__restrict__ Sp, DState st, const int *__restrict__ list, const int *__restrict__ count_ptr,
                                                           int count_fixed, const unsigned *__restrict__ tags, int tag_offset, int direct) {
    const DScene &S = *Sp;
    const int count = count_ptr ? *count_ptr : count_fixed;
    if ((int)((blockIdx.x * blockDim.x) >> 3) >= count) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 3, g = t & 7, B = S.B;
    const bool want = i < count && g < SMR_NJ;
    const int ii = i < count ? i : count - 1;
    const int e = list ? list[ii] : ii;
    const unsigned tag = direct ? 0u : (tags ? tags[ii] : st.reset_count[e] + (unsigned)tag_offset);
    const int j = g < SMR_NJ ? g : 0;
    const size_t k = direct ? (size_t)j * B + e : ((size_t)(tag & 1u) * SMR_NJ + j) * B + e;
    const float *sq = direct ? st.q : st.next_q, *sv = direct ? st.v : st.next_v, *sa = direct ? st.a : st.next_a;
    float lo, hi;
    int code;
    safe_acc_range_joint<8>(joint_lim(S, j), sq[k], sv[k], sa[k], lo, hi, code, want);      // (the env's own 8 lanes: the result does not depend on its neighbours in the list)
    if (want) { (direct ? st.range_lo : st.next_lo)[k] = lo; (direct ? st.range_hi : st.next_hi)[k] = hi; }
    if (direct && i < count && g == SMR_NJ) st.range_valid[e] = 1;
}
__device__ __forceinline__ void range_block(const DScene &S, const DState &st, int rb) {
    const int waves_per_joint = (S.B + 63) >> 6;
    const int w = rb * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    const int j = __builtin_amdgcn_readfirstlane(w / waves_per_joint);
    if (j >= SMR_NJ) return;
    range_wave(S, st, j, (w - j * waves_per_joint) * 64 + (int)__lane_id());
}

#ifdef SMR_PROFILE_ROWS
// profiling build (scripts/profile_rows.py): shader-clock cycles of the collide kernel's wavefronts, summed per grid row
struct RowClock {
    unsigned long long t0;
    int row;
    __device__ RowClock(int r) : t0(__builtin_readcyclecounter()), row(r) {}
    __device__ ~RowClock() {
        if (__lane_id() == 0 && row < 64) { atomicAdd(&g_row_cycles[row], __builtin_readcyclecounter() - t0); atomicAdd(&g_row_cycles[64 + row], 1ull); }
    }
};
extern "C" int smr_debug_job_hist(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_job_hist), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_job_hist), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
extern "C" int smr_debug_row_cycles(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_row_cycles), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[128] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_row_cycles), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// ... and the schedule of the collide kernel's blocks: real-time clock (100 MHz, the same on every XCD) at the start and the end
__device__ unsigned long long g_block_times[2 * 16384];
struct BlockClock {
    unsigned long long t0;
    __device__ BlockClock() : t0(wall_clock64()) {}
    __device__ ~BlockClock() {
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x == 0 && id < 16384u) { g_block_times[2 * id] = t0; g_block_times[2 * id + 1] = wall_clock64(); }
    }
};
extern "C" int smr_debug_block_times(unsigned long long *out, unsigned *wave_rec, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_times), sizeof(unsigned long long) * 2 * 16384) != hipSuccess) return -1;
    if (wave_rec && hipMemcpyFromSymbol(wave_rec, HIP_SYMBOL(g_wave_rec), sizeof(unsigned) * 8 * 65536) != hipSuccess) return -1;
    if (reset) { void *p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wave_rec)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned) * 8 * 65536) != hipSuccess) return -1; }
    return 0;
}
#ifdef SMR_PROFILE_BRAKE      // the same records for the brake_check kernel instead
#define SMR_ROW_CLOCK(r)
#define SMR_BRAKE_ROW_CLOCK(r) RowClock row_clock_(r); BlockClock block_clock_
#else
#define SMR_ROW_CLOCK(r) RowClock row_clock_(r); BlockClock block_clock_
#define SMR_BRAKE_ROW_CLOCK(r)
#endif
// ... and of the phases of the reset-side kernels: slots 32 + phase (first active lane of the wavefront)
struct PhaseClock {
    unsigned long long t;
    __device__ PhaseClock() : t(__builtin_readcyclecounter()) {}
    __device__ void mark(int i) {
        const unsigned long long now = __builtin_readcyclecounter();
        if ((int)__lane_id() == __ffsll((long long)__ballot(true)) - 1) { atomicAdd(&g_row_cycles[32 + i], now - t); atomicAdd(&g_row_cycles[96 + i], 1ull); }
        t = now;
    }
};
#define SMR_PHASE_INIT PhaseClock phase_clock_
#define SMR_PHASE(i) phase_clock_.mark(i)
#else
#define SMR_ROW_CLOCK(r)
#define SMR_BRAKE_ROW_CLOCK(r)
#define SMR_PHASE_INIT
#define SMR_PHASE(i)
#endif

// one lane per (env, body l), grid = (B / block, self pairs + n_bodies): the end-of-step sample k = N (distances of the reward + its
// contact test), then the intermediate contact samples the body still needs (smr_device.h, "A11") as work items that the lanes of
// the wavefront share out among themselves.
// The reset-side kernels (reset, spawn, target) run beside the step kernels.  Left to itself the compiler gives their wavefronts the
// whole register file of a SIMD (256 VGPRs + 256 AGPRs as spill space): one such wavefront then keeps every collide wavefront off
// its SIMD for as long as its rejection loops run.  SMR_SIDE_WAVES wavefronts per SIMD = 512 / SMR_SIDE_WAVES registers each.
#ifndef SMR_SIDE_WAVES
#define SMR_SIDE_WAVES 2
#endif
#define SMR_SIDE_KERNEL __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMR_SIDE_WAVES, SMR_SIDE_WAVES)))
// ... except the staged ball launches on the robot with the ball machine (spawn_kernel_extra): at 256 registers they spill 420-550, with the
// whole register file (256 + 256 as spill space; any value but 2 gives that at 60 KB of LDS per block) 51-77, and since the self pairs left
// the collide grid for the jobs kernel the shorter launches are worth more than the SIMDs they hold: ball + ball machine 64.2 -> 73.3 M
// env-steps/s, ball scene on the plain robot unchanged (measured; the same for reset_kernel: 68 M there, and -3 .. -7 % in the planet scenes)
#ifndef SMR_SPAWN_WAVES
#define SMR_SPAWN_WAVES 1
#endif
// (the staged target points likewise: 200 spilled registers at 256, none with the whole file; reaching task 103.8 -> 105.3 M, with the risk
// networks 54.8 -> 55.9 M, with the braking layer unchanged - measured)
#ifndef SMR_TARGET_WAVES
#define SMR_TARGET_WAVES 1
#endif
#define SMR_TARGET_KERNEL __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMR_TARGET_WAVES, SMR_TARGET_WAVES)))
#define SMR_SPAWN_KERNEL __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMR_SPAWN_WAVES, SMR_SPAWN_WAVES)))
#ifndef SMR_COLLIDE_WAVES
#define SMR_COLLIDE_WAVES 4      // 128 VGPRs (a 56-byte spill) for a fourth wavefront per SIMD: 0.96 -> 0.89 ms at 65536 envs
#endif
template <int NM, bool BLOCK = false>
__device__ __forceinline__ void push_jobs(const DState &st, int env, const int (&bodies)[NM], const uint32_t (&masks)[NM], int k, int kind);      // (pair-job path, below)
__device__ __forceinline__ void self_bounds_row(const DScene &S, const DState &st, int pair, int e0);
constexpr int kMaxItemsPerLane = kMaxChecksPerStep - 1;   // N - 1 intermediate samples (check_scene: N <= kMaxChecksPerStep)
// SELF: the scene has self-collision pairs (ball machine); the plain robot runs the instantiation without that code
// AUTO: lane-autonomous planet queries (planet_queries_auto); four wavefronts then share the LDS copies
#ifndef SMR_COLLIDE_BLOCK
#define SMR_COLLIDE_BLOCK 128    // ball scenes (no lane-autonomous planet queries): two wavefronts share the LDS copies (64 -> 128 lanes: ball 139.1 -> 141.7 M, 256: 131.2)
#endif
template <bool SELF, bool AUTO>
__global__ __launch_bounds__(AUTO ? 256 : SMR_COLLIDE_BLOCK) __attribute__((amdgpu_waves_per_eu(SMR_COLLIDE_WAVES, SMR_COLLIDE_WAVES))) void collide_kernel(const DScene *__restrict__ Sp, DState st, int pair_rows) {
    const DScene &S = *Sp;
    constexpr int kBlock = AUTO ? 256 : SMR_COLLIDE_BLOCK;
    SMR_ROW_CLOCK((int)blockIdx.y);
    // rows behind the bodies: the bound tests of the self-collision pairs (self_bounds_row: light rows that fill the grid's tail instead of a
    // launch of their own in front of it), then - a ball scene without a jobs launch - the next step's safe acceleration ranges, one row per
    // joint (range_wave)
    if ((int)blockIdx.y >= S.n_bodies) {
        const int r = (int)blockIdx.y - S.n_bodies, e0 = (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (r < pair_rows) { if (SELF) self_bounds_row(S, st, r, e0); }
        else range_wave(S, st, r - pair_rows, e0);
        return;
    }
    // grid y: the bodies, outer ones (most planet work, the longest blocks) leading.  (The self-collision pairs of a scene with mounted
    // bodies are not rows of this grid: self_bounds_kernel / jobs_kernel.)
    const int l = S.body_order[(int)blockIdx.y];
    const int B = S.B, N = S.checks_per_step, NB = S.n_bodies;
    {                                                    // nothing to do for this body: leave before anything is staged
        const bool any = ((S.static_mask | S.moving_mask | S.contact_mask) >> l) & 1u;
        if (!any) {
            const int e0 = blockIdx.x * blockDim.x + threadIdx.x;
            if (e0 < B) {
                st.dist[(size_t)l * B + e0] = S.static_max; st.dist[(size_t)(NB + l) * B + e0] = S.moving_max + 0.002f;   // finish reads these
                st.res[(size_t)l * B + e0] = 0.f;
                if (N > 1) st.lb_moving[(size_t)l * B + e0] = -1.f;
            }
            return;
        }
    }
    // the hull, edge graph and start map staged in LDS: every support call of this block climbs over them
    __shared__ __attribute__((aligned(16))) float lds_v[4 * kMaxPieceVerts];
    __shared__ uint32_t lds_ids[kMaxPieceIdGroups];
    __shared__ uint32_t lds_start[kStartCells / 4];
    __shared__ __attribute__((aligned(16))) float lds_pv[AUTO ? 4 * kMaxPlanetVerts : 4];
    __shared__ float lds_pd[AUTO ? kPlanetDescDwords * kMaxPlanetPieces : 4];
    __shared__ uint32_t lds_keys[AUTO ? 7 * kBlock : 4];
    __shared__ uint16_t lds_items[kBlock * kMaxItemsPerLane];   // per wavefront: 64 x 23 entries (source lane << 5 | sample k)
    __shared__ uint32_t lds_hit[kBlock];
    const int pl = S.link_piece[l];
    const DPiece LP = load_piece(S.pieces, pl);
    {
        const SMR_GLOBAL float4 *gv = (const SMR_GLOBAL float4 *)S.verts + LP.begin;
        const SMR_GLOBAL uint32_t *gi = (const SMR_GLOBAL uint32_t *)S.ids4 + LP.ids_begin,
                                  *gm = (const SMR_GLOBAL uint32_t *)S.start + (kStartCells / 4) * (size_t)pl;
        for (int i = threadIdx.x; i < LP.count; i += kBlock) {
            const SMR_GLOBAL float *f = (const SMR_GLOBAL float *)(gv + i);
            lds_v[4 * i] = f[0]; lds_v[4 * i + 1] = f[1]; lds_v[4 * i + 2] = f[2]; lds_v[4 * i + 3] = f[3];
        }
        for (int i = threadIdx.x; i < LP.ids_count; i += kBlock) lds_ids[i] = gi[i];
        for (int i = threadIdx.x; i < kStartCells / 4; i += kBlock) lds_start[i] = gm[i];
        if (AUTO) stage_planets(S, threadIdx.x, kBlock, lds_pv, lds_pd);
    }
    __syncthreads();
    const LocalHull hull{lds_v, lds_ids, (const uint8_t *)lds_start};
    const int e0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e0 < B;                           // the wavefront shares out its items: no lane leaves early
    const int e = valid ? e0 : B - 1;
    auto lane_input = [&](int env) {
        LaneInput in;
        in.q = st.q + env; in.v = st.v + env; in.a = st.a + env; in.a1 = st.a1 + env; in.stride = B;
        in.planet_idx[0] = st.planet_idx[env]; in.planet_idx[1] = st.planet_idx[B + env];
        in.moving_hit = st.moving_hit[env];
        in.ball_k0 = 0; in.ball_kmax = 0; in.ball_khit = 0; in.ball_active = 0; in.lb_start = -1.f;
        if (S.use_ball) {
            in.ball = load_ball(st, B, env);
            in.ball_k0 = st.ball_i[env]; in.ball_kmax = st.ball_i[B + env]; in.ball_khit = st.ball_i[2 * B + env]; in.ball_active = st.ball_i[3 * B + env];
        }
        return in;
    };
    LaneInput in = lane_input(e);
    // the bound the env's previous step left behind holds for t = 0 of this one while the episode goes on (a new episode starts with
    // step_count 0; state injection clears the array) and, in a ball scene, while it is the same ball (its update counter moved on)
    in.lb_start = (N > 1 && st.step_count[e] > 0 && !(S.use_ball && in.ball_k0 < S.updates_per_step)) ? st.lb_moving[(size_t)l * B + e] : -1.f;
    const LaneResult r = collide_lane<SELF, AUTO>(S, l, LP, hull, in, lds_pv, lds_pd, lds_keys + threadIdx.x, kBlock);
    if (valid) {
        st.dist[(size_t)l * B + e] = r.d_static; st.dist[(size_t)(NB + l) * B + e] = r.d_moving;
        if (N > 1) st.lb_moving[(size_t)l * B + e] = r.lb_end;
    }
    bool contact = r.contact;
    if (!AUTO && S.ball_jobs) {                           // the run against the ball the bounds could not settle: a job (piece field: mov | con << 1)
        const int bodies[1] = {l};
        const uint32_t masks[1] = {(valid && r.ball_pending) ? 1u << (r.ball_pending >> 1) : 0u};
        push_jobs<1, true>(st, e, bodies, masks, N, kJobBallEnd);
    }
#ifdef SMR_PROFILE_ROWS
    const unsigned long long items_t0 = SMR_DEV_CLOCK();
#endif
    if (N > 1) {
        // The samples k_first .. k_need of every lane become work items of its WAVEFRONT: a queue in LDS (source lane, sample),
        // worked off 64 items at a time by all lanes - dense rounds, however unevenly the items are spread over the lanes.
        const int n = (valid && r.k_need >= r.k_first) ? r.k_need - r.k_first + 1 : 0;
        if (__any(n > 0)) {
            const int lane = (int)__lane_id(), wave0 = (int)threadIdx.x - lane;
            uint16_t *queue = lds_items + wave0 * kMaxItemsPerLane;
            int incl = n;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            const int total = __shfl(incl, 63);
            for (int k = 0; k < n; ++k) queue[incl - n + k] = (uint16_t)((lane << 5) | (r.k_first + k));
            lds_hit[threadIdx.x] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // rounds of 64 items: frame of the body at the sample + lower bound; the few items whose bound does not clear the contact
            // distance are compacted to the front of the queue (behind the read position) ...
            int pending = 0;
            for (int base = 0; base < total; base += 64) {
                const bool live = base + lane < total;
                const int item = queue[live ? base + lane : 0];
                const LaneInput it = lane_input(__shfl(e, item >> 5));
                const bool cand = collide_item<SELF, AUTO, true>(S, l, LP, hull, it, item & 31, live, lds_pv, lds_pd, lds_keys + threadIdx.x, kBlock);
                const unsigned long long m = __ballot(cand);
                if (cand) queue[pending + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)item;
                pending += __popcll(m);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            SMR_WAVE_REC(5, total); SMR_WAVE_REC(6, pending);
            // ... and get the real test, again 64 at a time (ball jobs: as jobs of kind kJobBallItem, contact only)
            for (int base = 0; base < pending; base += 64) {
                const bool live = base + lane < pending;
                const int item = queue[live ? base + lane : 0], src = item >> 5;
                if (!AUTO && S.ball_jobs) {
                    const int bodies[1] = {l};
                    const uint32_t masks[1] = {live ? 1u << 2 : 0u};
                    push_jobs(st, __shfl(e, src), bodies, masks, item & 31, kJobBallItem);
                    continue;
                }
                const LaneInput it = lane_input(__shfl(e, src));
                if (collide_item<SELF, AUTO, false>(S, l, LP, hull, it, item & 31, live, lds_pv, lds_pd, lds_keys + threadIdx.x, kBlock) && live) lds_hit[wave0 + src] = 1u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            contact = contact || lds_hit[threadIdx.x] != 0u;
        }
    }
#ifdef SMR_PROFILE_ROWS
    SMR_WAVE_REC(4, (SMR_DEV_CLOCK() - items_t0) >> 4);
#endif
    if (valid) st.res[(size_t)l * B + e] = contact ? 1.f : 0.f;
}

// ---- pair-job path of the collide stage (planet scenes; smr_device.h "Pair jobs") ----
__device__ __forceinline__ LaneInput load_lane_input(const DScene &S, const DState &st, int env) {
    const int B = S.B;
    LaneInput in;
    in.q = st.q + env; in.v = st.v + env; in.a = st.a + env; in.a1 = st.a1 + env; in.stride = B;
    in.planet_idx[0] = st.planet_idx[env]; in.planet_idx[1] = st.planet_idx[B + env];
    in.moving_hit = st.moving_hit[env];
    in.ball_k0 = 0; in.ball_kmax = 0; in.ball_khit = 0; in.ball_active = 0; in.lb_start = -1.f;
    if (S.use_ball) {
        in.ball = load_ball(st, B, env);
        in.ball_k0 = st.ball_i[env]; in.ball_kmax = st.ball_i[B + env]; in.ball_khit = st.ball_i[2 * B + env]; in.ball_active = st.ball_i[3 * B + env];
    }
    return in;
}

// keys kernel: the collide kernel's grid and prologue (one lane per (env, body), the body's hull in LDS for the table shortcut),
// but no GJK run against a moving piece: keys, best key, items and the pending table run go to the lists the enumerate / jobs
// kernels work off
// (128 registers: four blocks per CU - 1280 working blocks over 1024 slots instead of 768; space 0.129 -> 0.117 ms, space + ball machine
// 0.193 -> 0.153 ms, reaching task 0.147 -> 0.124 ms; measured)
#ifndef SMR_KEYS_WAVES
#define SMR_KEYS_WAVES 4
#endif
template <bool SELF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMR_KEYS_WAVES, SMR_KEYS_WAVES))) void keys_kernel(const DScene *__restrict__ Sp, DState st) {
    const DScene &S = *Sp;
    constexpr int kBlock = 256;
    if ((int)blockIdx.y >= S.n_bodies) {                 // rows behind the bodies: the self-collision pairs' bound tests (see collide_kernel)
        if (SELF) self_bounds_row(S, st, (int)blockIdx.y - S.n_bodies, (int)(blockIdx.x * blockDim.x + threadIdx.x));
        return;
    }
    const int l = S.body_order[(int)blockIdx.y];
    const int B = S.B, N = S.checks_per_step, NB = S.n_bodies;
    {
        const bool any = ((S.static_mask | S.moving_mask | S.contact_mask) >> l) & 1u;
        if (!any) {
            const int e0 = blockIdx.x * blockDim.x + threadIdx.x;
            if (e0 < B) {
                st.dist[(size_t)l * B + e0] = S.static_max; st.dist[(size_t)(NB + l) * B + e0] = S.moving_max + 0.002f;
                st.res[(size_t)l * B + e0] = 0.f;
                if (N > 1) st.lb_moving[(size_t)l * B + e0] = -1.f;
            }
            return;
        }
    }
    __shared__ __attribute__((aligned(16))) float lds_v[4 * kMaxPieceVerts];
    __shared__ uint32_t lds_ids[kMaxPieceIdGroups];
    __shared__ uint32_t lds_start[kStartCells / 4];
    __shared__ uint32_t lds_keys[7 * kBlock];
    __shared__ __attribute__((aligned(16))) float lds_pv[4 * kMaxPlanetVerts];
    __shared__ float lds_pd[kPlanetDescDwords * kMaxPlanetPieces];
    const int pl = S.link_piece[l];
    const DPiece LP = load_piece(S.pieces, pl);
    {                                                                   // the hull: table shortcut, pair rows, upper bound of the best candidate
        const SMR_GLOBAL float4 *gv = (const SMR_GLOBAL float4 *)S.verts + LP.begin;
        const SMR_GLOBAL uint32_t *gi = (const SMR_GLOBAL uint32_t *)S.ids4 + LP.ids_begin,
                                  *gm = (const SMR_GLOBAL uint32_t *)S.start + (kStartCells / 4) * (size_t)pl;
        for (int i = threadIdx.x; i < LP.count; i += kBlock) {
            const SMR_GLOBAL float *f = (const SMR_GLOBAL float *)(gv + i);
            lds_v[4 * i] = f[0]; lds_v[4 * i + 1] = f[1]; lds_v[4 * i + 2] = f[2]; lds_v[4 * i + 3] = f[3];
        }
        for (int i = threadIdx.x; i < LP.ids_count; i += kBlock) lds_ids[i] = gi[i];
        for (int i = threadIdx.x; i < kStartCells / 4; i += kBlock) lds_start[i] = gm[i];
        stage_planets(S, threadIdx.x, kBlock, lds_pv, lds_pd);
        __syncthreads();
    }
    const LocalHull hull{lds_v, lds_ids, (const uint8_t *)lds_start};
    const int e0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e0 < B;
    const int e = valid ? e0 : B - 1;
    LaneInput in = load_lane_input(S, st, e);
    in.lb_start = (N > 1 && st.step_count[e] > 0) ? st.lb_moving[(size_t)l * B + e] : -1.f;
    const LaneResult r = collide_lane<SELF, true, true>(S, l, LP, hull, in, lds_pv, lds_pd, lds_keys + threadIdx.x, kBlock);
    if (valid) {
        st.dist[(size_t)l * B + e] = r.d_static; st.dist[(size_t)(NB + l) * B + e] = S.moving_max + 0.002f;
        st.res[(size_t)l * B + e] = 0.f;
        if (N > 1) st.lb_moving[(size_t)l * B + e] = r.lb_end;
        // the candidate keys of this (env, body) for the enumeration, and the upper bound of its best candidate: the env's moving
        // distance cannot exceed it
#pragma unroll
        for (int w = 0; w < 7; ++w) st.keyw[((size_t)w * NB + l) * B + e] = lds_keys[w * kBlock + threadIdx.x];
        {   // the body's frame for the jobs of this (env, body): 48 contiguous bytes per lane
            SMR_GLOBAL V4 *f = (SMR_GLOBAL V4 *)st.frames + 3 * ((size_t)l * B + e);
            const Xf &T = r.frame;
            f[0] = V4{T.r[0], T.r[1], T.r[2], T.r[3]}; f[1] = V4{T.r[4], T.r[5], T.r[6], T.r[7]}; f[2] = V4{T.r[8], T.t.x, T.t.y, T.t.z};
        }
        st.lb1[(size_t)l * B + e] = (r.best != 0xFFFFFFFFu && r.lb1 > 0.f) ? (((unsigned)__float_as_int(r.lb1) & ~31u) | (r.best & 31u)) : 0u;
        if (r.best != 0xFFFFFFFFu) {
            st.env_best[e] = 1u;                             // some body of the env has a candidate
            if (r.ub <= S.moving_max) atomicMin(&st.env_min[e], dist_bits(r.ub));
        }
    }
    // The table's GJK run (one lane in a hundred) becomes a job of its own, the samples k_first .. k_need of every lane go to the global
    // item list (one atomic per block, below).  Both counters are asked before either answer is used: an atomic that returns a value is a
    // round trip to the memory side, and the counter is a queue - every atomic in front of it adds ~3 ns.
    const bool pending = valid && r.static_pending;
    int slot = 0;
    if (pending) slot = atomicAdd(st.job_count, 1);
    const int n = (N > 1 && valid && r.k_need >= r.k_first) ? r.k_need - r.k_first + 1 : 0;
    const int lane = (int)__lane_id();
    int incl = n, base = 0;
    if (N > 1) {                                         // (one atomic per BLOCK: 5 000 wavefronts a step queued up at the counter, ~3 ns each)
        __shared__ int wave_items[kBlock / 64], block_items;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        const int w = (int)(threadIdx.x >> 6);
        if (lane == 63) wave_items[w] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int i = 0; i < kBlock / 64; ++i) t += wave_items[i];
            block_items = t > 0 ? atomicAdd(st.item_count, t) : 0;
        }
        __syncthreads();
        base = block_items;
        for (int i = 0; i < w; ++i) base += wave_items[i];
    }
    if (pending) {
        if (slot < st.job_split) st.jobs[slot] = job_word(e, l, S.auto_static, N, kJobStaticEnd);
        else st.overflow_env[e] = 1;
    }
    for (int k = 0; k < n; ++k) st.items[(size_t)base + incl - n + k] = job_word(e, l, 0, r.k_first + k, kJobItem);
}

// tables of the first / jobs kernels in dynamic LDS: the hull pool of the bodies that take part in moving / static queries
// (DScene::job_pool), the planets' padded vertex lists and descriptors (+ the table's corners)
extern __shared__ __attribute__((aligned(16))) uint32_t job_lds[];
struct JobTables { BodyPool BP; const float *pv, *pd, *cyl; };
constexpr int kJobCylDwords = 8 * 4;             // up to four world-fixed cylinders named by self-collision pairs (8 floats each)
__device__ __forceinline__ JobTables stage_job_tables(const DScene &S) {
    uint32_t *pool = job_lds;
    const int vn = S.planets_jobs ? S.auto_vn : (S.ball_jobs ? S.ball_jobs_vn : 0), pn = S.planets_jobs ? S.auto_pn : 0;      // (a scene without pair jobs stages no planets)
    float *pv = (float *)(job_lds + S.job_pool_dwords), *pd = pv + 4 * vn, *cyl = pd + kPlanetDescDwords * pn;
    const SMR_GLOBAL uint32_t *g = (const SMR_GLOBAL uint32_t *)S.job_pool;
    for (int i = threadIdx.x; i < S.job_pool_dwords; i += blockDim.x) pool[i] = g[i];
    if (S.planets_jobs) stage_planets(S, threadIdx.x, blockDim.x, pv, pd);
    if (S.ball_jobs) {                                   // the ball's vertices, padded to a multiple of 8 with its last one (ball jobs scan them eight at a time)
        const DPiece BPc = load_piece(S.pieces, S.ball_piece);
        const SMR_GLOBAL float *gv = (const SMR_GLOBAL float *)S.verts + 4 * (size_t)BPc.begin;
        for (int i = threadIdx.x; i < 4 * S.ball_jobs_vn; i += blockDim.x) pv[i] = gv[4 * min(i >> 2, BPc.count - 1) + (i & 3)];
    }
    if ((int)threadIdx.x < S.n_self_pairs && S.self_cyl_slot[threadIdx.x] >= 0) {       // centre, core radius, core half length, margin
        const DPiece A = load_piece_lane(S.pieces, -(S.self_pair[threadIdx.x][1] + 1));
        float *c = cyl + 8 * S.self_cyl_slot[threadIdx.x];
        c[0] = A.cx; c[1] = A.cy; c[2] = A.cz; c[3] = A.cyl_r; c[4] = A.cyl_h; c[5] = A.margin; c[6] = 0.f; c[7] = 0.f;
    }
    __syncthreads();
    return {BodyPool{pool}, pv, pd, cyl};
}

// what a job of (env state `in`, body) looks for: the distance of the reward (mov) and / or the contact latch (con) - collide_lane's rules
__device__ __forceinline__ void job_flags(const DScene &S, const LaneInput &in, int body, bool &mov, bool &con) {
    mov = (S.moving_mask >> body) & 1u;
    con = S.term_moving && S.n_planets > 0 && in.moving_hit == 0 && ((S.contact_mask >> body) & 1u);
}

// the result of a settled job.  Atomic minima on the bits of non-negative floats and stores of 1: independent of the order of the jobs.
// seen: the env's minimum the job read when it started - a result that does not improve on it changes nothing and is not sent (the
// atomics execute at the memory side, and a wavefront's next refill waits for every one it has in flight)
__device__ __forceinline__ void job_result(const DScene &S, const DState &st, const Job &jb, bool mov, bool con, float d, float seen) {
    const int B = S.B;
    if (jb.kind == kJobStaticEnd) {
        if (d <= S.static_max) {
            atomicMin(&st.env_dstatic[jb.env], dist_bits(d));
            atomicMin((unsigned *)st.dist + (size_t)jb.body * B + jb.env, dist_bits(d));       // the body's own entry too (smr_link_distances; non-negative floats order like their bits)
        }
        return;
    }
    if (jb.kind == kJobSelf) { if (d < S.self_cut && d < seen) atomicMin(&st.env_dself[jb.env], dist_bits(d)); return; }
    if (mov && d <= S.moving_max && d < seen) atomicMin(&st.env_min[jb.env], dist_bits(d));
    if (con && d <= S.contact_dist) st.res[(size_t)jb.body * B + jb.env] = 1.f;
}

// cutoff and early-stop distance of a job (planet_gjk's rules; an item only asks "within the contact distance?")
__device__ __forceinline__ void job_limits(const DScene &S, const Job &jb, bool mov, bool con, float env_min, float &cutoff, float &stop) {
    if (jb.kind == kJobStaticEnd) { cutoff = S.static_max; stop = S.collision_dist; return; }
    if (jb.kind == kJobSelf) { cutoff = fminf(env_min, S.self_cut); stop = S.collision_dist; return; }      // env_min: the env's self distance so far
    if (jb.kind == kJobItem || jb.kind == kJobBallItem) { cutoff = S.contact_dist; stop = S.contact_dist; return; }
    cutoff = mov ? fminf(env_min, S.moving_max) : S.contact_dist;
    if (con) cutoff = fmaxf(cutoff, S.contact_dist);
    stop = con ? S.contact_dist : S.collision_dist;
}

// Append the jobs (env, body b, piece of every set bit of masks[b], k, kind) of all lanes of the wavefront: ONE atomic on the list
// counter per call (a counter hit by every lane's every job serialises in the L2: 190 000 atomics per step cost 0.7 ms, measured).
// All lanes must call together.  A full list is not an error: the env is handed to the overflow kernel, which works ALL its queries
// in the lane form (a second GJK loop inline here costs the hot kernels 300 - 2400 spilled registers, measured).
// BLOCK: all wavefronts of the block call together and share ONE pair of atomics (the enumerate kernel: its 10 000 wavefronts per step that
// have a job hit the two counters one after the other, ~3 ns each)
template <int NM, bool BLOCK>
__device__ __forceinline__ void push_jobs(const DState &st, int env, const int (&bodies)[NM], const uint32_t (&masks)[NM], int k, int kind) {
    const int lane = (int)__lane_id();
    // planet pieces with a long vertex list go to the second region of the list (DState::job_split); self pairs / the table stay in front
    const unsigned big = (kind == kJobPlanetEnd || kind == kJobItem) ? st.big_piece_mask : 0u;
    int n = 0, nb = 0;
#pragma unroll
    for (int i = 0; i < NM; ++i) { n += __popc(masks[i] & ~big); nb += __popc(masks[i] & big); }
    if (!BLOCK && !__any(n + nb > 0)) return;
    int incl = n, inclb = nb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d), tb = __shfl_up(inclb, d); if (lane >= d) { incl += t; inclb += tb; } }
    const int total = __shfl(incl, 63), totalb = __shfl(inclb, 63);
    int base = 0, baseb = 0;
    if (BLOCK) {
        __shared__ int wave_total[2][16], block_base[2];
        const int w = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
        if (lane == 0) { wave_total[0][w] = total; wave_total[1][w] = totalb; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t0 = 0, t1 = 0;
            for (int i = 0; i < nw; ++i) { t0 += wave_total[0][i]; t1 += wave_total[1][i]; }
            block_base[0] = t0 > 0 ? atomicAdd(st.job_count, t0) : 0;
            block_base[1] = t1 > 0 ? atomicAdd(st.job_count_big, t1) : 0;
        }
        __syncthreads();
        base = block_base[0]; baseb = block_base[1];
        for (int i = 0; i < w; ++i) { base += wave_total[0][i]; baseb += wave_total[1][i]; }
        __syncthreads();                                 // (the totals are read: the next call may write them)
    } else if (lane == 0) {                              // (both counters asked before either answer is used)
        if (total > 0) base = atomicAdd(st.job_count, total);
        if (totalb > 0) baseb = atomicAdd(st.job_count_big, totalb);
    }
    if (!BLOCK) { base = __shfl(base, 0); baseb = __shfl(baseb, 0); }
    base += incl - n;
    baseb = st.job_split + baseb + inclb - nb;
    if ((n > 0 && base + n > st.job_split) || (nb > 0 && baseb + nb > st.job_cap)) st.overflow_env[env] = 1;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        uint32_t mask = masks[i];
        while (mask) {
            const int piece = __ffs((int)mask) - 1;
            mask &= mask - 1;
            if ((big >> piece) & 1u) { if (baseb < st.job_cap) st.jobs[baseb] = job_word(env, bodies[i], piece, k, kind); ++baseb; }
            else { if (base < st.job_split) st.jobs[base] = job_word(env, bodies[i], piece, k, kind); ++base; }
        }
    }
}

// self_bounds kernel (scenes with self-collision pairs): one lane per (env, pair), grid y = pair - the lower bounds of the pair at the
// end of the step (bound table, spheres, capsules: self_pair_cleared); a pair they cannot put beyond the distance range the reward
// reads (DScene::self_cut) becomes a job of kind kJobSelf: 0.3 per env on uniformly drawn states, where the 14 pairs of the ball
// machine used to be 14 rows of the collide grid with a GJK run in most lanes of the two rows against iiwa_link_5.
// (the rows of one pair: a kernel of its own - self_bounds_kernel - or rows behind the bodies of the keys / collide grid, round 4)
__device__ __forceinline__ void self_bounds_row(const DScene &S, const DState &st, int pair, int e0);
__global__ __launch_bounds__(256) void self_bounds_kernel(const DScene *__restrict__ Sp, DState st) {
    self_bounds_row(*Sp, st, (int)blockIdx.y, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}
__device__ __forceinline__ void self_bounds_row(const DScene &S, const DState &st, int pair, int e0) {
    const int B = S.B;
    const bool valid = e0 < B;
    const int e = valid ? e0 : B - 1;
    float q[SMR_NJ];
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) q[j] = interp_pos(st.q[j * B + e], st.v[j * B + e], st.a[j * B + e], st.a1[j * B + e], S.dt, S.dt);
    const bool job = valid && !(self_pair_lower_bound(S, pair, q, S.self_cut + kBoundSlack) > S.self_cut + kBoundSlack);
    const uint32_t masks[1] = {job ? 1u << pair : 0u};
    const int bodies[1] = {0};
    push_jobs<1, true>(st, e, bodies, masks, S.checks_per_step, kJobSelf);      // (every caller brings whole blocks)
}

// (128 registers: the kernel needs 125.)  The grid is ONE 512-lane block per CU - two wavefronts per SIMD: the runs are latency chains,
// and with two blocks per CU every wavefront took longer than the kernel gained from the extra lanes (65 536 envs: 0.250 -> 0.235 ms,
// one contact sample: 0.413 -> 0.398 ms, 4096 envs: 0.146 -> 0.131 ms; the reaching task liked two blocks better by 2 %; measured)
#ifndef SMR_JOB_WAVES
#define SMR_JOB_WAVES 4
#endif


// enumerate kernel (no GJK, no LDS).  Blocks [0, env_blocks): one lane per env - every pair of the env whose key bound does not exceed
// the upper bound the keys kernel left in env_min (or, for a body that is tested for contact, the contact distance) becomes a job.
// The blocks behind them stride over the item list: the bound test of one intermediate contact sample per lane; the pieces it cannot
// clear become jobs.
template <bool SELF>
__global__ __launch_bounds__(256) void enumerate_kernel(const DScene *__restrict__ Sp, DState st, int env_blocks) {
    const DScene &S = *Sp;
    const int B = S.B, N = S.checks_per_step, NB = S.n_bodies;
    if ((int)blockIdx.x < env_blocks) {
        const int e0 = blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = e0 < B;
        const int e = valid ? e0 : B - 1;
        const bool has = valid && st.env_best[e] != 0xFFFFFFFFu;
        const LaneInput in = load_lane_input(S, st, e);
        const float cut_first = fminf(__uint_as_float(st.env_min[e]), S.moving_max);
        int bodies[SMR_MAX_BODIES];
        uint32_t masks[SMR_MAX_BODIES];
#pragma unroll
        for (int b = 0; b < SMR_MAX_BODIES; ++b) {
            bodies[b] = b; masks[b] = 0u;
            if (b >= NB || !(((S.moving_mask | S.contact_mask) >> b) & 1u)) continue;
            bool mov2, con2;
            job_flags(S, in, b, mov2, con2);
            float cutoff = mov2 ? cut_first : S.contact_dist;
            if (con2) cutoff = fmaxf(cutoff, S.contact_dist);
            if (has && (mov2 || con2)) {
                uint32_t kws[7];
#pragma unroll
                for (int w = 0; w < 7; ++w) kws[w] = st.keyw[((size_t)w * NB + b) * B + e];      // seven loads in flight
                uint32_t mask = 0u;
#pragma unroll
                for (int w = 0; w < 7; ++w)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t key = (kws[w] >> (8 * k)) & 0xFFu;
                        if (key != 255u && (float)key * kKeyStep <= cutoff) mask |= 1u << (4 * w + k);
                    }
                // the best candidate's first GJK iteration was made by the keys kernel: where its separating plane already lies beyond the
                // cutoff the run would end there - no job (SMR_NO_LB1: A/B switch, compile time)
#ifndef SMR_NO_LB1
                const unsigned l1 = st.lb1[(size_t)b * B + e];
                if (__uint_as_float(l1 & ~31u) > cutoff) mask &= ~(1u << (l1 & 31u));
#endif
                masks[b] = mask;
            }
        }
        push_jobs<SMR_MAX_BODIES, true>(st, e, bodies, masks, N, kJobPlanetEnd);
        return;
    }
    const int n_items = N > 1 ? *st.item_count : 0, stride = ((int)gridDim.x - env_blocks) * (int)blockDim.x;
    const int rounds = (n_items + stride - 1) / stride;                 // block uniform: the lanes of a wavefront append together
    for (int r = 0; r < rounds; ++r) {
        const int i = r * stride + ((int)blockIdx.x - env_blocks) * (int)blockDim.x + (int)threadIdx.x;
        const bool valid = i < n_items;
        const Job it = job_unpack(st.items[valid ? i : 0]);
        const LaneInput in = load_lane_input(S, st, it.env);
        const Xf T = lane_frame_any<true>(S, it.body, in, S.dt * (float)it.k / (float)N);      // bounds: hardware sine / cosine, covered by the slack
        constexpr float kBoundSlack = 1e-4f;
        const int upd = S.updates_per_step / N;
        int idx[SMR_MAX_PLANETS] = {0, 0};
#pragma unroll
        for (int p = 0; p < SMR_MAX_PLANETS; ++p)
            if (p < S.n_planets) idx[p] = (in.planet_idx[p] + it.k * upd) % S.planet[p].n_orbit;
        const DPiece LP = load_piece_lane(S.pieces, S.link_piece[it.body]);
        uint32_t masks[1] = {0u};
        const int bodies[1] = {it.body};
        planet_candidates(S, LP, T, idx, valid, S.contact_dist + kBoundSlack, [&](int piece) { masks[0] |= 1u << piece; });
        // (one pair of atomics per block where the items fit one round; with several rounds the block's wavefronts would wait for each
        // other every round - reaching task: enumerate 0.102 -> 0.119 ms - and append wavefront by wavefront)
        if (rounds == 1) push_jobs<1, true>(st, it.env, bodies, masks, it.k, kJobItem);
        else push_jobs(st, it.env, bodies, masks, it.k, kJobItem);
    }
}

// jobs kernel: one lane per job of the list.  The wavefronts of the persistent grid draw their jobs from ONE stream (job_next):
// when all lanes of a wavefront have settled their runs it takes the next 64 jobs, the lanes set their runs up, then every lane
// with a run advances it by one GJK iteration per trip.  The stream is what keeps the kernel short beside the reset-side kernels:
// a block whose CU is held by two of their blocks starts hundreds of microseconds late, and with equal shares per wavefront
// (no counter at all) the kernel waited for it - 0.52 ms instead of 0.22 (measured).  The order in which the jobs are worked does
// not matter: their results are minima and flags (job_result).
// What was measured on the way: refilling as soon as 16 / 32 lanes are idle (slower: a refill stalls the running lanes for a setup),
// asking for the next chunk when 16 / 32 / 48 lanes have settled so that the counter's round trip runs beside the last trips
// (slower the earlier it asks: the chunk then waits for the wavefront's longest run while other wavefronts are idle).
template <bool SELF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(SMR_JOB_WAVES, SMR_JOB_WAVES))) void jobs_kernel(const DScene *__restrict__ Sp, DState st, int defer_after, int quota_max, int range_blocks) {
    const DScene &S = *Sp;
    if ((int)blockIdx.x >= (int)gridDim.x - range_blocks) { range_block(S, st, (int)blockIdx.x - ((int)gridDim.x - range_blocks)); return; }
#ifdef SMR_PROFILE_ROWS
    const unsigned long long pc0 = SMR_DEV_CLOCK();
    unsigned long long pc_setup = 0, pc_trip = 0, pc_atomic = 0, pc_loads = 0, pc_sa = 0, pc_sb = 0, pc_rest = 0;
    int prof_trips = 0, prof_refills = 0, prof_maxit = 0, prof_lanes = 0, prof_hist[5] = {0, 0, 0, 0, 0}, prof_n[3] = {0, 0, 0}, prof_bucket = -1;
    const unsigned prof_wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);       // scripts/profile_jobs.py: the schedule of the wavefronts
    const unsigned long long prof_w0 = wall_clock64();
    unsigned long long prof_w1 = 0;
#endif
    const int n_front = min(*st.job_count, st.job_split), n_jobs = n_front + min(*st.job_count_big, st.job_cap - st.job_split);
    // An agent-scope atomic on the list counter is a round trip to the memory side of ~10 us, more when 2048 wavefronts ask at once
    // (profile counters: 83 000 of a wavefront's 390 000 cycles at 65 536 envs went into 2.5 of them; a look at the counter before
    // staging, to let late blocks leave, cost the kernel 4 %; a look before every read-modify-write 30 %).  So the first hand-out
    // is taken per BLOCK and its round trip hides behind the staging; and when the first hand-outs cover the list nobody asks again.
    // With fewer jobs than lanes (small batches) the wavefronts share them out thinly: every SIMD gets a few runs instead of a few
    // CUs getting them all, and a wavefront's chain is the longest of four runs, not of sixty-four (a floor of 8 -> 4 with two blocks per
    // CU: 4096 envs 0.117 -> 0.106 ms; 2: 0.109 ms).
    const int n_waves = ((int)gridDim.x - range_blocks) * (int)(blockDim.x >> 6);
#ifndef SMR_JOB_QUOTA_MIN
#define SMR_JOB_QUOTA_MIN 4
#endif
    const int quota = min(quota_max, max(SMR_JOB_QUOTA_MIN, (n_jobs + n_waves - 1) / n_waves));
    __shared__ int first_base;
    if (threadIdx.x == 0) first_base = atomicAdd(st.job_next, (int)(blockDim.x >> 6) * quota);
    const JobTables JT = stage_job_tables(S);           // (ends with a barrier)
#ifdef SMR_PROFILE_ROWS
    SMR_DEV_COUNT(26, SMR_DEV_CLOCK() - pc0);
    prof_w1 = wall_clock64();
#endif
    const int lane = (int)__lane_id();
    bool active = false, mov = false, con = false, first = true;
    bool dry = (long long)n_waves * quota >= (long long)n_jobs;              // (for every ask after the first hand-out)
    Job jb = {0, 0, 0, 0, 0};
    unsigned long long jword = 0ull;
    float seen = 0.f;
    JobLane J;
    J.g.start(F3{1.f, 0.f, 0.f}, 0.f, 0.f);               // defined state for the lanes that never get a job
    J.XA = Xf{{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, F3{0.f, 0.f, 0.f}}; J.T = J.XA;
    J.hull = JT.BP.hull(0); J.av = JT.pv; J.ac = 0; J.margins = 0.f; J.is_static = false; J.hull_body = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if (n_idle == 64 && (first || !dry)) {
#ifdef SMR_PROFILE_ROWS
            const unsigned long long t0 = SMR_DEV_CLOCK();
            ++prof_refills;
#endif
            const int take = quota;
            int base = 0;
            if (first) { base = first_base + (int)(threadIdx.x >> 6) * quota; first = false; }
            else {
                if (lane == 0) base = atomicAdd(st.job_next, take);
                base = __builtin_amdgcn_readfirstlane(base);
            }
            dry = dry || base + take >= n_jobs;                // the stream is exhausted (for every wavefront that asks from now on)
#ifdef SMR_PROFILE_ROWS
            const unsigned long long t0a = SMR_DEV_CLOCK();
            pc_atomic += t0a - t0;
#endif
            const int rank = __popcll(idle & ((1ull << lane) - 1ull)), j = base + rank;
            if (!active && rank < take && j < n_jobs) {
                jword = st.jobs[j < n_front ? j : st.job_split + (j - n_front)];      // the short scans first, then the long ones
                jb = job_unpack(jword);
                const LaneInput in = load_lane_input(S, st, jb.env);
                job_flags(S, in, jb.body, mov, con);
                if (jb.kind == kJobItem) { mov = false; con = true; }
                if (jb.kind == kJobBallEnd || jb.kind == kJobBallItem) { mov = jb.piece & 1; con = (jb.piece >> 1) & 1; }      // decided by the collide kernel
                float cutoff, stop;
                // the env's minimum so far prunes the run.  A plain load: a value another XCD has lowered since is still an upper bound,
                // and the jobs of one env sit side by side in the list - they start together whatever the load returns
                const unsigned *running = (SELF && jb.kind == kJobSelf) ? st.env_dself : st.env_min;
                seen = __uint_as_float(running[jb.env]);
                job_limits(S, jb, mov, con, seen, cutoff, stop);
#ifdef SMR_PROFILE_ROWS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long t0b = SMR_DEV_CLOCK();
                pc_loads += t0b - t0a;
#endif
                job_setup<SELF>(S, JT.BP, JT.pv, JT.pd, JT.cyl, in, st.frames, jb, cutoff, stop, J);
                active = true;
#ifdef SMR_PROFILE_ROWS
                prof_bucket = -1;
                if (jb.kind == kJobPlanetEnd || jb.kind == kJobItem) {
                    float ratio = 1.f;
                    if (jb.kind == kJobPlanetEnd) {
                        const uint32_t kw = st.keyw[((size_t)(jb.piece >> 2) * S.n_bodies + jb.body) * S.B + jb.env];
                        ratio = (float)((kw >> (8 * (jb.piece & 3))) & 0xFFu) * kKeyStep / fmaxf(cutoff, 1e-6f);
                    }
                    {   // slack between the cutoff and the key bound, in key steps: < 1, < 2, < 4, < 8, < 16, < 32, more; row 7: the items
                        const float slack = (1.f - ratio) * fmaxf(cutoff, 1e-6f) / kKeyStep;
                        const int row = jb.kind == kJobItem ? 7 : (slack < 1.f ? 0 : slack < 2.f ? 1 : slack < 4.f ? 2 : slack < 8.f ? 3 : slack < 16.f ? 4 : slack < 32.f ? 5 : 6);
                        prof_bucket = 8 * row;
                    }
                    if (J.ac > kSmallScan) atomicAdd(&g_job_hist[prof_bucket + 5], 1ull);
                    atomicAdd(&g_job_hist[prof_bucket + 6], 1ull);
                }
#endif
            }
#ifdef SMR_PROFILE_ROWS
            pc_setup += SMR_DEV_CLOCK() - t0;
#endif
        }
        if (!__any(active)) break;
#ifdef SMR_PROFILE_ROWS
        const unsigned long long t1 = SMR_DEV_CLOCK();
        ++prof_trips;
#endif
        float d;
#ifdef SMR_PROFILE_ROWS
        int prof_its = 0;
        prof_lanes += __popcll(__ballot(active));
#ifdef SMR_PROFILE_TRIP      // (-DSMR_PROFILE_ROWS -DSMR_PROFILE_TRIP: clocks around the two support calls of every iteration - they double the trip)
        const int prof_src = __ffsll((long long)__ballot(active)) - 1;      // a lane that takes part in this trip: its clock deltas are the wavefront's
        const unsigned long long sa0 = __shfl(J.g.pc_sa, prof_src), sb0 = __shfl(J.g.pc_sb, prof_src), rest0 = __shfl(J.g.pc_rest, prof_src);
#endif
        { const int nn = J.g.n; prof_n[0] += __popcll(__ballot(active && nn <= 1)); prof_n[1] += __popcll(__ballot(active && nn == 2)); prof_n[2] += __popcll(__ballot(active && nn == 3)); }
#endif
        if (active && job_trip<SELF>(J, d)) {
            active = false;
            job_result(S, st, jb, mov, con, d, seen);
#ifdef SMR_PROFILE_ROWS
            prof_maxit = max(prof_maxit, J.g.it + 1);
            prof_its = J.g.it + 1;
            if (prof_bucket >= 0) atomicAdd(&g_job_hist[prof_bucket + (prof_its == 1 ? 0 : prof_its == 2 ? 1 : prof_its <= 4 ? 2 : prof_its <= 8 ? 3 : 4)], 1ull);
#endif
        }
        // a run that has used up its iterations in the lane form leaves as a record for jobs_coop_kernel (smr_device.h, "deferred runs"):
        // one atomic per wavefront and trip that has such runs, i.e. one per hand-out
        {
            const bool leave = active && defer_after > 0 && J.g.it >= defer_after;
            const unsigned long long lm = __ballot(leave);
            if (lm) {
                const int src = __ffsll((long long)lm) - 1;
                int base = 0;
                if (lane == src) base = atomicAdd(st.defer_count, __popcll(lm));
                base = __builtin_amdgcn_readlane(base, src);
                const int slot = base + __popcll(lm & ((1ull << lane) - 1ull));
                if (leave && slot < st.defer_cap) {          // (a full list: the run simply stays where it is)
                    defer_write((SMR_GLOBAL V4 *)st.defer + (size_t)slot * (kDeferWords / 4), J, (const float *)job_lds, jword, mov, con, seen);
                    active = false;
                }
            }
        }
#ifdef SMR_PROFILE_ROWS
        pc_trip += SMR_DEV_CLOCK() - t1;
#ifdef SMR_PROFILE_TRIP
        pc_sa += __shfl(J.g.pc_sa, prof_src) - sa0; pc_sb += __shfl(J.g.pc_sb, prof_src) - sb0; pc_rest += __shfl(J.g.pc_rest, prof_src) - rest0;
#endif
        prof_hist[0] += __popcll(__ballot(prof_its == 1)); prof_hist[1] += __popcll(__ballot(prof_its == 2));
        prof_hist[2] += __popcll(__ballot(prof_its == 3 || prof_its == 4)); prof_hist[3] += __popcll(__ballot(prof_its >= 5 && prof_its <= 8));
        prof_hist[4] += __popcll(__ballot(prof_its >= 9));
#endif
    }
#ifdef SMR_PROFILE_ROWS
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) prof_maxit = max(prof_maxit, __shfl_xor(prof_maxit, d));
    SMR_DEV_COUNT(27, pc_setup); SMR_DEV_COUNT(28, pc_trip); SMR_DEV_COUNT(29, prof_trips); SMR_DEV_MAX(30, prof_maxit);
    SMR_DEV_MAX(31, SMR_DEV_CLOCK() - pc0); SMR_DEV_COUNT(25, prof_refills); SMR_DEV_COUNT(19, pc_atomic); SMR_DEV_COUNT(20, pc_loads);
    // anatomy of the trips (scripts/profile_rows.py): cycles inside the two support calls and the rest of gjk_iterate (every lane of a
    // wavefront reads the same clock: lane 0's sums are the wavefront's), lanes with a run per trip, iterations per job, simplex sizes
    SMR_DEV_COUNT(7, pc_sa); SMR_DEV_COUNT(8, pc_sb); SMR_DEV_COUNT(9, pc_rest); SMR_DEV_COUNT(10, prof_lanes);
    SMR_DEV_COUNT(11, prof_hist[0]); SMR_DEV_COUNT(12, prof_hist[1]); SMR_DEV_COUNT(13, prof_hist[2]); SMR_DEV_COUNT(14, prof_hist[3]); SMR_DEV_COUNT(15, prof_hist[4]);
    SMR_DEV_COUNT(3, prof_n[0]); SMR_DEV_COUNT(5, prof_n[1]); SMR_DEV_COUNT(21, prof_n[2]);
    if (prof_wave < 16384u && (threadIdx.x & 63) == 0) {
        g_block_times[2 * prof_wave] = prof_w0; g_block_times[2 * prof_wave + 1] = wall_clock64();
        g_wave_rec[8 * prof_wave] = (unsigned)prof_refills; g_wave_rec[8 * prof_wave + 1] = (unsigned)prof_trips; g_wave_rec[8 * prof_wave + 2] = (unsigned)prof_maxit;
        g_wave_rec[8 * prof_wave + 3] = (unsigned)(prof_w1 - prof_w0);
    }
#endif
}

// jobs_coop kernel: the deferred runs, one group of 16 lanes per run (smr_device.h, "deferred runs").  The grid is persistent - one
// 1024-lane block per CU, sixteen wavefronts = 64 groups - and stages the tables of the jobs kernel at the same LDS offsets (the records
// name the scanned side by offset).  Group g takes the records g, g + G, g + 2 G ...: when its run has settled it loads the next record
// (all 16 lanes read the same 144 bytes) while the other groups of its wavefront go on iterating.  No counter: a run costs 1 to 13
// iterations, a few runs per group even out.
template <bool SELF>
__global__ __launch_bounds__(1024) void jobs_coop_kernel(const DScene *__restrict__ Sp, DState st) {
    const DScene &S = *Sp;
    const int count = *st.defer_count, n_def = count < st.defer_cap ? count : st.defer_cap;
    if (n_def <= 0) return;                                // (block uniform: no barrier is skipped by part of a block)
    const JobTables JT = stage_job_tables(S);
    const float *lds0 = (const float *)job_lds;
    const int G = (int)gridDim.x * (int)(blockDim.x >> 4);
    int next = (int)blockIdx.x * (int)(blockDim.x >> 4) + (int)(threadIdx.x >> 4);
    const bool head = (threadIdx.x & 15) == 0;
    bool active = false, is_static = false, mov = false, con = false;
    GjkRun g;
    g.start(F3{1.f, 0.f, 0.f}, 0.f, 0.f);
    Xf X = Xf{{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, F3{0.f, 0.f, 0.f}};
    const float *av = lds0, *hv = lds0;
    int ac = 0, hcount = 0;
    float margins = 0.f, seen = 0.f;
    Job jb = {0, 0, 0, 0, 0};
    for (;;) {
        if (!active && next < n_def) {
            const SMR_GLOBAL V4 *src = (const SMR_GLOBAL V4 *)st.defer + (size_t)next * (kDeferWords / 4);
            V4 r[kDeferWords / 4];
#pragma unroll
            for (int i = 0; i < kDeferWords / 4; ++i) r[i] = src[i];
            next += G;
            X.r[0] = r[0][0]; X.r[1] = r[0][1]; X.r[2] = r[0][2]; X.r[3] = r[0][3]; X.r[4] = r[1][0]; X.r[5] = r[1][1]; X.r[6] = r[1][2]; X.r[7] = r[1][3];
            X.r[8] = r[2][0]; X.t = F3{r[2][1], r[2][2], r[2][3]};
            g.v = F3{r[3][0], r[3][1], r[3][2]}; g.vv = dot(g.v, g.v); g.cutoff2 = r[3][3]; g.stop2 = r[4][0];
            margins = r[4][1]; seen = r[4][2];
            const unsigned meta = (unsigned)__float_as_int(r[4][3]);
            jb = job_unpack((unsigned long long)(unsigned)__float_as_int(r[5][0]) | ((unsigned long long)(unsigned)__float_as_int(r[5][1]) << 32));
            g.w0 = F3{r[5][2], r[5][3], r[6][0]}; g.w1 = F3{r[6][1], r[6][2], r[6][3]}; g.w2 = F3{r[7][0], r[7][1], r[7][2]}; g.w3 = F3{r[7][3], r[8][0], r[8][1]};
            const int packed = __float_as_int(r[8][2]);
            g.n = packed & 255; g.it = (packed >> 8) & 255; g.stalls = (packed >> 16) & 255;
            av = lds0 + (meta & 0xFFFFu); ac = (int)((meta >> 16) & 0x1FFu);
            const int hb = (int)((meta >> 25) & 15u);
            is_static = (meta >> 29) & 1u; mov = (meta >> 30) & 1u; con = (meta >> 31) & 1u;
            const uint32_t *head_w = JT.BP.pool + kBodyPoolHead * hb;
            hv = (const float *)JT.BP.pool + head_w[0];
            hcount = (int)JT.BP.pool[head_w[3] + 1];         // DPiece::count: the record's second dword
            active = true;
        }
        if (!__any(active)) break;
        float d;
        if (active && coop_iterate<SELF>(g, X, is_static, av, ac, hv, hcount, d)) {
            active = false;
            if (head) job_result(S, st, jb, mov, con, d - margins, seen);
        }
    }
}

// env.reset() from the staged start state (prepared by reset_kernel<true> for tag reset_count[e]): obs.py:144-149
__device__ inline void commit_staged(const DScene &S, const DState &st, int e, float *row) {
    const int B = S.B;
    const unsigned tag = st.reset_count[e];
    const size_t sl = (size_t)(tag & 1u);                                    // the slot of this tag
    if (st.staged_tag[sl * B + e] != tag) atomicAdd(st.stale_commits, 1);    // host checks: never expected
    float q[SMR_NJ], v[SMR_NJ], a[SMR_NJ];
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) {
        q[j] = st.next_q[(sl * SMR_NJ + j) * B + e]; v[j] = st.next_v[(sl * SMR_NJ + j) * B + e]; a[j] = st.next_a[(sl * SMR_NJ + j) * B + e];
        st.q[j * B + e] = q[j]; st.v[j * B + e] = v[j]; st.a[j * B + e] = a[j];
        st.range_lo[j * B + e] = st.next_lo[(sl * SMR_NJ + j) * B + e]; st.range_hi[j * B + e] = st.next_hi[(sl * SMR_NJ + j) * B + e];
    }
    int idx[SMR_MAX_PLANETS] = {st.next_planet_idx[(sl * 2) * B + e], st.next_planet_idx[(sl * 2 + 1) * B + e]};
    st.planet_idx[e] = idx[0]; st.planet_idx[B + e] = idx[1];
    st.range_valid[e] = 1;
    st.step_count[e] = 0;
    st.moving_hit[e] = st.next_hit[sl * B + e];
    st.start_attempts[e] = st.next_attempts[sl * B + e];
    st.reset_count[e] = tag + 1u;
    if (row) write_observation(S, q, v, a, idx, row);
}

__global__ void commit_kernel(const DScene *__restrict__ Sp, DState st, const int *__restrict__ list, const int *__restrict__ count_ptr,
                              int count_fixed, float *__restrict__ obs) {
    const DScene &S = *Sp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int count = count_ptr ? *count_ptr : count_fixed;
    if (i >= count) return;
    const int e = list ? list[i] : i;
    commit_staged(S, st, e, obs ? obs + (size_t)e * S.obs_size : nullptr);
}

// A14: the next target point of env e, installed from the staged one (drawn ahead on the side stream for tag tp_count[e]) with
// the distances and the observation the synchronous target_kernel writes; falls back to that kernel when nothing matching is staged
__device__ inline void install_target(const DScene &S, const DState &st, int e, F3 ee, float *row, bool staged) {
    const int B = S.B;
    const unsigned count = st.tp_count[e];
    const size_t sl = (size_t)(count & 1u);
    if (!staged || st.next_tp_tag[sl * B + e] != count) {
        st.tp_list[atomicAdd(st.tp_list_count, 1)] = e;
        if (staged) { const int ri = atomicAdd(st.tp_refill_count, 1); st.tp_refill_list[ri] = e; st.tp_refill_tags[ri] = count + 2u; }   // drawn now, staged again behind it
        return;
    }
    const float tp[3] = {st.next_tp[(sl * 3) * B + e], st.next_tp[(sl * 3 + 1) * B + e], st.next_tp[(sl * 3 + 2) * B + e]};
    const F3 d = F3{tp[0], tp[1], tp[2]} - ee;
    const float dist = sqrtf(dot(d, d));
    st.tp[e] = tp[0]; st.tp[B + e] = tp[1]; st.tp[2 * B + e] = tp[2]; st.tp[3 * B + e] = dist; st.tp[4 * B + e] = dist;
    st.tp_count[e] = count + 1u;
    write_target_observation(S, tp, ee, row);
    const int ri = atomicAdd(st.tp_refill_count, 1);
    st.tp_refill_list[ri] = e; st.tp_refill_tags[ri] = count + 2u;            // count + 1 sits in the other slot
}

__global__ __launch_bounds__(256) void finish_kernel(const DScene *__restrict__ Sp, DState st, const float *__restrict__ actions,
                                                     float *__restrict__ obs, float *__restrict__ reward, uint8_t *__restrict__ done,
                                                     int8_t *__restrict__ reason, float *__restrict__ info, int staged, int ranges_ahead) {
    const DScene &S = *Sp;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0 && st.refill_zero) *st.refill_zero = 0;                        // the list counters of the NEXT step
    if (e == 0 && st.tp_refill_zero) *st.tp_refill_zero = 0;
    if (e >= S.B) return;
    const int B = S.B;
    const int steps = st.step_count[e] + 1;                                   // base.py:1044
    int hit = st.moving_hit[e];
    int contact = 0;
    const int NB = S.n_bodies;
    for (int k = 0; k < NB; ++k) contact |= (st.res[(size_t)k * B + e] != 0.f) ? 1 : 0;   // any of the N contact samples of body k
    if (S.term_moving && S.n_planets > 0) hit |= contact;                     // ctlp.py:2631-2637
    float d_static = S.static_max, d_self = S.static_max, d_moving = S.moving_max + 0.002f;
    for (int l = 0; l < NB; ++l) {                                            // minimum over bodies, ctlp.py:3263-3278,3294-3372
        d_static = fminf(d_static, st.dist[(size_t)l * B + e]);
        d_moving = fminf(d_moving, st.dist[(size_t)(NB + l) * B + e]);
    }
    if (S.n_self_pairs > 0) {                                                 // the minimum over the env's self-pair jobs; at or beyond the range
        d_self = fminf(d_self, __uint_as_float(st.env_dself[e]));            // the reward reads (rewards.py:119-126) it is "none"
        if (d_self >= S.self_cut) d_self = S.static_max;
    }
    if (S.planets_jobs) {                                                     // pair-job path: the minima over the env's jobs
        d_moving = fminf(d_moving, __uint_as_float(st.env_min[e]));
        d_static = fminf(d_static, __uint_as_float(st.env_dstatic[e]));
    }
    if (S.ball_jobs) d_moving = fminf(d_moving, __uint_as_float(st.env_min[e]));    // ball jobs

    float q[SMR_NJ], v[SMR_NJ], a[SMR_NJ], amax = 0.f, vrel = 0.f, arel = 0.f;
    // A14: the target point is reached when the target link point comes within its radius at one of the updates of the
    // step (ctlp.py:2787-2821; the obstacle client follows the setpoints of the cubic)
    bool tp_reached = false;
    float tp[3] = {0.f, 0.f, 0.f}, tp_last = 0.f, tp_init = 0.f;
    F3 ee_end = {0.f, 0.f, 0.f};
    if (S.use_target) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tp[c] = st.tp[c * B + e];
        tp_last = st.tp[3 * B + e]; tp_init = st.tp[4 * B + e];
        // The literal test looks at every update (24 forward kinematics per env-step: 20 000 of this kernel's 21 000 instructions per
        // wavefront).  The point moves no farther per update than sum_j max|qdot_j| ee_R[j] dt / U (contact sweep, smr_device.h), so an
        // update at distance d rules out the next (d - radius) / that many.  Every lane evaluates the last update (its point is what
        // the reward and the observation read) and walks from the first one over the updates it cannot rule out - two evaluations
        // settle all but the lanes near their target.  Those are then taken one at a time by the whole wavefront: the lane's cubic is
        // handed round and every lane evaluates one of its remaining updates (the same test at the same updates: nothing changes but who
        // computes it).  The kernel is one wavefront per SIMD, so its length is the wavefront with the most rounds: 24 -> 3 + one per such lane.
        const int U = S.updates_per_step;
        float c_q[SMR_NJ], c_v[SMR_NJ], c_a[SMR_NJ], c_a1[SMR_NJ], step_u = 0.f;
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) {
            c_q[j] = st.q[j * B + e]; c_v[j] = st.v[j * B + e]; c_a[j] = st.a[j * B + e]; c_a1[j] = st.a1[j * B + e];
            step_u += cubic_speed_bound(c_v[j], c_a[j], c_a1[j], S.dt) * S.ee_R[j];
        }
        step_u *= S.dt / (float)U;
        const auto point_at = [&](const float (&Q)[SMR_NJ], const float (&V)[SMR_NJ], const float (&A)[SMR_NJ], const float (&A1)[SMR_NJ], int u) {
            const float t = u == U ? S.dt : S.dt * (float)u / (float)U;
            float qu[SMR_NJ];
#pragma unroll
            for (int j = 0; j < SMR_NJ; ++j) qu[j] = interp_pos(Q[j], V[j], A[j], A1[j], t, S.dt);
            return ee_point(S, qu);
        };
        const auto dist_to = [&](const F3 &p, float x, float y, float z) { const F3 d = F3{x, y, z} - p; return sqrtf(dot(d, d)); };
        ee_end = point_at(c_q, c_v, c_a, c_a1, U);
        tp_reached = dist_to(ee_end, tp[0], tp[1], tp[2]) < S.target_radius;
        int u_next = 1;
        bool pending = !tp_reached && U > 1;
        for (int it = 0; it < 2 && __any(pending); ++it) {
            if (pending) {
                const float dist = dist_to(point_at(c_q, c_v, c_a, c_a1, u_next), tp[0], tp[1], tp[2]);
                if (dist < S.target_radius) { tp_reached = true; pending = false; }
                else { u_next += 1 + sweep_skip(dist, S.target_radius, step_u); pending = u_next < U; }
            }
        }
        const unsigned long long here = __ballot(true);
        const int lane = (int)__lane_id(), n_here = __popcll(here), rank = __popcll(here & ((1ull << lane) - 1ull));
        for (unsigned long long m = __ballot(pending); m; m &= m - 1ull) {
            const int L = __ffsll((long long)m) - 1;
            float b_q[SMR_NJ], b_v[SMR_NJ], b_a[SMR_NJ], b_a1[SMR_NJ];
#pragma unroll
            for (int j = 0; j < SMR_NJ; ++j) { b_q[j] = __shfl(c_q[j], L); b_v[j] = __shfl(c_v[j], L); b_a[j] = __shfl(c_a[j], L); b_a1[j] = __shfl(c_a1[j], L); }
            const float bx = __shfl(tp[0], L), by = __shfl(tp[1], L), bz = __shfl(tp[2], L);
            const int u0 = __shfl(u_next, L);
            bool found = false;
            for (int base = u0; base < U && !found; base += n_here) {
                const int u = base + rank;
                const bool mine = u < U && dist_to(point_at(b_q, b_v, b_a, b_a1, u), bx, by, bz) < S.target_radius;
                found = __any(mine);
            }
            if (lane == L && found) tp_reached = true;
        }
    }
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) {                                        // base.py:1179-1181
        float q0 = st.q[j * B + e], v0 = st.v[j * B + e], a0 = st.a[j * B + e], a1 = st.a1[j * B + e];
        q[j] = interp_pos(q0, v0, a0, a1, S.dt, S.dt);
        v[j] = interp_vel(v0, a0, a1, S.dt, S.dt);
        a[j] = a1;
        amax = fmaxf(amax, fabsf(actions[(size_t)e * SMR_NJ + j]));
        vrel = fmaxf(vrel, fabsf(v[j] / S.vel_max[j]));
        arel = fmaxf(arel, fabsf(a[j] / S.acc_max[j]));
    }
    int idx[SMR_MAX_PLANETS] = {0, 0};
#pragma unroll
    for (int p = 0; p < SMR_MAX_PLANETS; ++p)
        if (p < S.n_planets) idx[p] = (st.planet_idx[p * B + e] + S.updates_per_step) % S.planet[p].n_orbit;   // ctlp.py:4504

    // _end_min_distance: the minimum distance (static obstacles, self-collision partners) at the end of the executed step,
    // rewards.py:322-332 (actions.py:340-369: the first entry of the valid braking trajectory's list, ctlp.py:3099-3102,3194)
    const float end_min_distance = fminf(d_static, d_self);
    // A12: rewards.py:95-162
    bool col_static = false, col_self = false, col_moving = false;
    if (d_static < S.collision_dist) { d_static = 0.f; col_static = true; }
    if (d_self < S.collision_dist) { d_self = 0.f; col_self = true; }
    if (hit) d_moving = 0.f;                                                  // ctlp.py:3229-3234
    if (S.n_planets == 0 && !S.use_ball) d_moving = S.moving_max + 0.002f;
    // ball bookkeeping (ctlp.py:2823-2861, 3217-3228): hit robot -> distance 0 and a new ball; passed the robot or
    // reached table/ground -> a new ball; otherwise 24 more updates
    BallState ball;
    int ball_k = 0, ball_active = 0, ball_event = 0;                           // 1 hit_robot, 2 hit_obstacle, 3 missed_robot
    bool ball_respawn = false;
    if (S.use_ball) {
        ball = load_ball(st, B, e);
        ball_k = st.ball_i[e];
        ball_active = st.ball_i[3 * B + e];
        if (ball_active) {
            const int kmax = st.ball_i[B + e];
            const int u_end = ball_retire_update(S, ball_k, kmax, st.ball_i[2 * B + e]);
            if (contact) { d_moving = 0.f; ball_active = 0; ball_event = 1; }
            else if (u_end <= S.updates_per_step) { d_moving = S.moving_max + 0.002f; ball_active = 0; ball_event = ball_k + u_end > kmax ? 3 : 2; }   // ctlp.py:2839-2848
            else ball_k += S.updates_per_step;
        } else {
            d_moving = S.moving_max + 0.002f;
        }
        ball_respawn = !ball_active;
    }
    if (d_moving < S.collision_dist) { d_moving = 0.f; col_moving = true; }
    float r_self = 0.f, r_static = 0.f, r_moving;
    if (S.self_max_reward != 0.f) { float r = fminf(1.f, d_self / S.self_max_dist); r_self = r * r; }
    if (S.static_max_reward != 0.f) { float r = fminf(1.f, d_static / S.static_max_dist); r_static = r * r; }
    { float r = fminf(1.f, d_moving / S.moving_max_dist); r_moving = r * r; }

    // A13: rewards.py:432-502
    float action_pun = 1.f;
    if (S.punish_action) {
        float p = (amax - S.action_thr) / (1.f - S.action_thr);
        p = fmaxf(fminf(p, 1.f), 0.f);
        action_pun = p * p;
    }
    float low_acc = 0.f, low_vel = 0.f;
    if (S.low_acc_reward != 0.f) { float r = fminf(1.f, arel / S.low_acc_thr); low_acc = (r - 1.f) * (r - 1.f); }
    if (S.low_vel_reward != 0.f) { float r = fminf(1.f, vrel / S.low_vel_thr); low_vel = (r - 1.f) * (r - 1.f); }
    const bool term_col = (S.term_self && col_self) || (S.term_static && col_static) || (S.term_moving && col_moving);
    const bool finished = steps >= S.episode_steps;                            // tm.py:187-192
    const float bonus = (finished && !term_col) ? S.term_bonus : 0.f;
    const float pun = term_col ? S.early_term_pun : 0.f;
    // A14: get_target_point_reward, ctlp.py:2309-2350
    float target_reward = 0.f;
    if (S.use_target) {
        float norm = 1.f;
        if (S.target_normalize) { norm = tp_init; if (norm == 0.f) norm += 0.0000001f; }
        if (tp_reached) target_reward = tp_last / (S.dt * norm) + S.target_bonus;
        else {
            const F3 dd = F3{tp[0], tp[1], tp[2]} - ee_end;
            const float cur = sqrtf(dot(dd, dd));
            target_reward = (tp_last - cur) / (S.dt * norm);
            st.tp[3 * B + e] = cur;                                           // the next observation, ctlp.py:2241-2243
        }
    }
    float end_min_pun = 0.f, bt_pun = 0.f;                                    // rewards.py:18-21,322-332; ctlp.py:2393-2415
    const int braked = S.check_braking ? st.braked[e] : 0;
    if (S.punish_end_min) {
        const float x = (S.end_min_thr - end_min_distance) / (S.end_min_thr - S.brake_safety);
        end_min_pun = fmaxf(fminf(x, 1.f), 0.f); end_min_pun *= end_min_pun;
    }
    if (S.punish_bt_min && S.check_braking) {
        const float bt = __uint_as_float(st.bt_min[e]);
        if (bt < S.brake_safety) bt_pun = 1.f;                                // (a hit leaves its distance below the safety distance in bt_min; a braking timeout is punished by
                                                                              //  the minimum over the samples walked like any other trajectory, ctlp.py:2393-2415,3185-3203)
        else if (bt < S.bt_thr) { const float x = fmaxf(fminf((S.bt_thr - bt) / (S.bt_thr - S.brake_safety), 1.f), 0.f); bt_pun = x * x; }
    }
    if (S.ca_mode)                                                            // CollisionAvoidanceReward, rewards.py:432-502
        reward[e] = (1.f - action_pun) * S.action_max_pun + r_self * S.self_max_reward + r_static * S.static_max_reward +
                    r_moving * S.moving_max_reward + low_acc * S.low_acc_reward + low_vel * S.low_vel_reward + bonus + pun;
    else                                                                      // TargetPointReachingReward, rewards.py:303-396
        reward[e] = target_reward * S.target_factor - (S.punish_action ? action_pun : 0.f) * S.action_max_pun -
                    ((S.punish_adaptation && braked) ? S.adaptation_max_pun : 0.f) - end_min_pun * S.end_min_max_pun +
                    r_self * S.self_max_reward + r_static * S.static_max_reward + r_moving * S.moving_max_reward - bt_pun * S.bt_max_pun;

    // A16: base.py:1775-1799
    int rs = SMR_TERMINATION_UNSET;
    bool d = false;
    if (S.term_self && col_self) { d = true; rs = SMR_TERMINATION_SELF_COLLISION; }
    if (!d && S.term_static && col_static) { d = true; rs = SMR_TERMINATION_COLLISION_WITH_STATIC_OBSTACLE; }
    if (!d && S.term_moving && col_moving) { d = true; rs = SMR_TERMINATION_COLLISION_WITH_MOVING_OBSTACLE; }
    if (!d && finished) { d = true; rs = SMR_TERMINATION_TRAJECTORY_LENGTH; }
    done[e] = d ? 1 : 0;
    reason[e] = (int8_t)rs;
    // range_kernel saw the episode go on; a staged start state brings its own range below.  (The store stays up here: placed behind the
    // ball branch's stores, in front of the episode counter's atomic, the kernel faulted on address 0 in ball scenes - ROCm 7.2 build, the
    // same store here or at the top of the kernel is fine; not understood, scripts/study/repro_ball.py reproduces it.)
    if (d || !ranges_ahead) st.range_valid[e] = 0;
    if (info) {
        float *row = info + (size_t)e * SMR_INFO_N;
        row[SMR_INFO_D_STATIC] = d_static; row[SMR_INFO_D_SELF] = d_self; row[SMR_INFO_D_MOVING] = d_moving;
        row[SMR_INFO_ACTION_PUNISHMENT] = action_pun; row[SMR_INFO_R_SELF] = r_self; row[SMR_INFO_R_STATIC] = r_static;
        row[SMR_INFO_R_MOVING] = r_moving; row[SMR_INFO_EPISODE_LENGTH] = (float)steps;
        row[SMR_INFO_BRAKED] = S.check_braking ? (float)st.braked[e] : 0.f; row[SMR_INFO_TARGET_REACHED] = tp_reached ? 1.f : 0.f;
        row[SMR_INFO_BALL_EVENT] = (float)ball_event; row[SMR_INFO_RISKY] = st.risky ? (float)st.risky[e] : 0.f;
        row[SMR_INFO_END_MIN_DISTANCE_PUNISHMENT] = end_min_pun; row[SMR_INFO_BRAKING_TRAJECTORY_PUNISHMENT] = bt_pun;
        row[SMR_INFO_START_ATTEMPTS] = (float)st.start_attempts[e];
    }

#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) { st.q[j * B + e] = q[j]; st.v[j * B + e] = v[j]; st.a[j * B + e] = a[j]; }
#pragma unroll
    for (int p = 0; p < SMR_MAX_PLANETS; ++p)
        if (p < S.n_planets) st.planet_idx[p * B + e] = idx[p];
    st.step_count[e] = steps;
    st.moving_hit[e] = hit;

    if (S.use_ball) { st.ball_i[e] = ball_k; st.ball_i[3 * B + e] = ball_active; }
    if (d && S.auto_reset) atomicAdd(st.episodes, 1ull);                       // (one atomic per wavefront after the compiler's aggregation)
    if (d && S.auto_reset && staged) {
        const unsigned tag = st.reset_count[e];
        commit_staged(S, st, e, obs + (size_t)e * S.obs_size);   // next episode's start state was prepared in the background
        const int ri = atomicAdd(st.refill_count, 1);            // hipcc folds this into one atomic per wave
        st.refill_list[ri] = e; st.refill_tags[ri] = tag + 2u;   // tag + 1 sits in the other slot already
        if (S.use_ball) {                                                         // its ball: spawn_kernel, as a reset
            const int si = atomicAdd(st.spawn_count, 1);
            st.spawn_list[si] = -(e + 1); st.spawn_tags[si] = (unsigned)st.ball_i[4 * B + e] + 2u;
        }
        if (S.use_target) {                                                       // its target point, seen from the new start pose
            float qn[SMR_NJ];
#pragma unroll
            for (int j = 0; j < SMR_NJ; ++j) qn[j] = st.q[j * B + e];
            install_target(S, st, e, ee_point(S, qn), obs + (size_t)e * S.obs_size, st.next_tp != nullptr);
        }
    } else if (d && S.auto_reset) {
        int slot = atomicAdd(st.done_count, 1);
        st.done_list[slot] = e;
    } else {
        float *row = obs + (size_t)e * S.obs_size;
        write_observation(S, q, v, a, idx, row);
        if (S.use_target) {
            if (tp_reached) install_target(S, st, e, ee_end, row, staged && st.next_tp != nullptr);   // a new one, ctlp.py:2217-2232
            else write_target_observation(S, tp, ee_end, row);
        }
        if (S.use_ball) {
            if (ball_respawn) {                                                   // ball columns written by spawn_kernel
                const int si = atomicAdd(st.spawn_count, 1);
                st.spawn_list[si] = e; st.spawn_tags[si] = (unsigned)st.ball_i[4 * B + e] + 2u;
            }
            else write_ball_observation(S, ball, ball_k, row + 3 * SMR_NJ + S.obs_target_size);
        }
    }
}

// ---- reset / spawn helpers: one rejection-sampling job is worked by a GROUP of 8 adjacent lanes, lane g = iiwa_link_g
// for every geometric test (its own FK chain, its own link hull, one GJK), lanes 0..6 = joint g for the kinematics.
// Candidates are tried one after the other exactly as the reference does (about 9 of 10 start positions and 19 of
// 20 planet phases are accepted at once), so the critical path of a candidate is one link's test, not eight.
struct Group {
    int g, gbase;
    __device__ __forceinline__ unsigned ballot(bool pred) const { return (unsigned)((__ballot(pred) >> gbase) & 0xFFull); }
    __device__ __forceinline__ bool all(bool pred) const { return ballot(pred) == 0xFFu; }
    __device__ __forceinline__ bool any(bool pred) const { return ballot(pred) != 0u; }
    // lane j of the group holds x_j: gather the 7 joint values
    __device__ __forceinline__ void gather(float x, float q[SMR_NJ]) const {
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) q[j] = __shfl(x, gbase + j);
    }
    // ... the joint angles: every lane gets the sines and cosines of all 7 (each computed once, by the joint's lane)
    __device__ __forceinline__ Trig trig(float x) const {
        float s, c;
        sincosf(x, &s, &c);
        Trig P;
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) { P.sn[j] = __shfl(s, gbase + j); P.cs[j] = __shfl(c, gbase + j); }
#pragma unroll
        for (int k = 0; k < 3; ++k) P.q_last[k] = __shfl(x, gbase + 4 + k);
        return P;
    }
};

// Lane g of a group works the bodies g, g + 8, ... (8 chain links; the ball machine's pieces fold onto lanes 0, 1).
// The lanes' bodies differ, so their hulls are climbed in the block's LDS copy of all bodies' tables (BodyPool).
constexpr int kSideBlock = 256;

// the block's copy of DScene::body_pool; false (for the whole block) when none of its groups has an entry of the list to work
__device__ inline bool stage_body_pool(const DScene &S, uint32_t *pool, int count, int group_stride) {
    const int grp0 = (int)((blockIdx.x * blockDim.x) >> 3);
    if ((grp0 + group_stride - 1) / group_stride >= count) return false;
    const SMR_GLOBAL uint32_t *g = (const SMR_GLOBAL uint32_t *)S.body_pool;
    for (int i = threadIdx.x; i < S.body_pool_dwords; i += kSideBlock) pool[i] = g[i];
    __syncthreads();
    return true;
}

// table clearance of ONE body (ctlp.py:2140-2208)
__device__ inline bool body_clear(const DScene &S, const BodyPool &BP, int b, const Xf &T, float thr) {
    const DPiece LP = BP.piece(b);
    const LocalHull hull = BP.hull(b);
    bool ok = true;
    if ((S.static_mask >> b) & 1u)
        for (int s = 0; s < S.n_static; ++s) {
            float d = piece_distance<true>(S, S.static_piece[s], T, LP, hull, T, thr + 0.005f, 0.f);
            if (d <= thr + 0.005f && d < thr) ok = false;
        }
    return ok;
}

// the self-collision pairs dealt to the lanes of a group by partner link: lane g tests the pairs whose partner is link g, with its own
// frame (DScene::pair_lane; ctlp.py:2170-2208, 1605-1619)
__device__ inline bool lane_self_pairs_clear(const DScene &S, const BodyPool &BP, const Group &G, const Trig &P, float thr_self) {
    bool ok = true;
    const Xf Tg = body_frame_sc(S, P, G.g);              // this lane's link (the partner of its pairs)
    for (int k = 0; k < S.pair_lane_max; ++k) {
        const int i = S.pair_lane[G.g][k];
        if (i < 0) continue;
        const int a = S.self_pair[i][0];
        float d = self_pair_distance<true>(S, i, nullptr, BP.piece(a), BP.hull(a), body_frame_sc(S, P, a), thr_self + 0.005f, 0.f, &BP, &P, nullptr, nullptr,
                                           S.self_pair[i][1] == G.g ? &Tg : nullptr);
        if (d <= thr_self + 0.005f && d < thr_self) ok = false;
    }
    return ok;
}

// table and self-collision clearance of a configuration, cheapest test first: the bodies against the table (mostly settled by the
// capsule / lowest-point shortcuts of piece_distance), then - only if all are clear - the self-collision pairs
// EXTRA: the scene has bodies beyond the chain and self-collision pairs (ball machine); false compiles that code out
template <bool EXTRA>
__device__ inline bool group_table_free(const DScene &S, const BodyPool &BP, const Group &G, const Trig &P, float thr, float thr_self) {
    bool ok = true;
#ifdef SMR_PROFILE_ROWS
    const unsigned long long c0 = SMR_DEV_CLOCK();
#endif
    for (int b = G.g; b < (EXTRA ? S.n_bodies : SMR_NLINK); b += 8) ok = ok && body_clear(S, BP, b, body_frame_sc(S, P, b), thr);
#ifdef SMR_PROFILE_ROWS
    const unsigned long long c1 = SMR_DEV_CLOCK();
    SMR_DEV_COUNT(22, c1 - c0);
#endif
    if (!G.all(ok)) return false;
    if (EXTRA) {
        const bool r = G.all(lane_self_pairs_clear(S, BP, G, P, thr_self));
#ifdef SMR_PROFILE_ROWS
        SMR_DEV_COUNT(23, SMR_DEV_CLOCK() - c1);
#endif
        return r;
    }
    return true;
}

// A17 part 1: _get_collision_free_robot_position, ctlp.py:1980-2138, for one candidate.  box / thr: cartesian
// range of the end-effector point and minimum table / self distances (start states ctlp.py:172-183,1468-1477; ball
// targets ctlp.py:191-229,1725-1726,2357).  (The one-physics-step gravity-torque check of ctlp.py:2091-2111 is
// outside the kinematic path.)  The box first (ctlp.py:2051-2061): one pass down the chain, the same on every lane - about half
// of the uniformly drawn candidates of a target point end there, before any distance query.
template <bool EXTRA>
__device__ inline bool group_pose_valid(const DScene &S, const BodyPool &BP, const Group &G, const Trig &P, const float box[3][2], float thr,
                                        float thr_self) {
    SMR_DEV_COUNT(24, 1);
    const F3 ee = apply(body_frame_sc(S, P, S.ee_body), F3{S.ee_offset[0], S.ee_offset[1], S.ee_offset[2]});
    if (ee.x < box[0][0] || ee.x > box[0][1] || ee.y < box[1][0] || ee.y > box[1][1] || ee.z < box[2][0] || ee.z > box[2][1]) return false;
    return group_table_free<EXTRA>(S, BP, G, P, thr, thr_self);
}

// Bullet contact between any planet piece and any robot body (Planet.check_if_object_is_colliding, ctlp.py:4570-4579)
template <bool EXTRA>
__device__ inline bool group_planet_contact(const DScene &S, const BodyPool &BP, const Group &G, const Trig &P, const int idx[SMR_MAX_PLANETS]) {
    bool contact = false;
    for (int b = G.g; b < (EXTRA ? S.n_bodies : SMR_NLINK); b += 8) {
        const Sample r = collide_link(S, b, BP.piece(b), BP.hull(b), body_frame_sc(S, P, b), idx, nullptr, false, false, true);
        contact = contact || r.contact;
    }
    return G.any(contact);
}

// is the ball (pose XB) within Bullet's contact distance of any robot body? (Ball.check_if_object_is_colliding, ctlp.py:4186-4194)
template <bool EXTRA>
__device__ inline bool group_ball_contact(const DScene &S, const BodyPool &BP, const Group &G, const Trig &P, const Xf &XB) {
    bool contact = false;
    for (int b = G.g; b < (EXTRA ? S.n_bodies : SMR_NLINK); b += 8)
        if ((S.contact_mask >> b) & 1u) {
            float d = piece_distance<false>(S, S.ball_piece, XB, BP.piece(b), BP.hull(b), body_frame_sc(S, P, b), S.contact_dist, S.contact_dist);
            contact = contact || d <= S.contact_dist;
        }
    return G.any(contact);
}

// ---- braking-trajectory safety layer: btg.py:44-93, ctlp.py:3026-3207,3468-3506 ----
// _compute_braking_acceleration (ctlp.py:3494-3506) for the joint of this lane: braking acceleration, clipped into the safe range where
// that changes it; the lanes of the wavefront share the root finding of the clipped joints (all must call together)
__device__ inline float braking_acceleration_lane(const DScene &S, const JointLim &L, int j, float p, float v, float a, bool active) {
    float b = braking_acceleration_joint(S, j, v, a);
    const bool want = active && braking_clip_needed(L, p, v, a, b);
    if (__any(want)) {
        float lo, hi;
        int code;
        safe_acc_range_joint<64>(L, p, v, a, lo, hi, code, want);
        if (want) b = fminf(fmaxf(b, lo), hi);                                            // btg.py:80-93
    }
    return b;
}

// The braking trajectory behind the planned step, once per env (ctlp.py:3155-3207 without the distance checks): the braking
// accelerations of its segments until the robot stands, or the braking timeout.  [0]: the braking acceleration at the current state.
// A group of 8 lanes per env, lane g < 7 = joint g.
__global__ __launch_bounds__(256) void brake_plan_kernel(const DScene *__restrict__ Sp, DState st) {
    const DScene &S = *Sp;
    const int B = S.B;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e0 = t >> 3, g = t & 7;
    const bool valid = e0 < B;
    const int e = valid ? e0 : B - 1;
    const Group G{g, (int)(threadIdx.x & 63) & ~7};
    const bool joint_lane = g < SMR_NJ;
    const int j = joint_lane ? g : 0;
    const JointLim L = joint_lim(S, j);
    float p = st.q[j * B + e], v = st.v[j * B + e], ap = st.a[j * B + e], an = st.a1[j * B + e];
    bool stopped = G.all(!joint_lane || (fabsf(v) < S.brake_stop_v && fabsf(ap) < S.brake_stop_a));
    float b = braking_acceleration_lane(S, L, j, p, v, ap, valid && joint_lane && !stopped);
    if (valid && joint_lane) st.brake_acc[(size_t)j * B + e] = stopped ? 0.f : b;
    bool active = valid;
    int steps = -1;
    for (int m = 0; m < S.brake_timeout_steps && __any(active); ++m) {                   // the state after segment m, the braking acceleration behind it
        const float p1 = interp_pos(p, v, ap, an, S.dt, S.dt), v1 = interp_vel(v, ap, an, S.dt, S.dt);
        p = p1; v = v1; ap = an;
        stopped = G.all(!joint_lane || (fabsf(v) < S.brake_stop_v && fabsf(ap) < S.brake_stop_a));
        if (active && stopped) { steps = m; active = false; }                             // robot stopped (btg.py:45-47)
        b = braking_acceleration_lane(S, L, j, p, v, ap, active && joint_lane);
        an = b;
        if (active && joint_lane) st.brake_acc[((size_t)(m + 1) * SMR_NJ + j) * B + e] = b;
    }
    if (valid && g == 7) {
        st.brake_steps[e] = steps;
        // no standstill within the braking timeout counts as a collision whatever is observed (ctlp.py:3201-3203): latched here, not in
        // brake_check_kernel, whose rows of unobserved bodies leave before they look at anything (plan_kernel cleared the flag)
        if (steps < 0) st.braked[e] = 1;
    }
}

// check_braking_trajectory_method (ctlp.py:3026-3053, _check_if_braking_trajectory_is_collision_free :3155-3207) on the grid of the
// collide kernel: one lane per (env, body) and per (env, self-collision pair), the body's hull staged in LDS.  The lane walks the
// planned step and the braking segments behind it and tests ITS body / pair at the samples: closer than the safety distance
// (get_minimum_distance with check_safety_distance, ctlp.py:3282-3374) -> braked[env] = 1.  A sample is only evaluated when the joints
// have moved far enough since the last evaluation to use up the clearance found there (every value piece_distance returns is a
// lower bound of the distance; joint travel x lever arms bounds the motion of the body's points).
#ifndef SMR_BRAKE_WAVES
#define SMR_BRAKE_WAVES 3        // wavefronts per SIMD (512 / SMR_BRAKE_WAVES registers each).  With the self pairs' bound tables and the
#endif                           // trajectory minimum the kernel spilled 169 registers at 128: 4 -> 3: 1.86 -> 1.72 ms; 2: 2.09 ms (measured)
// BT: punish_braking_trajectory_min_distance - the smallest distance of the walk is tracked as well (an instantiation of its own: the
// extra state cost the plain check a quarter of its speed in spilled registers)
template <bool SELF, bool BT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMR_BRAKE_WAVES, SMR_BRAKE_WAVES))) void brake_check_kernel(const DScene *__restrict__ Sp, DState st) {
    const DScene &S = *Sp;
    const int kBlock = (int)blockDim.x;           // 64 .. 256 lanes (smr_env::brake_block)
    // (Dealing the block ids out so that all rows of an env block run on ONE XCD, one after the other - they all read its 22 x 7
    // braking accelerations per env, and the kernel moves 3.8 GB through the L2s per launch - was slower, 1.72 -> 2.20 ms: the
    // heavy-rows-first order of the plain grid is worth more than the locality.)
    const int env_block = (int)blockIdx.x, row = (int)blockIdx.y;
    SMR_BRAKE_ROW_CLOCK(row);
    const int n_pair_blocks = SELF ? S.n_self_pairs : 0;
    const bool pair_block = SELF && row < n_pair_blocks;
    const int pair = pair_block ? S.pair_order[row] : 0;                                   // the pairs with the nearest partners (longest blocks) first
    const int body = pair_block ? S.self_pair[pair][0] : S.body_order[row - n_pair_blocks];
    const int partner = pair_block ? S.self_pair[pair][1] : -1;
    const int l = pair_block ? (partner >= 0 ? partner : body) : body;                    // the body whose hull is staged (see collide_kernel)
    if (!pair_block && !((S.static_mask >> body) & 1u)) return;                            // nothing of this body is observed by a static obstacle
    __shared__ __attribute__((aligned(16))) float lds_v[4 * kMaxPieceVerts];
    __shared__ uint32_t lds_ids[kMaxPieceIdGroups];
    __shared__ uint32_t lds_start[kStartCells / 4];
    const int pl = S.link_piece[l];
    const DPiece LP = load_piece(S.pieces, pl);
    {
        const SMR_GLOBAL float4 *gv = (const SMR_GLOBAL float4 *)S.verts + LP.begin;
        const SMR_GLOBAL uint32_t *gi = (const SMR_GLOBAL uint32_t *)S.ids4 + LP.ids_begin,
                                  *gm = (const SMR_GLOBAL uint32_t *)S.start + (kStartCells / 4) * (size_t)pl;
        for (int i = threadIdx.x; i < LP.count; i += kBlock) {
            const SMR_GLOBAL float *f = (const SMR_GLOBAL float *)(gv + i);
            lds_v[4 * i] = f[0]; lds_v[4 * i + 1] = f[1]; lds_v[4 * i + 2] = f[2]; lds_v[4 * i + 3] = f[3];
        }
        for (int i = threadIdx.x; i < LP.ids_count; i += kBlock) lds_ids[i] = gi[i];
        for (int i = threadIdx.x; i < kStartCells / 4; i += kBlock) lds_start[i] = gm[i];
    }
    __syncthreads();
    const LocalHull hull{lds_v, lds_ids, (const uint8_t *)lds_start};
    const int B = S.B;
    const int e = env_block * (int)blockDim.x + (int)threadIdx.x;
    if (e >= B) return;
    // braking timeout: brake_plan_kernel latched braked[e] already (ctlp.py:3201-3203).  The decision is made; what is left to know is
    // _braking_trajectory_minimum_distance, the minimum over the samples the reference walked until it noticed (ctlp.py:3185-3203): the
    // rows that keep it (BT) walk all brake_timeout_steps + 1 segments, the others leave
    const int steps_planned = st.brake_steps[e];
    if (steps_planned < 0 && !BT) return;
    const int steps = steps_planned < 0 ? S.brake_timeout_steps : steps_planned;
#ifdef SMR_PROFILE_ROWS
    { const unsigned long long am = __ballot(true); if ((int)__lane_id() == __ffsll((long long)am) - 1) { atomicAdd(&g_job_hist[51], 1ull); atomicAdd(&g_job_hist[52], (unsigned long long)__popcll(am)); } atomicAdd(&g_job_hist[53], (unsigned long long)(steps + 1)); }
#endif
    // the segment's cubic per joint in Horner form, q(t) = p + t (v + t (ha + t j6)) with ha = a0 / 2, j6 = (a1 - a0) / (6 dt): three
    // multiply-adds per joint and sample (fewer instructions and 8 fewer spilled registers than the monomial form; the kernel's time did not move)
    const float inv_dt6 = 1.0f / (6.0f * S.dt);
    float p[SMR_NJ], v[SMR_NJ], ha[SMR_NJ], j6[SMR_NJ], q_ref[SMR_NJ];
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) {
        const float a0 = st.a[j * B + e], a1 = st.a1[j * B + e];
        p[j] = st.q[j * B + e]; v[j] = st.v[j * B + e]; ha[j] = 0.5f * a0; j6[j] = (a1 - a0) * inv_dt6; q_ref[j] = p[j];
    }
    // levers of the joints on the points of what this lane tests.  A pair of bodies on the chain: the joints below the partner link
    // carry both along, only the joints from the partner link on move the body relative to it.
    float lever[SMR_NJ];
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) lever[j] = (partner >= 0 && j < partner) ? 0.f : S.sweep_R[body][j];
    const float thr = S.brake_safety;
    // with punish_braking_trajectory_min_distance the smallest distance of the walk is asked for below bt_thr as well: a sample is
    // only skipped when its distance is proven beyond that (get_braking_trajectory_punishment, ctlp.py:2393-2415)
    const float thr_skip = BT ? fmaxf(thr, S.bt_thr) : thr;
    float dmin = 3.0e38f;
    float margin = -1.f;                                 // clearance beyond the safety distance found at q_ref (< 0: nothing evaluated yet)
    // consecutive evaluations look at nearly the same configuration: each GJK run starts in the direction the last one ended with
    // (world frame; the first support points then already span the closest feature)
    F3 warm = {0.f, 0.f, 0.f};
    for (int m = 0; m <= steps; ++m) {
        for (int sidx = 1; sidx <= S.brake_checks; ++sidx) {                              // linspace(dt / Nb, dt, Nb), ctlp.py:3167-3169
            const float t = sidx == S.brake_checks ? S.dt : S.dt * (float)sidx / (float)S.brake_checks;
            float qs[SMR_NJ], moved = 0.f;
#pragma unroll
            for (int j = 0; j < SMR_NJ; ++j) { qs[j] = fmaf(t, fmaf(t, fmaf(t, j6[j], ha[j]), v[j]), p[j]); moved += fabsf(qs[j] - q_ref[j]) * lever[j]; }
            // (Letting the whole wavefront evaluate whenever one lane has to - the lanes then fall into step - was measured: 1.57 -> 1.60 ms.
            // With no evaluation at all (measurement build, SMR_BRAKE_WALK_ONLY) the braking kernels take 0.67 of their 1.56 ms: two fifths
            // of the time is this walk, repeated in each of the 24 rows.  Walking only the joints that move what the row tests (the
            // others interpolated when a sample is evaluated) was measured too: 1.57 -> 1.77 ms, 116 -> 132 spilled registers.)
#ifdef SMR_PROFILE_ROWS      // (scripts/study/brake_hist.py) wave-level sample steps | evaluation passes and evaluating lanes per row
            { const unsigned long long am = __ballot(true); if ((int)__lane_id() == __ffsll((long long)am) - 1) atomicAdd(&g_job_hist[50], 1ull); }
#endif
            if (moved * 1.0001f + 1e-6f < margin) continue;
#ifdef SMR_PROFILE_ROWS
            { const unsigned long long am = __ballot(true); if ((int)__lane_id() == __ffsll((long long)am) - 1 && row < 24) { atomicAdd(&g_job_hist[2 * row], 1ull); atomicAdd(&g_job_hist[2 * row + 1], (unsigned long long)__popcll(am)); } }
#endif
            float d = -kNoBound;
            // a pair with a bound table (partner iiwa_link_4 / _5: the rows with the smallest clearances, evaluated at nearly every
            // sample): two bytes indexed by the last joints' angles BEFORE any frame - an evaluation is a chain of seven dependent
            // joint transforms whatever it ends in, the table read is not.  A bound that clears the threshold is as good as the
            // distance here: the walk only asks "closer than thr?" and how far the joints may travel before it has to ask again.
            if (pair_block && partner >= 0) d = self_tab_bound(S, pair, qs[4], qs[5], qs[6]);
#ifdef SMR_BRAKE_WALK_ONLY                                 // measurement build: what the walk alone costs (no evaluation ever)
            d = 1.0e30f;
#endif
            if (d > thr_skip + 2e-5f) {
            } else if (pair_block) {
                const int link_a = body < SMR_NLINK ? body : SMR_NLINK - 1;
                Xf Ta = fk_base(S), Tp = Ta;
#pragma unroll
                for (int j = 0; j < SMR_NJ; ++j) {
                    if (j == partner) Tp = Ta;
                    if (j < link_a) fk_advance(S, j, qs[j], Ta);
                }
                if (partner >= link_a) Tp = body_frame(S, qs, partner);
                if (body >= SMR_NLINK) Ta = mounted_frame(S, body, Ta);
                if (partner >= 0) d = piece_distance<false>(S, S.link_piece[body], Ta, LP, hull, Tp, S.static_max, thr, &warm);
                else d = self_pair_distance(S, pair, qs, LP, hull, Ta, S.static_max, thr, nullptr, nullptr, &warm);
            } else {
                const Xf T = body_frame(S, qs, body);
                d = 3.0e38f;
                for (int sp = 0; sp < S.n_static; ++sp)
                    d = fminf(d, piece_distance<true>(S, S.static_piece[sp], T, LP, hull, T, S.static_max, thr, S.n_static == 1 ? &warm : nullptr));
            }
            if (BT) dmin = fminf(dmin, d);
            if (d < thr) {                                                                // ctlp.py:3307-3312,3364-3368; several lanes of an env may write the same 1
                st.braked[e] = 1;
                if (BT) atomicMin(&st.bt_min[e], dist_bits(dmin));
                return;
            }
            margin = d - thr_skip - 1e-5f;
#pragma unroll
            for (int j = 0; j < SMR_NJ; ++j) q_ref[j] = qs[j];
        }
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) {                                                 // the knot: p(dt), v(dt), a(dt) = a1; the next segment's a1
            const float dt = S.dt, a_end = fmaf(6.0f * dt, j6[j], 2.0f * ha[j]);
            p[j] = fmaf(dt, fmaf(dt, fmaf(dt, j6[j], ha[j]), v[j]), p[j]);
            v[j] = fmaf(dt, fmaf(3.0f * dt, j6[j], 2.0f * ha[j]), v[j]);
            ha[j] = 0.5f * a_end;
            // (asked for a segment ahead of its use: 1.39 -> 1.41 ms, seven more registers to spill)
            j6[j] = m < steps ? (st.brake_acc[((size_t)(m + 1) * SMR_NJ + j) * B + e] - a_end) * inv_dt6 : 0.f;
        }
    }
    if (BT && dmin <= S.static_max) atomicMin(&st.bt_min[e], dist_bits(dmin));
}

// adapt_action (ctlp.py:3055-3153, actions.py:340-361): where the braking trajectory of the planned end acceleration is not collision free
// the braking acceleration replaces it (the stored valid braking trajectory of the reference continues from exactly this state)
__global__ __launch_bounds__(256) void brake_apply_kernel(const DScene *__restrict__ Sp, DState st) {
    const DScene &S = *Sp;
    const int e = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (e >= S.B || !st.braked[e]) return;
    st.a1[(size_t)j * S.B + e] = st.brake_acc[(size_t)j * S.B + e];
}

// end-effector point in float64 (the ball is aimed at it; its launch arithmetic runs in float64 so that the
// integer update counters derived from it agree with the oracle)
__device__ inline void ee_point_f64(const DScene &S, const float q[SMR_NJ], double ee[3]) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {S.base[0], S.base[1], S.base[2]};
    for (int k = 0; k < SMR_NJ; ++k) {
        double o[3] = {S.joint_xyz[k][0], S.joint_xyz[k][1], S.joint_xyz[k][2]};
        for (int i = 0; i < 3; ++i) t[i] += R[3 * i] * o[0] + R[3 * i + 1] * o[1] + R[3 * i + 2] * o[2];
        double ax = S.joint_axis[k][0], ay = S.joint_axis[k][1], az = S.joint_axis[k][2];
        double c = cos((double)q[k]), s = sin((double)q[k]), C = 1 - c;
        double J[9] = {c + ax * ax * C, ax * ay * C - az * s, ax * az * C + ay * s, ay * ax * C + az * s, c + ay * ay * C, ay * az * C - ax * s,
                       az * ax * C - ay * s, az * ay * C + ax * s, c + az * az * C};
        double n[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) n[3 * i + j] = R[3 * i] * J[j] + R[3 * i + 1] * J[3 + j] + R[3 * i + 2] * J[6 + j];
        for (int i = 0; i < 9; ++i) R[i] = n[i];
    }
    double off[3] = {S.ee_offset[0], S.ee_offset[1], S.ee_offset[2]};
    if (S.ee_body >= SMR_NLINK) {          // the target link is mounted on iiwa_link_7 (ball machine): offset in ITS frame
        const int b = S.ee_body;
        double o2[3];
        for (int i = 0; i < 3; ++i)
            o2[i] = S.body_xyz[b][i] + S.body_rot[b][3 * i] * off[0] + S.body_rot[b][3 * i + 1] * off[1] + S.body_rot[b][3 * i + 2] * off[2];
        for (int i = 0; i < 3; ++i) off[i] = o2[i];
    }
    for (int i = 0; i < 3; ++i) ee[i] = t[i] + R[3 * i] * off[0] + R[3 * i + 1] * off[1] + R[3 * i + 2] * off[2];
}

// Ball.get_target_height_time, ctlp.py:4346-4361
__device__ inline double ball_height_time(double h0, double vz, double target, double udt, double *steps) {
    const double g = 9.81;
    double sq = vz * vz + 2 * g * (h0 - target), t = nan("");
    if (steps) *steps = nan("");
    if (sq >= 0) {
        t = (vz + sqrt(sq)) / g;
        if (t > 0) { if (steps) *steps = nearbyint(t / udt); }
        else t = nan("");
    }
    return t;
}

struct BallSpawn { float p0[3], v0[3], e0[3], omega; int kmax, khit; };

// _add_moving_object, ctlp.py:1723-1931, by a group of 8 lanes: candidate target poses one after the other, each
// tested link-parallel; the scalar launch arithmetic is computed redundantly by all lanes, in float64.
// (forced inline: as a function of its own it would not know that the scene sits in constant memory and the hulls in LDS - every
// read of either became a per-lane flat load)
template <bool EXTRA>
__device__ __forceinline__ BallSpawn spawn_ball_group(const DScene &S, const BodyPool &BP, const DState &st, const Rng &rng, const Group &G) {
    const int jj = G.g < SMR_NJ ? G.g : 0;
    const double udt = S.dt64 / S.updates_per_step, v = S.ball_speed, grav = 9.81;
    double rel[3] = {0, 0, 0}, vel[3] = {0, 0, 1};
    unsigned pose_attempt = 0;
    bool launched = false;
    for (unsigned n = 0; n < 25000u; ++n) {
        const double h = rng.uniformd(S_BALL_RELEASE, 3u * n, S.ball_height[0], S.ball_height[1]);       // ctlp.py:1943-1952
        const double r = sqrt(S.ball_sphere_radius * S.ball_sphere_radius - h * h);
        const double ang = rng.uniformd(S_BALL_RELEASE, 3u * n + 1u, S.ball_angle[0], S.ball_angle[1]);
        rel[0] = S.ball_center[0] + r * cos(ang); rel[1] = S.ball_center[1] + r * sin(ang); rel[2] = S.ball_center[2] + h;
        double target[3] = {0, 0, 0};
        bool have_target = false;
        for (int tries = 0; tries < 25000; ++tries) {                                                       // ctlp.py:1782-1790
            float q[SMR_NJ];
            const float qme = rng.uniform(S_BALL_POSE, 7u * pose_attempt + jj, S.pos_lo[jj], S.pos_hi[jj]);
            G.gather(qme, q);
            ++pose_attempt;
            if (group_pose_valid<EXTRA>(S, BP, G, G.trig(qme), S.ball_target_box, S.ball_target_min_static, S.ball_target_min_self)) { ee_point_f64(S, q, target); have_target = true; break; }
        }
        if (!have_target) { if (G.g == 0) sampling_exhausted(st, kSamplingBallTarget); break; }            // ctlp.py:2127-2136 raises
        const double dx = target[0] - rel[0], dy = target[1] - rel[1], dh = target[2] - rel[2];
        const double dxy = sqrt(dx * dx + dy * dy);
        const double sq = v * v * v * v - grav * (grav * dxy * dxy + 2 * dh * v * v);                      // ctlp.py:1797
        bool valid = sq >= 0;
        if (valid) {
            const bool high = rng.uniform01d(S_BALL_RELEASE, 3u * n + 2u) <= S.ball_high_prob;
            const double theta = atan((v * v + (high ? 1.0 : -1.0) * sqrt(sq)) / (grav * dxy));
            vel[2] = v * sin(theta);
            vel[0] = v * cos(theta) * dx / dxy; vel[1] = v * cos(theta) * dy / dxy;
            if (S.ball_check_invalid) {                                                                     // ctlp.py:1813-1848
                const double tt = ball_height_time(rel[2], vel[2], S.ball_invalid_max[2] + (double)S.ball_radius, udt, nullptr);
                const double tb = ball_height_time(rel[2], vel[2], S.ball_invalid_min[2] + (double)S.ball_radius, udt, nullptr);
                bool in_top = false, in_bot = false;
                if (tt == tt) { double x = rel[0] + vel[0] * tt, y = rel[1] + vel[1] * tt;
                    in_top = S.ball_invalid_min[0] <= x && x <= S.ball_invalid_max[0] && S.ball_invalid_min[1] <= y && y <= S.ball_invalid_max[1]; }
                if (tb == tb) { double x = rel[0] + vel[0] * tb, y = rel[1] + vel[1] * tb;
                    in_bot = S.ball_invalid_min[0] <= x && x <= S.ball_invalid_max[0] && S.ball_invalid_min[1] <= y && y <= S.ball_invalid_max[1]; }
                if (in_top || in_bot) valid = false;
            }
        }
        if (valid) { launched = true; break; }
    }
    if (!launched && G.g == 0) sampling_exhausted(st, kSamplingBallLaunch);
    BallSpawn out;
    for (int c = 0; c < 3; ++c) { out.p0[c] = (float)rel[c]; out.v0[c] = (float)vel[c]; }
    out.omega = (float)rng.uniformd(S_BALL_MISC, 0u, 0.0, 2.0 * 3.14159265358979323846);                   // ctlp.py:4056-4057
    // initial orientation: rotation z -> direction, quaternion of its transpose, euler angles (ctlp.py:4029-4034,5389-5421)
    {
        const double nv = sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
        const double d0 = vel[0] / nv, d1 = vel[1] / nv, d2 = vel[2] / nv;
        const double vx = -d1, vy = d0, c = d2, s2 = vx * vx + vy * vy;
        const double K[3][3] = {{0, 0, vy}, {0, 0, -vx}, {-vy, vx, 0}};
        double m[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double kk = 0;
                for (int k = 0; k < 3; ++k) kk += K[i][k] * K[k][j];
                m[j][i] = (i == j ? 1.0 : 0.0) + K[i][j] + kk * ((1 - c) / s2);       // transposed
            }
        double t, q[4];
        if (m[2][2] < 0) {
            if (m[0][0] > m[1][1]) { t = 1 + m[0][0] - m[1][1] - m[2][2]; q[0] = t; q[1] = m[0][1] + m[1][0]; q[2] = m[2][0] + m[0][2]; q[3] = m[1][2] - m[2][1]; }
            else { t = 1 - m[0][0] + m[1][1] - m[2][2]; q[0] = m[0][1] + m[1][0]; q[1] = t; q[2] = m[1][2] + m[2][1]; q[3] = m[2][0] - m[0][2]; }
        } else {
            if (m[0][0] < -m[1][1]) { t = 1 - m[0][0] - m[1][1] + m[2][2]; q[0] = m[2][0] + m[0][2]; q[1] = m[1][2] + m[2][1]; q[2] = t; q[3] = m[0][1] - m[1][0]; }
            else { t = 1 + m[0][0] + m[1][1] + m[2][2]; q[0] = m[1][2] - m[2][1]; q[1] = m[2][0] - m[0][2]; q[2] = m[0][1] - m[1][0]; q[3] = t; }
        }
        const double sc = 0.5 / sqrt(t);
        const double x = q[0] * sc, y = q[1] * sc, z = q[2] * sc, w = q[3] * sc;
        const double sarg = -2 * (x * z - w * y);                                          // pybullet.getEulerFromQuaternion
        double e0, e1, e2;
        if (sarg <= -0.99999) { e1 = -0.5 * 3.14159265358979323846; e0 = 0; e2 = 2 * atan2(x, -y); }
        else if (sarg >= 0.99999) { e1 = 0.5 * 3.14159265358979323846; e0 = 0; e2 = 2 * atan2(-x, y); }
        else { e0 = atan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z); e1 = asin(sarg); e2 = atan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z); }
        out.e0[0] = (float)e0; out.e0[1] = (float)e1; out.e0[2] = (float)e2;
    }
    // Ball.get_final_ball_position, ctlp.py:4303-4344
    double tf = nan("");
    for (int i = 0; i < 2; ++i) {
        const int a = (i + 1) % 3, b = (i + 2) % 3;
        if (vel[i] == 0) continue;
        for (int j = 0; j < 2; ++j) {
            const double lim = j == 0 ? S.ball_final_min[i] : S.ball_final_max[i];
            const double tt = (lim - rel[i]) / vel[i];
            const double pa = rel[a] + vel[a] * tt, pb = rel[b] + vel[b] * tt;
            if (S.ball_final_min[a] <= pa && pa <= S.ball_final_max[a] && S.ball_final_min[b] <= pb && pb <= S.ball_final_max[b])
                if (!(tf == tf) || tt > tf) tf = tt;
        }
    }
    const double tmin = ball_height_time(rel[2], vel[2], S.ball_final_min[2], udt, nullptr);
    if (tmin == tmin && (!(tf == tf) || tmin < tf)) tf = tmin;
    out.kmax = (tf == tf) ? (int)floor(tf / udt) : 0;                                                      // ctlp.py:4101
    double steps;
    const double tt = ball_height_time(rel[2], vel[2], (double)S.ball_radius, udt, &steps);               // ctlp.py:1889-1917
    bool on_table = false;
    if (S.n_static > 0 && tt == tt) {
        const double x = rel[0] + vel[0] * tt, y = rel[1] + vel[1] * tt;
        on_table = -S.table_half_xy[0] <= x && x <= S.table_half_xy[0] && -S.table_half_xy[1] <= y && y <= S.table_half_xy[1];
    }
    if (!on_table) ball_height_time(rel[2], vel[2], S.plane_z + (double)S.ball_radius, udt, &steps);
    out.khit = (steps == steps) ? (int)steps - 1 : 0x7fffffff;
    return out;
}

__device__ inline void store_ball(const DState &st, int B, int e, const BallSpawn &b, int k, unsigned spawns) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { st.ball[c * B + e] = b.p0[c]; st.ball[(3 + c) * B + e] = b.v0[c]; st.ball[(6 + c) * B + e] = b.e0[c]; }
    st.ball[9 * B + e] = b.omega;
    st.ball_i[e] = k; st.ball_i[B + e] = b.kmax; st.ball_i[2 * B + e] = b.khit; st.ball_i[3 * B + e] = 1; st.ball_i[4 * B + e] = (int)spawns;
}

// a new ball for the envs whose ball was retired in this step (get_moving_object_observation, ctlp.py:2354-2360)
// list entry e >= 0: the ball of env e was retired in flight; -(e + 1): env e was reset (staged robot state committed) and
// gets the ball of its new episode, advanced to a random contact-free time step (ctlp.py:1052-1111).  all_reset: every
// entry is a reset (explicit env.reset()).
// STAGE: only the launch of the env's NEXT ball (tag = its spawn count) is computed, into next_ball*; nothing live is touched.
// Otherwise the ball is installed - from the staged launch when one with the right tag is there, else drawn on the spot.
// tags / tag_offset (STAGE): the spawn count the launch is for - tags[i], or the env's current count + tag_offset
template <bool EXTRA, bool STAGE>
__device__ __forceinline__ void spawn_body(const DScene *__restrict__ Sp, const DState &st, const int *__restrict__ list,
                                           const int *__restrict__ count_ptr, int count_fixed, int all_reset, int group_stride,
                                           float *__restrict__ obs, const unsigned *__restrict__ tags, int tag_offset) {
    const DScene &S = *Sp;
    __shared__ __attribute__((aligned(16))) uint32_t pool[kBodyPoolDwords];
    const int count = count_ptr ? *count_ptr : count_fixed;
    if (!stage_body_pool(S, pool, count, group_stride)) return;
    const BodyPool BP{pool};
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = t >> 3, g = t & 7;
    if (grp % group_stride) return;
    const int i = grp / group_stride;
    if (i >= count) return;
    const int entry = list ? list[i] : i;
    const bool is_reset = all_reset || entry < 0;
    const int e = entry < 0 ? -(entry + 1) : entry;
    const int B = S.B;
    const Group G{g, (int)(threadIdx.x & 63) & ~7};
    const unsigned spawns = STAGE ? (tags ? tags[i] : (unsigned)st.ball_i[4 * B + e] + (unsigned)tag_offset) : (unsigned)st.ball_i[4 * B + e];
    const Rng rng = {S.seed_lo, S.seed_hi, S.env_base + (unsigned)e, spawns};
    const size_t sl = (size_t)(spawns & 1u);
    SMR_PHASE_INIT;
    if (STAGE) {
        const BallSpawn b = spawn_ball_group<EXTRA>(S, BP, st, rng, G);
        SMR_PHASE(4);
        if (g == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { st.next_ball[(sl * 10 + c) * B + e] = b.p0[c]; st.next_ball[(sl * 10 + 3 + c) * B + e] = b.v0[c]; st.next_ball[(sl * 10 + 6 + c) * B + e] = b.e0[c]; }
            st.next_ball[(sl * 10 + 9) * B + e] = b.omega;
            st.next_ball_i[(sl * 2) * B + e] = b.kmax; st.next_ball_i[(sl * 2 + 1) * B + e] = b.khit;
            st.next_ball_tag[sl * B + e] = spawns;
        }
        return;
    }
    BallSpawn b;
    const bool have = st.next_ball != nullptr && st.next_ball_tag[sl * B + e] == spawns;
    if (have) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { b.p0[c] = st.next_ball[(sl * 10 + c) * B + e]; b.v0[c] = st.next_ball[(sl * 10 + 3 + c) * B + e]; b.e0[c] = st.next_ball[(sl * 10 + 6 + c) * B + e]; }
        b.omega = st.next_ball[(sl * 10 + 9) * B + e];
        b.kmax = st.next_ball_i[(sl * 2) * B + e]; b.khit = st.next_ball_i[(sl * 2 + 1) * B + e];
    }
    if (G.any(!have)) {                                  // nothing matching staged for this group: draw it now
        const BallSpawn b2 = spawn_ball_group<EXTRA>(S, BP, st, rng, G);
        if (!have) b = b2;
    }
    BallState bs;
    for (int c = 0; c < 3; ++c) { bs.p0[c] = b.p0[c]; bs.v0[c] = b.v0[c]; bs.e0[c] = b.e0[c]; }
    bs.omega = b.omega;
    int k0 = 0;
    if (is_reset && S.ball_random_initial) {
        const Trig P = G.trig(st.q[(g < SMR_NJ ? g : 0) * B + e]);
        int kcap = b.kmax < b.khit ? b.kmax : b.khit;
        int max_ts = kcap / S.updates_per_step;
        if (max_ts < 0) max_ts = 0;
        bool free_step = false;
        for (unsigned n = 0; n < 100000u && !free_step; ++n) {
            k0 = rng.randint(S_BALL_MISC, 1u + n, max_ts + 1) * S.updates_per_step;
            free_step = !group_ball_contact<EXTRA>(S, BP, G, P, ball_pose(S, bs, k0, nullptr));
        }
        if (!free_step && g == 0) sampling_exhausted(st, kSamplingBallInitialStep);
    }
    if (g == 0) {
        store_ball(st, B, e, b, k0, spawns + 1u);
        if (obs) write_ball_observation(S, bs, k0, obs + (size_t)e * S.obs_size + 3 * SMR_NJ + S.obs_target_size);
    }
}

// Two instantiation families.  Round 2 held the one with the ball machine (EXTRA) to 256 registers like the other reset-side kernels
// (a wavefront with 512 kept the collide kernel off its SIMD: 67 against 65 M env-steps/s then); since the self pairs are jobs it has
// the whole register file again (SMR_SPAWN_WAVES above: 64 -> 73 M).  On the plain robot the float64 launch arithmetic spills 100-200
// registers at 256 and the kernel is short - it keeps the AGPRs as spill space (ball scene 113 -> 120 M).
template <bool STAGE>
__global__ SMR_SPAWN_KERNEL void spawn_kernel_extra(const DScene *__restrict__ Sp, DState st, const int *__restrict__ list, const int *__restrict__ count_ptr,
                                                   int count_fixed, int all_reset, int group_stride, float *__restrict__ obs,
                                                   const unsigned *__restrict__ tags, int tag_offset) {
    spawn_body<true, STAGE>(Sp, st, list, count_ptr, count_fixed, all_reset, group_stride, obs, tags, tag_offset);
}
template <bool STAGE>
__global__ __launch_bounds__(256) void spawn_kernel_plain(const DScene *__restrict__ Sp, DState st, const int *__restrict__ list, const int *__restrict__ count_ptr,
                                                          int count_fixed, int all_reset, int group_stride, float *__restrict__ obs,
                                                          const unsigned *__restrict__ tags, int tag_offset) {
    spawn_body<false, STAGE>(Sp, st, list, count_ptr, count_fixed, all_reset, group_stride, obs, tags, tag_offset);
}

// A14: a new target point for the envs of the list (reached in this step, or reset): _add_target_point, ctlp.py:1658-1721 -
// the target link point of a random valid configuration inside the target box (attempt m = draws 7m..7m+6 of stream
// S_TARGET, tag = target points sampled so far), with the distances the next observation records (ctlp.py:2229-2243).
// STAGED: the point for the env's CURRENT target count goes to next_tp / next_tp_tag and nothing of the live env is touched
// tags / tag_offset (STAGED): the target count the point is for - tags[i], or the env's current count + tag_offset
template <bool EXTRA, bool STAGED>
__global__ SMR_TARGET_KERNEL void target_kernel(const DScene *__restrict__ Sp, DState st, const int *__restrict__ list,
                                                    const int *__restrict__ count_ptr, int count_fixed, int group_stride,
                                                    float *__restrict__ obs, const unsigned *__restrict__ tags, int tag_offset) {
    const DScene &S = *Sp;
    __shared__ __attribute__((aligned(16))) uint32_t pool[kBodyPoolDwords];
    const int n_entries = count_ptr ? *count_ptr : count_fixed;
    if (!stage_body_pool(S, pool, n_entries, group_stride)) return;
    const BodyPool BP{pool};
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = t >> 3, g = t & 7;
    if (grp % group_stride) return;
    const int i = grp / group_stride;
    if (i >= n_entries) return;
    const int e = list ? list[i] : i;
    const int B = S.B;
    const Group G{g, (int)(threadIdx.x & 63) & ~7};
    const int jj = g < SMR_NJ ? g : 0;
    const unsigned count = STAGED ? (tags ? tags[i] : st.tp_count[e] + (unsigned)tag_offset) : st.tp_count[e];
    const Rng rng = {S.seed_lo, S.seed_hi, S.env_base + (unsigned)e, count};
    float q[SMR_NJ];
    bool found = false;
    SMR_PHASE_INIT;
    for (unsigned m = 0; m < 100000u && !found; ++m) {
        const float qme = rng.uniform(S_TARGET, 7u * m + jj, S.pos_lo[jj], S.pos_hi[jj]);
        G.gather(qme, q);
        found = group_pose_valid<EXTRA>(S, BP, G, G.trig(qme), S.ball_target_box, S.target_min_static, S.target_min_self);
    }
    SMR_PHASE(6);
    if (!found && g == 0) sampling_exhausted(st, kSamplingTargetPoint);                                   // ctlp.py:2127-2136 raises
    if (g == 0 && STAGED) {
        const F3 target = ee_point(S, q);
        const size_t sl = (size_t)(count & 1u);
        st.next_tp[(sl * 3) * B + e] = target.x; st.next_tp[(sl * 3 + 1) * B + e] = target.y; st.next_tp[(sl * 3 + 2) * B + e] = target.z;
        st.next_tp_tag[sl * B + e] = count;
    }
    if (g == 0 && !STAGED) {
        const F3 target = ee_point(S, q);
        float qc[SMR_NJ];
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) qc[j] = st.q[j * B + e];
        const F3 ee = ee_point(S, qc);
        const F3 d = target - ee;
        const float dist = sqrtf(dot(d, d));
        st.tp[e] = target.x; st.tp[B + e] = target.y; st.tp[2 * B + e] = target.z; st.tp[3 * B + e] = dist; st.tp[4 * B + e] = dist;
        st.tp_count[e] = count + 1u;
        if (obs) { const float tp[3] = {target.x, target.y, target.z}; write_target_observation(S, tp, ee, obs + (size_t)e * S.obs_size); }
    }
}

// every n-th 8-lane group works (tuning knob SMR_RESET_GROUP_STRIDE: 1, 2, 4 or 8).  Sparse wavefronts end sooner (a wavefront
// runs as long as the group with the most rejected candidates), dense ones are fewer: since every block stages the bodies' hulls
// in LDS and a wavefront holds half a SIMD's registers, fewer wins - measured at 65 536 envs with stride 1 / 2 / 4: space 82.0 /
// 79.3 / 75.6, reaching task 75.9 / 68.6 / 62.7, ball + ball machine 65.7 / 56.9 / 45.7 M env-steps/s.
static int reset_group_stride(bool staged = false) {
    static const int v = [] { const char *e = std::getenv("SMR_RESET_GROUP_STRIDE"); int x = e ? std::atoi(e) : 0; return (x == 1 || x == 2 || x == 4 || x == 8) ? x : 0; }();
    (void)staged;
    return v ? v : 1;
}

// env.reset() for the envs in list[0..count): obs.py:144-149 -> base.py:913-1022.
// One reset is worked by a GROUP of 8 adjacent lanes: they evaluate 8 candidate start positions / 8 candidate
// planet phases at once and take the first valid one (ballot + ffs = the sequential "first valid" of
// ctlp.py:2030-2138,4477-4490), and lanes 0..6 handle one joint each for the velocity/acceleration tries
// and the random walk (ctlp.py:1499-1654).  Rejection sampling is a chain of data-dependent loops: latency,
// not lane count, bounds this kernel, so only every group_stride-th group takes an entry.
// STAGED: the result goes to the staging arrays (next_*, staged_tag) and nothing of the live env is touched -
// the kernel then runs on the side stream while the main stream steps the envs (scenes without a ball).
// tags / tag_offset (STAGED): the reset count the state is for - tags[i], or the env's current count + tag_offset
template <bool STAGED, bool EXTRA>
__global__ SMR_SIDE_KERNEL void reset_kernel(const DScene *__restrict__ Sp, DState st, const int *__restrict__ list,
                                                   const int *__restrict__ count_ptr, int count_fixed, int group_stride,
                                                   float *__restrict__ obs, const unsigned *__restrict__ tags, int tag_offset) {
    const DScene &S = *Sp;
    __shared__ __attribute__((aligned(16))) uint32_t pool[kBodyPoolDwords];
    const int count = count_ptr ? *count_ptr : count_fixed;
    if (!stage_body_pool(S, pool, count, group_stride)) return;
    const BodyPool BP{pool};
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = t >> 3, g = t & 7;
    if (grp % group_stride) return;
    const int i = grp / group_stride;
    if (i >= count) return;
    const int e = list ? list[i] : i;
    const int B = S.B;
    const Group G{g, (int)(threadIdx.x & 63) & ~7};
    const int gbase = G.gbase;
    const Rng rng = {S.seed_lo, S.seed_hi, S.env_base + (unsigned)e, tags ? tags[i] : st.reset_count[e] + (unsigned)tag_offset};
    const int jj = g < SMR_NJ ? g : 0;

    // ---- start position: attempts in sequence, each tested link-parallel (stream S_POS, attempt n = draws 7n..7n+6)
    float q[SMR_NJ];
    unsigned attempt = 0;
    auto next_position = [&]() {
        for (int tries = 0; tries < 100000; ++tries) {
            const float qme = rng.uniform(S_POS, 7u * attempt + jj, S.pos_lo[jj], S.pos_hi[jj]);          // ctlp.py:2035-2037
            G.gather(qme, q);
            ++attempt;
            if (group_pose_valid<EXTRA>(S, BP, G, G.trig(qme), S.start_box, S.start_min_static, S.start_min_self)) return;
        }
        if (g == 0) sampling_exhausted(st, kSamplingStart);                                               // ctlp.py:2127-2136 raises
    };
    SMR_PHASE_INIT;
    next_position();
    const unsigned first_attempts = attempt;                                            // ctlp.py:1480-1488
    SMR_PHASE(0);

    // ---- joint velocity / acceleration.  After this block lane j (< 7) holds joint j in (qj, vj, aj).
    float vj = 0.f, aj = 0.f, qj = 0.f;
    const JointLim L = joint_lim(S, jj);
    if (!S.ca_mode) {                                                                   // the episode starts at rest, ctlp.py:1490-1491
        qj = q[jj];
    } else if (S.kin_mode && rng.uniform01(S_MODE, 0) < S.kin_prob) {                   // ctlp.py:1494-1556
        for (unsigned round = 0;; ++round) {
            if (round >= 4096u) { if (g == 0) sampling_exhausted(st, kSamplingKinematicState); break; }    // the reference loops on (ctlp.py:1499-1556)
            bool found = (g >= SMR_NJ);
            const float qme = q[jj];
            // the group walks the 5 x 10 (velocity, acceleration) tries in lockstep; a lane that has its pair tags along
            for (unsigned tv = 0; tv < 5 && !G.all(found); ++tv) {
                const float vt = rng.uniform(S_JOINT + jj, round * 55u + tv * 11u, -L.V, L.V);
                for (unsigned ta = 0; ta < 10 && !G.all(found); ++ta) {
                    const float at = rng.uniform(S_JOINT + jj, round * 55u + tv * 11u + 1u + ta, -L.A, L.A);
                    const int code = safe_state_code<8>(L, qme, vt, at, !found);        // only feasibility is asked for
                    if (!found && code == 0) { found = true; vj = vt; aj = at; }
                }
            }
            if (G.all(found)) break;
            next_position();                                                            // ctlp.py:1547-1556
        }
        qj = q[jj];
    } else {                                                                            // ctlp.py:1558-1654
        // random walk of safe actions; each lane keeps the ring of the three latest states of ITS joint
        float hq0 = q[jj], hq1 = 0.f, hq2 = 0.f, hv0 = 0.f, hv1 = 0.f, hv2 = 0.f, ha0 = 0.f, ha1 = 0.f, ha2 = 0.f;
        int cnt = 1;
        unsigned w = 0;
        for (int walk = 0; walk < 64; ++walk) {
            if (rng.uniform01(S_WALK, w++) < S.stay_prob) break;
            const int cur = (cnt - 1) % 3, nxt = cnt % 3;
            const float cq = cur == 0 ? hq0 : cur == 1 ? hq1 : hq2, cv = cur == 0 ? hv0 : cur == 1 ? hv1 : hv2,
                        ca = cur == 0 ? ha0 : cur == 1 ? ha1 : ha2;
            float lo, hi;
            int code;
            safe_acc_range_joint<8>(L, cq, cv, ca, lo, hi, code, g < SMR_NJ);
            const float u = rng.uniform(S_WALK, w + (unsigned)jj, -1.f, 1.f);
            w += SMR_NJ;
            const float na = lo + 0.5f * (u + 1.f) * (hi - lo);
            const float nv = interp_vel(cv, ca, na, S.dt, S.dt), nq = interp_pos(cq, cv, ca, na, S.dt, S.dt);
            if (group_table_free<EXTRA>(S, BP, G, G.trig(nq), S.start_min_static, S.start_min_self)) {
                if (nxt == 0) { hq0 = nq; hv0 = nv; ha0 = na; } else if (nxt == 1) { hq1 = nq; hv1 = nv; ha1 = na; } else { hq2 = nq; hv2 = nv; ha2 = na; }
                ++cnt;
            } else {
                const int sel = 1 + rng.randint(S_WALK, w++, cnt < 3 ? cnt : 3);        // ctlp.py:1641-1646
                cnt = cnt - sel + 1;
                break;
            }
        }
        const int cur = (cnt - 1) % 3;
        qj = cur == 0 ? hq0 : cur == 1 ? hq1 : hq2;
        vj = cur == 0 ? hv0 : cur == 1 ? hv1 : hv2;
        aj = cur == 0 ? ha0 : cur == 1 ? ha1 : ha2;
        G.gather(qj, q);
    }

    SMR_PHASE(1);
    const Trig P = G.trig(qj);            // the start configuration
    // ---- planet phase: candidate phases in sequence, first contact-free one wins (ctlp.py:4470-4501)
    int idx[SMR_MAX_PLANETS] = {0, 0};
    int hit = 0;
    if (S.n_planets > 0) {
        int n_indep = 0;
#pragma unroll
        for (int p = 0; p < SMR_MAX_PLANETS; ++p) n_indep += (p < S.n_planets && !S.planet[p].coupled) ? 1 : 0;
        bool free_phase = false;
        for (unsigned n = 0; n < 100000u && !free_phase; ++n) {
            int k = 0;
#pragma unroll
            for (int p = 0; p < SMR_MAX_PLANETS; ++p) {
                if (p >= S.n_planets) continue;
                const int no = S.planet[p].n_orbit;
                if (S.planet[p].coupled) idx[p] = ((idx[0] + S.planet[p].index_shift) % no + no) % no;
                else idx[p] = rng.randint(S_PLANET, n * (unsigned)n_indep + (unsigned)(k++), no);
            }
            free_phase = !group_planet_contact<EXTRA>(S, BP, G, P, idx);
        }
        if (!free_phase && g == 0) sampling_exhausted(st, kSamplingPlanetPhase);                          // the reference loops on (ctlp.py:4477-4490)
        // obstacle_wrapper.update(update_after_reset=True): one planet update + contact check,
        // base.py:980-985, ctlp.py:2626-2637
#pragma unroll
        for (int p = 0; p < SMR_MAX_PLANETS; ++p)
            if (p < S.n_planets) idx[p] = (idx[p] + 1) % S.planet[p].n_orbit;
        if (S.term_moving) hit = group_planet_contact<EXTRA>(S, BP, G, P, idx) ? 1 : 0;
    }

    SMR_PHASE(2);
    if (STAGED) {      // robot part only; the ball of a reset depends on the env's spawn history: spawn_kernel draws it at commit time
        const size_t sl = (size_t)(rng.tag & 1u);
        if (g < SMR_NJ) { st.next_q[(sl * SMR_NJ + g) * B + e] = qj; st.next_v[(sl * SMR_NJ + g) * B + e] = vj; st.next_a[(sl * SMR_NJ + g) * B + e] = aj; }
        else { st.next_planet_idx[(sl * 2) * B + e] = idx[0]; st.next_planet_idx[(sl * 2 + 1) * B + e] = idx[1]; st.next_hit[sl * B + e] = hit; st.next_attempts[sl * B + e] = first_attempts; st.staged_tag[sl * B + e] = rng.tag; }
        return;
    }
    // ---- ball: spawn + random initial time step, candidates in sequence (ctlp.py:1052-1111)
    BallSpawn bsp;
    int ball_k0 = 0;
    unsigned ball_spawns = 0;
    if (S.use_ball) {
        ball_spawns = (unsigned)st.ball_i[4 * B + e];
        const Rng brng = {S.seed_lo, S.seed_hi, S.env_base + (unsigned)e, ball_spawns};
        bsp = spawn_ball_group<EXTRA>(S, BP, st, brng, G);
        if (S.ball_random_initial) {
            BallState bs;
            for (int c = 0; c < 3; ++c) { bs.p0[c] = bsp.p0[c]; bs.v0[c] = bsp.v0[c]; bs.e0[c] = bsp.e0[c]; }
            bs.omega = bsp.omega;
            int kcap = bsp.kmax < bsp.khit ? bsp.kmax : bsp.khit;
            int max_ts = kcap / S.updates_per_step;
            if (max_ts < 0) max_ts = 0;
            bool free_step = false;
            for (unsigned n = 0; n < 100000u && !free_step; ++n) {
                ball_k0 = brng.randint(S_BALL_MISC, 1u + n, max_ts + 1) * S.updates_per_step;
                free_step = !group_ball_contact<EXTRA>(S, BP, G, P, ball_pose(S, bs, ball_k0, nullptr));
            }
            if (!free_step && g == 0) sampling_exhausted(st, kSamplingBallInitialStep);
        }
    }

    // ---- write back: lane j < 7 owns joint j, lane 7 the scalars
    float *row = obs ? obs + (size_t)e * S.obs_size : nullptr;
    if (g < SMR_NJ) {
        st.q[g * B + e] = qj; st.v[g * B + e] = vj; st.a[g * B + e] = aj;
        if (row) {
            row[g] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (qj - L.plo) / (L.phi - L.plo)));
            row[SMR_NJ + g] = fminf(1.f, fmaxf(-1.f, vj / L.V));
            row[2 * SMR_NJ + g] = fminf(1.f, fmaxf(-1.f, aj / L.A));
        }
    } else {
        st.planet_idx[e] = idx[0];
        st.planet_idx[B + e] = idx[1];
        st.range_valid[e] = 0;                             // (the first plan kernel computes the range of an unstaged start state itself)
        st.step_count[e] = 0;
        st.moving_hit[e] = hit;
        st.start_attempts[e] = first_attempts;
        st.reset_count[e] = rng.tag + 1u;
        if (row) write_planet_observation(S, idx, row + 3 * SMR_NJ + S.obs_target_size);
        if (S.use_ball) {
            store_ball(st, B, e, bsp, ball_k0, ball_spawns + 1u);
            BallState bs;
            for (int c = 0; c < 3; ++c) { bs.p0[c] = bsp.p0[c]; bs.v0[c] = bsp.v0[c]; bs.e0[c] = bsp.e0[c]; }
            bs.omega = bsp.omega;
            if (row) write_ball_observation(S, bs, ball_k0, row + 3 * SMR_NJ + S.obs_target_size);
        }
    }
}

__global__ void mask_to_list_kernel(const uint8_t *__restrict__ mask, int B, int *__restrict__ list, int *__restrict__ count) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    if (mask[e]) list[atomicAdd(count, 1)] = e;
}

__global__ void iota_list_kernel(int B, int *__restrict__ list, int *__restrict__ count) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < B) list[e] = e;
    if (e == 0) *count = B;
}

__global__ void observe_kernel(const DScene *__restrict__ Sp, DState st, float *__restrict__ obs) {
    const DScene &S = *Sp;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.B) return;
    const int B = S.B;
    float q[SMR_NJ], v[SMR_NJ], a[SMR_NJ];
    for (int j = 0; j < SMR_NJ; ++j) { q[j] = st.q[j * B + e]; v[j] = st.v[j * B + e]; a[j] = st.a[j * B + e]; }
    int idx[SMR_MAX_PLANETS] = {st.planet_idx[e], st.planet_idx[B + e]};
    write_observation(S, q, v, a, idx, obs + (size_t)e * S.obs_size);
    if (S.use_ball) write_ball_observation(S, load_ball(st, B, e), st.ball_i[e], obs + (size_t)e * S.obs_size + 3 * SMR_NJ + S.obs_target_size);
    if (S.use_target) {
        float tp[3] = {st.tp[e], st.tp[B + e], st.tp[2 * B + e]};
        write_target_observation(S, tp, ee_point(S, q), obs + (size_t)e * S.obs_size);
    }
}

// ---- component kernels for the parity tests ----
__global__ void test_fk_kernel(const DScene *__restrict__ Sp, const float *__restrict__ q, int n, float *__restrict__ pos, float *__restrict__ rotm) {
    const DScene &S = *Sp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xf T = fk_base(S);
    for (int l = 0; l < SMR_NLINK; ++l) {
        if (l > 0) fk_advance(S, l - 1, q[i * SMR_NJ + l - 1], T);
        float *p = pos + ((size_t)i * SMR_NLINK + l) * 3;
        p[0] = T.t.x; p[1] = T.t.y; p[2] = T.t.z;
        for (int c = 0; c < 9; ++c) rotm[((size_t)i * SMR_NLINK + l) * 9 + c] = T.r[c];
    }
}

__global__ void test_range_kernel(const DScene *__restrict__ Sp, const float *__restrict__ p, const float *__restrict__ v,
                                  const float *__restrict__ a, int n, float *__restrict__ lo, float *__restrict__ hi, int *__restrict__ code) {
    const DScene &S = *Sp;
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < n;
    const int i = live ? i0 : n - 1;
    for (int j = 0; j < SMR_NJ; ++j) {
        int k = i * SMR_NJ + j, c;
        float l, h;
        safe_acc_range_joint<64>(joint_lim(S, j), p[k], v[k], a[k], l, h, c, live);
        if (live) { lo[k] = l; hi[k] = h; code[k] = c; }
    }
}

__global__ void test_distances_kernel(const DScene *__restrict__ Sp, const float *__restrict__ q, const int *__restrict__ pidx, int n,
                                      float *__restrict__ out) {
    const DScene &S = *Sp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float qq[SMR_NJ];
    for (int j = 0; j < SMR_NJ; ++j) qq[j] = q[i * SMR_NJ + j];
    int idx[SMR_MAX_PLANETS] = {pidx[2 * i], pidx[2 * i + 1]};
    Sample d = collide_sample(S, qq, idx, true, false);
    Sample c = collide_sample(S, qq, idx, false, true);
    out[4 * i] = d.d_static; out[4 * i + 1] = d.d_self; out[4 * i + 2] = d.d_moving; out[4 * i + 3] = c.contact ? 1.f : 0.f;
}

__global__ void test_ball_distances_kernel(const DScene *__restrict__ Sp, const float *__restrict__ q, const float *__restrict__ ball,
                                           const int *__restrict__ kk, int n, float *__restrict__ out) {
    const DScene &S = *Sp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BallState b;
    for (int c = 0; c < 3; ++c) { b.p0[c] = ball[10 * i + c]; b.v0[c] = ball[10 * i + 3 + c]; b.e0[c] = ball[10 * i + 6 + c]; }
    b.omega = ball[10 * i + 9];
    const Xf XB = ball_pose(S, b, kk[i], nullptr);
    float dst = S.static_max, dmv = S.moving_max + 0.002f;
    bool contact = false;
    float qq[SMR_NJ], dsf = S.static_max;
    for (int j = 0; j < SMR_NJ; ++j) qq[j] = q[i * SMR_NJ + j];
    for (int l = 0; l < S.n_bodies; ++l) {
        const Xf T = body_frame(S, qq, l);
        const DPiece LP = load_piece(S.pieces, S.link_piece[l]);
        const GlobalHull hull = global_hull(S, S.link_piece[l], LP);
        Sample d = collide_link(S, l, LP, hull, T, nullptr, &XB, true, true, false, qq);
        Sample c = collide_link(S, l, LP, hull, T, nullptr, &XB, false, false, true);
        dst = fminf(dst, d.d_static); dmv = fminf(dmv, d.d_moving); dsf = fminf(dsf, d.d_self); contact = contact || c.contact;
    }
    out[4 * i] = dst; out[4 * i + 1] = dsf; out[4 * i + 2] = dmv; out[4 * i + 3] = contact ? 1.f : 0.f;
}

// ---- self-pair bound tables (smr_device.h, "Self pairs: bound tables"): one lane per cell of pair i's grid ----
// The entry is a rigorous lower bound of the pair's distance at the cell centre: the smaller of the value the run returns and the
// largest support-plane bound it saw (a stalled run returns an upper bound), margins subtracted, rounded down to kSelfTabStep.
__global__ __launch_bounds__(256) void self_table_kernel(const DScene *__restrict__ Sp, int i, uint16_t *__restrict__ out, int n_cells) {
    const DScene &S = *Sp;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= n_cells) return;
    const int nd = S.self_tab_nd[i];
    float q[SMR_NJ] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int rest = cell;
    for (int d = nd - 1; d >= 0; --d) {
        const int n = S.self_tab_n[i][d], c = rest % n, j = SMR_NJ - nd + d;
        rest /= n;
        const float centre = S.pos_lo[j] + ((float)c + 0.5f) * S.self_tab_h[i][d];
        if (j == 4) q[4] = centre; else if (j == 5) q[5] = centre; else q[6] = centre;
    }
    const int a = S.self_pair[i][0];
    const DPiece LP = load_piece(S.pieces, S.link_piece[a]);
    const GlobalHull hull = global_hull(S, S.link_piece[a], LP);
    float lower = 0.f;
    const float d = self_pair_distance<false, false>(S, i, q, LP, hull, body_frame(S, q, a), kSelfTabMax, 0.f, nullptr, nullptr, nullptr, &lower);
    // a value beyond the cutoff is a lower bound already (bounding volumes, or a separating plane farther than the cutoff)
    float v = d > kSelfTabMax ? kSelfTabMax : fminf(d, lower);
    v = fminf(fmaxf(v - 2e-6f, 0.f), kSelfTabMax);
    out[cell] = (uint16_t)floorf(v * (1.0f / kSelfTabStep));
}

// A step whose job list ran full (push_jobs / keys_kernel set overflow_env): the envs concerned get ALL their queries from this kernel,
// one lane per env, every body and every sample in sequence with the plain per-sample routine of the component tests (collide_sample) -
// slow, and exact.  The list holds 64 jobs per env, twelve times the average of uniformly drawn states, so this runs when most of a
// batch sits inside the planets' bounding volumes at once (or the tests shrink the list with SMR_JOB_CAP); otherwise a block reads
// one word per lane and leaves.  The results are minima / flags like those of the jobs, so a job that did fit changes nothing.
__global__ __launch_bounds__(64) void overflow_kernel(const DScene *__restrict__ Sp, DState st) {
    const DScene &S = *Sp;
    const int B = S.B, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B || !st.overflow_env[e]) return;
    const int N = S.checks_per_step, upd = S.updates_per_step / N;
    const bool want_contact = S.term_moving && S.n_planets > 0 && st.moving_hit[e] == 0;
    bool contact = false;
    for (int k = N; k >= 1; --k) {
        if (k < N && !(want_contact && !contact)) break;
        const float t = k == N ? S.dt : S.dt * (float)k / (float)N;
        float q[SMR_NJ];
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) q[j] = interp_pos(st.q[j * B + e], st.v[j * B + e], st.a[j * B + e], st.a1[j * B + e], t, S.dt);
        int idx[SMR_MAX_PLANETS] = {0, 0};
#pragma unroll
        for (int p = 0; p < SMR_MAX_PLANETS; ++p)
            if (p < S.n_planets) idx[p] = (st.planet_idx[p * B + e] + k * upd) % S.planet[p].n_orbit;
        if (k == N) {
            const Sample r = collide_sample(S, q, idx, true, false);
            atomicMin(&st.env_dself[e], dist_bits(r.d_self));
            if (S.planets_jobs) {
                atomicMin(&st.env_dstatic[e], dist_bits(r.d_static));
                if (r.d_moving <= S.moving_max) atomicMin(&st.env_min[e], dist_bits(r.d_moving));
            }
        }
        if (S.planets_jobs && want_contact) contact = contact || collide_sample(S, q, idx, false, true).contact;
    }
    if (S.ball_jobs) {                                   // the ball, every body and sample in the lane form (collide_lane / collide_item's rules)
        const LaneInput in = load_lane_input(S, st, e);
        const int u_end = ball_retire_update(S, in.ball_k0, in.ball_kmax, in.ball_khit);
        for (int k = N; k >= 1 && in.ball_active; --k) {
            if (!(k * upd < u_end)) continue;               // retired before this sample
            const Xf XB = ball_pose(S, in.ball, in.ball_k0 + k * upd, nullptr);
            const bool near = XB.t.x * XB.t.x + XB.t.y * XB.t.y < S.ball_active_xy_radius * S.ball_active_xy_radius;
            for (int l = 0; l < S.n_bodies; ++l) {
                const bool mov = k == N && ((S.moving_mask >> l) & 1u), con = near && !contact && ((S.contact_mask >> l) & 1u);
                if (!(mov || con)) continue;
                const int pl = S.link_piece[l];
                const DPiece LP = load_piece(S.pieces, pl);
                const Xf T = lane_frame_any<false>(S, l, in, k == N ? S.dt : S.dt * (float)k / (float)N);
                float cutoff = mov ? S.moving_max : S.contact_dist;
                if (con) cutoff = fmaxf(cutoff, S.contact_dist);
                const float d = piece_distance<false>(S, S.ball_piece, XB, LP, global_hull(S, pl, LP), T, cutoff, con ? S.contact_dist : S.collision_dist);
                if (mov && d <= S.moving_max) atomicMin(&st.env_min[e], dist_bits(d));
                if (con && d <= S.contact_dist) contact = true;
            }
        }
    }
    if (contact) st.res[e] = 1.f;                       // finish ORs the bodies' flags: body 0's slot will do
}

// the lower bounds the step's pruning uses for the self-collision pairs of n joint vectors (component test): out [n][SMR_MAX_SELF_PAIRS]
__global__ void test_self_bounds_kernel(const DScene *__restrict__ Sp, const float *__restrict__ q, int n, float *__restrict__ out) {
    const DScene &S = *Sp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int p = 0; p < SMR_MAX_SELF_PAIRS; ++p)
        out[(size_t)i * SMR_MAX_SELF_PAIRS + p] = p < S.n_self_pairs ? self_tab_bound(S, p, q[i * SMR_NJ + 4], q[i * SMR_NJ + 5], q[i * SMR_NJ + 6]) : -kNoBound;
}

// ---- risk-aware stepping: actions.py:303-340, base.py:1498-1502,1598-1603 ----
// row e of x: the risk network's input for env e; row e of xb: the backup policy's - the risk observation of the CURRENT state
// (obs.py:419-431: the observation without its target-point columns).  mode (base.py:1498-1517):
//   SMR_RISK_STATE_ACTION     x = [risk observation | requested action]                       (a network with action_size = 7)
//   SMR_RISK_CURRENT_STATE    x = the risk observation                                         (RISK_CHECK_CURRENT_STATE)
//   SMR_RISK_NEXT_KINEMATIC   x = the risk observation the NEXT step would see for the requested action: joint positions, velocities
//                                 and accelerations at the end of the planned cubic (plan_kernel has run on the requested action:
//                                 a1), the planets / the ball as now (_get_next_risk_observation, obs.py:192-231)
//   SMR_RISK_NEXT_FULL        ... and the planets / the ball one step on as well (forecast_non_kinematic_components, obs.py:262-271,
//                                 ctlp.py:2364-2377)
__global__ void risk_input_kernel(const DScene *__restrict__ Sp, DState st, const float *__restrict__ actions, float *__restrict__ x, int ldx,
                                  float *__restrict__ xb, int ldxb, int mode) {
    const DScene &S = *Sp;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.B) return;
    const int B = S.B;
    float q[SMR_NJ], v[SMR_NJ], a[SMR_NJ];
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) { q[j] = st.q[j * B + e]; v[j] = st.v[j * B + e]; a[j] = st.a[j * B + e]; }
    const bool forecast = mode == SMR_RISK_NEXT_KINEMATIC || mode == SMR_RISK_NEXT_FULL, next_all = mode == SMR_RISK_NEXT_FULL;
    float *rb = xb + (size_t)e * ldxb, *rx = x + (size_t)e * ldx;
    // the backup policy's row: the current state; the risk network's row is the same unless it forecasts
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) {
        const float pj = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (q[j] - S.pos_lo[j]) / (S.pos_hi[j] - S.pos_lo[j])));
        const float vj = fminf(1.f, fmaxf(-1.f, v[j] / S.vel_max[j])), aj = fminf(1.f, fmaxf(-1.f, a[j] / S.acc_max[j]));
        rb[j] = pj; rb[SMR_NJ + j] = vj; rb[2 * SMR_NJ + j] = aj;
        if (!forecast) { rx[j] = pj; rx[SMR_NJ + j] = vj; rx[2 * SMR_NJ + j] = aj; }
        else {
            const float a1 = st.a1[j * B + e];
            const float qn = interp_pos(q[j], v[j], a[j], a1, S.dt, S.dt), vn = interp_vel(v[j], a[j], a1, S.dt, S.dt);
            rx[j] = fminf(1.f, fmaxf(-1.f, -1.f + 2.f * (qn - S.pos_lo[j]) / (S.pos_hi[j] - S.pos_lo[j])));
            rx[SMR_NJ + j] = fminf(1.f, fmaxf(-1.f, vn / S.vel_max[j])); rx[2 * SMR_NJ + j] = fminf(1.f, fmaxf(-1.f, a1 / S.acc_max[j]));
        }
    }
    int o = 3 * SMR_NJ;
    const int rest = (S.use_ball ? 6 : 0) + S.obs_num_planets * S.obs_planet_size;
    if (S.use_ball) write_ball_observation(S, load_ball(st, B, e), st.ball_i[e], rb + o);                  // moving object, then planets: obs.py:424-427
    const int idx[SMR_MAX_PLANETS] = {st.planet_idx[e], st.planet_idx[B + e]};
    write_planet_observation(S, idx, rb + o + (S.use_ball ? 6 : 0));
    if (!next_all) for (int c = 0; c < rest; ++c) rx[o + c] = rb[o + c];
    else {
        if (S.use_ball) write_ball_observation(S, load_ball(st, B, e), st.ball_i[e] + S.updates_per_step, rx + o);
        int idn[SMR_MAX_PLANETS] = {idx[0], idx[1]};
#pragma unroll
        for (int p = 0; p < SMR_MAX_PLANETS; ++p) if (p < S.n_planets) idn[p] = (idn[p] + S.updates_per_step) % S.planet[p].n_orbit;
        write_planet_observation(S, idn, rx + o + (S.use_ball ? 6 : 0));
    }
    o += rest;
    if (mode == SMR_RISK_STATE_ACTION)
#pragma unroll
        for (int j = 0; j < SMR_NJ; ++j) rx[o + j] = actions[(size_t)e * SMR_NJ + j];
}

// threshold test of the rating (base.py:1602-1603), the backup policy's action where it objects (actions.py:327-335), and - where asked for -
// risk_ignore_estimation_probability (base.py:1652-1655): a risky rating is dropped with probability ignore_p; the draw is keyed by the
// env's reset and step counters (the reference draws from the global numpy stream)
constexpr unsigned S_RISK_IGNORE = 15;
__global__ void risk_apply_kernel(const DScene *__restrict__ Sp, DState st, const float *__restrict__ actions, const float *__restrict__ risk,
                                  const float *__restrict__ backup, int ldb, float threshold, float ignore_p, float *__restrict__ actions_out) {
    const DScene &S = *Sp;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S.B) return;
    bool risky = risk[e] >= threshold;                                                      // base.py:1602-1603
    if (risky && ignore_p > 0.f) {
        const Rng rng{S.seed_lo, S.seed_hi, S.env_base + (unsigned)e, st.reset_count[e]};
        if (rng.uniform01(S_RISK_IGNORE, (unsigned)st.step_count[e]) < ignore_p) risky = false;
    }
    st.risky[e] = risky ? 1 : 0;
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j)
        actions_out[(size_t)e * SMR_NJ + j] = risky ? fminf(1.f, fmaxf(-1.f, backup[(size_t)e * ldb + j])) : actions[(size_t)e * SMR_NJ + j];
}

// ---- the background copy of the env (base.py:1518-1577,1605-1650; include/safemotions_hip.h: smr_set_risk_rollout) ----
// the envs' state -> the shadow handle's (switch_to_backup_client, base.py:1801-1823: the copy starts with _episode_length = 0)
__global__ void rollout_snapshot_kernel(const DScene *__restrict__ Sp, DState src, DState dst, int *__restrict__ alive, int *__restrict__ outcome) {
    const DScene &S = *Sp;
    const int e = blockIdx.x * blockDim.x + threadIdx.x, B = S.B;
    if (e >= B) return;
#pragma unroll
    for (int j = 0; j < SMR_NJ; ++j) { dst.q[j * B + e] = src.q[j * B + e]; dst.v[j * B + e] = src.v[j * B + e]; dst.a[j * B + e] = src.a[j * B + e];
                                       dst.range_lo[j * B + e] = src.range_lo[j * B + e]; dst.range_hi[j * B + e] = src.range_hi[j * B + e]; }
    dst.range_valid[e] = src.range_valid[e];
    dst.planet_idx[e] = src.planet_idx[e]; dst.planet_idx[B + e] = src.planet_idx[B + e];
    dst.step_count[e] = 0;
    dst.moving_hit[e] = src.moving_hit[e];
    dst.reset_count[e] = src.reset_count[e]; dst.start_attempts[e] = src.start_attempts[e];
    for (int l = 0; l < S.n_bodies; ++l) dst.lb_moving[(size_t)l * B + e] = -1.f;              // (no bound of a previous step)
    if (S.use_ball) {
        for (int c = 0; c < 10; ++c) dst.ball[c * B + e] = src.ball[c * B + e];
        for (int c = 0; c < 5; ++c) dst.ball_i[c * B + e] = src.ball_i[c * B + e];
    }
    if (S.use_target) {
        for (int c = 0; c < 5; ++c) dst.tp[c * B + e] = src.tp[c * B + e];
        dst.tp_count[e] = src.tp_count[e];
    }
    alive[e] = 1; outcome[e] = 0;
}
// after simulated step i of the copy: an env whose copy is done leaves the rollout (the reference breaks out of its loop), risky unless
// the copy's trajectory length ended it.  outcome bits: 1 the copy collided (at any step), 2 its FIRST step ended it, 4 ... by a collision
__global__ void rollout_track_kernel(int B, int first, const uint8_t *__restrict__ done, const int8_t *__restrict__ reason, int *__restrict__ alive,
                                     int *__restrict__ outcome) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B || !alive[e] || !done[e]) return;
    const bool collided = reason[e] != SMR_TERMINATION_TRAJECTORY_LENGTH;
    alive[e] = 0;
    outcome[e] |= (collided ? 1 : 0) | (first ? 2 : 0) | ((first && collided) ? 4 : 0);
}
// the backup policy's deterministic action (_get_action_from_backup_action: its first 7 outputs, clipped)
__global__ void rollout_action_kernel(int B, const float *__restrict__ backup, int ldb, float *__restrict__ actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * SMR_NJ) return;
    actions[i] = fminf(1.f, fmaxf(-1.f, backup[(size_t)(i / SMR_NJ) * ldb + (i % SMR_NJ)]));
}
// the rating of the simulating modes (base.py:1548-1566): mode 4: 1 where the copy collided, else 0; mode 3: 1 where the copy's first step
// collided, the network's rating of the copy's next risk observation where that step did not end the copy's episode, else no rating (0)
__global__ void rollout_rating_kernel(int B, int mode, const int *__restrict__ outcome, const float *__restrict__ net, float *__restrict__ rating) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    const int o = outcome[e];
    if (mode == SMR_RISK_SIMULATE_BACKUP_TRAJECTORY) rating[e] = (o & 1) ? 1.f : 0.f;
    else rating[e] = (o & 4) ? 1.f : ((o & 2) ? 0.f : net[e]);
}
// risk_store_ground_truth (base.py:1567-1586,1605-1650): rows of the two tables, see smr_set_risk_rollout
__global__ void rollout_ground_truth_kernel(int B, int R, int mode, const float *__restrict__ actions, const float *__restrict__ obs_now,
                                            const float *__restrict__ obs_first, const int *__restrict__ outcome, const float *__restrict__ rating,
                                            float *__restrict__ gt_sa, float *__restrict__ gt_s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    const int o = outcome[e];
    const float label = (o & 1) ? 1.f : 0.f;
    const bool first_ok = !(o & 2);                          // the first simulated step did not end the copy's episode
    // whose state the single table carries: the current one for a state-action network and mode 0, the copy's next one for modes 1 - 3
    const bool next_state = mode == SMR_RISK_NEXT_KINEMATIC || mode == SMR_RISK_NEXT_FULL || mode == SMR_RISK_SIMULATE_NEXT_STEP;
    float *row = gt_sa + (size_t)e * (R + SMR_NJ + 3);
    const float *state = next_state ? obs_first + (size_t)e * R : obs_now + (size_t)e * R;
    for (int c = 0; c < R; ++c) row[c] = state[c];
    for (int j = 0; j < SMR_NJ; ++j) row[R + j] = actions[(size_t)e * SMR_NJ + j];
    row[R + SMR_NJ] = label; row[R + SMR_NJ + 1] = rating[e]; row[R + SMR_NJ + 2] = (!next_state || first_ok) ? 1.f : 0.f;
    if (gt_s && mode == SMR_RISK_SIMULATE_BACKUP_TRAJECTORY) {
        float *r2 = gt_s + (size_t)e * (R + 2);
        for (int c = 0; c < R; ++c) r2[c] = obs_first[(size_t)e * R + c];
        r2[R] = label; r2[R + 1] = first_ok ? 1.f : 0.f;
    }
}

}  // namespace smr

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
using namespace smr;

struct smr_env {
    int device = 0;
    hipStream_t stream = nullptr;
    int B = 0, obs_size = 0, N = 1;
    int auto_reset = 0;
    bool staged = false;        // start states are prepared on `side` and committed by finish (no ball in the scene)
    hipStream_t side = nullptr;
    hipStream_t side2 = nullptr;      // staged target points run beside the staged start states (forked from / joined into `side`)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // the refills of consecutive steps overlap (each has two step periods): a second set of streams / events, used by odd steps
    hipStream_t side_b = nullptr, side2_b = nullptr;
    int self_job_blocks = 0;          // grid of a jobs kernel launch that only has self-pair jobs (0.3 per env)
    hipEvent_t ev_commit_b = nullptr, ev_refill_b = nullptr, ev_fork_b = nullptr, ev_join_b = nullptr;
    int rot = 0;                      // which of the three list slots this step uses (a list is re-used two steps after its refill was awaited)
    int *refill_lists = nullptr;      // [3][B]
    unsigned *refill_tag_lists = nullptr;   // [3][B]
    int *tp_refill_lists = nullptr;   // [3][B]
    unsigned *tp_refill_tag_lists = nullptr, *spawn_tag_lists = nullptr;   // [3][B] each
    hipEvent_t ev_commit = nullptr, ev_refill = nullptr;
    // the next step's safe acceleration ranges are computed beside the distance kernels (range_kernel; SMR_RANGE_AHEAD=0: by the plan kernel, as before)
    int range_mode = 1;                               // SMR_RANGE_AHEAD=0: the plan kernel computes the ranges itself, as it did until round 4 (A/B switch)
    bool range_host = true;                           // as extra blocks of the jobs kernel where the step has one (SMR_RANGE_HOST=0: always the stream)
    bool self_rows = false;                           // SMR_SELF_ROWS=1: the self-collision pairs' bound tests as rows of the keys / collide grid instead of self_bounds_kernel (measured: -1.6 .. +2 %, opt-in)
    hipStream_t range_stream = nullptr;
    hipEvent_t ev_planned = nullptr, ev_ranges = nullptr;
    DScene hscene;              // host copy (device pointers inside)
    int *refill_slots = nullptr; // two list counters: step s uses slot s & 1, its finish kernel clears the other
    int *tp_refill_slots = nullptr;   // the same for the staged target points (null: not staged)
    int *spawn_lists = nullptr, *spawn_counts = nullptr;   // two spawn lists: the side stream re-stages from one while finish fills the other
    bool stage_balls = false;
    int parity = 0;
    struct smr_mlp *risk_net = nullptr, *backup_net = nullptr;     // risk-aware stepping (smr_set_risk_networks)
    float risk_threshold = 0.05f;
    float *risk_x = nullptr, *risk_y = nullptr, *backup_x = nullptr, *backup_y = nullptr, *actions_eff = nullptr;
    int risk_state_config = SMR_RISK_CURRENT_STATE;   // what a risk network WITHOUT an action input looks at (smr_set_risk_state_config)
    int risk_mode = SMR_RISK_STATE_ACTION;            // the mode in force (state-action when the network takes the action)
    int *risky_buf = nullptr;
    // the background copy (smr_set_risk_rollout): a second handle of the same scene that smr_step steps ahead
    struct smr_env *shadow = nullptr;
    int rollout_steps = 0;            // risk_state_backup_trajectory_steps
    float risk_ignore_p = 0.f;        // risk_ignore_estimation_probability
    float *gt_sa = nullptr, *gt_s = nullptr;                 // caller-owned ground-truth rows (or null)
    int *ro_alive = nullptr, *ro_outcome = nullptr;
    float *ro_first = nullptr, *ro_rollx = nullptr, *ro_riskx = nullptr, *ro_actions = nullptr, *ro_obs = nullptr, *ro_reward = nullptr, *ro_rating = nullptr;
    uint8_t *ro_done = nullptr; int8_t *ro_reason = nullptr;
    size_t job_lds_bytes = 0;         // dynamic LDS of the first / jobs kernels (pair-job path)
    int job_blocks = 0;               // their persistent grid: one 512-lane block per CU
    int job_block_lanes = 512;        // (SMR_JOB_BLOCK_LANES: 256 / 128 for measurements)
    int defer_after = 0;              // jobs_kernel: a run leaves its lane after this many iterations for jobs_coop_kernel (0: never; SMR_JOB_DEFER)
    int coop_blocks = 0;              // ... whose grid is one 1024-lane block per CU
    size_t side_lds_pad = 0;          // dynamic LDS added to the staged refill kernels' blocks (SMR_SIDE_LDS_PAD, bytes): with their 60 KB of
                                      // tables > 80 KB per block = one block per CU, so that a jobs block (64 KB) always finds room beside it
    std::vector<int> host_planet_padded;      // padded vertex count of every planet descriptor (auto_desc[..][1]), for DState::big_piece_mask
    int job_quota_max = 64;           // jobs a wavefront takes per hand-out at most (SMR_JOB_QUOTA_MAX: fewer = idle lanes, shorter chains)
    // lanes per block of brake_check: one wavefront - with three wavefronts per SIMD a 4-wavefront block rarely finds its four slots at
    // once (560 of 768 blocks resident; 256 -> 128 -> 64 lanes: 1.58 -> 1.42 -> 1.39 ms).  SMR_BRAKE_BLOCK, read at smr_create
    int brake_block = 64;
    int item_blocks = 0;              // enumerate kernel: 256-lane blocks striding over the item list
    bool primed = false;              // staged start states exist (a reset, or state injection with reset counters, has run)
    DScene *dscene = nullptr;
    DState st;
    std::vector<void *> allocs;
    std::string err;
    bool profiling = false;     // events are recorded around the launches of the current call (smr_set_profiling; every profile_every-th step)
    int profile_every = 0;      // 0: off, 1: every step, n: every n-th step
    long long step_seq = 0;
    hipEvent_t ev[SMR_K_COUNT][2];
    bool ev_ok = false;
    struct Pending { int k; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[SMR_K_COUNT] = {};
    int64_t prof_n[SMR_K_COUNT] = {};
};

static thread_local std::string g_create_error;

struct smr_mlp {
    int device = 0;
    DMlp d;
    std::vector<void *> allocs;
};

#define SMR_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            env->err = std::string(#call) + ": " + hipGetErrorString(_e);                               \
            return SMR_E_HIP;                                                                           \
        }                                                                                               \
    } while (0)

template <class T>
static int dev_alloc(smr_env *env, T **out, size_t count) {
    void *p = nullptr;
    SMR_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    env->allocs.push_back(p);
    *out = (T *)p;
    return SMR_OK;
}

template <class T>
static int dev_upload(smr_env *env, T **out, const std::vector<T> &host) {
    int rc = dev_alloc(env, out, host.size());
    if (rc) return rc;
    if (!host.empty()) SMR_HIP(hipMemcpy(*out, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return SMR_OK;
}

static void copy3(float *dst, const double *src, int n = 3) { for (int i = 0; i < n; ++i) dst[i] = (float)src[i]; }

static int build_device_scene(smr_env *env, const smr_scene_t &s, uint64_t seed, uint32_t env_index_base) {
    DScene &d = env->hscene;
    std::memset(&d, 0, sizeof d);
    d.B = env->B; d.obs_size = s.obs_size; d.episode_steps = s.episode_steps; d.updates_per_step = s.updates_per_step;
    d.checks_per_step = s.collision_checks_per_step; d.auto_reset = s.auto_reset; d.dt = (float)s.dt;
    copy3(d.pos_lo, s.pos_lo, SMR_NJ); copy3(d.pos_hi, s.pos_hi, SMR_NJ); copy3(d.vel_max, s.vel_max, SMR_NJ);
    copy3(d.acc_max, s.acc_max, SMR_NJ); copy3(d.jerk_max, s.jerk_max, SMR_NJ);
    for (int j = 0; j < SMR_NJ; ++j) {
        d.pos_lo_res[j] = (float)(s.pos_lo[j] - (double)d.pos_lo[j]);
        d.pos_hi_res[j] = (float)(s.pos_hi[j] - (double)d.pos_hi[j]);
        d.vel_max_res[j] = (float)(s.vel_max[j] - (double)d.vel_max[j]);
    }
    copy3(d.base, s.base_xyz); copy3(d.ee_offset, s.ee_offset);
    for (int j = 0; j < SMR_NJ; ++j) { copy3(d.joint_xyz[j], s.joint_xyz[j]); copy3(d.joint_axis[j], s.joint_axis[j]); }

    HostGeometry G;
    if (const char *msg = build_host_geometry(s, G)) { env->err = msg; return SMR_E_INVALID; }
    if (const char *msg = scene_bodies(s, d)) { env->err = msg; return SMR_E_INVALID; }
    for (int l = 0; l < s.n_bodies; ++l)
        if (G.pieces[s.link_piece[l]].ids_count > kMaxPieceIdGroups) { env->err = "link hull adjacency exceeds the LDS budget"; return SMR_E_INVALID; }
    std::vector<DPiece> &pieces = G.pieces;
    float4 *dverts; DPiece *dpieces; uint32_t *dspan, *dids; uint8_t *dstart;
    int rc;
    while (G.start.size() % 4) G.start.push_back(0);
    if ((rc = dev_upload(env, &dverts, G.verts)) || (rc = dev_upload(env, &dpieces, G.pieces)) || (rc = dev_upload(env, &dspan, G.span)) ||
        (rc = dev_upload(env, &dids, G.ids4)) || (rc = dev_upload(env, &dstart, G.start))) return rc;
    d.verts = dverts; d.pieces = dpieces; d.span = dspan; d.ids4 = dids; d.start = dstart;
    auto piece_ok = [&](int p) { return p >= 0 && p < s.n_pieces; };
    if (s.n_static < 0 || s.n_static > SMR_MAX_STATIC) { env->err = "bad n_static"; return SMR_E_INVALID; }
    d.n_static = s.n_static;
    for (int i = 0; i < s.n_static; ++i) { if (!piece_ok(s.static_piece[i])) { env->err = "bad static_piece"; return SMR_E_INVALID; } d.static_piece[i] = s.static_piece[i]; }
    d.static_mask = reachable_static_mask(s); d.static_max = (float)s.static_max_distance;
    d.moving_mask = s.moving_link_mask; d.contact_mask = reachable_contact_mask(s);
    d.moving_max = (float)s.moving_max_distance; d.contact_dist = (float)s.contact_distance;
    if (s.n_planets < 0 || s.n_planets > SMR_MAX_PLANETS) { env->err = "bad n_planets"; return SMR_E_INVALID; }
    d.n_planets = s.n_planets; d.obs_planet_size = s.obs_planet_size_per_planet; d.obs_num_planets = s.obs_num_planets;
    for (int p = 0; p < s.n_planets; ++p) {
        const smr_planet_t &pl = s.planet[p];
        if (pl.n_orbit <= 0 || !pl.orbit_pose || !pl.orbit_local || !piece_ok(pl.piece_begin) || pl.piece_count <= 0 ||
            !piece_ok(pl.piece_begin + pl.piece_count - 1)) { env->err = "bad planet table"; return SMR_E_INVALID; }
        std::vector<float> pose((size_t)pl.n_orbit * 8, 0.f), local((size_t)pl.n_orbit * 2);
        for (int i = 0; i < pl.n_orbit; ++i) {
            for (int c = 0; c < 7; ++c) pose[(size_t)i * 8 + c] = (float)pl.orbit_pose[(size_t)i * 7 + c];
            local[(size_t)i * 2] = (float)pl.orbit_local[(size_t)i * 2]; local[(size_t)i * 2 + 1] = (float)pl.orbit_local[(size_t)i * 2 + 1];
        }
        float *dpose, *dlocal;
        if ((rc = dev_upload(env, &dpose, pose))) return rc;
        if ((rc = dev_upload(env, &dlocal, local))) return rc;
        DPlanet &q = d.planet[p];
        q.piece_begin = pl.piece_begin; q.piece_count = pl.piece_count; q.n_orbit = pl.n_orbit; q.index_shift = pl.index_shift;
        q.coupled = pl.coupled; q.radius_x = (float)pl.radius_xy[0]; q.radius_y = (float)pl.radius_xy[1];
        q.orbit_pose = dpose; q.orbit_local = dlocal;
        planet_bounds(pieces, q);
        d.planet_step[p] = planet_step_bound(G, q, pose.data());
    }
    sweep_tables(s, G, d);
    copy3(d.planet_obs_min, s.planet_obs_pos_min); copy3(d.planet_obs_max, s.planet_obs_pos_max);
    d.use_ball = s.use_ball; d.dt64 = s.dt;
    planet_tables(G, d);
    if (const char *v = getenv("SMR_LOCKSTEP_PLANETS")) if (atoi(v)) d.planets_auto = 0;     // A/B switch
    {   // the reset-side kernels' LDS pool with every body's hull tables
        std::vector<uint32_t> body_pool;
        if (!body_pool_tables(s, G, body_pool)) { env->err = "the bodies' hull tables exceed the LDS pool of the reset kernels"; return SMR_E_INVALID; }
        uint32_t *dpool;
        if ((rc = dev_upload(env, &dpool, body_pool))) return rc;
        d.body_pool = dpool; d.body_pool_dwords = (int)body_pool.size();
    }
    {   // dispatch order of the bodies in the collide grid: the bodies with planet queries first, the ones with most work
        // (mounted pieces, then iiwa_link_5, 6, 4, 7, 3: measured) leading, so that the lightest of them start last
        int n = 0;
        for (int pass = 0; pass < 2; ++pass) {
            const bool want_moving = pass == 0;
            for (int b = d.n_bodies - 1; b >= SMR_NLINK; --b) if ((((d.moving_mask >> b) & 1u) != 0) == want_moving) d.body_order[n++] = b;
            const int chain[SMR_NLINK] = {5, 6, 4, 7, 3, 2, 1, 0};
            for (int i = 0; i < SMR_NLINK; ++i) if ((((d.moving_mask >> chain[i]) & 1u) != 0) == want_moving) d.body_order[n++] = chain[i];
        }
    }
    {   // the pairs by partner lane (reset-side kernels)
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < 8; ++g) for (int k = 0; k < 4; ++k) d.pair_lane[g][k] = -1;
        bool fits = true;
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < d.n_self_pairs; ++i) {
                const int partner = d.self_pair[i][1];
                if ((partner >= 0) != (pass == 0)) continue;
                int g = partner;
                if (partner < 0) { g = 0; for (int c = 1; c < 8; ++c) if (cnt[c] < cnt[g]) g = c; }      // a world-fixed partner: the emptiest lane
                if (cnt[g] >= 4) { fits = false; continue; }
                d.pair_lane[g][cnt[g]++] = i;
            }
        if (!fits) { env->err = "more than four self-collision pairs share one partner link"; return SMR_E_INVALID; }
        d.pair_lane_max = 0;
        for (int g = 0; g < 8; ++g) d.pair_lane_max = std::max(d.pair_lane_max, cnt[g]);
    }
    {   // ... and of the self-collision pairs: highest partner link (nearest to the mounted bodies) first, the world-fixed piece last
        std::vector<int> order(d.n_self_pairs);
        for (int i = 0; i < d.n_self_pairs; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d.self_pair[a][1] > d.self_pair[b][1]; });
        for (int i = 0; i < d.n_self_pairs; ++i) d.pair_order[i] = order[i];
    }
    if (const char *v = getenv("SMR_BODY_ORDER")) {                                          // experiment: e.g. 54367210
        int n = 0;
        for (const char *c = v; *c && n < d.n_bodies; ++c) if (*c >= '0' && *c - '0' < d.n_bodies) d.body_order[n++] = *c - '0';
    }
    if (d.planets_auto) {
        float4 *dav; float *dad;
        if ((rc = dev_upload(env, &dav, G.auto_verts)) || (rc = dev_upload(env, &dad, G.auto_desc))) return rc;
        d.auto_verts = (const float *)dav; d.auto_desc = dad;
        env->host_planet_padded.clear();
        for (size_t c = 0; c * kPlanetDescDwords < G.auto_desc.size(); ++c) { int padded; std::memcpy(&padded, &G.auto_desc[c * kPlanetDescDwords + 1], 4); env->host_planet_padded.push_back(padded); }
        // pair-job path (the default in planet scenes; SMR_COLLIDE_LANES=1: the lane-per-(env, body) collide kernel, for A/B runs)
        const char *lanes = getenv("SMR_COLLIDE_LANES");
        d.planets_jobs = !(lanes && atoi(lanes));
    }
    {   // the job kernels' LDS pool: the hull tables of the bodies that take part in moving / contact / static queries (pair jobs) and
        // of the bodies the self-collision pairs name; behind it the planets' vertices and descriptors and the pairs' cylinders
        uint32_t self_mask = 0u;
        for (int i = 0; i < d.n_self_pairs; ++i) { self_mask |= 1u << d.self_pair[i][0]; if (d.self_pair[i][1] >= 0) self_mask |= 1u << d.self_pair[i][1]; }
        // ball scenes (round 4): the GJK runs against the ball can be jobs as well - its padded vertex list takes the planets' place in the
        // tables.  The default where the scene has self-collision pairs, whose jobs need a launch of the jobs kernel anyway (ball machine
        // mounted, 65 536 envs: 72.3 -> 82.3 M env-steps/s); without them the launch costs what the collide kernel saves (its span is the
        // longest run of the batch, 0.16 ms: 139 -> 127 M), so the runs stay in the collide kernel's lanes.  SMR_BALL_JOBS=0 / 1: A/B switch
        d.ball_jobs = 0; d.ball_jobs_vn = 0;
        if (s.use_ball && !d.planets_auto && s.ball_piece >= 0 && s.ball_piece < s.n_pieces) {
            d.ball_jobs = d.n_self_pairs > 0;
            if (const char *v = getenv("SMR_BALL_JOBS")) d.ball_jobs = atoi(v) != 0;
            if (const char *lanes = getenv("SMR_COLLIDE_LANES")) if (atoi(lanes)) d.ball_jobs = 0;
            d.ball_jobs_vn = (G.pieces[s.ball_piece].count + 7) & ~7;
        }
        auto lds_bytes = [&](uint32_t mask, bool planets, std::vector<uint32_t> &pool) {
            body_pool_tables(s, G, pool, mask);
            return 4 * (pool.size() + (planets ? 4 * (size_t)d.auto_vn + kPlanetDescDwords * (size_t)d.auto_pn : 0) + (d.ball_jobs ? 4 * (size_t)d.ball_jobs_vn : 0) + kJobCylDwords);
        };
        std::vector<uint32_t> job_pool;
        const uint32_t pair_mask = d.static_mask | d.moving_mask | d.contact_mask;
        size_t bytes = lds_bytes(((d.planets_jobs || d.ball_jobs) ? pair_mask : 0u) | self_mask, d.planets_jobs != 0, job_pool);
        if ((d.planets_jobs || d.ball_jobs) && bytes > 64 * 1024) {             // does not fit two blocks per CU: the planets / the ball stay with the lane form
            d.planets_jobs = 0; d.ball_jobs = 0;
            bytes = lds_bytes(self_mask, false, job_pool);
        }
        if (d.n_self_pairs > 0 && bytes > 64 * 1024) { env->err = "the hull tables of the self-collision pairs' bodies exceed the LDS pool of the job kernels"; return SMR_E_INVALID; }
        d.jobs_on = (d.planets_jobs || d.ball_jobs || d.n_self_pairs > 0) ? 1 : 0;
        if (d.jobs_on) {
            uint32_t *dpool;
            if ((rc = dev_upload(env, &dpool, job_pool))) return rc;
            d.job_pool = dpool; d.job_pool_dwords = (int)job_pool.size();
            env->job_lds_bytes = bytes;
        }
    }
    if (s.use_ball) {
        if (!piece_ok(s.ball_piece)) { env->err = "bad ball_piece"; return SMR_E_INVALID; }
        d.ball_piece = s.ball_piece; d.ball_check_invalid = s.ball_check_invalid_target; d.ball_random_initial = s.ball_random_initial_position;
        d.ball_radius = (float)s.ball_radius; d.ball_active_xy_radius = (float)s.ball_active_xy_radius;
        copy3(d.ball_obs_pos_min, s.ball_obs_pos_min); copy3(d.ball_obs_pos_max, s.ball_obs_pos_max);
        copy3(d.ball_obs_vel_min, s.ball_obs_vel_min); copy3(d.ball_obs_vel_max, s.ball_obs_vel_max);
        for (int c = 0; c < 3; ++c) {
            d.ball_target_box[c][0] = (float)s.ball_target_box[c][0]; d.ball_target_box[c][1] = (float)s.ball_target_box[c][1];
            d.ball_center[c] = s.ball_sphere_center[c]; d.ball_invalid_min[c] = s.ball_invalid_min[c]; d.ball_invalid_max[c] = s.ball_invalid_max[c];
            d.ball_final_min[c] = s.ball_final_min[c]; d.ball_final_max[c] = s.ball_final_max[c];
        }
        d.ball_target_min_static = (float)s.ball_target_min_dist_static; d.ball_target_min_self = (float)s.ball_target_min_dist_self; d.ball_speed = s.ball_speed; d.ball_sphere_radius = s.ball_sphere_radius;
        d.ball_height[0] = s.ball_sphere_height[0]; d.ball_height[1] = s.ball_sphere_height[1];
        d.ball_angle[0] = s.ball_sphere_angle[0]; d.ball_angle[1] = s.ball_sphere_angle[1]; d.ball_high_prob = s.ball_high_angle_prob;
        d.plane_z = s.plane_z; d.table_half_xy[0] = s.table_half_xy[0]; d.table_half_xy[1] = s.table_half_xy[1];
    }
    d.use_target = s.use_target_points; d.obs_target_pos = s.obs_target_pos; d.obs_target_rel = s.obs_target_rel;
    d.obs_target_size = s.use_target_points ? 3 * ((s.obs_target_pos ? 1 : 0) + (s.obs_target_rel ? 1 : 0)) : 0;
    d.target_normalize = s.target_normalize; d.ca_mode = s.collision_avoidance_mode;
    d.target_min_static = (float)s.target_min_dist_static; d.target_min_self = (float)s.target_min_dist_self;
    d.target_radius = (float)s.target_radius; d.target_bonus = (float)s.target_bonus; d.target_factor = (float)s.target_factor;
    copy3(d.target_rel_min, s.target_rel_min); copy3(d.target_rel_max, s.target_rel_max);
    if (s.use_target_points && !s.use_ball)
        for (int c = 0; c < 3; ++c) { d.ball_target_box[c][0] = (float)s.ball_target_box[c][0]; d.ball_target_box[c][1] = (float)s.ball_target_box[c][1]; }
    d.punish_action = s.punish_action; d.term_self = s.terminate_on_self_collision; d.term_static = s.terminate_on_static_collision;
    d.term_moving = s.terminate_on_moving_collision;
    d.action_thr = (float)s.action_punishment_min_threshold; d.action_max_pun = (float)s.action_max_punishment;
    d.self_max_reward = (float)s.self_max_reward; d.self_max_dist = (float)s.self_max_reward_distance;
    d.static_max_reward = (float)s.static_max_reward; d.static_max_dist = (float)s.static_max_reward_distance;
    d.moving_max_reward = (float)s.moving_max_reward; d.moving_max_dist = (float)s.moving_max_reward_distance;
    d.low_acc_reward = (float)s.low_acc_max_reward; d.low_acc_thr = (float)s.low_acc_threshold;
    d.low_vel_reward = (float)s.low_vel_max_reward; d.low_vel_thr = (float)s.low_vel_threshold;
    d.term_bonus = (float)s.termination_bonus; d.early_term_pun = (float)s.early_termination_punishment;
    d.collision_dist = (float)s.collision_distance;
    for (int c = 0; c < 3; ++c) { d.start_box[c][0] = (float)s.start_box[c][0]; d.start_box[c][1] = (float)s.start_box[c][1]; }
    d.start_min_static = (float)s.start_min_dist_static; d.start_min_self = (float)s.start_min_dist_self;
    d.check_braking = s.check_braking_trajectory_collisions; d.brake_checks = s.brake_checks_per_step; d.brake_timeout_steps = s.brake_timeout_steps;
    d.brake_safety = (float)s.brake_safety_distance; d.brake_stop_v = (float)s.brake_stop_velocity; d.brake_stop_a = (float)s.brake_stop_acceleration;
    copy3(d.brake_acc_max, s.brake_acc_max, SMR_NJ); copy3(d.brake_jerk_max, s.brake_jerk_max, SMR_NJ);
    d.kin_mode = s.kinematic_state_sampling_mode; d.kin_prob = (float)s.kinematic_state_sampling_probability;
    d.stay_prob = (float)s.stay_in_state_probability;
    d.seed_lo = (unsigned)seed; d.seed_hi = (unsigned)(seed >> 32); d.env_base = env_index_base;
    // the range of self distances anything reads: the reward's min(1, d / max_reward_distance) and the d < collision_distance flag
    // (rewards.py:119-126); never beyond the cut-off of the queries (ctlp.py:354-366)
    d.punish_adaptation = s.punish_adaptation; d.punish_end_min = s.punish_end_min_distance; d.punish_bt_min = s.punish_braking_trajectory_min_distance;
    d.adaptation_max_pun = (float)s.adaptation_max_punishment; d.end_min_max_pun = (float)s.end_min_distance_max_punishment;
    d.end_min_thr = (float)s.end_min_distance_max_threshold; d.bt_max_pun = (float)s.braking_trajectory_max_punishment;
    d.bt_thr = (float)s.braking_trajectory_min_distance_max_threshold;
    d.self_cut = std::min(d.static_max, std::max(std::max(d.self_max_reward != 0.f ? d.self_max_dist : 0.f, d.collision_dist),
                                                 d.punish_end_min ? d.end_min_thr : 0.f));      // ... and the end-of-step minimum distance (rewards.py:322-332)
    if ((rc = dev_alloc(env, &env->dscene, 1))) return rc;
    // self-pair bound tables: laid out here, filled by the device GJK itself (the scene goes up without them first)
    const size_t tab_entries = getenv("SMR_NO_SELF_TABLES") ? 0 : self_table_layout(s, G, d);
    if (tab_entries > 0) {
        uint16_t *tab;
        if ((rc = dev_alloc(env, &tab, tab_entries))) return rc;
        SMR_HIP(hipMemcpy(env->dscene, &d, sizeof d, hipMemcpyHostToDevice));        // layout in, self_tab still null: the builder makes the literal queries
        for (int i = 0; i < d.n_self_pairs; ++i) {
            if (d.self_tab_nd[i] == 0) continue;
            int cells = 1;
            for (int k = 0; k < d.self_tab_nd[i]; ++k) cells *= d.self_tab_n[i][k];
            hipLaunchKernelGGL(self_table_kernel, dim3((cells + 255) / 256), dim3(256), 0, nullptr, env->dscene, i, tab + d.self_tab_off[i], cells);
        }
        SMR_HIP(hipGetLastError());
        SMR_HIP(hipDeviceSynchronize());
        d.self_tab = tab;
    } else {
        for (int i = 0; i < SMR_MAX_SELF_PAIRS; ++i) d.self_tab_nd[i] = 0;
    }
    SMR_HIP(hipMemcpy(env->dscene, &d, sizeof d, hipMemcpyHostToDevice));
    return SMR_OK;
}

static int check_scene(const smr_scene_t *s, int32_t num_envs, std::string &err) {
    if (!s) { err = "scene is NULL"; return SMR_E_INVALID; }
    if (s->version != SMR_SCENE_VERSION) { err = "smr_scene_t version mismatch"; return SMR_E_INVALID; }
    if (num_envs <= 0 || num_envs > (1 << 26)) { err = "num_envs must be in 1 .. 2^26"; return SMR_E_INVALID; }
    if (s->collision_checks_per_step <= 0 || s->updates_per_step <= 0 || s->updates_per_step % s->collision_checks_per_step) {
        err = "collision_checks_per_step must divide updates_per_step"; return SMR_E_INVALID;
    }
    if (s->collision_checks_per_step > kMaxChecksPerStep) {      // the sample index of job words / item queue entries, the items per lane
        err = "collision_checks_per_step > 24 is not supported (trajectory_time_step > 0.1 s: pass a divisor of updates_per_step that is <= 24)"; return SMR_E_INVALID;
    }
    if (s->use_ball && s->n_planets > 0) { err = "ball and planets in one scene are not supported"; return SMR_E_INVALID; }
    if (!s->verts || !s->pieces || s->n_verts <= 0 || s->n_pieces <= 0) { err = "scene has no geometry"; return SMR_E_INVALID; }
    int expect = 3 * SMR_NJ + s->obs_num_planets * s->obs_planet_size_per_planet + (s->use_ball ? 6 : 0) +
                 (s->use_target_points ? 3 * ((s->obs_target_pos ? 1 : 0) + (s->obs_target_rel ? 1 : 0)) : 0);
    if (s->obs_size != expect) { err = "obs_size does not match the scene"; return SMR_E_INVALID; }
    if (!(s->dt > 0) || s->episode_steps <= 0) { err = "bad time parameters"; return SMR_E_INVALID; }
    if (s->check_braking_trajectory_collisions) {
        if (s->collision_avoidance_mode) { err = "check_braking_trajectory_collisions with collision_avoidance_mode is not supported (start-state sampling, ctlp.py:1499-1545)"; return SMR_E_INVALID; }
        if (s->brake_checks_per_step < 1 || s->brake_timeout_steps < 1 || !(s->brake_safety_distance >= 0)) { err = "bad braking parameters"; return SMR_E_INVALID; }
    }
    return SMR_OK;
}

extern "C" {

int smr_abi_version(void) { return SMR_ABI_VERSION; }
size_t smr_scene_sizeof(void) { return sizeof(smr_scene_t); }

const char *smr_last_error(const smr_env_t *env) { return env ? env->err.c_str() : g_create_error.c_str(); }

int smr_destroy(smr_env_t *env) {
    if (!env) return SMR_OK;
    (void)hipSetDevice(env->device);
    (void)hipStreamSynchronize(env->stream);
    if (env->range_stream) { (void)hipStreamSynchronize(env->range_stream); (void)hipStreamDestroy(env->range_stream); }
    if (env->ev_planned) (void)hipEventDestroy(env->ev_planned);
    if (env->ev_ranges) (void)hipEventDestroy(env->ev_ranges);
    if (env->side) { (void)hipStreamSynchronize(env->side); (void)hipStreamDestroy(env->side); }
    if (env->side2) { (void)hipStreamSynchronize(env->side2); (void)hipStreamDestroy(env->side2); }
    if (env->side_b) { (void)hipStreamSynchronize(env->side_b); (void)hipStreamDestroy(env->side_b); }
    if (env->side2_b) { (void)hipStreamSynchronize(env->side2_b); (void)hipStreamDestroy(env->side2_b); }
    for (hipEvent_t ev : {env->ev_commit_b, env->ev_refill_b, env->ev_fork_b, env->ev_join_b}) if (ev) (void)hipEventDestroy(ev);
    if (env->ev_fork) (void)hipEventDestroy(env->ev_fork);
    if (env->ev_join) (void)hipEventDestroy(env->ev_join);
    if (env->ev_commit) (void)hipEventDestroy(env->ev_commit);
    if (env->ev_refill) (void)hipEventDestroy(env->ev_refill);
    for (auto &p : env->pending) { env->event_pool.push_back(p.a); env->event_pool.push_back(p.b); }
    for (auto e : env->event_pool) (void)hipEventDestroy(e);
    for (void *p : env->allocs) (void)hipFree(p);
    delete env;
    return SMR_OK;
}

int smr_create(const smr_scene_t *scene, int32_t num_envs, uint32_t env_index_base, int32_t device, uint64_t seed, void *stream, smr_env_t **out) {
    if (!out) { g_create_error = "out is NULL"; return SMR_E_INVALID; }
    *out = nullptr;
    int rc = check_scene(scene, num_envs, g_create_error);
    if (rc) return rc;
    if ((uint64_t)env_index_base + (uint64_t)num_envs > 0xFFFFFFFFull) { g_create_error = "env_index_base + num_envs must fit 32 bits (the env word of the Philox counter)"; return SMR_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device: libsmr_hip has no CPU path (the CPU restatement lives in oracle/ and is test-only)";
        return SMR_E_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_create_error = "device ordinal out of range"; return SMR_E_INVALID; }
    smr_env *env = new smr_env();
    env->device = device; env->stream = (hipStream_t)stream; env->B = num_envs; env->obs_size = scene->obs_size;
    env->N = scene->collision_checks_per_step; env->auto_reset = scene->auto_reset;
    auto fail = [&](int code) { g_create_error = env->err; smr_destroy(env); return code; };
    if (hipSetDevice(device) != hipSuccess) { env->err = "hipSetDevice failed"; return fail(SMR_E_HIP); }
    if ((rc = build_device_scene(env, *scene, seed, env_index_base))) return fail(rc);
    const size_t B = (size_t)num_envs;
    DState &st = env->st;
    if ((rc = dev_alloc(env, &st.q, 7 * B)) || (rc = dev_alloc(env, &st.v, 7 * B)) || (rc = dev_alloc(env, &st.a, 7 * B)) ||
        (rc = dev_alloc(env, &st.planet_idx, 2 * B)) || (rc = dev_alloc(env, &st.step_count, B)) ||
        (rc = dev_alloc(env, &st.moving_hit, B)) || (rc = dev_alloc(env, &st.reset_count, B)) || (rc = dev_alloc(env, &st.start_attempts, B)) || (rc = dev_alloc(env, &st.a1, 7 * B)) ||
        (rc = dev_alloc(env, &st.range_lo, 7 * B)) || (rc = dev_alloc(env, &st.range_hi, 7 * B)) || (rc = dev_alloc(env, &st.range_valid, B)) ||
        (rc = dev_alloc(env, &st.res, (size_t)scene->n_bodies * B)) || (rc = dev_alloc(env, &st.lb_moving, (size_t)scene->n_bodies * B)) || (rc = dev_alloc(env, &st.braked, B)) || (rc = dev_alloc(env, &st.brake_steps, B)) || (rc = dev_alloc(env, &st.bt_min, B)) || 
        (rc = dev_alloc(env, &st.brake_acc, scene->check_braking_trajectory_collisions ? (size_t)(scene->brake_timeout_steps + 1) * SMR_NJ * B : 1)) || (rc = dev_alloc(env, &st.dist, (size_t)(2 * scene->n_bodies + scene->n_self_pairs) * B)) ||
        (rc = dev_alloc(env, &st.done_list, B)) || (rc = dev_alloc(env, &st.done_count, 1)) || (rc = dev_alloc(env, &st.ball, 10 * B)) ||
        (rc = dev_alloc(env, &st.ball_i, 5 * B)) || (rc = dev_alloc(env, &env->spawn_lists, 3 * B)) || (rc = dev_alloc(env, &env->spawn_counts, 3)) || (rc = dev_alloc(env, &env->spawn_tag_lists, 3 * B)))
        return fail(rc);
    if (hipMemset(st.ball, 0, 10 * B * 4) || hipMemset(st.ball_i, 0, 5 * B * 4) || hipMemset(env->spawn_counts, 0, 12) || hipMemset(st.lb_moving, 0xFF, (size_t)scene->n_bodies * B * 4) || hipMemset(st.braked, 0, B * 4)) { env->err = "hipMemset failed"; return fail(SMR_E_HIP); }
    st.spawn_list = env->spawn_lists; st.spawn_count = env->spawn_counts; st.spawn_tags = env->spawn_tag_lists;
    hipError_t e1 = hipMemset(st.q, 0, 7 * B * 4), e2 = hipMemset(st.v, 0, 7 * B * 4), e3 = hipMemset(st.a, 0, 7 * B * 4),
               e4 = hipMemset(st.planet_idx, 0, 2 * B * 4), e5 = hipMemset(st.step_count, 0, B * 4), e6 = hipMemset(st.moving_hit, 0, B * 4),
               e7 = hipMemset(st.reset_count, 0, B * 4) ? hipErrorUnknown : hipMemset(st.start_attempts, 0, B * 4), e8 = hipMemset(st.done_count, 0, 4), e9 = hipMemset(st.a1, 0, 7 * B * 4);
    if (e1 || e2 || e3 || e4 || e5 || e6 || e7 || e8 || e9 || hipMemset(st.range_valid, 0, B * 4)) { env->err = "hipMemset failed"; return fail(SMR_E_HIP); }
    if (const char *v = getenv("SMR_RANGE_AHEAD")) env->range_mode = atoi(v) != 0 ? 1 : 0;
    if (const char *v = getenv("SMR_RANGE_HOST")) env->range_host = atoi(v) != 0;
    if (const char *v = getenv("SMR_SELF_ROWS")) env->self_rows = atoi(v) != 0;
    if (const char *v = getenv("SMR_BRAKE_BLOCK")) if (atoi(v) == 64 || atoi(v) == 128 || atoi(v) == 192 || atoi(v) == 256) env->brake_block = atoi(v);
    if (env->hscene.jobs_on) {
        const size_t NB = (size_t)scene->n_bodies, N = (size_t)scene->collision_checks_per_step;
        const bool pj = env->hscene.planets_jobs != 0;
        // job list: 64 entries per env - uniformly drawn states make 4.6 pair jobs, 0.3 self-pair jobs and 0.06 table runs per env.  A
        // step that needs more hands the envs concerned to overflow_kernel.
        size_t cap = B * 64 + 65536;
        if (const char *v = getenv("SMR_JOB_CAP")) cap = (size_t)std::max(0, atoi(v));       // tests: force the full-list path
        if (cap > 0x7FFFFFFFull) { env->err = "num_envs too large for the job list"; return fail(SMR_E_INVALID); }
        st.job_cap = (int)cap;
        // the long scans: a quarter of the list.  One list for small batches, where the step is a chain of latencies and the second counter's
        // round trip costs more than the shorter scans save (4096 envs: enumerate 0.020 -> 0.024 ms, measured); SMR_JOB_ONE_LIST=0 / 1: A/B switch
        bool one_list = B <= 16384;
        if (const char *v = getenv("SMR_JOB_ONE_LIST")) one_list = atoi(v) != 0;
        st.job_split = one_list ? (int)cap : (int)(cap - cap / 4);
        st.big_piece_mask = 0u;
        if (pj && !one_list)
            for (int c = 0; c < env->hscene.planet_pn && c < 32 && c < (int)env->host_planet_padded.size(); ++c)
                if (env->host_planet_padded[c] > kSmallScan) st.big_piece_mask |= 1u << c;
        if ((rc = dev_alloc(env, &st.keyw, pj ? 7 * NB * B : 1)) || (rc = dev_alloc(env, &st.env_best, pj ? B : 1)) || (rc = dev_alloc(env, &st.env_min, (pj || env->hscene.ball_jobs) ? B : 1)) ||
            (rc = dev_alloc(env, &st.env_dstatic, pj ? B : 1)) || (rc = dev_alloc(env, &st.frames, pj ? 12 * NB * B : 4)) || (rc = dev_alloc(env, &st.lb1, pj ? NB * B : 1)) || (rc = dev_alloc(env, &st.env_dself, B)) || (rc = dev_alloc(env, &st.items, (pj && N > 1) ? B * NB * (N - 1) : 1)) ||
            (rc = dev_alloc(env, &st.jobs, std::max<size_t>(cap, 1))) || (rc = dev_alloc(env, &st.item_count, 1)) || (rc = dev_alloc(env, &st.job_count, 1)) || (rc = dev_alloc(env, &st.job_count_big, 1)) || (rc = dev_alloc(env, &st.job_next, 1)) ||
            (rc = dev_alloc(env, &st.overflow_env, B)) || (rc = dev_alloc(env, &st.defer_count, 1))) return fail(rc);
        st.defer_cap = (int)std::min<size_t>(4 * B + 4096, 1u << 22);      // up to ~2 deferred runs per env when every run leaves after its first iteration; a full list defers no more
        if ((rc = dev_alloc(env, &st.defer, (size_t)st.defer_cap * kDeferWords))) return fail(rc);
        if (hipMemset(st.defer_count, 0, 4)) { env->err = "hipMemset failed"; return fail(SMR_E_HIP); }
        if (hipMemset(st.item_count, 0, 4) || hipMemset(st.job_count, 0, 4) || hipMemset(st.job_count_big, 0, 4) || hipMemset(st.job_next, 0, 4) || hipMemset(st.overflow_env, 0, B * 4)) { env->err = "hipMemset failed"; return fail(SMR_E_HIP); }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { env->err = "hipGetDeviceProperties failed"; return fail(SMR_E_HIP); }
        // one 512-lane block per CU at the full batch; two up to 32 768 envs, where the kernel is the chain of its longest wavefront and
        // thinner shares shorten it (measured, blocks per CU 1 -> 2: 4096 envs 0.119 -> 0.106 ms with SMR_JOB_QUOTA_MIN 4, 8192 envs
        // 0.140 -> 0.128, 16 384 0.171 -> 0.153, 32 768 0.206 -> 0.191; 65 536: 0.217 -> 0.224 ms in the space scene, 0.173 -> 0.162 in the
        // reaching task; 256-lane blocks - one wavefront per SIMD - at 4096 envs: 0.119 -> 0.142 ms)
        env->job_blocks = std::max(prop.multiProcessorCount, 1) * (B <= 32768 ? 2 : 1);
        // ball jobs (1.3 per env, runs of up to 31 iterations): one block per two CUs - denser wavefronts, fewer of them beside the refill
        // kernels (65 536 envs, ball machine mounted: jobs 0.270 -> 0.230 ms; 96 / 192 blocks: 0.242 / 0.262)
        if (env->hscene.ball_jobs && B > 32768) env->job_blocks = std::max(prop.multiProcessorCount / 2, 1);
        if (const char *v = getenv("SMR_JOB_BLOCKS")) if (atoi(v) > 0) env->job_blocks = atoi(v);
        if (const char *v = getenv("SMR_JOB_BLOCK_LANES")) if (atoi(v) == 256 || atoi(v) == 512 || atoi(v) == 128) env->job_block_lanes = atoi(v);
        if (const char *v = getenv("SMR_JOB_QUOTA_MAX")) env->job_quota_max = std::max(4, std::min(64, atoi(v)));
        if (const char *v = getenv("SMR_JOB_DEFER")) env->defer_after = std::max(0, std::min(31, atoi(v)));
        env->coop_blocks = std::max(prop.multiProcessorCount, 1);
        if (const char *v = getenv("SMR_COOP_BLOCKS")) if (atoi(v) > 0) env->coop_blocks = atoi(v);
        env->self_job_blocks = std::max(8, std::min(env->job_blocks, (int)((B + 1023) / 1024)));      // ~40 jobs per wavefront
        if (const char *v = getenv("SMR_SELF_JOB_BLOCKS")) if (atoi(v) > 0) env->self_job_blocks = atoi(v);
        // the item part of the enumerate grid: one round for the usual 1.2 items per (env, body) - every round of a wavefront ends with a
        // round trip to the list counter (SMR_ITEM_BLOCKS: blocks per CU, A/B runs)
        {
            const int cu = std::max(prop.multiProcessorCount, 1);
            const long long want = (3LL * (long long)B * NB / 2 + 255) / 256;
            env->item_blocks = (int)std::min<long long>(16LL * cu, std::max<long long>(cu, want));
            if (const char *v = getenv("SMR_ITEM_BLOCKS")) if (atoi(v) > 0) env->item_blocks = atoi(v) * cu;
        }
    }
    if ((rc = dev_alloc(env, &st.episodes, 1))) return fail(rc);
    if (hipMemset(st.episodes, 0, 8)) { env->err = "hipMemset failed"; return fail(SMR_E_HIP); }
    if ((rc = dev_alloc(env, &st.tp, 5 * B)) || (rc = dev_alloc(env, &st.tp_count, B)) || (rc = dev_alloc(env, &st.tp_list, B)) ||
        (rc = dev_alloc(env, &st.tp_list_count, 1)) || (rc = dev_alloc(env, &st.error_flags, 1))) return fail(rc);
    if (hipMemset(st.tp, 0, 5 * B * 4) || hipMemset(st.tp_count, 0, B * 4) || hipMemset(st.tp_list_count, 0, 4) || hipMemset(st.error_flags, 0, 4)) { env->err = "hipMemset failed"; return fail(SMR_E_HIP); }
    // staged start states (robot part; in ball scenes spawn_kernel draws the ball at commit time)
    {
        const char *off = std::getenv("SMR_SYNC_RESET");          // debugging / A-B measurement: reset on the critical path
        env->staged = !(off && off[0] == '1');
    }
    if (const char *v = getenv("SMR_SIDE_LDS_PAD")) {
        env->side_lds_pad = (size_t)std::max(0, std::min(96 * 1024, atoi(v)));
        const int total = (int)env->side_lds_pad;
        const void *fns[] = {(const void *)reset_kernel<true, true>, (const void *)reset_kernel<true, false>, (const void *)spawn_kernel_extra<true>, (const void *)spawn_kernel_plain<true>,
                             (const void *)target_kernel<true, true>, (const void *)target_kernel<false, true>};
        for (const void *f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, total);
    }
    if (env->staged) {
        if ((rc = dev_alloc(env, &st.next_q, 14 * B)) || (rc = dev_alloc(env, &st.next_v, 14 * B)) || (rc = dev_alloc(env, &st.next_a, 14 * B)) ||
            (rc = dev_alloc(env, &st.next_lo, 14 * B)) || (rc = dev_alloc(env, &st.next_hi, 14 * B)) ||
            (rc = dev_alloc(env, &st.next_planet_idx, 4 * B)) || (rc = dev_alloc(env, &st.next_hit, 2 * B)) || (rc = dev_alloc(env, &st.next_attempts, 2 * B)) || (rc = dev_alloc(env, &st.staged_tag, 2 * B)) ||
            (rc = dev_alloc(env, &env->refill_lists, 3 * B)) || (rc = dev_alloc(env, &env->refill_tag_lists, 3 * B)) ||
            (rc = dev_alloc(env, &env->refill_slots, 3)) || (rc = dev_alloc(env, &st.stale_commits, 1)))
            return fail(rc);
        if (hipMemset(st.staged_tag, 0xFF, 2 * B * 4) || hipMemset(env->refill_slots, 0, 12) || hipMemset(st.stale_commits, 0, 4) ||
            hipMemset(st.next_planet_idx, 0, 4 * B * 4) || hipMemset(st.next_hit, 0, 2 * B * 4)) { env->err = "hipMemset failed"; return fail(SMR_E_HIP); }
        st.refill_list = env->refill_lists; st.refill_tags = env->refill_tag_lists; st.refill_count = env->refill_slots; st.refill_zero = nullptr;
        if (scene->use_ball) {
            if ((rc = dev_alloc(env, &st.next_ball, 20 * B)) || (rc = dev_alloc(env, &st.next_ball_i, 4 * B)) || (rc = dev_alloc(env, &st.next_ball_tag, 2 * B))) return fail(rc);
            if (hipMemset(st.next_ball_tag, 0xFF, 2 * B * 4)) { env->err = "hipMemset failed"; return fail(SMR_E_HIP); }
            env->stage_balls = true;
        }
        if (scene->use_target_points) {
            if ((rc = dev_alloc(env, &st.next_tp, 6 * B)) || (rc = dev_alloc(env, &st.next_tp_tag, 2 * B)) || (rc = dev_alloc(env, &env->tp_refill_lists, 3 * B)) ||
                (rc = dev_alloc(env, &env->tp_refill_tag_lists, 3 * B)) || (rc = dev_alloc(env, &env->tp_refill_slots, 3))) return fail(rc);
            if (hipMemset(st.next_tp_tag, 0xFF, 2 * B * 4) || hipMemset(env->tp_refill_slots, 0, 12)) { env->err = "hipMemset failed"; return fail(SMR_E_HIP); }
            st.tp_refill_list = env->tp_refill_lists; st.tp_refill_tags = env->tp_refill_tag_lists; st.tp_refill_count = env->tp_refill_slots; st.tp_refill_zero = nullptr;
        }
        if (hipStreamCreateWithFlags(&env->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&env->ev_commit, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&env->ev_refill, hipEventDisableTiming) != hipSuccess) { env->err = "could not create the side stream"; return fail(SMR_E_HIP); }
        if (hipStreamCreateWithFlags(&env->side_b, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&env->ev_commit_b, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&env->ev_refill_b, hipEventDisableTiming) != hipSuccess) { env->err = "could not create the side stream"; return fail(SMR_E_HIP); }
        if (hipEventRecord(env->ev_refill, env->side) != hipSuccess || hipEventRecord(env->ev_refill_b, env->side_b) != hipSuccess) { env->err = "hipEventRecord failed"; return fail(SMR_E_HIP); }
        if ((env->tp_refill_slots || env->stage_balls) &&
            (hipStreamCreateWithFlags(&env->side2_b, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&env->ev_fork_b, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&env->ev_join_b, hipEventDisableTiming) != hipSuccess)) { env->err = "could not create the second side stream"; return fail(SMR_E_HIP); }
        if ((env->tp_refill_slots || env->stage_balls) &&
            (hipStreamCreateWithFlags(&env->side2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&env->ev_fork, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&env->ev_join, hipEventDisableTiming) != hipSuccess)) { env->err = "could not create the second side stream"; return fail(SMR_E_HIP); }
    }
    if (hipDeviceSynchronize() != hipSuccess) { env->err = "hipDeviceSynchronize failed"; return fail(SMR_E_HIP); }
    *out = env;
    return SMR_OK;
}

int32_t smr_num_envs(const smr_env_t *env) { return env ? env->B : 0; }
int32_t smr_obs_size(const smr_env_t *env) { return env ? env->obs_size : 0; }

}  // extern "C"

// ---- profiling helpers: a pair of events around one launch, resolved lazily ----
static hipEvent_t pool_get(smr_env *env) {
    if (!env->event_pool.empty()) { hipEvent_t e = env->event_pool.back(); env->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);                            // (hipEventReleaseToDevice makes no difference to what a pair costs, measured)
    return e;
}

struct LaunchTimer {
    smr_env *env; int k; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    LaunchTimer(smr_env *e, int kk, hipStream_t stream = nullptr) : env(e), k(kk), s(stream ? stream : e->stream) {
        if (env->profiling) { a = pool_get(env); b = pool_get(env); (void)hipEventRecord(a, s); }
    }
    ~LaunchTimer() {
        if (env->profiling) { (void)hipEventRecord(b, s); env->pending.push_back({k, a, b}); }
    }
};

static int resolve_profile(smr_env *env) {
    SMR_HIP(hipStreamSynchronize(env->stream));
    if (env->side) { SMR_HIP(hipStreamSynchronize(env->side)); SMR_HIP(hipStreamSynchronize(env->side_b)); }
    for (auto &p : env->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { env->prof_ms[p.k] += ms; env->prof_n[p.k] += 1; }
        env->event_pool.push_back(p.a); env->event_pool.push_back(p.b);
    }
    env->pending.clear();
    return SMR_OK;
}

// host [B][C] row-major <-> device [C][B]
template <class T>
static int state_get(smr_env *env, T *host, const T *dev, int C) {
    if (!host) return SMR_OK;
    const size_t B = (size_t)env->B;
    std::vector<T> tmp(B * C);
    SMR_HIP(hipMemcpy(tmp.data(), dev, tmp.size() * sizeof(T), hipMemcpyDeviceToHost));
    for (size_t e = 0; e < B; ++e) for (int c = 0; c < C; ++c) host[e * C + c] = tmp[(size_t)c * B + e];
    return SMR_OK;
}
template <class T>
static int state_set(smr_env *env, const T *host, T *dev, int C) {
    if (!host) return SMR_OK;
    const size_t B = (size_t)env->B;
    std::vector<T> tmp(B * C);
    for (size_t e = 0; e < B; ++e) for (int c = 0; c < C; ++c) tmp[(size_t)c * B + e] = host[e * C + c];
    SMR_HIP(hipMemcpy(dev, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
    return SMR_OK;
}

// ---- component entry points ----
struct TmpBuf {
    smr_env *env; std::vector<void *> ptrs;
    ~TmpBuf() { for (void *p : ptrs) (void)hipFree(p); }
    template <class T> T *up(const T *host, size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        if (host && n && hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return (T *)p;
    }
};

static bool scene_extra(const smr_env *env) { return env->hscene.n_bodies > SMR_NLINK || env->hscene.n_self_pairs > 0; }

// grid of a reset-side kernel: a group of 8 lanes per entry, every stride-th group works
static dim3 side_grid(int entries, int stride) { return dim3((unsigned)(((size_t)entries * 8 * stride + kSideBlock - 1) / kSideBlock)); }

template <bool STAGED>
static void launch_reset(smr_env *env, dim3 grid, hipStream_t stream, const int *list, const int *count_ptr, int count_fixed, int stride, float *obs,
                         const unsigned *tags = nullptr, int tag_offset = 0) {
    const size_t pad = STAGED ? env->side_lds_pad : 0;     // staged refills run beside the step: one block per CU leaves room for a jobs block (smr_create)
    if (scene_extra(env)) hipLaunchKernelGGL((reset_kernel<STAGED, true>), grid, dim3(kSideBlock), pad, stream, env->dscene, env->st, list, count_ptr, count_fixed, stride, obs, tags, tag_offset);
    else hipLaunchKernelGGL((reset_kernel<STAGED, false>), grid, dim3(kSideBlock), pad, stream, env->dscene, env->st, list, count_ptr, count_fixed, stride, obs, tags, tag_offset);
    hipLaunchKernelGGL(staged_range_kernel, dim3((unsigned)(((size_t)env->B * 8 + 255) / 256)), dim3(256), 0, stream, env->dscene, env->st, list, count_ptr, count_fixed, tags, tag_offset, STAGED ? 0 : 1);
}

static void launch_spawn(smr_env *env, dim3 grid, const int *list, const int *count_ptr, int count_fixed, int all_reset, int stride, float *obs) {
    if (scene_extra(env)) hipLaunchKernelGGL((spawn_kernel_extra<false>), grid, dim3(kSideBlock), 0, env->stream, env->dscene, env->st, list, count_ptr, count_fixed, all_reset, stride, obs, (const unsigned *)nullptr, 0);
    else hipLaunchKernelGGL((spawn_kernel_plain<false>), grid, dim3(kSideBlock), 0, env->stream, env->dscene, env->st, list, count_ptr, count_fixed, all_reset, stride, obs, (const unsigned *)nullptr, 0);
}
// the staged launch of the NEXT ball of the listed envs (after their current one was installed), on stream s
static void launch_spawn_staged(smr_env *env, hipStream_t s, dim3 grid, const int *list, const int *count_ptr, int count_fixed, int stride,
                                const unsigned *tags = nullptr, int tag_offset = 0) {
    if (scene_extra(env)) hipLaunchKernelGGL((spawn_kernel_extra<true>), grid, dim3(kSideBlock), env->side_lds_pad, s, env->dscene, env->st, list, count_ptr, count_fixed, 0, stride, (float *)nullptr, tags, tag_offset);
    else hipLaunchKernelGGL((spawn_kernel_plain<true>), grid, dim3(kSideBlock), env->side_lds_pad, s, env->dscene, env->st, list, count_ptr, count_fixed, 0, stride, (float *)nullptr, tags, tag_offset);
}

static void launch_target(smr_env *env, dim3 grid, const int *list, const int *count_ptr, int count_fixed, int stride, float *obs) {
    if (scene_extra(env)) hipLaunchKernelGGL((target_kernel<true, false>), grid, dim3(kSideBlock), 0, env->stream, env->dscene, env->st, list, count_ptr, count_fixed, stride, obs, (const unsigned *)nullptr, 0);
    else hipLaunchKernelGGL((target_kernel<false, false>), grid, dim3(kSideBlock), 0, env->stream, env->dscene, env->st, list, count_ptr, count_fixed, stride, obs, (const unsigned *)nullptr, 0);
}
// the staged next target of the listed envs (after their current one was installed), on stream s
static void launch_target_staged(smr_env *env, hipStream_t s, dim3 grid, const int *list, const int *count_ptr, int count_fixed, int stride,
                                 const unsigned *tags = nullptr, int tag_offset = 0) {
    if (scene_extra(env)) hipLaunchKernelGGL((target_kernel<true, true>), grid, dim3(kSideBlock), env->side_lds_pad, s, env->dscene, env->st, list, count_ptr, count_fixed, stride, (float *)nullptr, tags, tag_offset);
    else hipLaunchKernelGGL((target_kernel<false, true>), grid, dim3(kSideBlock), env->side_lds_pad, s, env->dscene, env->st, list, count_ptr, count_fixed, stride, (float *)nullptr, tags, tag_offset);
}

static int launch_reset_list(smr_env *env, float *obs_dev) {
    // entries of done_list[0..*done_count) are reset; the grid covers the worst case B, idle lanes exit at once
    LaunchTimer t(env, SMR_K_RESET);
    const int B = env->B;
    launch_reset<false>(env, side_grid(B, reset_group_stride()), env->stream, env->st.done_list, env->st.done_count, 0, reset_group_stride(), obs_dev);
    SMR_HIP(hipGetLastError());
    if (env->hscene.use_target) {                       // the target points of the new episodes
        launch_target(env, side_grid(B, reset_group_stride()), env->st.done_list, env->st.done_count, 0, reset_group_stride(), obs_dev);
        SMR_HIP(hipGetLastError());
    }
    return SMR_OK;
}

// ---- staged start states -------------------------------------------------------------------
// main stream:  ... finish/commit (consumes staged states, appends to refill_list) -> ev_commit
// side stream:  wait ev_commit -> reset_kernel<true>(refill_list) -> ev_refill
// main stream:  wait ev_refill before the next commit (and before refill_count is cleared for it)
static int staged_wait_refill(smr_env *env) {          // main stream: every refill in flight must have landed (reset paths, state injection)
    SMR_HIP(hipStreamWaitEvent(env->stream, env->ev_refill, 0));
    SMR_HIP(hipStreamWaitEvent(env->stream, env->ev_refill_b, 0));
    return SMR_OK;
}
static int staged_wait_lane(smr_env *env, int p) {     // main stream, step t: the refill launched at step t - 2 (same parity) must have landed
    SMR_HIP(hipStreamWaitEvent(env->stream, p ? env->ev_refill_b : env->ev_refill, 0));
    return SMR_OK;
}
// side streams of parity p: prepare the start states behind the consumed ones (and the next balls / target points).  The refill has
// two step periods: the env's other slot covers a restart in the very next step.
static int staged_launch_refill(smr_env *env, int p) {
    hipStream_t side = p ? env->side_b : env->side, side2 = p ? env->side2_b : env->side2;
    hipEvent_t ev_commit = p ? env->ev_commit_b : env->ev_commit, ev_refill = p ? env->ev_refill_b : env->ev_refill,
               ev_fork = p ? env->ev_fork_b : env->ev_fork, ev_join = p ? env->ev_join_b : env->ev_join;
    SMR_HIP(hipEventRecord(ev_commit, env->stream));
    SMR_HIP(hipStreamWaitEvent(side, ev_commit, 0));
    {
        LaunchTimer t(env, SMR_K_RESET, side);
        const int B = env->B;
        if (side2) {                                    // the next target / ball of the envs whose staged one was installed: its own
            SMR_HIP(hipEventRecord(ev_fork, side));                                // stream, beside the start states (all are latency chains)
            SMR_HIP(hipStreamWaitEvent(side2, ev_fork, 0));
            const dim3 grid = side_grid(B, reset_group_stride(true));
            if (env->tp_refill_slots) launch_target_staged(env, side2, grid, env->st.tp_refill_list, env->st.tp_refill_count, 0, reset_group_stride(true), env->st.tp_refill_tags, 0);
            if (env->stage_balls) launch_spawn_staged(env, side2, grid, env->st.spawn_list, env->st.spawn_count, 0, reset_group_stride(true), env->st.spawn_tags, 0);
            SMR_HIP(hipEventRecord(ev_join, side2));
        }
        launch_reset<true>(env, side_grid(B, reset_group_stride(true)), side, env->st.refill_list, env->st.refill_count, 0, reset_group_stride(true),
                           (float *)nullptr, env->st.refill_tags, 0);
        if (side2) SMR_HIP(hipStreamWaitEvent(side, ev_join, 0));
    }
    SMR_HIP(hipGetLastError());
    SMR_HIP(hipEventRecord(ev_refill, side));
    return SMR_OK;
}
// main stream, synchronous: (re)compute the staged state of every env for its reset_count + tag_offset
static int staged_prepare_all(smr_env *env, int tag_offset = 0) {
    int rc;
    if ((rc = staged_wait_refill(env))) return rc;
    LaunchTimer t(env, SMR_K_RESET);
    launch_reset<true>(env, side_grid(env->B, 1), env->stream, (const int *)nullptr, (const int *)nullptr, env->B, 1, (float *)nullptr, nullptr, tag_offset);
    SMR_HIP(hipGetLastError());
    return SMR_OK;
}
// the list slots of this step (three rotate: a list is written again two steps after its refill was awaited)
static void staged_select_lists(smr_env *env, int r) {
    const size_t B = (size_t)env->B;
    env->st.refill_list = env->refill_lists + r * B; env->st.refill_tags = env->refill_tag_lists + r * B; env->st.refill_count = env->refill_slots + r;
    if (env->tp_refill_slots) { env->st.tp_refill_list = env->tp_refill_lists + r * B; env->st.tp_refill_tags = env->tp_refill_tag_lists + r * B; env->st.tp_refill_count = env->tp_refill_slots + r; }
}
static int staged_check(smr_env *env) {                // after a synchronisation: device-side error conditions of the steps since the last one
    int flags = 0;
    SMR_HIP(hipMemcpy(&flags, env->st.error_flags, sizeof(int), hipMemcpyDeviceToHost));
    if (flags) {
        SMR_HIP(hipMemset(env->st.error_flags, 0, sizeof(int)));
        env->err = std::string("Could not find a valid sample within the attempt limit (the reference raises ValueError, ctlp.py:2127-2136):") +
                   ((flags & kSamplingStart) ? " collision-free robot start position;" : "") + ((flags & kSamplingPlanetPhase) ? " contact-free planet phase;" : "") +
                   ((flags & kSamplingBallLaunch) ? " ball launch;" : "") + ((flags & kSamplingBallTarget) ? " ball target pose;" : "") +
                   ((flags & kSamplingTargetPoint) ? " target point;" : "") + ((flags & kSamplingBallInitialStep) ? " contact-free initial ball time step;" : "") +
                   ((flags & kSamplingKinematicState) ? " kinematically feasible start velocity / acceleration;" : "");
        return SMR_E_SAMPLING;
    }
    if (!env->st.stale_commits) return SMR_OK;
    int stale = 0;
    SMR_HIP(hipMemcpy(&stale, env->st.stale_commits, sizeof(int), hipMemcpyDeviceToHost));
    if (stale) { env->err = "internal error: a reset was committed without a matching staged start state"; return SMR_E_INVALID; }
    return SMR_OK;
}

extern "C" {

int smr_reset(smr_env_t *env, const uint8_t *mask_dev, float *obs_dev) {
    if (!env) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    const int B = env->B;
    int rc;
    // explicit resets are timed whenever timing is on at all (the sampling period of smr_set_profiling counts steps); the flag is this
    // call's own - it does not inherit the phase the last step left behind
    struct ProfilingScope { smr_env *e; bool saved; ~ProfilingScope() { e->profiling = saved; } } scope{env, env->profiling};
    env->profiling = env->profile_every > 0;
    if (env->staged && mask_dev && !env->primed) {     // a partial reset as the very first call: nothing is staged yet
        if ((rc = staged_prepare_all(env, 0)) || (rc = staged_prepare_all(env, 1))) return rc;
    }
    env->primed = true;
    if (env->staged) {
        // Explicit resets are worked synchronously on the caller's stream: every refill in flight is awaited first, the lists of the
        // current slot are used and left empty again, and the states BEHIND the committed ones (tags + 1, + 2) are prepared here.
        if ((rc = staged_wait_refill(env))) return rc;
        staged_select_lists(env, env->rot);
        env->st.refill_zero = nullptr; env->st.tp_refill_zero = nullptr;
        if (env->stage_balls) { env->st.spawn_list = env->spawn_lists + (size_t)env->rot * B; env->st.spawn_tags = env->spawn_tag_lists + (size_t)env->rot * B; env->st.spawn_count = env->spawn_counts + env->rot; }
        const int stride = mask_dev ? reset_group_stride() : 1;
        const dim3 ggrid = side_grid(B, stride);
        const int *list = nullptr, *count = nullptr;
        int fixed = B;
        if (mask_dev) {
            SMR_HIP(hipMemsetAsync(env->st.refill_count, 0, sizeof(int), env->stream));
            hipLaunchKernelGGL(mask_to_list_kernel, dim3((B + 255) / 256), dim3(256), 0, env->stream, mask_dev, B, env->st.refill_list, env->st.refill_count);
            SMR_HIP(hipGetLastError());
            list = env->st.refill_list; count = env->st.refill_count; fixed = 0;
        } else {
            // every env: (re)compute the state for the current tag (a mask reset finds both of an env's slots staged already)
            launch_reset<true>(env, ggrid, env->stream, nullptr, nullptr, B, 1, (float *)nullptr, nullptr, 0);
            SMR_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(commit_kernel, dim3((B + 255) / 256), dim3(256), 0, env->stream, env->dscene, env->st, list, count, fixed, obs_dev);
        SMR_HIP(hipGetLastError());
        // behind the committed tag r (reset_count is r + 1 now): r + 1 (already staged for a mask reset) and r + 2
        if (!mask_dev) { launch_reset<true>(env, ggrid, env->stream, list, count, fixed, stride, (float *)nullptr, nullptr, 0); SMR_HIP(hipGetLastError()); }
        launch_reset<true>(env, ggrid, env->stream, list, count, fixed, stride, (float *)nullptr, nullptr, 1);
        SMR_HIP(hipGetLastError());
        if (env->hscene.use_ball) {                     // the balls of the new episodes (after the robot states are committed), and the ones behind them
            launch_spawn(env, ggrid, list, count, fixed, 1, stride, obs_dev);
            SMR_HIP(hipGetLastError());
            if (env->stage_balls) { launch_spawn_staged(env, env->stream, ggrid, list, count, fixed, stride, nullptr, 0); launch_spawn_staged(env, env->stream, ggrid, list, count, fixed, stride, nullptr, 1); }
            SMR_HIP(hipGetLastError());
        }
        if (env->hscene.use_target) {
            launch_target(env, ggrid, list, count, fixed, stride, obs_dev);
            SMR_HIP(hipGetLastError());
            if (env->tp_refill_slots) { launch_target_staged(env, env->stream, ggrid, list, count, fixed, stride, nullptr, 0); launch_target_staged(env, env->stream, ggrid, list, count, fixed, stride, nullptr, 1); }
            SMR_HIP(hipGetLastError());
        }
        if (mask_dev) SMR_HIP(hipMemsetAsync(env->st.refill_count, 0, sizeof(int), env->stream));     // the slot is the next step's: leave it empty
        return SMR_OK;
    }
    if (!mask_dev) {
        LaunchTimer t(env, SMR_K_RESET);
        launch_reset<false>(env, side_grid(B, 1), env->stream, (const int *)nullptr, (const int *)nullptr, B, 1, obs_dev);
        SMR_HIP(hipGetLastError());
        if (env->hscene.use_target) {
            launch_target(env, side_grid(B, 1), nullptr, nullptr, B, 1, obs_dev);
            SMR_HIP(hipGetLastError());
        }
        return SMR_OK;
    }
    SMR_HIP(hipMemsetAsync(env->st.done_count, 0, sizeof(int), env->stream));
    hipLaunchKernelGGL(mask_to_list_kernel, dim3((B + 255) / 256), dim3(256), 0, env->stream, mask_dev, B, env->st.done_list, env->st.done_count);
    SMR_HIP(hipGetLastError());
    return launch_reset_list(env, obs_dev);
}

int smr_step(smr_env_t *env, const float *actions_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev, int8_t *reason_dev,
             float *info_dev) {
    if (!env) return SMR_E_INVALID;
    if (!actions_dev || !obs_dev || !reward_dev || !done_dev || !reason_dev) { env->err = "smr_step: NULL buffer"; return SMR_E_INVALID; }
    if (env->staged && env->auto_reset && !env->primed) {
        env->err = "smr_step before the first smr_reset: with auto_reset a finished env restarts from a start state that reset() prepares";
        return SMR_E_INVALID;
    }
    SMR_HIP(hipSetDevice(env->device));
    const int B = env->B;
    // a pair of timing events is a pair of barrier packets in the queue: around every launch of every step they cost 65 us per step
    // (0.485 -> 0.550 ms at 65 536 envs, measured) - with a period the timed steps are a sample
    struct ProfilingScope { smr_env *e; ~ProfilingScope() { e->profiling = false; } } profiling_scope{env};      // this call's own flag
    env->profiling = env->profile_every > 0 && (env->step_seq++ % env->profile_every) == 0;
    if (env->hscene.use_ball) {
        // this step's spawn list: the side stream may still be re-staging from the other one
        env->st.spawn_list = env->spawn_lists + (size_t)env->rot * B; env->st.spawn_tags = env->spawn_tag_lists + (size_t)env->rot * B; env->st.spawn_count = env->spawn_counts + env->rot;
    }
    // (done_count / spawn_count / tp_list_count of this step: cleared by the plan kernel)
    // The next step's safe acceleration ranges ahead of time: range_block inside the step's jobs launch, range_kernel on a stream of its own
    // in the scenes without one (65 536 envs: space 133 -> 146 M env-steps/s, ball 141 -> 148, reaching task 109 -> 115, with the risk
    // networks 57 -> 67, ball machine + ball 81.4 -> 81.9)
    const bool range_ahead = env->range_mode != 0;
    const float *planned_actions = actions_dev;         // what the plan kernel maps into the safe range
    if (env->backup_net) {                              // actions.py:303-340: the backup policy's action where the risk rating objects
        LaunchTimer t(env, SMR_K_RISK);
        const int robs = env->obs_size - env->hscene.obs_target_size, ldx = robs + SMR_NJ, ldb = env->backup_net->d.out_size;
        const int mode = env->risk_mode;
        const bool simulate = mode == SMR_RISK_SIMULATE_NEXT_STEP || mode == SMR_RISK_SIMULATE_BACKUP_TRAJECTORY;
        const bool want_gt = env->gt_sa != nullptr;
        if ((simulate || want_gt) && !env->shadow) { env->err = "smr_step: this risk mode steps a background copy of the env - smr_set_risk_rollout first"; return SMR_E_INVALID; }
        if (!simulate && !env->risk_net) { env->err = "smr_step: no risk network"; return SMR_E_INVALID; }
        const dim3 egrid((B + 255) / 256), mgrid((B + kMlpRows - 1) / kMlpRows);
        if (mode == SMR_RISK_NEXT_KINEMATIC || mode == SMR_RISK_NEXT_FULL)      // the forecast needs the requested action's end acceleration
            hipLaunchKernelGGL(plan_kernel, dim3((B + 255) / 256, SMR_NJ), dim3(256), 0, env->stream, env->dscene, env->st, actions_dev, range_ahead ? 1 : 0);
        // the network rows of the CURRENT state: the backup policy's (what replaces a risky action), and the risk network's unless it rates a simulated state
        hipLaunchKernelGGL(risk_input_kernel, egrid, dim3(256), 0, env->stream, env->dscene, env->st, actions_dev, env->risk_x, ldx, env->backup_x, robs,
                           simulate ? SMR_RISK_CURRENT_STATE : mode);
        const float *rating = env->risk_y;
        if (simulate || want_gt) {
            // The background copy.  What it is stepped with (base.py:1529-1541,1611-1632): the requested action first - except where the
            // ground truth of a current-state network is asked for - then the backup policy's actions; a rating by simulation needs the first
            // step (mode 3) or all of them (mode 4), the ground truth label always all of them.
            smr_env *sh = env->shadow;
            const bool first_action = !(want_gt && !simulate && mode == SMR_RISK_CURRENT_STATE);
            const int K = (want_gt || mode == SMR_RISK_SIMULATE_BACKUP_TRAJECTORY) ? env->rollout_steps : 0;
            const int n_steps = (first_action ? 1 : 0) + K;
            hipLaunchKernelGGL(rollout_snapshot_kernel, egrid, dim3(256), 0, env->stream, env->dscene, env->st, sh->st, env->ro_alive, env->ro_outcome);
            const float *src = env->backup_x;             // the copy's risk observation before its next step (at first: the env's own)
            for (int i = 0; i < n_steps; ++i) {
                const float *act = actions_dev;
                if (!(i == 0 && first_action)) {
                    hipLaunchKernelGGL(mlp_forward_kernel, mgrid, dim3(64), 0, env->stream, env->backup_net->d, src, robs, env->backup_y, ldb, B);
                    hipLaunchKernelGGL(rollout_action_kernel, dim3((B * SMR_NJ + 255) / 256), dim3(256), 0, env->stream, B, (const float *)env->backup_y, ldb, env->ro_actions);
                    act = env->ro_actions;
                }
                const int rc = smr_step(sh, act, env->ro_obs, env->ro_reward, env->ro_done, env->ro_reason, nullptr);
                if (rc) { env->err = "background copy: " + sh->err; return rc; }
                hipLaunchKernelGGL(rollout_track_kernel, egrid, dim3(256), 0, env->stream, B, i == 0 ? 1 : 0, (const uint8_t *)env->ro_done, (const int8_t *)env->ro_reason,
                                   env->ro_alive, env->ro_outcome);
                if (i + 1 < n_steps || i == 0) {          // the copy's risk observation after this step: the backup policy's next input / the state the tables carry
                    float *dst = i == 0 ? env->ro_first : env->ro_rollx;
                    hipLaunchKernelGGL(risk_input_kernel, egrid, dim3(256), 0, env->stream, sh->dscene, sh->st, actions_dev, env->ro_riskx, ldx, dst, robs, SMR_RISK_CURRENT_STATE);
                    src = dst;
                }
            }
            if (simulate) {
                if (mode == SMR_RISK_SIMULATE_NEXT_STEP)  // the network rates the copy's next risk observation (base.py:1553-1555,1598-1603)
                    hipLaunchKernelGGL(mlp_forward_kernel, mgrid, dim3(64), 0, env->stream, env->risk_net->d, (const float *)env->ro_first, robs, env->risk_y, 1, B);
                hipLaunchKernelGGL(rollout_rating_kernel, egrid, dim3(256), 0, env->stream, B, mode, (const int *)env->ro_outcome, (const float *)env->risk_y, env->ro_rating);
                rating = env->ro_rating;
            }
        }
        if (!simulate) hipLaunchKernelGGL(mlp_forward_kernel, mgrid, dim3(64), 0, env->stream, env->risk_net->d, (const float *)env->risk_x, ldx, env->risk_y, 1, B);
        hipLaunchKernelGGL(mlp_forward_kernel, mgrid, dim3(64), 0, env->stream, env->backup_net->d, (const float *)env->backup_x, robs, env->backup_y, ldb, B);
        if (want_gt)
            hipLaunchKernelGGL(rollout_ground_truth_kernel, egrid, dim3(256), 0, env->stream, B, robs, mode, actions_dev, (const float *)env->backup_x, (const float *)env->ro_first,
                               (const int *)env->ro_outcome, rating, env->gt_sa, env->gt_s);
        hipLaunchKernelGGL(risk_apply_kernel, egrid, dim3(256), 0, env->stream, env->dscene, env->st, actions_dev, rating,
                           (const float *)env->backup_y, ldb, env->risk_threshold, env->risk_ignore_p, env->actions_eff);
        SMR_HIP(hipGetLastError());
        planned_actions = env->actions_eff;
    }
    {
        LaunchTimer t(env, SMR_K_PLAN);
        hipLaunchKernelGGL(plan_kernel, dim3((B + 255) / 256, SMR_NJ), dim3(256), 0, env->stream, env->dscene, env->st, planned_actions, range_ahead ? 1 : 0);
    }
    SMR_HIP(hipGetLastError());
    if (env->hscene.check_braking) {                    // adapt_action: the braking acceleration replaces actions whose braking trajectory collides
        LaunchTimer t(env, SMR_K_BRAKE);
        hipLaunchKernelGGL(brake_plan_kernel, dim3(((size_t)B * 8 + 255) / 256), dim3(256), 0, env->stream, env->dscene, env->st);
        const int bb = env->brake_block;
        const dim3 grid((B + bb - 1) / bb, env->hscene.n_bodies + env->hscene.n_self_pairs);
        const bool bt = env->hscene.punish_bt_min != 0;
        if (env->hscene.n_self_pairs > 0) {
            if (bt) hipLaunchKernelGGL((brake_check_kernel<true, true>), grid, dim3(bb), 0, env->stream, env->dscene, env->st);
            else hipLaunchKernelGGL((brake_check_kernel<true, false>), grid, dim3(bb), 0, env->stream, env->dscene, env->st);
        } else if (bt) hipLaunchKernelGGL((brake_check_kernel<false, true>), grid, dim3(bb), 0, env->stream, env->dscene, env->st);
        else hipLaunchKernelGGL((brake_check_kernel<false, false>), grid, dim3(bb), 0, env->stream, env->dscene, env->st);
        hipLaunchKernelGGL(brake_apply_kernel, dim3((B + 255) / 256, SMR_NJ), dim3(256), 0, env->stream, env->dscene, env->st);
        SMR_HIP(hipGetLastError());
    }
    const bool range_in_jobs = range_ahead && env->range_host && (env->hscene.planets_jobs || env->hscene.ball_jobs);
    const bool range_in_collide = range_ahead && env->range_host && !range_in_jobs && !env->hscene.planets_jobs;      // the lane-form collide kernel runs: rows behind its bodies
    const bool range_hosted = range_in_jobs || range_in_collide;
    if (range_ahead && !range_hosted) {            // the step is planned: the next step's ranges, beside the distance kernels
        // (the stream is made when a step first needs it: every stream of the process shifts the others' places in the runtime's four hardware
        // queues - with an idle sixth stream the staged refills of the ball-machine scenes shared a queue and took 1.8 instead of 1.1 ms)
        if (!env->range_stream && (hipStreamCreateWithFlags(&env->range_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&env->ev_planned, hipEventDisableTiming) != hipSuccess ||
                                   hipEventCreateWithFlags(&env->ev_ranges, hipEventDisableTiming) != hipSuccess)) { env->err = "could not create the range stream"; return SMR_E_HIP; }
        SMR_HIP(hipEventRecord(env->ev_planned, env->stream));
        SMR_HIP(hipStreamWaitEvent(env->range_stream, env->ev_planned, 0));
        hipLaunchKernelGGL(range_kernel, dim3((B + 255) / 256, SMR_NJ), dim3(256), 0, env->range_stream, env->dscene, env->st);
        SMR_HIP(hipEventRecord(env->ev_ranges, env->range_stream));
    }
    {
        LaunchTimer t(env, SMR_K_COLLIDE);
        const bool self = scene_extra(env), autop = env->hscene.planets_auto != 0, pairs = env->hscene.n_self_pairs > 0;
        const int block = autop ? 256 : SMR_COLLIDE_BLOCK;
        const dim3 grid((B + block - 1) / block, env->hscene.n_bodies);
        // the next step's ranges ride along with the step's full jobs launch (range_block); scenes without one get them from range_kernel
        const int range_blocks = range_in_jobs ? (7 * ((B + 63) / 64) + env->job_block_lanes / 64 - 1) / (env->job_block_lanes / 64) : 0;
        // the self-collision pairs' bound tests: a kernel of their own in front (SMR_SELF_ROWS=1: rows behind the bodies of the keys / collide grid)
        const bool ball_jobs = env->hscene.ball_jobs != 0;
        const bool self_jobs_alone = pairs && !env->hscene.planets_jobs && !ball_jobs;      // no pair / ball jobs to work them behind
        const int pair_rows = (pairs && !self_jobs_alone && env->self_rows) ? env->hscene.n_self_pairs : 0;
        const dim3 cgrid(grid.x, grid.y + pair_rows + (range_in_collide ? SMR_NJ : 0));
        const dim3 jgrid(env->job_blocks + range_blocks);
        // the self-collision pairs: bounds per env, the survivors as jobs (worked by the jobs kernel behind the pair jobs of a planet
        // scene, by a launch of its own otherwise)
        // (on a stream of their own beside the other distance kernels they finished sooner, but took from the staged refills what they
        // gained: reaching task 94 -> 86 M env-steps/s; ball scene beside the collide kernel: 63.6 -> 51.8 M; both measured)
        if (pairs && !pair_rows) {
            hipLaunchKernelGGL(self_bounds_kernel, dim3((B + 255) / 256, env->hscene.n_self_pairs), dim3(256), 0, env->stream, env->dscene, env->st);
            if (self_jobs_alone) hipLaunchKernelGGL((jobs_kernel<true>), dim3(env->self_job_blocks), dim3(512), env->job_lds_bytes, env->stream, env->dscene, env->st, 0, 64, 0);
        }
        if (env->hscene.planets_jobs) {
            const int env_blocks = (B + 255) / 256;
            const dim3 egrid(env_blocks + (env->N > 1 ? env->item_blocks : 0));      // behind the env blocks: a fixed grid striding over the items
            {
                LaunchTimer tk(env, SMR_K_KEYS);
                if (self) hipLaunchKernelGGL((keys_kernel<true>), dim3(grid.x, grid.y + pair_rows), dim3(256), 0, env->stream, env->dscene, env->st);
                else hipLaunchKernelGGL((keys_kernel<false>), grid, dim3(256), 0, env->stream, env->dscene, env->st);
            }
            {
                LaunchTimer te(env, SMR_K_ENUMERATE);
                if (self) hipLaunchKernelGGL((enumerate_kernel<true>), egrid, dim3(256), 0, env->stream, env->dscene, env->st, env_blocks);
                else hipLaunchKernelGGL((enumerate_kernel<false>), egrid, dim3(256), 0, env->stream, env->dscene, env->st, env_blocks);
            }
        } else if (self && autop) hipLaunchKernelGGL((collide_kernel<true, true>), cgrid, dim3(block), 0, env->stream, env->dscene, env->st, pair_rows);
        else if (self) hipLaunchKernelGGL((collide_kernel<true, false>), cgrid, dim3(block), 0, env->stream, env->dscene, env->st, pair_rows);
        else if (autop) hipLaunchKernelGGL((collide_kernel<false, true>), cgrid, dim3(block), 0, env->stream, env->dscene, env->st, pair_rows);
        else hipLaunchKernelGGL((collide_kernel<false, false>), cgrid, dim3(block), 0, env->stream, env->dscene, env->st, pair_rows);
        if (env->hscene.jobs_on) {
            if (ball_jobs) {                                  // the self pairs' and the ball's runs: one launch behind the collide kernel
                LaunchTimer tj(env, SMR_K_JOBS);
                if (self) hipLaunchKernelGGL((jobs_kernel<true>), jgrid, dim3(env->job_block_lanes), env->job_lds_bytes, env->stream, env->dscene, env->st, env->defer_after, env->job_quota_max, range_blocks);
                else hipLaunchKernelGGL((jobs_kernel<false>), jgrid, dim3(env->job_block_lanes), env->job_lds_bytes, env->stream, env->dscene, env->st, env->defer_after, env->job_quota_max, range_blocks);
                if (env->defer_after > 0) {
                    if (self) hipLaunchKernelGGL((jobs_coop_kernel<true>), dim3(env->coop_blocks), dim3(1024), env->job_lds_bytes, env->stream, env->dscene, env->st);
                    else hipLaunchKernelGGL((jobs_coop_kernel<false>), dim3(env->coop_blocks), dim3(1024), env->job_lds_bytes, env->stream, env->dscene, env->st);
                }
            }
            if (env->hscene.planets_jobs) {
                LaunchTimer tj(env, SMR_K_JOBS);
                if (self) hipLaunchKernelGGL((jobs_kernel<true>), jgrid, dim3(env->job_block_lanes), env->job_lds_bytes, env->stream, env->dscene, env->st, env->defer_after, env->job_quota_max, range_blocks);
                else hipLaunchKernelGGL((jobs_kernel<false>), jgrid, dim3(env->job_block_lanes), env->job_lds_bytes, env->stream, env->dscene, env->st, env->defer_after, env->job_quota_max, range_blocks);
                if (env->defer_after > 0) {
                    if (self) hipLaunchKernelGGL((jobs_coop_kernel<true>), dim3(env->coop_blocks), dim3(1024), env->job_lds_bytes, env->stream, env->dscene, env->st);
                    else hipLaunchKernelGGL((jobs_coop_kernel<false>), dim3(env->coop_blocks), dim3(1024), env->job_lds_bytes, env->stream, env->dscene, env->st);
                }
            }
            // (beside the jobs kernel on a stream of its own - its flags are final once the lists are written, its results commute with
            // the jobs' - the step got slower: space 115 -> 109 M, reaching task 92 -> 78 M env-steps/s, measured)
            hipLaunchKernelGGL(overflow_kernel, dim3((B + 63) / 64), dim3(64), 0, env->stream, env->dscene, env->st);
        }
    }
    SMR_HIP(hipGetLastError());
    const bool staged = env->staged && env->auto_reset;
    const int lane = env->parity;
    if (staged) {
        int rc = staged_wait_lane(env, lane);               // the refill of two steps ago: its slots are committed from / its lists re-used below
        if (rc) return rc;
        staged_select_lists(env, env->rot);                 // this slot's counters were cleared by the previous step's finish kernel
        const int nxt = (env->rot + 1) % 3;                 // ... as this one clears the next (last used two steps ago, awaited just now)
        env->st.refill_zero = env->refill_slots + nxt;
        env->st.tp_refill_zero = env->tp_refill_slots ? env->tp_refill_slots + nxt : nullptr;
    } else { env->st.refill_zero = nullptr; env->st.tp_refill_zero = nullptr; }
    if (range_ahead && !range_hosted) SMR_HIP(hipStreamWaitEvent(env->stream, env->ev_ranges, 0));      // finish overwrites the ranges of the envs that restart
    {
        LaunchTimer t(env, SMR_K_FINISH);
        hipLaunchKernelGGL(finish_kernel, dim3((B + 255) / 256), dim3(256), 0, env->stream, env->dscene, env->st, actions_dev, obs_dev,
                           reward_dev, done_dev, reason_dev, info_dev, staged ? 1 : 0, range_ahead ? 1 : 0);
    }
    SMR_HIP(hipGetLastError());
    if (env->hscene.use_ball) {
        LaunchTimer t(env, SMR_K_SPAWN);
        launch_spawn(env, side_grid(B, reset_group_stride()), env->st.spawn_list, env->st.spawn_count, 0, 0, reset_group_stride(), obs_dev);
        SMR_HIP(hipGetLastError());
    }
    if (env->hscene.use_target) {                       // reached target points, and those of staged resets
        LaunchTimer t(env, SMR_K_SPAWN);
        launch_target(env, side_grid(B, reset_group_stride()), env->st.tp_list, env->st.tp_list_count, 0, reset_group_stride(), obs_dev);
        SMR_HIP(hipGetLastError());
    }
    if (staged) {
        const int rc = staged_launch_refill(env, lane);
        env->parity ^= 1; env->rot = (env->rot + 1) % 3;
        return rc;
    }
    if (env->hscene.use_ball) env->rot = (env->rot + 1) % 3;
    if (env->auto_reset) return launch_reset_list(env, obs_dev);
    return SMR_OK;
}

int smr_observe(smr_env_t *env, float *obs_dev) {
    if (!env || !obs_dev) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    hipLaunchKernelGGL(observe_kernel, dim3((env->B + 255) / 256), dim3(256), 0, env->stream, env->dscene, env->st, obs_dev);
    SMR_HIP(hipGetLastError());
    return SMR_OK;
}

int smr_synchronize(smr_env_t *env) {
    if (!env) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    SMR_HIP(hipStreamSynchronize(env->stream));
    if (env->range_stream) SMR_HIP(hipStreamSynchronize(env->range_stream));
    if (env->side) { SMR_HIP(hipStreamSynchronize(env->side)); SMR_HIP(hipStreamSynchronize(env->side_b)); }
    return staged_check(env);
}

int smr_get_state(smr_env_t *env, const smr_host_state_t *h) {
    if (!env || !h) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    SMR_HIP(hipStreamSynchronize(env->stream));
    int rc;
    if ((rc = staged_check(env))) return rc;
    if ((rc = state_get(env, h->q, env->st.q, 7)) || (rc = state_get(env, h->v, env->st.v, 7)) || (rc = state_get(env, h->a, env->st.a, 7)) ||
        (rc = state_get(env, h->planet_idx, env->st.planet_idx, 2)) || (rc = state_get(env, h->step_count, env->st.step_count, 1)) ||
        (rc = state_get(env, h->moving_hit, env->st.moving_hit, 1)) || (rc = state_get(env, h->reset_count, env->st.reset_count, 1)) ||
        (rc = state_get(env, h->ball, env->st.ball, 10)) || (rc = state_get(env, h->ball_i, env->st.ball_i, 5)) ||
        (rc = state_get(env, h->target, env->st.tp, 5)) || (rc = state_get(env, h->target_count, env->st.tp_count, 1)))
        return rc;
    return SMR_OK;
}

int smr_set_state(smr_env_t *env, const smr_host_state_t *h) {
    if (!env || !h) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    SMR_HIP(hipStreamSynchronize(env->stream));
    if (env->side) { SMR_HIP(hipStreamSynchronize(env->side)); SMR_HIP(hipStreamSynchronize(env->side_b)); }
    if (h->planet_idx)
        for (size_t e = 0; e < (size_t)env->B; ++e)
            for (int p = 0; p < env->hscene.n_planets; ++p) {
                int v = h->planet_idx[e * 2 + p];
                if (v < 0 || v >= env->hscene.planet[p].n_orbit) { env->err = "smr_set_state: planet_idx out of range"; return SMR_E_INVALID; }
            }
    int rc;
    if ((rc = state_set(env, h->q, env->st.q, 7)) || (rc = state_set(env, h->v, env->st.v, 7)) || (rc = state_set(env, h->a, env->st.a, 7)) ||
        (rc = state_set(env, h->planet_idx, env->st.planet_idx, 2)) || (rc = state_set(env, h->step_count, env->st.step_count, 1)) ||
        (rc = state_set(env, h->moving_hit, env->st.moving_hit, 1)) || (rc = state_set(env, h->reset_count, env->st.reset_count, 1)) ||
        (rc = state_set(env, h->ball, env->st.ball, 10)) || (rc = state_set(env, h->ball_i, env->st.ball_i, 5)) ||
        (rc = state_set(env, h->target, env->st.tp, 5)) || (rc = state_set(env, h->target_count, env->st.tp_count, 1)))
        return rc;
    SMR_HIP(hipMemset(env->st.lb_moving, 0xFF, (size_t)env->hscene.n_bodies * env->B * 4));     // NaN: the bounds of the last step no longer hold
    SMR_HIP(hipMemset(env->st.range_valid, 0, (size_t)env->B * 4));                              // ... nor do the ranges computed ahead
    SMR_HIP(hipDeviceSynchronize());
    if (env->staged && h->reset_count) {               // the staged start states belong to the old reset counters
        if ((rc = staged_prepare_all(env, 0)) || (rc = staged_prepare_all(env, 1))) return rc;
        SMR_HIP(hipStreamSynchronize(env->stream));
    }
    if (env->stage_balls && h->ball_i) {               // ... the staged balls to the old spawn counters
        if ((rc = staged_wait_refill(env))) return rc;
        launch_spawn_staged(env, env->stream, side_grid(env->B, 1), nullptr, nullptr, env->B, 1, nullptr, 0);
        launch_spawn_staged(env, env->stream, side_grid(env->B, 1), nullptr, nullptr, env->B, 1, nullptr, 1);
        SMR_HIP(hipGetLastError());
        SMR_HIP(hipStreamSynchronize(env->stream));
    }
    if (env->tp_refill_slots && h->target_count) {     // ... and the staged target points to the old target counters
        if ((rc = staged_wait_refill(env))) return rc;
        launch_target_staged(env, env->stream, side_grid(env->B, 1), nullptr, nullptr, env->B, 1, nullptr, 0);
        launch_target_staged(env, env->stream, side_grid(env->B, 1), nullptr, nullptr, env->B, 1, nullptr, 1);
        SMR_HIP(hipGetLastError());
        SMR_HIP(hipStreamSynchronize(env->stream));
    }
    if (h->reset_count) env->primed = true;            // every staged slot was just prepared for the injected counters
    return staged_check(env);
}

int smr_set_profiling(smr_env_t *env, int32_t enable) {
    if (!env) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    int rc = resolve_profile(env);
    env->profile_every = enable > 0 ? enable : 0;      // (a negative value switches it off like 0)
    env->profiling = false;                             // decided per call: smr_step for its sampled steps, smr_reset whenever timing is on
    return rc;
}

int smr_get_profile(smr_env_t *env, double ms_out[SMR_K_COUNT], int64_t launches_out[SMR_K_COUNT]) {
    if (!env) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    int rc = resolve_profile(env);
    if (rc) return rc;
    for (int k = 0; k < SMR_K_COUNT; ++k) {
        if (ms_out) ms_out[k] = env->prof_ms[k];
        if (launches_out) launches_out[k] = env->prof_n[k];
        env->prof_ms[k] = 0; env->prof_n[k] = 0;
    }
    return staged_check(env);
}

int smr_episode_count(smr_env_t *env, int64_t *auto_resets_out) {
    if (!env || !auto_resets_out) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    SMR_HIP(hipStreamSynchronize(env->stream));
    unsigned long long n = 0;
    SMR_HIP(hipMemcpy(&n, env->st.episodes, sizeof n, hipMemcpyDeviceToHost));
    *auto_resets_out = (int64_t)n;
    return SMR_OK;
}

int smr_link_distances(smr_env_t *env, float *static_dev) {
    if (!env || !static_dev) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    SMR_HIP(hipMemcpyAsync(static_dev, env->st.dist, (size_t)env->hscene.n_bodies * env->B * sizeof(float), hipMemcpyDeviceToDevice, env->stream));
    return SMR_OK;
}

int smr_mlp_create(int32_t n_layers, const int32_t *sizes, const float *const *weights_host, const float *const *biases_host,
                   const int32_t *activations, int32_t device, smr_mlp_t **out) {
    if (!out) { g_create_error = "out is NULL"; return SMR_E_INVALID; }
    *out = nullptr;
    if (n_layers < 1 || n_layers > SMR_MLP_MAX_LAYERS || !sizes || !weights_host || !biases_host || !activations) { g_create_error = "smr_mlp_create: bad arguments"; return SMR_E_INVALID; }
    for (int l = 0; l <= n_layers; ++l)
        if (sizes[l] < 1 || sizes[l] > SMR_MLP_MAX_WIDTH) { g_create_error = "smr_mlp_create: layer width out of range (1 .. 256)"; return SMR_E_INVALID; }
    for (int l = 0; l < n_layers; ++l)
        if (!weights_host[l] || !biases_host[l] || activations[l] < 0 || activations[l] > SMR_ACT_SELU) { g_create_error = "smr_mlp_create: bad layer"; return SMR_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_create_error = "smr_mlp_create: no such HIP device (the library has no CPU path)"; return SMR_E_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return SMR_E_HIP; }
    smr_mlp *m = new smr_mlp();
    m->device = device;
    std::memset(&m->d, 0, sizeof m->d);
    m->d.n_layers = n_layers; m->d.in_size = sizes[0]; m->d.out_size = sizes[n_layers];
    auto fail = [&](const char *msg) { g_create_error = msg; smr_mlp_destroy(m); return SMR_E_HIP; };
    for (int l = 0; l < n_layers; ++l) {
        // zero padded: K to a multiple of 16 (four MFMA k steps are fetched together), N to a multiple of 64 (four 16-column tiles per
        // pass); a hidden layer's padded N is the next layer's K
        const int K = l == 0 ? (sizes[0] + 15) / 16 * 16 : m->d.N[l - 1];
        const int N = (sizes[l + 1] + 63) / 64 * 64;
        m->d.K[l] = K; m->d.N[l] = N; m->d.act[l] = activations[l];
        std::vector<float> W((size_t)K * N, 0.f), b((size_t)N, 0.f);
        for (int k = 0; k < sizes[l]; ++k)
            for (int n = 0; n < sizes[l + 1]; ++n) W[(size_t)k * N + n] = weights_host[l][(size_t)k * sizes[l + 1] + n];
        for (int n = 0; n < sizes[l + 1]; ++n) b[n] = biases_host[l][n];
        void *dW = nullptr, *db = nullptr;
        if (hipMalloc(&dW, W.size() * 4) != hipSuccess) return fail("hipMalloc failed");
        m->allocs.push_back(dW);
        if (hipMalloc(&db, b.size() * 4) != hipSuccess) return fail("hipMalloc failed");
        m->allocs.push_back(db);
        if (hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            return fail("hipMemcpy failed");
        m->d.W[l] = (const float *)dW; m->d.b[l] = (const float *)db;
    }
    *out = m;
    return SMR_OK;
}

int smr_mlp_destroy(smr_mlp_t *mlp) {
    if (!mlp) return SMR_OK;
    (void)hipSetDevice(mlp->device);
    (void)hipDeviceSynchronize();
    for (void *p : mlp->allocs) (void)hipFree(p);
    delete mlp;
    return SMR_OK;
}

int smr_mlp_forward(smr_mlp_t *mlp, const float *x_dev, int32_t ldx, float *y_dev, int32_t ldy, int32_t n_rows, void *stream) {
    if (!mlp || !x_dev || !y_dev || n_rows <= 0 || ldx < mlp->d.in_size || ldy < mlp->d.out_size) { g_create_error = "smr_mlp_forward: bad arguments"; return SMR_E_INVALID; }
    for (int l = 0; l < mlp->d.n_layers; ++l)             // the kernel's contract (smr_mlp.h): what smr_mlp_create pads to
        if (mlp->d.K[l] % 8 || mlp->d.N[l] % 64 || mlp->d.K[l] > kMlpMaxWidth || mlp->d.N[l] > kMlpMaxWidth || mlp->d.K[l] <= 0 || mlp->d.N[l] <= 0) {
            g_create_error = "smr_mlp_forward: layer sizes violate the kernel's padding contract (K % 8, N % 64, both <= 256)"; return SMR_E_INVALID;
        }
    if (hipSetDevice(mlp->device) != hipSuccess) return SMR_E_HIP;
    hipLaunchKernelGGL(mlp_forward_kernel, dim3((n_rows + kMlpRows - 1) / kMlpRows), dim3(64), 0, (hipStream_t)stream, mlp->d, x_dev, ldx, y_dev, ldy, n_rows);
    return hipGetLastError() == hipSuccess ? SMR_OK : SMR_E_HIP;
}

int smr_set_risk_networks(smr_env_t *env, smr_mlp_t *risk, smr_mlp_t *backup, float risk_threshold) {
    if (!env) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    SMR_HIP(hipStreamSynchronize(env->stream));
    if (!risk && !backup) { env->risk_net = env->backup_net = nullptr; env->st.risky = nullptr; return SMR_OK; }
    const int robs = env->obs_size - env->hscene.obs_target_size;                         // obs.py:419-431: no target-point columns
    const bool by_simulation = env->risk_state_config == SMR_RISK_SIMULATE_BACKUP_TRAJECTORY;          // rates without a network (base.py:1560-1566)
    if (!backup || (!risk && !by_simulation)) { env->err = "smr_set_risk_networks: both networks are needed (the risk network may be NULL with risk_state_config 4 only)"; return SMR_E_INVALID; }
    if ((risk && risk->device != env->device) || backup->device != env->device) { env->err = "smr_set_risk_networks: networks live on another device"; return SMR_E_INVALID; }
    if (risk && ((risk->d.in_size != robs + SMR_NJ && risk->d.in_size != robs) || risk->d.out_size != 1)) {
        env->err = "smr_set_risk_networks: the risk network must map (risk observation, action) or the risk observation alone to one value"; return SMR_E_INVALID;
    }
    if (backup->d.in_size != robs || backup->d.out_size < SMR_NJ) { env->err = "smr_set_risk_networks: the backup policy must map the risk observation to >= 7 outputs"; return SMR_E_INVALID; }
    const size_t B = (size_t)env->B;
    int rc;
    if (!env->risk_x && ((rc = dev_alloc(env, &env->risk_x, B * (robs + SMR_NJ))) || (rc = dev_alloc(env, &env->risk_y, B)) || (rc = dev_alloc(env, &env->backup_x, B * robs)) ||
                         (rc = dev_alloc(env, &env->backup_y, B * SMR_MLP_MAX_WIDTH)) || (rc = dev_alloc(env, &env->actions_eff, B * SMR_NJ)) ||
                         (rc = dev_alloc(env, &env->risky_buf, B)))) return rc;
    SMR_HIP(hipMemset(env->risky_buf, 0, B * 4));
    SMR_HIP(hipMemset(env->risk_y, 0, B * 4));
    env->st.risky = env->risky_buf;
    env->risk_net = risk; env->backup_net = backup; env->risk_threshold = risk_threshold;
    env->risk_mode = (risk && risk->d.in_size == robs + SMR_NJ) ? SMR_RISK_STATE_ACTION : env->risk_state_config;      // base.py:609-617: action_size decides
    return SMR_OK;
}

int smr_set_risk_state_config(smr_env_t *env, int32_t config) {
    if (!env) return SMR_E_INVALID;
    if (config < SMR_RISK_CURRENT_STATE || config > SMR_RISK_SIMULATE_BACKUP_TRAJECTORY) {
        env->err = "smr_set_risk_state_config: 0 .. 4 (safe_motions_base.py:73-83)";
        return SMR_E_INVALID;
    }
    if (env->backup_net && !env->risk_net && config != SMR_RISK_SIMULATE_BACKUP_TRAJECTORY) { env->err = "smr_set_risk_state_config: this mode needs a risk network"; return SMR_E_INVALID; }
    env->risk_state_config = config;
    if (env->backup_net && env->risk_mode != SMR_RISK_STATE_ACTION) env->risk_mode = config;
    return SMR_OK;
}

int smr_set_risk_rollout(smr_env_t *env, smr_env_t *shadow, int32_t backup_steps, float ignore_probability, float *gt_state_action_dev, float *gt_state_dev) {
    if (!env) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    SMR_HIP(hipStreamSynchronize(env->stream));
    if (!(ignore_probability >= 0.f && ignore_probability <= 1.f)) { env->err = "smr_set_risk_rollout: ignore_probability must lie in [0, 1]"; return SMR_E_INVALID; }
    env->risk_ignore_p = ignore_probability;
    if (!shadow) {
        if (gt_state_action_dev || gt_state_dev) { env->err = "smr_set_risk_rollout: the ground truth comes from the background copy"; return SMR_E_INVALID; }
        env->shadow = nullptr; env->gt_sa = env->gt_s = nullptr;
        return SMR_OK;
    }
    const DScene &a = env->hscene, &b = shadow->hscene;
    if (shadow == env || shadow->shadow) { env->err = "smr_set_risk_rollout: the background copy must be a handle of its own without a copy of its own"; return SMR_E_INVALID; }
    if (backup_steps < 0 || shadow->B != env->B || shadow->device != env->device || shadow->stream != env->stream || shadow->obs_size != env->obs_size ||
        a.n_bodies != b.n_bodies || a.n_self_pairs != b.n_self_pairs || a.use_ball != b.use_ball || a.use_target != b.use_target || a.n_planets != b.n_planets ||
        a.checks_per_step != b.checks_per_step || a.seed_lo != b.seed_lo || a.seed_hi != b.seed_hi || a.env_base != b.env_base || a.dt != b.dt ||
        a.check_braking != b.check_braking) {
        env->err = "smr_set_risk_rollout: the background copy must be a handle of the same scene, batch, seed, env_index_base, device and stream"; return SMR_E_INVALID;
    }
    if (shadow->auto_reset || b.auto_reset || b.episode_steps != backup_steps + 1) {
        env->err = "smr_set_risk_rollout: the background copy needs auto_reset = 0 and episode_steps = backup_steps + 1 (trajectory_length = backup_steps + 2, safe_motions_base.py:1819)";
        return SMR_E_INVALID;
    }
    if (gt_state_dev && !gt_state_action_dev) { env->err = "smr_set_risk_rollout: gt_state_dev without gt_state_action_dev"; return SMR_E_INVALID; }
    const size_t B = (size_t)env->B;
    const int robs = env->obs_size - a.obs_target_size;
    int rc;
    if (!env->ro_alive && ((rc = dev_alloc(env, &env->ro_alive, B)) || (rc = dev_alloc(env, &env->ro_outcome, B)) || (rc = dev_alloc(env, &env->ro_first, B * robs)) ||
                           (rc = dev_alloc(env, &env->ro_rollx, B * robs)) || (rc = dev_alloc(env, &env->ro_riskx, B * (robs + SMR_NJ))) ||
                           (rc = dev_alloc(env, &env->ro_actions, B * SMR_NJ)) || (rc = dev_alloc(env, &env->ro_obs, B * (size_t)env->obs_size)) ||
                           (rc = dev_alloc(env, &env->ro_reward, B)) || (rc = dev_alloc(env, &env->ro_rating, B)) || (rc = dev_alloc(env, &env->ro_done, B)) ||
                           (rc = dev_alloc(env, &env->ro_reason, B)))) return rc;
    SMR_HIP(hipMemset(env->ro_first, 0, B * robs * 4));
    SMR_HIP(hipMemset(env->ro_outcome, 0, B * 4));
    env->shadow = shadow; env->rollout_steps = backup_steps; env->gt_sa = gt_state_action_dev; env->gt_s = gt_state_dev;
    return SMR_OK;
}

int smr_test_fk(smr_env_t *env, const float *q_host, int32_t n, float *pos_out, float *rot_out) {
    if (!env || !q_host || n <= 0 || !pos_out || !rot_out) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    TmpBuf t{env, {}};
    float *q = t.up(q_host, (size_t)n * 7), *pos = t.up<float>(nullptr, (size_t)n * 24), *r = t.up<float>(nullptr, (size_t)n * 72);
    if (!q || !pos || !r) { env->err = "smr_test_fk: allocation failed"; return SMR_E_HIP; }
    hipLaunchKernelGGL(test_fk_kernel, dim3((n + 63) / 64), dim3(64), 0, env->stream, env->dscene, q, n, pos, r);
    SMR_HIP(hipGetLastError());
    SMR_HIP(hipStreamSynchronize(env->stream));
    SMR_HIP(hipMemcpy(pos_out, pos, (size_t)n * 24 * 4, hipMemcpyDeviceToHost));
    SMR_HIP(hipMemcpy(rot_out, r, (size_t)n * 72 * 4, hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_test_safe_acc_range(smr_env_t *env, const float *p_host, const float *v_host, const float *a_host, int32_t n, float *lo_out,
                            float *hi_out, int32_t *code_out) {
    if (!env || !p_host || !v_host || !a_host || n <= 0 || !lo_out || !hi_out || !code_out) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    TmpBuf t{env, {}};
    size_t m = (size_t)n * 7;
    float *p = t.up(p_host, m), *v = t.up(v_host, m), *a = t.up(a_host, m), *lo = t.up<float>(nullptr, m), *hi = t.up<float>(nullptr, m);
    int *code = t.up<int>(nullptr, m);
    if (!p || !v || !a || !lo || !hi || !code) { env->err = "smr_test_safe_acc_range: allocation failed"; return SMR_E_HIP; }
    hipLaunchKernelGGL(test_range_kernel, dim3((n + 63) / 64), dim3(64), 0, env->stream, env->dscene, p, v, a, n, lo, hi, code);
    SMR_HIP(hipGetLastError());
    SMR_HIP(hipStreamSynchronize(env->stream));
    SMR_HIP(hipMemcpy(lo_out, lo, m * 4, hipMemcpyDeviceToHost));
    SMR_HIP(hipMemcpy(hi_out, hi, m * 4, hipMemcpyDeviceToHost));
    SMR_HIP(hipMemcpy(code_out, code, m * 4, hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_test_distances(smr_env_t *env, const float *q_host, const int32_t *planet_idx_host, int32_t n, float *out) {
    if (!env || !q_host || !planet_idx_host || n <= 0 || !out) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    for (int i = 0; i < n; ++i)
        for (int p = 0; p < env->hscene.n_planets; ++p) {
            int v = planet_idx_host[2 * i + p];
            if (v < 0 || v >= env->hscene.planet[p].n_orbit) { env->err = "smr_test_distances: planet_idx out of range"; return SMR_E_INVALID; }
        }
    TmpBuf t{env, {}};
    float *q = t.up(q_host, (size_t)n * 7), *o = t.up<float>(nullptr, (size_t)n * 4);
    int *idx = t.up(planet_idx_host, (size_t)n * 2);
    if (!q || !o || !idx) { env->err = "smr_test_distances: allocation failed"; return SMR_E_HIP; }
    hipLaunchKernelGGL(test_distances_kernel, dim3((n + 63) / 64), dim3(64), 0, env->stream, env->dscene, q, idx, n, o);
    SMR_HIP(hipGetLastError());
    SMR_HIP(hipStreamSynchronize(env->stream));
    SMR_HIP(hipMemcpy(out, o, (size_t)n * 16, hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_test_brake_plan(smr_env_t *env, float *acc_out, int32_t *steps_out, int32_t *braked_out) {
    if (!env || !env->hscene.check_braking) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    SMR_HIP(hipStreamSynchronize(env->stream));
    const size_t B = (size_t)env->B, n = (size_t)(env->hscene.brake_timeout_steps + 1) * SMR_NJ * B;
    if (acc_out) SMR_HIP(hipMemcpy(acc_out, env->st.brake_acc, n * 4, hipMemcpyDeviceToHost));
    if (steps_out) SMR_HIP(hipMemcpy(steps_out, env->st.brake_steps, B * 4, hipMemcpyDeviceToHost));
    if (braked_out) SMR_HIP(hipMemcpy(braked_out, env->st.braked, B * 4, hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_test_self_bounds(smr_env_t *env, const float *q_host, int32_t n, float *out) {
    if (!env || !q_host || n <= 0 || !out) return SMR_E_INVALID;
    SMR_HIP(hipSetDevice(env->device));
    TmpBuf t{env, {}};
    float *q = t.up(q_host, (size_t)n * 7), *o = t.up<float>(nullptr, (size_t)n * SMR_MAX_SELF_PAIRS);
    if (!q || !o) { env->err = "smr_test_self_bounds: allocation failed"; return SMR_E_HIP; }
    hipLaunchKernelGGL(test_self_bounds_kernel, dim3((n + 63) / 64), dim3(64), 0, env->stream, env->dscene, q, n, o);
    SMR_HIP(hipGetLastError());
    SMR_HIP(hipStreamSynchronize(env->stream));
    SMR_HIP(hipMemcpy(out, o, (size_t)n * SMR_MAX_SELF_PAIRS * 4, hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_test_ball_distances(smr_env_t *env, const float *q_host, const float *ball_host, const int32_t *k_host, int32_t n, float *out) {
    if (!env || !q_host || !ball_host || !k_host || n <= 0 || !out) return SMR_E_INVALID;
    if (!env->hscene.use_ball) { env->err = "smr_test_ball_distances: not a ball scene"; return SMR_E_INVALID; }
    SMR_HIP(hipSetDevice(env->device));
    TmpBuf t{env, {}};
    float *q = t.up(q_host, (size_t)n * 7), *b = t.up(ball_host, (size_t)n * 10), *o = t.up<float>(nullptr, (size_t)n * 4);
    int *k = t.up(k_host, (size_t)n);
    if (!q || !b || !o || !k) { env->err = "smr_test_ball_distances: allocation failed"; return SMR_E_HIP; }
    hipLaunchKernelGGL(test_ball_distances_kernel, dim3((n + 63) / 64), dim3(64), 0, env->stream, env->dscene, q, b, k, n, o);
    SMR_HIP(hipGetLastError());
    SMR_HIP(hipStreamSynchronize(env->stream));
    SMR_HIP(hipMemcpy(out, o, (size_t)n * 16, hipMemcpyDeviceToHost));
    return SMR_OK;
}

}  // extern "C"
