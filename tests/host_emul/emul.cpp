// Host build of the DEVICE functions in safemotionsrisk_amd/csrc/smr_device.h (float32, same source), for
// debugging GJK / range code paths without a GPU.  Test infrastructure only - never loaded by the product.
#define SMR_HOST_EMUL 1
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <algorithm>
#define __device__
#define __forceinline__ inline
#define SMR_CONST
#define SMR_GLOBAL
struct float4 { float x, y, z, w; };
static inline int __builtin_amdgcn_readfirstlane(int x) { return x; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline int __popc(int x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }     // a "wavefront" of one lane
static inline unsigned __lane_id() { return 0u; }
template <class T> static inline T __shfl(T x, int) { return x; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline bool __any(bool p) { return p; }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline void sincosf_(float x, float *s, float *c) { *s = std::sin(x); *c = std::cos(x); }
#define sincosf sincosf_
#include <math.h>
extern "C" { long emul_counters[8] = {0, 0, 0, 0, 0, 0, 0, 0}; }
extern "C" { long emul_slot = 0, emul_slot_counters[64] = {0}; }
#define SMR_CNT(i) ((i) == 4 ? ++emul_slot_counters[emul_slot & 63] : ++emul_counters[i])
#define SMR_CNT_SLOT(i) (emul_slot = (i))
#include "../../include/smr_scene.h"
#include "../../safemotionsrisk_amd/csrc/smr_device.h"
#include "../../safemotionsrisk_amd/csrc/smr_tables.h"
using namespace smr;

extern "C" {
// distance between vertex sets A (identity frame) and B (transform R,t): float32 device GJK
float emul_gjk(const float *va, int na, const float *vb, int nb, const float *Rb, const float *tb, float cutoff, float stop_below, int identity_a, const float *Ra, const float *ta) {
    static float4 verts[4096];
    for (int i = 0; i < na; ++i) verts[i] = {va[3*i], va[3*i+1], va[3*i+2], 0};
    for (int i = 0; i < nb; ++i) verts[na+i] = {vb[3*i], vb[3*i+1], vb[3*i+2], 0};
    DPiece A{}, B{};
    A.begin = 0; A.count = na; B.begin = na; B.count = nb;
    Xf XA, XB;
    for (int i = 0; i < 9; ++i) { XB.r[i] = Rb[i]; XA.r[i] = Ra[i]; }
    XB.t = {tb[0], tb[1], tb[2]}; XA.t = {ta[0], ta[1], ta[2]};
    // initial direction as piece_distance: center difference
    F3 ca{0,0,0}, cb{0,0,0};
    for (int i = 0; i < na; ++i) { ca.x += va[3*i]/na; ca.y += va[3*i+1]/na; ca.z += va[3*i+2]/na; }
    for (int i = 0; i < nb; ++i) { cb.x += vb[3*i]/nb; cb.y += vb[3*i+1]/nb; cb.z += vb[3*i+2]/nb; }
    if (!identity_a) ca = apply(XA, ca);
    cb = apply(XB, cb);
    F3 d = ca - cb;
    const ScanHull hb{(const float *)verts, B.begin, B.count};
    if (identity_a) return gjk_distance<true>((const float *)verts, A, XA, hb, XB, d, cutoff, stop_below);
    return gjk_distance<false>((const float *)verts, A, XA, hb, XB, d, cutoff, stop_below);
}
// the violation code alone, as the kinematic start-state sampling asks for it (safe_state_code)
int emul_state_code(double plo, double phi, double V, double A, double J, double dt, float p, float v, float a) {
    JointLim L{(float)plo, (float)phi, (float)V, (float)A, (float)J, (float)dt, 0.f, 0.f, 0.f, 0.f, 0.f};
    L.derive();
    L.plo_res = (float)(plo - (double)L.plo); L.phi_res = (float)(phi - (double)L.phi); L.V_res = (float)(V - (double)L.V);
    return safe_state_code<1>(L, p, v, a);
}
void emul_safe_acc_range(double plo, double phi, double V, double A, double J, double dt, float p, float v, float a, float *lo, float *hi, int *code) {
    JointLim L{(float)plo, (float)phi, (float)V, (float)A, (float)J, (float)dt, 0.f, 0.f, 0.f, 0.f, 0.f};
    L.derive();
    L.plo_res = (float)(plo - (double)L.plo); L.phi_res = (float)(phi - (double)L.phi); L.V_res = (float)(V - (double)L.V);
    safe_acc_range_joint<1>(L, p, v, a, *lo, *hi, *code);
}
}

// ---- whole-scene emulation: float32 tables as smr_create builds them, collide_sample on the host ----
#include <vector>
struct HostScene {
    DScene d;
    HostGeometry G; std::vector<float> pose[2], local[2];
};
static void cp(float *dst, const double *src, int n) { for (int i = 0; i < n; ++i) dst[i] = (float)src[i]; }
extern "C" void *emul_scene_create(const smr_scene_t *sp) {
    const smr_scene_t &s = *sp;
    HostScene *H = new HostScene();
    DScene &d = H->d;
    std::memset(&d, 0, sizeof d);
    d.B = 1; d.obs_size = s.obs_size; d.episode_steps = s.episode_steps; d.updates_per_step = s.updates_per_step;
    d.checks_per_step = s.collision_checks_per_step; d.dt = (float)s.dt;
    cp(d.pos_lo, s.pos_lo, 7); cp(d.pos_hi, s.pos_hi, 7); cp(d.vel_max, s.vel_max, 7); cp(d.acc_max, s.acc_max, 7); cp(d.jerk_max, s.jerk_max, 7);
    cp(d.base, s.base_xyz, 3); cp(d.ee_offset, s.ee_offset, 3);
    for (int j = 0; j < 7; ++j) { cp(d.joint_xyz[j], s.joint_xyz[j], 3); cp(d.joint_axis[j], s.joint_axis[j], 3); }
    if (const char *msg = build_host_geometry(s, H->G)) { std::fprintf(stderr, "emul_scene_create: %s\n", msg); delete H; return nullptr; }
    d.verts = H->G.verts.data(); d.pieces = H->G.pieces.data(); d.span = H->G.span.data(); d.ids4 = H->G.ids4.data(); d.start = H->G.start.data();
    if (const char *msg = scene_bodies(s, d)) { std::fprintf(stderr, "emul_scene_create: %s\n", msg); delete H; return nullptr; }
    d.n_static = s.n_static; for (int i = 0; i < s.n_static; ++i) d.static_piece[i] = s.static_piece[i];
    d.static_mask = reachable_static_mask(s); d.static_max = (float)s.static_max_distance; d.moving_mask = s.moving_link_mask; d.contact_mask = reachable_contact_mask(s);
    d.moving_max = (float)s.moving_max_distance; d.contact_dist = (float)s.contact_distance; d.n_planets = s.n_planets;
    for (int p = 0; p < s.n_planets; ++p) {
        const smr_planet_t &pl = s.planet[p];
        H->pose[p].assign((size_t)pl.n_orbit * 8, 0.f); H->local[p].resize((size_t)pl.n_orbit * 2);
        for (int i = 0; i < pl.n_orbit; ++i) { for (int c = 0; c < 7; ++c) H->pose[p][(size_t)i*8+c] = (float)pl.orbit_pose[(size_t)i*7+c];
            H->local[p][2*(size_t)i] = (float)pl.orbit_local[2*(size_t)i]; H->local[p][2*(size_t)i+1] = (float)pl.orbit_local[2*(size_t)i+1]; }
        DPlanet &q = d.planet[p];
        q.piece_begin = pl.piece_begin; q.piece_count = pl.piece_count; q.n_orbit = pl.n_orbit; q.index_shift = pl.index_shift; q.coupled = pl.coupled;
        q.orbit_pose = H->pose[p].data(); q.orbit_local = H->local[p].data();
        planet_bounds(H->G.pieces, q);
        d.planet_step[p] = planet_step_bound(H->G, q, H->pose[p].data());
    }
    sweep_tables(s, H->G, d);
    d.use_ball = s.use_ball; d.ball_piece = s.ball_piece; d.ball_radius = (float)s.ball_radius; d.ball_active_xy_radius = (float)s.ball_active_xy_radius;
    planet_tables(H->G, d);
    d.auto_verts = (const float *)H->G.auto_verts.data(); d.auto_desc = H->G.auto_desc.data();
    d.term_moving = s.terminate_on_moving_collision; d.collision_dist = (float)s.collision_distance;
    return H;
}
extern "C" void emul_distances(void *h, const float *q, const int *idx, int want_dist, int want_contact, float *out) {
    HostScene *H = (HostScene *)h;
    Sample r = collide_sample(H->d, q, idx, want_dist != 0, want_contact != 0);
    out[0] = r.d_static; out[1] = r.d_moving; out[2] = r.contact ? 1.f : 0.f; out[3] = r.d_self;
}

// the LDS pool image of the reset-side kernels (body_pool_tables): dwords of the image, or -1 when it does not fit; *mismatch counts the
// support points over `n_dirs` pseudo-random directions per body that differ between a climb over the pool copy and one over the
// global tables, plus the piece records that differ
extern "C" int emul_body_pool_check(void *h, const smr_scene_t *sp, int n_dirs, int *mismatch) {
    HostScene *H = (HostScene *)h;
    std::vector<uint32_t> pool;
    if (!body_pool_tables(*sp, H->G, pool)) return -1;
    const BodyPool BP{pool.data()};
    *mismatch = 0;
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / 8388608.0f - 1.0f; };
    for (int b = 0; b < sp->n_bodies; ++b) {
        const int pi = sp->link_piece[b];
        const DPiece P = load_piece(H->d.pieces, pi), Q = BP.piece(b);
        if (std::memcmp(&P, &Q, sizeof(DPiece)) != 0) ++*mismatch;
        const GlobalHull g = global_hull(H->d, pi, P);
        const LocalHull l = BP.hull(b);
        for (int k = 0; k < n_dirs; ++k) {
            const F3 d{rnd(), rnd(), rnd()};
            int h1 = 0, h2 = 0;
            const F3 a = g.support(d, h1, true), c = l.support(d, h2, true);
            if (a.x != c.x || a.y != c.y || a.z != c.z || h1 != h2) ++*mismatch;
        }
    }
    return (int)pool.size();
}

// support of piece `pi` along d (piece frame): climbing result vs exhaustive scan; returns the two dot products
extern "C" void emul_support_check(void *h, int pi, const float *dir, int warm_start_vertex, float *out /*climb, scan, steps unused*/) {
    HostScene *H = (HostScene *)h;
    const DPiece P = load_piece(H->d.pieces, pi);
    const GlobalHull hull = global_hull(H->d, pi, P);
    F3 d{dir[0], dir[1], dir[2]};
    int hint = warm_start_vertex < 0 ? 0 : warm_start_vertex;
    F3 a = hull.support(d, hint, warm_start_vertex < 0);
    const ScanHull sc{(const float *)H->d.verts, P.begin, P.count};
    int dummy = 0;
    F3 b = sc.support(d, dummy, true);
    out[0] = dot(a, d); out[1] = dot(b, d); out[2] = (float)hint;
}
extern "C" unsigned emul_contact_mask(void *h) { return ((HostScene *)h)->d.contact_mask; }
extern "C" unsigned emul_static_mask(void *h) { return ((HostScene *)h)->d.static_mask; }
extern "C" int emul_piece_count(void *h, int pi) { return ((HostScene *)h)->d.pieces[pi].count; }
extern "C" int emul_piece_kind(void *h, int pi) { return ((HostScene *)h)->d.pieces[pi].kind; }

// planets: the end-of-step query set of one (env, link) lane of the collide kernel
// want_moving / want_contact bit 1 (value 2): lane-autonomous planet queries (planet_queries_auto) instead of the lockstep loop
extern "C" void emul_link_query(void *h, const float *q, const int *idx, int l, int want_static, int want_moving, int want_contact, float *out) {
    HostScene *H = (HostScene *)h;
    const DScene &S = H->d;
    Xf T = fk_base(S);
    for (int j = 0; j < l; ++j) fk_advance(S, j, q[j], T);
    const DPiece LP = load_piece(S.pieces, S.link_piece[l]);
    const bool autop = ((want_moving | want_contact) & 2) && S.planets_auto;
    want_moving &= 1; want_contact &= 1;
    Sample r = collide_link(S, l, LP, global_hull(S, S.link_piece[l], LP), T, idx, nullptr, want_static != 0, want_moving != 0, want_contact != 0, nullptr, autop);
    if (autop) {
        static std::vector<float> pv, pd;
        pv.assign(4 * kMaxPlanetVerts, 0.f); pd.assign(kPlanetDescDwords * kMaxPlanetPieces, 0.f);
        stage_planets(S, 0, 1, pv.data(), pd.data());
        uint32_t keys[7];
        planet_queries_auto(S, pv.data(), pd.data(), keys, 1, LP, global_hull(S, S.link_piece[l], LP), T, idx,
                            want_moving && ((S.moving_mask >> l) & 1u), want_contact && ((S.contact_mask >> l) & 1u), r);
    }
    out[0] = r.d_static; out[1] = r.d_moving; out[2] = r.contact ? 1.f : 0.f;
}

// ball scene: per-link moving distance (no cut-off sharing between links) for configuration q and the ball after k updates
extern "C" void emul_ball_link_distances(void *h, const float *q, const float *ball, int k, float *out /*8*/, int use_scan) {
    HostScene *H = (HostScene *)h;
    const DScene &S = H->d;
    BallState b;
    for (int c = 0; c < 3; ++c) { b.p0[c] = ball[c]; b.v0[c] = ball[3 + c]; b.e0[c] = ball[6 + c]; }
    b.omega = ball[9];
    const Xf XB = ball_pose(S, b, k, nullptr);
    Xf T = fk_base(S);
    for (int l = 0; l < SMR_NLINK; ++l) {
        if (l > 0) fk_advance(S, l - 1, q[l - 1], T);
        const DPiece LP = load_piece(S.pieces, S.link_piece[l]);
        Sample r = use_scan ? collide_link(S, l, LP, ScanHull{(const float *)S.verts, LP.begin, LP.count}, T, nullptr, &XB, false, true, false)
                            : collide_link(S, l, LP, global_hull(S, S.link_piece[l], LP), T, nullptr, &XB, false, true, false);
        out[l] = r.d_moving;
    }
}

extern "C" int emul_simplex(float *w /*4x3 in/out*/, int n, float *v_out) {
    F3 a{w[0],w[1],w[2]}, b{w[3],w[4],w[5]}, c{w[6],w[7],w[8]}, d{w[9],w[10],w[11]}, v{0,0,0};
    bool inside = simplex_solve(a, b, c, d, n, v);
    float o[12] = {a.x,a.y,a.z,b.x,b.y,b.z,c.x,c.y,c.z,d.x,d.y,d.z};
    for (int i = 0; i < 12; ++i) w[i] = o[i];
    v_out[0] = v.x; v_out[1] = v.y; v_out[2] = v.z;
    return inside ? -1 : n;
}
extern "C" float emul_pos_margin(double plo, double phi, double V, double A, double J, double dt, float gap, float v, float a, float a1) {
    JointLim L{(float)plo, (float)phi, (float)V, (float)A, (float)J, (float)dt, 0.f, 0.f, 0.f, 0.f, 0.f};
    L.derive();
    L.V_res = (float)(V - (double)L.V);
    return pos_margin(gap, v, a, a1, 1e-5f * L.V, L);
}

// one lane of the collide kernel (collide_lane) and the work items it asks for (collide_item): body l over the step's cubic
// (q, v, a -> a1).  mode 0: out = contact at the end-of-step sample, d_static, d_moving, k_need (intermediate samples still to test),
// contact at any of those samples (the item kernel's work), the lower bound at t = dt.  lb_start: the bound at t = 0 (< 0: none).  mode 1: every sample 1 .. N-1 is tested (what mode 0 must
// reproduce): out[4] = contact at any of them.  ball: NULL or the 10 floats of smr_host_state_t.ball + ball_i counters
extern "C" void emul_collide_lane(void *h, int l, const float *q, const float *v, const float *a, const float *a1, const int *idx0, int N,
                                  int mode, const float *ball, const int *ball_i, float lb_start, float *out) {
    HostScene *H = (HostScene *)h;
    DScene &S = H->d;
    S.checks_per_step = N;
    const DPiece LP = load_piece(S.pieces, S.link_piece[l]);
    const GlobalHull hull = global_hull(S, S.link_piece[l], LP);
    static std::vector<float> pv, pd;
    pv.assign(4 * kMaxPlanetVerts, 0.f); pd.assign(kPlanetDescDwords * kMaxPlanetPieces, 0.f);
    if (S.planets_auto) stage_planets(S, 0, 1, pv.data(), pd.data());
    uint32_t keys[7];
    LaneInput in;
    in.q = q; in.v = v; in.a = a; in.a1 = a1; in.stride = 1;
    in.planet_idx[0] = idx0[0]; in.planet_idx[1] = idx0[1]; in.moving_hit = 0;
    in.ball_k0 = in.ball_kmax = in.ball_khit = in.ball_active = 0;
    in.lb_start = lb_start;
    if (ball) {
        for (int c = 0; c < 3; ++c) { in.ball.p0[c] = ball[c]; in.ball.v0[c] = ball[3 + c]; in.ball.e0[c] = ball[6 + c]; }
        in.ball.omega = ball[9];
        in.ball_k0 = ball_i[0]; in.ball_kmax = ball_i[1]; in.ball_khit = ball_i[2]; in.ball_active = ball_i[3];
    }
    const bool con_link = (S.contact_mask >> l) & 1u;
    auto item_full = [&](int k) {
        return S.planets_auto ? collide_item<true, true, false>(S, l, LP, hull, in, k, true, pv.data(), pd.data(), keys, 1)
                              : collide_item<true, false, false>(S, l, LP, hull, in, k, true, pv.data(), pd.data(), keys, 1);
    };
    // as the kernel works an item: the bound first, the real test only when it does not clear the contact distance
    auto item = [&](int k) {
        const bool cand = S.planets_auto ? collide_item<true, true, true>(S, l, LP, hull, in, k, true, pv.data(), pd.data(), keys, 1)
                                         : collide_item<true, false, true>(S, l, LP, hull, in, k, true, pv.data(), pd.data(), keys, 1);
        return cand && item_full(k);
    };
    if (mode == 0) {
        const LaneResult r = S.planets_auto ? collide_lane<true, true>(S, l, LP, hull, in, pv.data(), pd.data(), keys, 1)
                                            : collide_lane<true, false>(S, l, LP, hull, in, pv.data(), pd.data(), keys, 1);
        bool any = false;
        for (int k = r.k_first; k <= r.k_need; ++k) any = any || item(k);
        out[0] = r.contact ? 1.f : 0.f; out[1] = r.d_static; out[2] = r.d_moving; out[3] = (float)(r.k_need >= r.k_first ? r.k_need - r.k_first + 1 : 0);
        out[4] = any ? 1.f : 0.f; out[5] = r.lb_end;
        return;
    }
    const int upd = S.updates_per_step / N;
    bool contact = false;
    const int u_end = ball ? ball_retire_update(S, in.ball_k0, in.ball_kmax, in.ball_khit) : 0;
    for (int k = 1; k < N && !contact && con_link; ++k) {
        if (ball && !(in.ball_active && k * upd < u_end)) continue;
        contact = item_full(k);
    }
    out[0] = out[1] = out[2] = out[3] = 0.f; out[4] = contact ? 1.f : 0.f;
}
extern "C" float emul_sweep_R(void *h, int b, int j) { return ((HostScene *)h)->d.sweep_R[b][j]; }
extern "C" float emul_planet_step(void *h, int p) { return ((HostScene *)h)->d.planet_step[p]; }
