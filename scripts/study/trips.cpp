// CPU study (not a product path): how many GJK iterations ("trips") the lanes of the collide kernel's planet loop need under
// different cutoff policies, on the host build of the device functions.  See scripts/study/trips.py.
#include "../../tests/host_emul/emul.cpp"

// one (env, link) lane: keys, then best-first runs; iterations are counted per run.  cutoff0: initial cutoff on the moving distance
// (the scene's moving_max, or an env-level bound).  Returns the number of iterations; d_out = the lane's moving distance (or > cutoff)
static int lane_trips(const DScene &S, const float *pv, const float *pd, int l, const float *q, const int *idx, float cutoff0, float *d_out,
                      int *runs_out, float *lbmin_out, int *first_iters) {
    Xf T = fk_base(S);
    for (int j = 0; j < l; ++j) fk_advance(S, j, q[j], T);
    const DPiece LP = load_piece(S.pieces, S.link_piece[l]);
    const GlobalHull hull = global_hull(S, S.link_piece[l], LP);
    uint32_t keys[7];
    const float lb = planet_keys(S, keys, 1, LP, T, idx, true, S.moving_max);
    *lbmin_out = lb;
    Sample out; out.d_static = S.static_max; out.d_self = S.static_max; out.d_moving = cutoff0; out.contact = false;
    const long it0 = 0;
    long before = 0; for (int i = 0; i < 64; ++i) before += emul_slot_counters[i];
    const long runs0 = emul_counters[2];
    planet_gjk(S, pv, pd, keys, 1, LP, hull, T, idx, true, false, S.moving_max, out);
    long after = 0; for (int i = 0; i < 64; ++i) after += emul_slot_counters[i];
    *d_out = out.d_moving;
    *runs_out = (int)(emul_counters[2] - runs0) - 1;      // g.start() of the idle state counts one
    (void)it0; (void)first_iters;
    return (int)(after - before);
}

extern "C" void study_trips(void *h, int n, const float *q /*[n][7]*/, const int *idx /*[n][2]*/, const int *links, int n_links,
                            int *trips0 /*[n][n_links]*/, int *trips_ideal, int *trips_first, float *dmin /*[n]*/, int *runs0, int *runs_ideal, float *lbmin /*[n][n_links]*/) {
    HostScene *H = (HostScene *)h;
    const DScene &S = H->d;
    static std::vector<float> pv, pd;
    pv.assign(4 * kMaxPlanetVerts, 0.f); pd.assign(kPlanetDescDwords * kMaxPlanetPieces, 0.f);
    stage_planets(S, 0, 1, pv.data(), pd.data());
    for (int e = 0; e < n; ++e) {
        float dm = S.moving_max + 0.002f;
        std::vector<float> dl(n_links);
        for (int i = 0; i < n_links; ++i) {
            int r; float lb;
            trips0[e * n_links + i] = lane_trips(S, pv.data(), pd.data(), links[i], q + 7 * e, idx + 2 * e, S.moving_max + 0.002f, &dl[i], &r, &lb, nullptr);
            runs0[e * n_links + i] = r; lbmin[e * n_links + i] = lb;
            dm = std::min(dm, dl[i]);
        }
        dmin[e] = dm;
        // ideal: every lane starts with the env's true minimum as cutoff (slightly above, so that the attaining pair is not cut)
        for (int i = 0; i < n_links; ++i) {
            int r; float lb, d;
            trips_ideal[e * n_links + i] = lane_trips(S, pv.data(), pd.data(), links[i], q + 7 * e, idx + 2 * e, std::min(dm * 1.00001f + 1e-6f, S.moving_max + 0.002f), &d, &r, &lb, nullptr);
            runs_ideal[e * n_links + i] = r;
        }
        // "first run" policy: the link with the smallest key bound runs alone first (its own full query), its result is the cutoff of the others
        int best = 0;
        for (int i = 1; i < n_links; ++i) if (lbmin[e * n_links + i] < lbmin[e * n_links + best]) best = i;
        for (int i = 0; i < n_links; ++i) {
            int r; float lb, d;
            const float c = i == best ? S.moving_max + 0.002f : std::min(dl[best] * 1.00001f + 1e-6f, S.moving_max + 0.002f);
            trips_first[e * n_links + i] = lane_trips(S, pv.data(), pd.data(), links[i], q + 7 * e, idx + 2 * e, c, &d, &r, &lb, nullptr);
        }
    }
}

// job form: the env's best pair (smallest key over all links) runs first with the scene's cutoff; every other (link, piece) whose key
// bound does not exceed that distance becomes an independent job with that distance as cutoff.  out: per env jobs, total iterations,
// max iterations of a job; first-run iterations
extern "C" void study_jobs(void *h, int n, const float *q, const int *idx, const int *links, int n_links, int *n_jobs, int *iters_total, int *iters_max,
                           int *iters_first, float *dmin_out, int *hist /*[33]*/) {
    HostScene *H = (HostScene *)h;
    const DScene &S = H->d;
    static std::vector<float> pv, pd;
    pv.assign(4 * kMaxPlanetVerts, 0.f); pd.assign(kPlanetDescDwords * kMaxPlanetPieces, 0.f);
    stage_planets(S, 0, 1, pv.data(), pd.data());
    for (int e = 0; e < n; ++e) {
        uint32_t keys[16][7];
        Xf T[16]; DPiece LP[16];
        int bl = -1, bp = -1; uint32_t bk = 255;
        for (int i = 0; i < n_links; ++i) {
            const int l = links[i];
            T[i] = body_frame(S, q + 7 * e, l);
            LP[i] = load_piece(S.pieces, S.link_piece[l]);
            planet_keys(S, keys[i], 1, LP[i], T[i], idx + 2 * e, true, S.moving_max);
            for (int c = 0; c < S.planet_pn; ++c) { const uint32_t k = (keys[i][c >> 2] >> (8 * (c & 3))) & 255u; if (k < bk) { bk = k; bl = i; bp = c; } }
        }
        n_jobs[e] = 0; iters_total[e] = 0; iters_max[e] = 0; iters_first[e] = 0; dmin_out[e] = S.moving_max + 0.002f;
        if (bl < 0) continue;
        auto run = [&](int i, int c, float cutoff) {
            const float *d = pd.data() + kPlanetDescDwords * c;
            const float *av = pv.data() + 4 * __float_as_int(d[0]);
            const int ac = __float_as_int(d[1]);
            const float margins = d[2] + LP[i].margin;
            const int p = __float_as_int(d[3]);
            const Xf XA = inv_compose(T[i], quat_pose((p ? S.planet[1].orbit_pose : S.planet[0].orbit_pose) + 8 * (size_t)(p ? idx[2 * e + 1] : idx[2 * e])));
            const GlobalHull hull = global_hull(S, S.link_piece[links[i]], LP[i]);
            const auto sb = [&](F3 dd, int &hint, bool first) { return hull.support(dd, hint, first); };
            const auto sa = [&](F3 dd, int &, bool) {
                const F3 dl = rot_t(XA, dd); float best = -3e38f; int bi = 0;
                for (int k = 0; k < ac; ++k) { const float *qq = av + 4 * k; const float sq = qq[2] * dl.z + qq[1] * dl.y + qq[0] * dl.x; if (sq > best) { best = sq; bi = k; } }
                return apply(XA, F3{av[4 * bi], av[4 * bi + 1], av[4 * bi + 2]});
            };
            GjkRun g; g.start(apply(XA, F3{d[4], d[5], d[6]}) - F3{LP[i].cx, LP[i].cy, LP[i].cz}, cutoff + margins, S.collision_dist + margins);
            float dist = 0.f; int it = 0;
            for (;;) { ++it; if (gjk_iterate(g, sa, sb, dist)) break; }
            return std::make_pair(dist - margins, it);
        };
        auto r0 = run(bl, bp, S.moving_max);
        iters_first[e] = r0.second; hist[std::min(r0.second, 32)]++;
        float dm = r0.first <= S.moving_max ? r0.first : S.moving_max + 0.002f;
        const float cut = std::min(dm, S.moving_max);
        for (int i = 0; i < n_links; ++i)
            for (int c = 0; c < S.planet_pn; ++c) {
                if (i == bl && c == bp) continue;
                const uint32_t k = (keys[i][c >> 2] >> (8 * (c & 3))) & 255u;
                if (k == 255u || (float)k * kKeyStep > cut) continue;
                auto r = run(i, c, cut);
                n_jobs[e]++; iters_total[e] += r.second; iters_max[e] = std::max(iters_max[e], r.second); hist[std::min(r.second, 32)]++;
                if (r.first <= S.moving_max) dm = std::min(dm, r.first);
            }
        dmin_out[e] = dm;
    }
}

// how many pairs survive (key bound <= cutoff) when the cutoff is the UPPER bound the env's best pair leaves after k GJK iterations
// (k = 1 .. 6; k = 0: exact distance), and when every body uses the bound of ITS best piece after k iterations instead (no env-level exchange)
extern "C" void study_ub(void *h, int n, const float *q, const int *idx, const int *links, int n_links, double *jobs_env /*[7]*/, double *jobs_body /*[7]*/) {
    HostScene *H = (HostScene *)h;
    const DScene &S = H->d;
    static std::vector<float> pv, pd;
    pv.assign(4 * kMaxPlanetVerts, 0.f); pd.assign(kPlanetDescDwords * kMaxPlanetPieces, 0.f);
    stage_planets(S, 0, 1, pv.data(), pd.data());
    for (int k = 0; k < 7; ++k) jobs_env[k] = jobs_body[k] = 0;
    for (int e = 0; e < n; ++e) {
        uint32_t keys[16][7];
        Xf T[16]; DPiece LP[16];
        int bl = -1, bp = -1; uint32_t bk = 255;
        int bpl[16]; uint32_t bkl[16];
        for (int i = 0; i < n_links; ++i) {
            const int l = links[i];
            T[i] = body_frame(S, q + 7 * e, l);
            LP[i] = load_piece(S.pieces, S.link_piece[l]);
            planet_keys(S, keys[i], 1, LP[i], T[i], idx + 2 * e, true, S.moving_max);
            bkl[i] = 255; bpl[i] = -1;
            for (int c = 0; c < S.planet_pn; ++c) { const uint32_t k = (keys[i][c >> 2] >> (8 * (c & 3))) & 255u; if (k < bk) { bk = k; bl = i; bp = c; } if (k < bkl[i]) { bkl[i] = k; bpl[i] = c; } }
        }
        if (bl < 0) continue;
        auto ub_after = [&](int i, int c, int kmax) {      // kmax = 0: run to the end
            const float *d = pd.data() + kPlanetDescDwords * c;
            const float *av = pv.data() + 4 * __float_as_int(d[0]);
            const int ac = __float_as_int(d[1]);
            const float margins = d[2] + LP[i].margin;
            const int p = __float_as_int(d[3]);
            const Xf XA = inv_compose(T[i], quat_pose((p ? S.planet[1].orbit_pose : S.planet[0].orbit_pose) + 8 * (size_t)(p ? idx[2 * e + 1] : idx[2 * e])));
            const GlobalHull hull = global_hull(S, S.link_piece[links[i]], LP[i]);
            const auto sb = [&](F3 dd, int &hint, bool first) { return hull.support(dd, hint, first); };
            const auto sa = [&](F3 dd, int &, bool) {
                const F3 dl = rot_t(XA, dd); float best = -3e38f; int bi = 0;
                for (int k = 0; k < ac; ++k) { const float *qq = av + 4 * k; const float sq = qq[2] * dl.z + qq[1] * dl.y + qq[0] * dl.x; if (sq > best) { best = sq; bi = k; } }
                return apply(XA, F3{av[4 * bi], av[4 * bi + 1], av[4 * bi + 2]});
            };
            GjkRun g; g.start(apply(XA, F3{d[4], d[5], d[6]}) - F3{LP[i].cx, LP[i].cy, LP[i].cz}, S.moving_max + margins, S.collision_dist + margins);
            float dist = 0.f;
            for (int it = 0; kmax == 0 || it < kmax; ++it) if (gjk_iterate(g, sa, sb, dist)) return std::min(dist - margins, S.moving_max);
            return std::min(std::sqrt(g.vv) - margins, S.moving_max);
        };
        auto count = [&](int i, float cut) { int nn = 0; for (int c = 0; c < S.planet_pn; ++c) { const uint32_t k = (keys[i][c >> 2] >> (8 * (c & 3))) & 255u; if (k != 255u && (float)k * kKeyStep <= cut) ++nn; } return nn; };
        for (int k = 0; k < 7; ++k) {
            const float ub = ub_after(bl, bp, k);
            for (int i = 0; i < n_links; ++i) jobs_env[k] += count(i, ub);
            for (int i = 0; i < n_links; ++i) if (bpl[i] >= 0) jobs_body[k] += count(i, ub_after(i, bpl[i], k));
        }
    }
    for (int k = 0; k < 7; ++k) { jobs_env[k] /= n; jobs_body[k] /= n; }
}

// iterations of a job against the ratio (key bound / cutoff): hist[bin of ratio (10)][min(iterations, 15)]
extern "C" void study_ratio(void *h, int n, const float *q, const int *idx, const int *links, int n_links, int *hist /*[10][16]*/) {
    HostScene *H = (HostScene *)h;
    const DScene &S = H->d;
    static std::vector<float> pv, pd;
    pv.assign(4 * kMaxPlanetVerts, 0.f); pd.assign(kPlanetDescDwords * kMaxPlanetPieces, 0.f);
    stage_planets(S, 0, 1, pv.data(), pd.data());
    for (int e = 0; e < n; ++e) {
        uint32_t keys[16][7];
        Xf T[16]; DPiece LP[16];
        float ub = S.moving_max;
        auto run = [&](int i, int c, float cutoff, int kmax, int *iters) {
            const float *d = pd.data() + kPlanetDescDwords * c;
            const float *av = pv.data() + 4 * __float_as_int(d[0]);
            const int ac = __float_as_int(d[1]);
            const float margins = d[2] + LP[i].margin;
            const int p = __float_as_int(d[3]);
            const Xf XA = inv_compose(T[i], quat_pose((p ? S.planet[1].orbit_pose : S.planet[0].orbit_pose) + 8 * (size_t)(p ? idx[2 * e + 1] : idx[2 * e])));
            const GlobalHull hull = global_hull(S, S.link_piece[links[i]], LP[i]);
            const auto sb = [&](F3 dd, int &hint, bool first) { return hull.support(dd, hint, first); };
            const auto sa = [&](F3 dd, int &, bool) {
                const F3 dl = rot_t(XA, dd); float best = -3e38f; int bi = 0;
                for (int k = 0; k < ac; ++k) { const float *qq = av + 4 * k; const float sq = qq[2] * dl.z + qq[1] * dl.y + qq[0] * dl.x; if (sq > best) { best = sq; bi = k; } }
                return apply(XA, F3{av[4 * bi], av[4 * bi + 1], av[4 * bi + 2]});
            };
            GjkRun g; g.start(apply(XA, F3{d[4], d[5], d[6]}) - F3{LP[i].cx, LP[i].cy, LP[i].cz}, cutoff + margins, S.collision_dist + margins);
            float dist = 0.f; int it = 0;
            for (;;) { ++it; if (gjk_iterate(g, sa, sb, dist)) break; if (kmax && it >= kmax) { *iters = it; return std::sqrt(g.vv) - margins; } }
            *iters = it;
            return dist - margins;
        };
        for (int i = 0; i < n_links; ++i) {
            const int l = links[i];
            T[i] = body_frame(S, q + 7 * e, l);
            LP[i] = load_piece(S.pieces, S.link_piece[l]);
            planet_keys(S, keys[i], 1, LP[i], T[i], idx + 2 * e, true, S.moving_max);
            uint32_t bk = 255; int bp = -1;
            for (int c = 0; c < S.planet_pn; ++c) { const uint32_t k = (keys[i][c >> 2] >> (8 * (c & 3))) & 255u; if (k < bk) { bk = k; bp = c; } }
            if (bp >= 0) { int it; ub = std::min(ub, run(i, bp, S.moving_max, 1, &it)); }
        }
        for (int i = 0; i < n_links; ++i)
            for (int c = 0; c < S.planet_pn; ++c) {
                const uint32_t k = (keys[i][c >> 2] >> (8 * (c & 3))) & 255u;
                if (k == 255u || (float)k * kKeyStep > ub) continue;
                int it; run(i, c, ub, 0, &it);
                const int bin = std::min(9, (int)(10.f * (float)k * kKeyStep / std::max(ub, 1e-6f)));
                hist[bin * 16 + std::min(it, 15)]++;
            }
    }
}
