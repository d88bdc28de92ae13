// smr_tables.h - host side: the geometry of an smr_scene_t (float64, include/smr_scene.h) converted to the float32
// tables the device functions of smr_device.h read.  Shared by the HIP library (smr_hip.hip uploads the vectors)
// and by tests/host_emul (which points a DScene at them in host memory).  No HIP runtime calls in here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "smr_device.h"

namespace smr {

constexpr int kMaxPieceVerts = 256;    // neighbour ids are bytes
constexpr int kMaxPieceIdGroups = 1024; // LDS budget of the collide kernel for one link hull (groups of 4 ids)

struct HostGeometry {
    std::vector<float4> verts;     // every piece starts on a 64-byte boundary (wide scalar loads of the scan never straddle it)
    std::vector<DPiece> pieces;
    std::vector<uint32_t> span;    // per vertex slot (indexed like verts): first id group relative to the piece | groups << 16
    std::vector<uint32_t> ids4;    // groups of 4 neighbour ids (one byte each, local to the piece), padded with the vertex itself
    std::vector<uint8_t> start;    // kStartCells per piece: the support vertex of each cube-map cell centre
    std::vector<float4> auto_verts; // lane-autonomous planet queries (planet_tables): padded vertex lists of the planets' pieces + table
    std::vector<float> auto_desc;   // ... and their descriptors
};

// image of the LDS pool of the reset-side kernels: kBodyPoolHead dwords per body (offsets of its vertices / id groups / start map /
// DPiece), then every body's tables as the global ones hold them.  Returns false when they do not fit.
// mask: the bodies whose tables go in (the others keep zero offsets: nobody asks for them)
inline bool body_pool_tables(const smr_scene_t &s, const HostGeometry &G, std::vector<uint32_t> &pool, uint32_t mask = 0xFFFFFFFFu) {
    pool.assign((size_t)kBodyPoolHead * SMR_MAX_BODIES, 0u);
    for (int b = 0; b < s.n_bodies; ++b) {
        if (!((mask >> b) & 1u)) continue;
        const int piece = s.link_piece[b];
        const DPiece &P = G.pieces[piece];
        // the vertex list is padded to a multiple of 8 with copies of its last vertex: the job kernels scan it eight at a time
        const uint32_t padded = ((uint32_t)P.count + 7u) & ~7u;
        pool.resize((pool.size() + 3u) & ~(size_t)3u);          // vertices at 16-byte boundaries: they are read as 128-bit words (load_v4)
        const uint32_t v_off = (uint32_t)pool.size(), ids_off = v_off + 4u * padded, start_off = ids_off + (uint32_t)P.ids_count;
        const uint32_t piece_off = start_off + kStartCells / 4;
        pool[(size_t)kBodyPoolHead * b] = v_off; pool[(size_t)kBodyPoolHead * b + 1] = ids_off; pool[(size_t)kBodyPoolHead * b + 2] = start_off;
        pool[(size_t)kBodyPoolHead * b + 3] = piece_off;
        pool.resize((size_t)piece_off + kPieceDwords);
        std::memcpy(pool.data() + piece_off, &P, 4 * kPieceDwords);
        std::memcpy(pool.data() + v_off, G.verts.data() + P.begin, 16 * (size_t)P.count);
        for (uint32_t k = (uint32_t)P.count; k < padded; ++k) std::memcpy(pool.data() + v_off + 4u * k, G.verts.data() + P.begin + P.count - 1, 16);
        std::memcpy(pool.data() + ids_off, G.ids4.data() + P.ids_begin, 4 * (size_t)P.ids_count);
        std::memcpy(pool.data() + start_off, G.start.data() + (size_t)kStartCells * piece, kStartCells);
    }
    pool.resize((pool.size() + 3u) & ~(size_t)3u);              // ... and so is whatever a kernel puts behind the pool (the planets' vertices)
    return (int)pool.size() <= kBodyPoolDwords;
}

// direction of start-map cell (m, ia, ib): see climb_start() in smr_device.h
inline void start_cell_direction(int m, int ia, int ib, double d[3]) {
    const double a = -1.0 + (2.0 * ia + 1.0) / kStartBins, b = -1.0 + (2.0 * ib + 1.0) / kStartBins, sgn = (m & 1) ? -1.0 : 1.0;
    if (m < 2) { d[0] = sgn; d[1] = a; d[2] = b; }
    else if (m < 4) { d[0] = a; d[1] = sgn; d[2] = b; }
    else { d[0] = a; d[1] = b; d[2] = sgn; }
}

// bounding capsule of n points (float4 xyz): axis = principal axis of the points through their mean (power iteration on the
// covariance), end points pulled in from the extreme projections on a grid, radius = largest distance from the segment
inline void fit_capsule(const float4 *v, int n, float a_out[3], float b_out[3], float &r_out) {
    double m[3] = {0, 0, 0}, C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int k = 0; k < n; ++k) { m[0] += v[k].x; m[1] += v[k].y; m[2] += v[k].z; }
    for (int c = 0; c < 3; ++c) m[c] /= n;
    for (int k = 0; k < n; ++k) {
        const double d[3] = {v[k].x - m[0], v[k].y - m[1], v[k].z - m[2]};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i][j] += d[i] * d[j];
    }
    double ax[3] = {1.0, 0.7, 0.4};
    for (int it = 0; it < 200; ++it) {
        double y[3], nn = 0;
        for (int i = 0; i < 3; ++i) { y[i] = C[i][0] * ax[0] + C[i][1] * ax[1] + C[i][2] * ax[2]; nn += y[i] * y[i]; }
        if (nn < 1e-300) break;
        nn = std::sqrt(nn);
        for (int i = 0; i < 3; ++i) ax[i] = y[i] / nn;
    }
    { double nn = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]); for (int i = 0; i < 3; ++i) ax[i] /= nn; }
    double t0 = 1e300, t1 = -1e300;
    for (int k = 0; k < n; ++k) {
        const double t = (v[k].x - m[0]) * ax[0] + (v[k].y - m[1]) * ax[1] + (v[k].z - m[2]) * ax[2];
        t0 = std::min(t0, t); t1 = std::max(t1, t);
    }
    double best = 1e300;
    const int G = 20;
    for (int i0 = 0; i0 <= G; ++i0)
        for (int i1 = 0; i1 <= G; ++i1) {
            const double L = t1 - t0, s0 = t0 + 0.5 * L * i0 / G, s1 = t1 - 0.5 * L * i1 / G;
            const double a[3] = {m[0] + ax[0] * s0, m[1] + ax[1] * s0, m[2] + ax[2] * s0}, len = s1 - s0;
            double r = 0;
            for (int k = 0; k < n; ++k) {
                const double d[3] = {v[k].x - a[0], v[k].y - a[1], v[k].z - a[2]};
                double t = d[0] * ax[0] + d[1] * ax[1] + d[2] * ax[2];
                t = std::min(std::max(t, 0.0), len);
                const double e[3] = {d[0] - t * ax[0], d[1] - t * ax[1], d[2] - t * ax[2]};
                r = std::max(r, std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]));
            }
            if (r < best) {
                best = r;
                for (int c = 0; c < 3; ++c) { a_out[c] = (float)a[c]; b_out[c] = (float)(a[c] + ax[c] * len); }
            }
        }
    // radius re-measured against the float32 end points, slightly inflated: the bound stays conservative
    double r = 0;
    const double a[3] = {a_out[0], a_out[1], a_out[2]}, e[3] = {b_out[0] - a[0], b_out[1] - a[1], b_out[2] - a[2]};
    const double ee = std::max(e[0] * e[0] + e[1] * e[1] + e[2] * e[2], 1e-300);
    for (int k = 0; k < n; ++k) {
        const double d[3] = {v[k].x - a[0], v[k].y - a[1], v[k].z - a[2]};
        const double t = std::min(std::max((d[0] * e[0] + d[1] * e[1] + d[2] * e[2]) / ee, 0.0), 1.0);
        const double f[3] = {d[0] - t * e[0], d[1] - t * e[1], d[2] - t * e[2]};
        r = std::max(r, std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]));
    }
    r_out = (float)(r * 1.00001 + 1e-6);
}

// returns nullptr on success, else a message
inline const char *build_host_geometry(const smr_scene_t &s, HostGeometry &G) {
    if (!s.adj_offsets || !s.adj) return "scene has no hull adjacency (adj_offsets / adj)";
    G.pieces.assign((size_t)s.n_pieces, DPiece{});
    for (int i = 0; i < s.n_pieces; ++i) {
        const smr_piece_t &p = s.pieces[i];
        if (p.vert_begin < 0 || p.vert_count <= 0 || p.vert_begin + p.vert_count > s.n_verts) return "piece vertex range out of bounds";
        if (p.vert_count > kMaxPieceVerts) return "a piece has more than 256 hull vertices";
        DPiece &q = G.pieces[i];
        while (G.verts.size() % 4) { G.verts.push_back(float4{0.f, 0.f, 0.f, 0.f}); G.span.push_back(0u); }
        q.begin = (int)G.verts.size(); q.count = p.vert_count;
        for (int k = 0; k < p.vert_count; ++k) {
            const double *src = s.verts + 3 * (size_t)(p.vert_begin + k);
            G.verts.push_back(float4{(float)src[0], (float)src[1], (float)src[2], 0.f});
        }
        q.cx = (float)p.center[0]; q.cy = (float)p.center[1]; q.cz = (float)p.center[2];
        q.margin = (float)p.margin;
        q.kind = p.kind; q.cyl_r = (float)p.cyl_radius; q.cyl_h = (float)p.cyl_half_length;
        if (p.kind != SMR_PIECE_HULL && p.kind != SMR_PIECE_CYLINDER_Z) return "unknown piece kind";
        if (p.kind == SMR_PIECE_CYLINDER_Z && !(p.cyl_radius > 0 && p.cyl_half_length > 0)) return "bad cylinder piece";
        // bounding data recomputed in float32 around the float32 centre, slightly inflated, so culling stays conservative
        float r = 0.f, hx = 0.f, hy = 0.f, hz = 0.f;
        for (int k = 0; k < p.vert_count; ++k) {
            const float4 &vv = G.verts[(size_t)q.begin + k];
            float dx = vv.x - q.cx, dy = vv.y - q.cy, dz = vv.z - q.cz;
            r = std::max(r, std::sqrt(dx * dx + dy * dy + dz * dz));
            hx = std::max(hx, std::fabs(dx)); hy = std::max(hy, std::fabs(dy)); hz = std::max(hz, std::fabs(dz));
        }
        q.radius = r * 1.00001f + 1e-7f; q.hx = hx + 1e-7f; q.hy = hy + 1e-7f; q.hz = hz + 1e-7f;
        fit_capsule(&G.verts[(size_t)q.begin], p.vert_count, q.cap_a, q.cap_b, q.cap_r);
        // an axis-aligned box given as its 8 corners (the table): every vertex is a corner of the vertices' bounding box
        q.bx0 = q.bx1 = q.by0 = q.by1 = q.btop = 0.f;
        if (p.kind == SMR_PIECE_HULL && p.vert_count == 8) {
            float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
            for (int k = 0; k < 8; ++k) {
                const float4 &vv = G.verts[(size_t)q.begin + k];
                const float xyz[3] = {vv.x, vv.y, vv.z};
                for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], xyz[c]); hi[c] = std::max(hi[c], xyz[c]); }
            }
            bool box = true; unsigned seen = 0;
            for (int k = 0; k < 8; ++k) {
                const float4 &vv = G.verts[(size_t)q.begin + k];
                const float xyz[3] = {vv.x, vv.y, vv.z};
                unsigned code = 0;
                for (int c = 0; c < 3; ++c) { if (xyz[c] == hi[c]) code |= 1u << c; else if (xyz[c] != lo[c]) box = false; }
                seen |= 1u << code;
            }
            if (box && seen == 0xFFu && hi[0] > lo[0] && hi[1] > lo[1] && hi[2] > lo[2]) {
                q.kind = kPieceAlignedBox;
                q.bx0 = lo[0]; q.bx1 = hi[0]; q.by0 = lo[1]; q.by1 = hi[1]; q.btop = hi[2];
            }
        }
        // three interior points: mean of all vertices, and of the halves above / below the box centre along the longest box axis
        {
            const int ax = (q.hx >= q.hy && q.hx >= q.hz) ? 0 : (q.hy >= q.hz ? 1 : 2);
            double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            int n[3] = {0, 0, 0};
            const float cen[3] = {q.cx, q.cy, q.cz};
            for (int k = 0; k < p.vert_count; ++k) {
                const float4 &vv = G.verts[(size_t)q.begin + k];
                const float xyz[3] = {vv.x, vv.y, vv.z};
                const int half = xyz[ax] >= cen[ax] ? 1 : 2;
                for (int c = 0; c < 3; ++c) { m[0][c] += xyz[c]; m[half][c] += xyz[c]; }
                n[0]++; n[half]++;
            }
            for (int h = 0; h < 3; ++h)
                for (int c = 0; c < 3; ++c) q.in[h][c] = (float)(n[h] ? m[h][c] / n[h] : m[0][c] / n[0]);
        }
        // adjacency: byte ids in groups of four
        q.ids_begin = (int)G.ids4.size();
        for (int k = 0; k < p.vert_count; ++k) {
            const int a0 = s.adj_offsets[p.vert_begin + k], a1 = s.adj_offsets[p.vert_begin + k + 1];
            if (a0 < 0 || a1 < a0) return "bad adjacency offsets";
            const int first = (int)G.ids4.size() - q.ids_begin, groups = (a1 - a0 + 3) / 4;
            if (first + groups > 0xffff || groups > 0xffff) return "hull adjacency too large";
            for (int g = 0; g < groups; ++g) {
                uint32_t word = 0;
                for (int b = 0; b < 4; ++b) {
                    const int a = a0 + 4 * g + b;
                    int id = a < a1 ? s.adj[a] : k;
                    if (id < 0 || id >= p.vert_count) return "adjacency id out of the piece";
                    word |= (uint32_t)id << (8 * b);
                }
                G.ids4.push_back(word);
            }
            G.span.push_back((uint32_t)first | ((uint32_t)groups << 16));
            std::memcpy(&G.verts[(size_t)q.begin + k].w, &G.span.back(), 4);        // the climb reads it with the coordinates
        }
        q.ids_count = (int)G.ids4.size() - q.ids_begin;
        // start map: exhaustive support of the cell-centre directions
        for (int m = 0; m < 6; ++m)
            for (int ia = 0; ia < kStartBins; ++ia)
                for (int ib = 0; ib < kStartBins; ++ib) {
                    double d[3];
                    start_cell_direction(m, ia, ib, d);
                    int best = 0; double bs = -1e300;
                    for (int k = 0; k < p.vert_count; ++k) {
                        const float4 &vv = G.verts[(size_t)q.begin + k];
                        const double sv = vv.x * d[0] + vv.y * d[1] + vv.z * d[2];
                        if (sv > bs) { bs = sv; best = k; }
                    }
                    G.start.push_back((uint8_t)best);
                }
    }
    for (int k = 0; k < 8; ++k) { G.verts.push_back(float4{0.f, 0.f, 0.f, 0.f}); G.span.push_back(0u); }   // slack for the last unrolled load
    G.ids4.push_back(0u);
    return nullptr;
}

// bodies, their mounting transforms and the self-collision pairs (DScene image of the smr_scene_t fields); nullptr or a message
inline const char *scene_bodies(const smr_scene_t &s, DScene &d) {
    if (s.n_bodies < SMR_NLINK || s.n_bodies > SMR_MAX_BODIES) return "bad n_bodies";
    d.n_bodies = s.n_bodies;
    for (int b = 0; b < s.n_bodies; ++b) {
        if (s.link_piece[b] < 0 || s.link_piece[b] >= s.n_pieces) return "bad link_piece";
        if (s.pieces[s.link_piece[b]].kind != SMR_PIECE_HULL) return "a robot body must be a convex hull piece";
        d.link_piece[b] = s.link_piece[b];
        d.body_parent[b] = s.body_parent[b];
        if (b < SMR_NLINK ? s.body_parent[b] != b : s.body_parent[b] != SMR_NLINK - 1) return "bodies beyond the chain must be mounted on iiwa_link_7";
        for (int c = 0; c < 3; ++c) d.body_xyz[b][c] = (float)s.body_xyz[b][c];
        for (int c = 0; c < 9; ++c) d.body_rot[b][c] = (float)s.body_rot[b][c];
    }
    if (s.ee_body < 0 || s.ee_body >= s.n_bodies) return "bad ee_body";
    d.ee_body = s.ee_body;
    if (s.n_self_pairs < 0 || s.n_self_pairs > SMR_MAX_SELF_PAIRS) return "bad n_self_pairs";
    d.n_self_pairs = s.n_self_pairs;
    for (int i = 0; i < s.n_self_pairs; ++i) {
        const int a = s.self_pair[i][0], b = s.self_pair[i][1];
        if (a < 0 || a >= s.n_bodies) return "bad self pair";
        if (b >= 0 ? (b >= SMR_NLINK || b == a) : (-(b + 1) >= s.n_pieces)) return "bad self pair partner";
        d.self_pair[i][0] = a; d.self_pair[i][1] = b;
    }
    return nullptr;
}

// Links that can NEVER come within contact distance of a planet are dropped from the contact queries (their blocks of the
// step kernel would stage their tables, run FK and find nothing, at every sample).  Conservative test per chain link l:
// when the axes of all joints before it pass through its frame origin c (iiwa: links 0, 1, 2 - the shoulder), every point
// of the link keeps its distance from c whatever the joints do, so the link stays inside the ball (c, max |v - c| + margin);
// the link is dropped when that ball stays clear of every planet piece's bounding sphere at every orbit index.
inline uint32_t reachable_contact_mask(const smr_scene_t &s) {
    uint32_t mask = s.contact_link_mask;
    if (s.use_ball || s.n_planets <= 0) return mask;
    double o[SMR_NLINK][3];
    for (int c = 0; c < 3; ++c) o[0][c] = s.base_xyz[c];
    for (int k = 0; k + 1 < SMR_NLINK; ++k)
        for (int c = 0; c < 3; ++c) o[k + 1][c] = o[k][c] + s.joint_xyz[k][c];       // link frames at q = 0 (fk_advance)
    for (int l = 0; l < SMR_NLINK && l < s.n_bodies; ++l) {
        if (!((mask >> l) & 1u)) continue;
        bool fixed_centre = true;
        for (int k = 0; k + 1 < l; ++k) {                   // joint k turns about joint_axis[k] through o[k + 1]
            const double d[3] = {o[l][0] - o[k + 1][0], o[l][1] - o[k + 1][1], o[l][2] - o[k + 1][2]};
            const double *a = s.joint_axis[k];
            const double cx = d[1] * a[2] - d[2] * a[1], cy = d[2] * a[0] - d[0] * a[2], cz = d[0] * a[1] - d[1] * a[0];
            if (std::sqrt(cx * cx + cy * cy + cz * cz) > 1e-9) fixed_centre = false;
        }
        if (!fixed_centre) continue;
        const smr_piece_t &lp = s.pieces[s.link_piece[l]];
        double rl = 0;
        for (int k = 0; k < lp.vert_count; ++k) {
            const double *v = s.verts + 3 * (size_t)(lp.vert_begin + k);
            rl = std::max(rl, std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
        }
        rl += lp.margin;
        bool reach = false;
        for (int p = 0; p < s.n_planets && !reach; ++p) {
            const smr_planet_t &pl = s.planet[p];
            for (int c = 0; c < pl.piece_count && !reach; ++c) {
                const smr_piece_t &pp = s.pieces[pl.piece_begin + c];
                double rp = 0;
                for (int k = 0; k < pp.vert_count; ++k) {
                    const double *v = s.verts + 3 * (size_t)(pp.vert_begin + k);
                    const double dx = v[0] - pp.center[0], dy = v[1] - pp.center[1], dz = v[2] - pp.center[2];
                    rp = std::max(rp, std::sqrt(dx * dx + dy * dy + dz * dz));
                }
                rp += pp.margin;
                for (int i = 0; i < pl.n_orbit && !reach; ++i) {
                    const double *row = pl.orbit_pose + 7 * (size_t)i;
                    const double x = row[3], y = row[4], z = row[5], w = row[6], *m = pp.center;
                    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
                    double dd = 0;
                    for (int r = 0; r < 3; ++r) {
                        const double cw = R[3 * r] * m[0] + R[3 * r + 1] * m[1] + R[3 * r + 2] * m[2] + row[r] - o[l][r];
                        dd += cw * cw;
                    }
                    if (std::sqrt(dd) - rp - rl <= s.contact_distance + 1e-3) reach = true;
                }
            }
        }
        if (!reach) mask &= ~(1u << l);
    }
    return mask;
}

// ... and the same for the static pieces (the table): a link whose ball (c, r) stays farther from every static piece's
// bounding box than the largest distance the queries report leaves the static queries (iiwa_link_2: the shoulder is 0.36 m up).
inline uint32_t reachable_static_mask(const smr_scene_t &s) {
    uint32_t mask = s.static_link_mask;
    double o[SMR_NLINK][3];
    for (int c = 0; c < 3; ++c) o[0][c] = s.base_xyz[c];
    for (int k = 0; k + 1 < SMR_NLINK; ++k)
        for (int c = 0; c < 3; ++c) o[k + 1][c] = o[k][c] + s.joint_xyz[k][c];
    for (int l = 0; l < SMR_NLINK && l < s.n_bodies; ++l) {
        if (!((mask >> l) & 1u)) continue;
        bool fixed_centre = true;
        for (int k = 0; k + 1 < l; ++k) {
            const double d[3] = {o[l][0] - o[k + 1][0], o[l][1] - o[k + 1][1], o[l][2] - o[k + 1][2]};
            const double *a = s.joint_axis[k];
            const double cx = d[1] * a[2] - d[2] * a[1], cy = d[2] * a[0] - d[0] * a[2], cz = d[0] * a[1] - d[1] * a[0];
            if (std::sqrt(cx * cx + cy * cy + cz * cz) > 1e-9) fixed_centre = false;
        }
        if (!fixed_centre) continue;
        const smr_piece_t &lp = s.pieces[s.link_piece[l]];
        double rl = 0;
        for (int k = 0; k < lp.vert_count; ++k) {
            const double *v = s.verts + 3 * (size_t)(lp.vert_begin + k);
            rl = std::max(rl, std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
        }
        rl += lp.margin;
        bool reach = false;
        for (int i = 0; i < s.n_static && !reach; ++i) {
            const smr_piece_t &pp = s.pieces[s.static_piece[i]];
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            if (pp.kind == SMR_PIECE_HULL && pp.vert_count > 0) {
                for (int k = 0; k < pp.vert_count; ++k)
                    for (int c = 0; c < 3; ++c) { const double x = s.verts[3 * (size_t)(pp.vert_begin + k) + c]; lo[c] = std::min(lo[c], x); hi[c] = std::max(hi[c], x); }
            } else { reach = true; break; }              // no cheap box for other kinds: keep the link
            double dd = 0;
            for (int c = 0; c < 3; ++c) { const double e = std::max(std::max(lo[c] - o[l][c], o[l][c] - hi[c]), 0.0); dd += e * e; }
            // the largest distance any static query asks about: the reported range, and the clearances of start states / targets
            const double ask = std::max(std::max(s.static_max_distance, s.start_min_dist_static),
                                        std::max(s.use_ball ? s.ball_target_min_dist_static : 0.0, s.use_target_points ? s.target_min_dist_static : 0.0)) + 0.006;
            if (std::sqrt(dd) - rl - pp.margin <= ask) reach = true;
        }
        if (!reach) mask &= ~(1u << l);
    }
    return mask;
}

// Tables of the lane-autonomous planet queries (planet_queries_auto in smr_device.h), when the planets fit their budget:
// the pieces' vertex lists padded to multiples of 4 (copies of the last vertex: the scan needs no bounds test) and
// one descriptor per piece.  d.auto_verts / d.auto_desc are left to the caller (device copies of G.auto_verts / G.auto_desc).
inline void planet_tables(HostGeometry &G, DScene &d) {
    const std::vector<DPiece> &pieces = G.pieces;
    d.planets_auto = 0; d.planet_p0 = d.planet_pn = d.planet_vn = 0;
    d.auto_static = -1; d.auto_pn = d.auto_vn = 0;
    G.auto_verts.clear(); G.auto_desc.clear();
    if (d.n_planets <= 0 || d.use_ball) return;
    int p0 = 1 << 30, p1 = 0, covered = 0;
    for (int p = 0; p < d.n_planets; ++p) {
        p0 = std::min(p0, d.planet[p].piece_begin); p1 = std::max(p1, d.planet[p].piece_begin + d.planet[p].piece_count);
        covered += d.planet[p].piece_count;
    }
    if (covered != p1 - p0 || p1 - p0 > kMaxPlanetPieces) return;               // contiguous piece range, one key byte each
    if (d.n_planets > 1 && d.planet[1].piece_begin < d.planet[0].piece_begin) return;
    auto add = [&](int pi, int frame) {
        const DPiece &P = pieces[pi];
        const int first = (int)G.auto_verts.size(), padded = (P.count + 7) / 8 * 8;      // the scans read 4 or 8 vertices per round
        for (int k = 0; k < padded; ++k) G.auto_verts.push_back(G.verts[(size_t)P.begin + std::min(k, P.count - 1)]);
        float w[kPlanetDescDwords] = {0, 0, P.margin, 0, P.cx, P.cy, P.cz, 0};
        std::memcpy(&w[0], &first, 4); std::memcpy(&w[1], &padded, 4); std::memcpy(&w[3], &frame, 4);
        G.auto_desc.insert(G.auto_desc.end(), w, w + kPlanetDescDwords);
    };
    for (int i = p0; i < p1; ++i) add(i, (d.n_planets > 1 && i >= d.planet[1].piece_begin) ? 1 : 0);
    if ((int)G.auto_verts.size() > kMaxPlanetVerts) { G.auto_verts.clear(); G.auto_desc.clear(); return; }
    d.planets_auto = 1; d.planet_p0 = p0; d.planet_pn = p1 - p0; d.planet_vn = (int)G.auto_verts.size();
    d.auto_pn = d.planet_pn; d.auto_vn = d.planet_vn;
    // the static piece behind them (the table: an axis-aligned box), world frame - its GJK runs join the planet runs' loop
    if (d.n_static == 1 && pieces[d.static_piece[0]].kind == kPieceAlignedBox && d.planet_pn < kMaxPlanetPieces &&
        d.planet_vn + (pieces[d.static_piece[0]].count + 7) / 8 * 8 <= kMaxPlanetVerts) {
        add(d.static_piece[0], 2);
        d.auto_static = d.planet_pn; d.auto_pn = d.planet_pn + 1; d.auto_vn = (int)G.auto_verts.size();
    }
}

// Contact sweep constants (smr_device.h, "contact sweep").  sweep_R[b][j]: no point of body b (margin included) is farther from the
// axis of joint j than the link lengths between them plus the body's extent about its link frame, so it moves no faster than
// sum_j |qdot_j| sweep_R[b][j] (joint j turns about an axis through the origin of link j + 1, fk_advance).
inline void sweep_tables(const smr_scene_t &s, const HostGeometry &G, DScene &d) {
    for (int b = 0; b < s.n_bodies; ++b) {
        const DPiece &P = G.pieces[s.link_piece[b]];
        double rho = 0;
        for (int k = 0; k < P.count; ++k) {
            const float4 &v = G.verts[(size_t)P.begin + k];
            double x[3] = {v.x, v.y, v.z};
            if (b >= SMR_NLINK) {
                double y[3];
                for (int r = 0; r < 3; ++r) y[r] = s.body_xyz[b][r] + s.body_rot[b][3 * r] * x[0] + s.body_rot[b][3 * r + 1] * x[1] + s.body_rot[b][3 * r + 2] * x[2];
                for (int r = 0; r < 3; ++r) x[r] = y[r];
            }
            rho = std::max(rho, std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]));
        }
        rho += P.margin;
        const int chain = b < SMR_NLINK ? b : SMR_NLINK - 1;
        for (int j = 0; j < SMR_NJ; ++j) {
            double R = rho;
            for (int i = j + 1; i < chain; ++i) R += std::sqrt(s.joint_xyz[i][0] * s.joint_xyz[i][0] + s.joint_xyz[i][1] * s.joint_xyz[i][1] + s.joint_xyz[i][2] * s.joint_xyz[i][2]);
            d.sweep_R[b][j] = j < chain ? (float)(R * 1.00001 + 1e-6) : 0.f;
        }
    }
    {   // the target link point: ee_offset in the frame of ee_body (a chain link, or a body mounted on the last one)
        const int b = s.ee_body;
        double x[3] = {s.ee_offset[0], s.ee_offset[1], s.ee_offset[2]};
        if (b >= SMR_NLINK) {
            double y[3];
            for (int r = 0; r < 3; ++r) y[r] = s.body_xyz[b][r] + s.body_rot[b][3 * r] * x[0] + s.body_rot[b][3 * r + 1] * x[1] + s.body_rot[b][3 * r + 2] * x[2];
            for (int r = 0; r < 3; ++r) x[r] = y[r];
        }
        const double rho = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        const int chain = (b >= 0 && b < SMR_NLINK) ? b : SMR_NLINK - 1;
        for (int j = 0; j < SMR_NJ; ++j) {
            double R = rho;
            for (int i = j + 1; i < chain; ++i) R += std::sqrt(s.joint_xyz[i][0] * s.joint_xyz[i][0] + s.joint_xyz[i][1] * s.joint_xyz[i][1] + s.joint_xyz[i][2] * s.joint_xyz[i][2]);
            d.ee_R[j] = j < chain ? (float)(R * 1.00001 + 1e-6) : 0.f;
        }
    }
    d.ball_bound_radius = 0.f;
    if (s.use_ball && s.ball_piece >= 0 && s.ball_piece < s.n_pieces) {
        const DPiece &P = G.pieces[s.ball_piece];
        double r = 0;
        for (int k = 0; k < P.count; ++k) { const float4 &v = G.verts[(size_t)P.begin + k]; r = std::max(r, std::sqrt((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z)); }
        d.ball_bound_radius = (float)((r + P.margin) * 1.00001 + 1e-6);
    }
}

// Layout of the self-pair bound tables (smr_device.h, "Self pairs: bound tables"): a pair (body mounted on iiwa_link_7, partner link k)
// with k >= 4 gets a grid over the joints k .. 6 - fine for two axes, coarse for three.  Returns the number of uint16 entries in all.
// lever of joint j: the body's points stay within sweep_R of the joint's axis; the last joint's axis is fixed in the body's parent
// frame, so there the distance from the axis itself is taken (a third less than the distance from the frame origin).
// Also the LDS slots of the world-fixed cylinders the pairs name (self_cyl_slot).
inline size_t self_table_layout(const smr_scene_t &s, const HostGeometry &G, DScene &d) {
    size_t total = 0;
    int n_cyl = 0;
    for (int i = 0; i < SMR_MAX_SELF_PAIRS; ++i) { d.self_tab_nd[i] = 0; d.self_tab_off[i] = 0; d.self_cyl_slot[i] = -1; }
    for (int i = 0; i < s.n_self_pairs; ++i) {
        const int a = s.self_pair[i][0], partner = s.self_pair[i][1];
        if (partner < 0) { d.self_cyl_slot[i] = n_cyl++; continue; }
        if (a < SMR_NLINK || partner < 4 || partner > 5) continue;
        const int nd = SMR_NJ - partner;
        // cells: ~0.016 rad for two axes (98 k entries), ~0.09 rad for three (180 k)
        const double width = nd == 2 ? 0.016 : 0.092;
        size_t cells = 1;
        for (int k = 0; k < nd; ++k) {
            const int j = SMR_NJ - nd + k;
            const double range = s.pos_hi[j] - s.pos_lo[j];
            const int n = std::max(1, (int)std::ceil(range / width));
            d.self_tab_n[i][k] = n;
            d.self_tab_h[i][k] = (float)(range / n);
            d.self_tab_ih[i][k] = (float)(n / range);
            double lever = d.sweep_R[a][j];
            if (j == SMR_NJ - 1) {                          // distance from the axis of the last joint (through the parent frame's origin)
                const DPiece &P = G.pieces[s.link_piece[a]];
                const double *ax = s.joint_axis[j];
                double r = 0;
                for (int v = 0; v < P.count; ++v) {
                    const float4 &p = G.verts[(size_t)P.begin + v];
                    double x[3];
                    for (int c = 0; c < 3; ++c) x[c] = s.body_xyz[a][c] + s.body_rot[a][3 * c] * p.x + s.body_rot[a][3 * c + 1] * p.y + s.body_rot[a][3 * c + 2] * p.z;
                    const double t = x[0] * ax[0] + x[1] * ax[1] + x[2] * ax[2];
                    const double e[3] = {x[0] - t * ax[0], x[1] - t * ax[1], x[2] - t * ax[2]};
                    r = std::max(r, std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]));
                }
                lever = std::min(lever, (r + P.margin) * 1.00001 + 1e-6);
            }
            d.self_tab_lever[i][k] = (float)lever;
            cells *= (size_t)n;
        }
        d.self_tab_nd[i] = nd;
        d.self_tab_off[i] = (int)total;
        total += cells;
    }
    return total;
}

// planet_step: the farthest any vertex of the planet's pieces moves from one orbit index to the next (float32 pose table
// [n_orbit][8] as the kernels read it, same quaternion -> matrix formula as quat_pose), rounded up; the margin moves with it
inline float planet_step_bound(const HostGeometry &G, const DPlanet &q, const float *pose) {
    auto xf = [&](int i, double R[9], double t[3]) {
        const float *row = pose + 8 * (size_t)i;
        const double x = row[3], y = row[4], z = row[5], w = row[6], n = x * x + y * y + z * z + w * w, sc = n > 0 ? 2.0 / n : 0.0;
        R[0] = 1 - sc * (y * y + z * z); R[1] = sc * (x * y - z * w); R[2] = sc * (x * z + y * w);
        R[3] = sc * (x * y + z * w); R[4] = 1 - sc * (x * x + z * z); R[5] = sc * (y * z - x * w);
        R[6] = sc * (x * z - y * w); R[7] = sc * (y * z + x * w); R[8] = 1 - sc * (x * x + y * y);
        t[0] = row[0]; t[1] = row[1]; t[2] = row[2];
    };
    double worst = 0;
    for (int i = 0; i < q.n_orbit; ++i) {
        double Ra[9], ta[3], Rb[9], tb[3];
        xf(i, Ra, ta); xf((i + 1) % q.n_orbit, Rb, tb);
        for (int c = 0; c < q.piece_count; ++c) {
            const DPiece &P = G.pieces[q.piece_begin + c];
            for (int k = 0; k < P.count; ++k) {
                const float4 &v = G.verts[(size_t)P.begin + k];
                double dd = 0;
                for (int r = 0; r < 3; ++r) {
                    const double a = Ra[3 * r] * v.x + Ra[3 * r + 1] * v.y + Ra[3 * r + 2] * v.z + ta[r];
                    const double b = Rb[3 * r] * v.x + Rb[3 * r + 1] * v.y + Rb[3 * r + 2] * v.z + tb[r];
                    dd += (a - b) * (a - b);
                }
                worst = std::max(worst, std::sqrt(dd));
            }
        }
    }
    return (float)(worst * 1.0001 + 2e-6);
}

// bounding sphere of a whole planet: centre = mean of the piece centres, radius covers every piece sphere + margin
inline void planet_bounds(const std::vector<DPiece> &pieces, DPlanet &q) {
    double cx = 0, cy = 0, cz = 0;
    for (int c = 0; c < q.piece_count; ++c) { const DPiece &pp = pieces[q.piece_begin + c]; cx += pp.cx; cy += pp.cy; cz += pp.cz; }
    q.bx = (float)(cx / q.piece_count); q.by = (float)(cy / q.piece_count); q.bz = (float)(cz / q.piece_count);
    float br = 0.f;
    for (int c = 0; c < q.piece_count; ++c) {
        const DPiece &pp = pieces[q.piece_begin + c];
        float dx = pp.cx - q.bx, dy = pp.cy - q.by, dz = pp.cz - q.bz;
        br = std::max(br, std::sqrt(dx * dx + dy * dy + dz * dz) + pp.radius + pp.margin);
    }
    q.bradius = br * 1.00001f + 1e-6f;
}

}  // namespace smr
