// host_geom.hip -- two pieces of the asset / editor side of the pose path that decide what the device code is given, as plain host
// functions of the C ABI (no context, no GPU):
//   fyx_curve_simplify            which keys an imported curve keeps (glTF importer: fyrox-impl/src/resource/gltf/simplify.rs:28-140,
//                                 run on every curve by gltf/animation.rs:155-163, :292 with the binding's epsilon / max_step, :50-65)
//   fyx_blend_space_triangulate   the triangles of a BlendSpace after set_points (fyrox-animation/src/machine/node/blendspace.rs:416-447)
// so that a shim that builds tracks or edits blend spaces outside the engine's own importer gets the engine's answers.
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <array>
#include <utility>
#include <vector>

#include "../../include/fyrox_hip.h"

extern "C" {

int fyx_curve_simplify(const float* x, const float* y, uint32_t n, float epsilon, float max_step, uint32_t* out_indices, uint32_t* out_count) {
    if (!out_count || (n && (!x || !y || !out_indices))) return FYX_ERR_INVALID_ARG;
    *out_count = 0;
    if (n == 0) return FYX_OK;                                          // simplify.rs:44-46
    try {
        std::vector<uint8_t> keep(n, 0);
        const size_t end = (size_t)n - 1;
        keep[0] = keep[end] = 1;
        // find_points_in_span (simplify.rs:106-140) with the recursion on an explicit stack; the two halves of a span are
        // independent, so the order they are visited in changes nothing
        std::vector<std::pair<size_t, size_t>> spans;
        spans.emplace_back(0, end);
        while (!spans.empty()) {
            const auto [s, e] = spans.back();
            spans.pop_back();
            if (e <= s + 1) continue;
            const float x0 = x[s], y0 = y[s];
            const float slope = (y[e] - y0) / (x[e] - x0);
            size_t far = 0;
            float far_dist = 0.0f;
            for (size_t i = s + 1; i < e; ++i) {
                const float y_line = y0 + slope * (x[i] - x0);
                const float dist = std::fabs(y[i] - y_line);
                if (far_dist < dist) { far_dist = dist; far = i; }
            }
            if (far == 0 || far_dist < epsilon) continue;
            keep[far] = 1;
            spans.emplace_back(s, far);
            spans.emplace_back(far, e);
        }
        if (std::isfinite(max_step)) {                                  // limit_step_size + find_step (simplify.rs:69-102)
            size_t i = 1;
            while (i < end) {
                if (keep[i]) { ++i; continue; }
                const size_t start = i - 1;
                size_t next = end;
                for (size_t k = start + 1; k < n; ++k) {
                    if (std::fabs(y[k] - y[start]) > max_step) { next = std::max(k - 1, start + 1); break; }
                    if (keep[k]) { next = k; break; }
                }
                keep[next] = 1;
                i = std::max(next + 1, i + 1);
            }
        }
        uint32_t m = 0;
        for (uint32_t i = 0; i < n; ++i)
            if (keep[i]) out_indices[m++] = i;
        if (m == 2 && std::fabs(y[out_indices[0]] - y[out_indices[1]]) < epsilon) m = 1;     // simplify.rs:62-64
        *out_count = m;
        return FYX_OK;
    } catch (...) {
        return FYX_ERR_OOM;
    }
}

// BlendSpace::triangulate: the reference feeds the points, in order, to spade::DelaunayTriangulation (crate `spade`, absent here)
// and lists every inner face as the origins of its three edges.  What is reproduced: the Delaunay triangulation with insertion
// order deciding between co-circular alternatives, every triangle counter-clockwise and starting at its newest point, triangles
// listed by newest point -- which is what the reference's one fixture shows (blendspace.rs:455-484: the unit square ->
// [2, 0, 1], [3, 0, 2]); spade's listing order for larger inputs is pinned by nothing in the reference.
// Incremental insertion into a triangle list with edge -> triangle adjacency found by search (blend spaces have a handful of
// points); predicates in double on the f32 inputs.
int fyx_blend_space_triangulate(const float* points_xy, uint32_t n_points, uint32_t* out_triangles, uint32_t capacity, uint32_t* out_count) {
    if (!out_count || (n_points && !points_xy)) return FYX_ERR_INVALID_ARG;
    *out_count = 0;
    for (uint32_t i = 0; i < 2 * n_points; ++i)
        if (!std::isfinite(points_xy[i])) return FYX_ERR_INVALID_ARG;    // spade's insert fails: triangulate() returns false
    if (n_points < 3) return FYX_OK;
    try {
        struct P2 { double x, y; };
        std::vector<P2> P(n_points + 3);
        double lo[2] = {points_xy[0], points_xy[1]}, hi[2] = {points_xy[0], points_xy[1]};
        for (uint32_t i = 0; i < n_points; ++i) {
            P[i] = P2{points_xy[2 * i], points_xy[2 * i + 1]};
            lo[0] = std::min(lo[0], P[i].x); hi[0] = std::max(hi[0], P[i].x);
            lo[1] = std::min(lo[1], P[i].y); hi[1] = std::max(hi[1], P[i].y);
        }
        double ext = std::max(hi[0] - lo[0], hi[1] - lo[1]);
        if (!(ext > 0.0)) ext = 1.0;
        const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), R = 3.0e4 * ext;
        const uint32_t g0 = n_points, g1 = n_points + 1, g2 = n_points + 2;      // the far bounding triangle
        P[g0] = P2{cx - 1.8 * R, cy - R}; P[g1] = P2{cx + 1.8 * R, cy - R}; P[g2] = P2{cx, cy + 2.2 * R};
        struct Tri { uint32_t v[3]; bool dead; };
        std::vector<Tri> T;
        T.push_back(Tri{{g0, g1, g2}, false});
        const auto orient = [&](uint32_t a, uint32_t b, uint32_t c) { return (P[b].x - P[a].x) * (P[c].y - P[a].y) - (P[b].y - P[a].y) * (P[c].x - P[a].x); };
        const auto in_circle = [&](const Tri& t, uint32_t p) {
            const double ax = P[t.v[0]].x - P[p].x, ay = P[t.v[0]].y - P[p].y, bx = P[t.v[1]].x - P[p].x, by = P[t.v[1]].y - P[p].y;
            const double cx_ = P[t.v[2]].x - P[p].x, cy_ = P[t.v[2]].y - P[p].y;
            return (ax * ax + ay * ay) * (bx * cy_ - cx_ * by) - (bx * bx + by * by) * (ax * cy_ - cx_ * ay) + (cx_ * cx_ + cy_ * cy_) * (ax * by - bx * ay) > 0.0;
        };
        std::vector<std::pair<uint32_t, uint32_t>> rim;
        for (uint32_t p = 0; p < n_points; ++p) {
            bool seen = false;
            for (uint32_t q = 0; q < p && !seen; ++q) seen = points_xy[2 * q] == points_xy[2 * p] && points_xy[2 * q + 1] == points_xy[2 * p + 1];
            if (seen) continue;                                           // a repeated point adds no face
            rim.clear();
            for (Tri& t : T) {
                if (t.dead || !in_circle(t, p)) continue;
                t.dead = true;
                for (int k = 0; k < 3; ++k) rim.emplace_back(t.v[k], t.v[(k + 1) % 3]);
            }
            for (size_t i = 0; i < rim.size(); ++i) {
                bool inner = false;                                       // shared by two dying triangles: appears reversed as well
                for (size_t j = 0; j < rim.size() && !inner; ++j) inner = rim[j].first == rim[i].second && rim[j].second == rim[i].first;
                if (inner || !(orient(rim[i].first, rim[i].second, p) > 0.0)) continue;
                T.push_back(Tri{{p, rim[i].first, rim[i].second}, false});
            }
            T.erase(std::remove_if(T.begin(), T.end(), [](const Tri& t) { return t.dead; }), T.end());
        }
        std::vector<std::array<uint32_t, 3>> faces;
        for (const Tri& t : T) {
            if (t.v[0] >= n_points || t.v[1] >= n_points || t.v[2] >= n_points) continue;
            const int top = (t.v[0] > t.v[1] && t.v[0] > t.v[2]) ? 0 : (t.v[1] > t.v[2] ? 1 : 2);
            faces.push_back({t.v[top], t.v[(top + 1) % 3], t.v[(top + 2) % 3]});
        }
        std::sort(faces.begin(), faces.end());
        *out_count = (uint32_t)faces.size();
        if (out_triangles)
            for (size_t i = 0; i < faces.size() && i < capacity; ++i)
                for (int k = 0; k < 3; ++k) out_triangles[3 * i + k] = faces[i][k];
        return FYX_OK;
    } catch (...) {
        return FYX_ERR_OOM;
    }
}

}  // extern "C"
