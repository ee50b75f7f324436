// anim_api.hip -- C ABI of the pose path (see include/fyrox_hip.h, second half) and its host
// control plane.
//
// The control plane mirrors, per instance, the scalar logic of the reference:
//   Animation::tick / set_time_position / has_ended      fyrox-animation/src/lib.rs:432-496,736
//   Machine::evaluate_pose                                machine/mod.rs:344-382
//   MachineLayer::evaluate_pose                           machine/layer.rs:590-706
//   Transition::update / is_done, LogicNode               machine/transition.rs:141-173,301-322
//   PlayAnimation / BlendAnimations / ..ByIndex / BlendSpace::eval_pose
//                                                         machine/node/{play,blend,blendspace}.rs
//   StateAction::apply                                    machine/state.rs:48-80
// but instead of touching poses it RECORDS what each pose node would have blended (a "recipe")
// and flattens the consumed recipes into a fold program the pose_update kernel executes per
// bone.  No per-bone arithmetic happens here.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_set>
#include <vector>

#include "fyx_ctx.h"
#include "anim_leaves.h"

#include "anim_model.h"
#include "anim_planner.h"
#include "anim_device.h"

using namespace fyx;

// A one-launch frame reported a timed-out in-grid wait (check_device_error): everything in flight is waited for, then the latest frame of
// the animator -- of the whole scene, if that is how it was last updated -- runs again on the frame stream it ran on (palette pairs: the
// same buffer), as separate launches (the caller has switched anim.one_launch off), and is waited for.  The host control plane is not run
// again: clocks, programs and events are the planned frame's.  Pose records, hints and transforms take the same values twice.
int fyx::reissue_frame(fyx_ctx* c, uint64_t tag) {
    auto it = store(c).animators.find(tag);
    if (it == store(c).animators.end()) return FYX_OK;      // (freed since: nothing of it is left to be wrong)
    Animator& A = *it->second;
    if (A.last_frame_kind == 0) return FYX_OK;
    if (int rc = sync_all(c)) return rc;
    if (c->alt_stream) FYX_HIP(c, hipStreamSynchronize(c->alt_stream));
    if (c->pose_overlap) c->frame_idx ^= 1;      // enter_pose toggles it back: the frame's own stream and palette buffers
    SceneBatch& S = store(c).scene;
    const bool in_scene = A.last_frame_kind == 3 && S.members_gen == store(c).animators_gen &&
                          std::find(S.animators.begin(), S.animators.end(), &A) != S.animators.end();
    const int rc = in_scene ? scene_frame(c, S, 0.0f, true) : run_frame(c, A, A.last_frame_kind != 2);
    if (rc) return rc;
    if (int rc2 = sync_all(c)) return rc2;
    if (c->alt_stream) FYX_HIP(c, hipStreamSynchronize(c->alt_stream));
    return FYX_OK;
}

extern "C" {

int fyx_init_control_only(fyx_ctx** out_ctx) {
    if (!out_ctx) return FYX_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    FYX_GUARD_BEGIN_NOCTX
    fyx_ctx* c = new fyx_ctx();
    c->device = -1;
    *out_ctx = c;
    return FYX_OK;
    FYX_GUARD_END(nullptr)
}

// ---- tracks data ---------------------------------------------------------------------------

int fyx_tracks_data_upload(fyx_ctx* c, uint64_t tracks_id, uint32_t n_tracks, const fyx_track_desc* tracks,
                           uint32_t n_keys, const float* key_location, const float* key_value,
                           const uint8_t* key_kind, const float* key_left_tangent,
                           const float* key_right_tangent) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (n_tracks && !tracks) return fail(c, FYX_ERR_INVALID_ARG, "tracks is null");
    if (n_keys && (!key_location || !key_value || !key_kind))
        return fail(c, FYX_ERR_INVALID_ARG, "key arrays are null");
    uint64_t total = 0;
    for (uint32_t t = 0; t < n_tracks; ++t) {
        const fyx_track_desc& d = tracks[t];
        if (d.binding >= FYX_BIND_PROPERTY0) {
            // Property{name, value_type}: the id stands for the name; every TrackValueKind
            if (d.kind < FYX_KIND_REAL || d.kind > FYX_KIND_QUAT)
                return fail(c, FYX_ERR_INVALID_ARG, "track %u: value kind %d", t, d.kind);
            if (d.n_curves > 4) return fail(c, FYX_ERR_INVALID_ARG, "track %u has %u curves", t, d.n_curves);
            for (uint32_t k = 0; k < d.n_curves; ++k) total += d.curve_n_keys[k];
            continue;
        }
        if (d.binding != FYX_BIND_POSITION && d.binding != FYX_BIND_SCALE && d.binding != FYX_BIND_ROTATION)
            return fail(c, FYX_ERR_INVALID_ARG, "track %u: binding %d", t, d.binding);
        const bool vec3 = d.kind == FYX_KIND_VEC3;
        const bool quat = d.kind == FYX_KIND_QUAT || d.kind == FYX_KIND_QUAT_EULER;
        // (a kind that its binding cannot take -- a Real track bound to Position -- is accepted as the reference accepts it: the value
        // sits in its node's list, blends with nothing and is never applied: scene/animation/mod.rs:147-186, value.rs:221-230)
        (void)vec3; (void)quat;
        if (d.kind < FYX_KIND_REAL || d.kind > FYX_KIND_QUAT) return fail(c, FYX_ERR_INVALID_ARG, "track %u: value kind %d", t, d.kind);
        if (d.n_curves > 4) return fail(c, FYX_ERR_INVALID_ARG, "track %u has %u curves", t, d.n_curves);
        for (uint32_t k = 0; k < d.n_curves; ++k) total += d.curve_n_keys[k];
    }
    if (total != n_keys)
        return fail(c, FYX_ERR_INVALID_ARG, "tracks describe %llu keys but n_keys = %u", (unsigned long long)total, n_keys);
    for (uint32_t k = 0; k < n_keys; ++k)
        if (key_kind[k] > FYX_KEY_CUBIC) return fail(c, FYX_ERR_INVALID_ARG, "key %u has kind %u", k, key_kind[k]);
    // Curve keeps its keys sorted by location (Curve::from / add_key sort them, fyrox-math/src/curve.rs:176-200, 215-236)
    // and value_at's partition_point, span hints and end clamps rely on it: keys arrive here as the Curve holds them.
    // Anything else (unsorted, NaN / infinite locations) would sample silently different values: refused.
    {
        uint32_t key = 0;
        for (uint32_t t = 0; t < n_tracks; ++t)
            for (uint32_t k = 0; k < tracks[t].n_curves; ++k) {
                const uint32_t nk = tracks[t].curve_n_keys[k];
                for (uint32_t i = 0; i < nk; ++i) {
                    const float loc = key_location[key + i];
                    if (!(loc - loc == 0.0f))
                        return fail(c, FYX_ERR_INVALID_ARG, "track %u curve %u: key %u has a non-finite location", t, k, i);
                    if (i && loc < key_location[key + i - 1])
                        return fail(c, FYX_ERR_INVALID_ARG, "track %u curve %u: key locations are not sorted (key %u)", t, k, i);
                }
                key += nk;
            }
    }
    TracksData td;
    td.n_tracks = n_tracks;
    td.tracks.assign(tracks, tracks + n_tracks);
    if (has_device(c)) {
        if (int rc = enter_primary(c)) return rc;
        std::vector<TrackDev> hd(n_tracks);
        uint32_t key = 0;
        for (uint32_t t = 0; t < n_tracks; ++t) {
            hd[t].kind = tracks[t].kind;
            hd[t].n_curves = tracks[t].n_curves;
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t nk = k < tracks[t].n_curves ? tracks[t].curve_n_keys[k] : 0;
                hd[t].first_key[k] = key;
                hd[t].n_keys[k] = nk;
                hd[t].first_loc[k] = nk ? key_location[key] : 0.f;
                hd[t].last_loc[k] = nk ? key_location[key + nk - 1] : 0.f;
                hd[t].first_val[k] = nk ? key_value[key] : 0.f;
                hd[t].last_val[k] = nk ? key_value[key + nk - 1] : 0.f;
                key += nk;
            }
        }
        std::vector<float4> aux(n_keys);
        for (uint32_t k = 0; k < n_keys; ++k) {
            const uint32_t kind = key_kind[k];
            float kb;
            memcpy(&kb, &kind, 4);
            const bool cubic = kind == FYX_KEY_CUBIC;
            aux[k] = make_float4(key_value[k], kb, cubic && key_left_tangent ? key_left_tangent[k] : 0.f,
                                 cubic && key_right_tangent ? key_right_tangent[k] : 0.f);
        }
        std::vector<KeyRec> recs(n_keys);
        for (uint32_t k = 0; k < n_keys; ++k) {
            recs[k].aux = aux[k];
            recs[k].loc = key_location[k];
            recs[k].pad[0] = recs[k].pad[1] = recs[k].pad[2] = 0.f;
        }
        // span records (TrackHot / AnimDev::spans): tracks whose first `need` curves share their key times
        std::vector<TrackHot> hot(n_tracks);
        std::vector<float4> spans;
        for (uint32_t t = 0; t < n_tracks; ++t) {
            const int kind = tracks[t].kind;
            const uint32_t need = kind == FYX_KIND_QUAT ? 4u : (kind == FYX_KIND_VEC3 || kind == FYX_KIND_QUAT_EULER) ? 3u : 0u;
            hot[t] = TrackHot{kind, tracks[t].n_curves, 0u, kNoSpans};
            if (need == 0 || tracks[t].n_curves < need || tracks[t].binding >= FYX_BIND_PROPERTY0) continue;
            const uint32_t nk = hd[t].n_keys[0];
            bool same = nk >= 2;
            for (uint32_t k = 1; k < need && same; ++k)
                same = hd[t].n_keys[k] == nk &&
                       memcmp(key_location + hd[t].first_key[k], key_location + hd[t].first_key[0], (size_t)nk * 4) == 0;
            if (!same) continue;
            const uint32_t stride = span_stride(need);           // f4 per record: 64 bytes (three curves) / 80 (four)
            if (spans.size() + (size_t)(nk - 1) * stride > 0x7fffffffull) continue;
            hot[t].n_keys = nk;
            hot[t].span_first = (uint32_t)spans.size();
            spans.resize(spans.size() + (size_t)(nk - 1) * stride, make_float4(0.f, 0.f, 0.f, 0.f));
            for (uint32_t i = 1; i < nk; ++i) {
                float4* r = spans.data() + hot[t].span_first + (size_t)(i - 1) * stride;
                // what CurveKey::interpolate reads of keys i - 1 and i (curve.rs:87-132; aux = {value, kind bits, left tangent, right tangent}):
                // the two values, the left key's kind and right tangent, the right key's left tangent when THAT key is cubic (else 0)
                uint32_t kinds = 0;
                for (uint32_t k = 0; k < need; ++k) {
                    const float4 la = aux[hd[t].first_key[k] + i - 1], ra = aux[hd[t].first_key[k] + i];
                    uint32_t lk = 0, rk = 0;
                    memcpy(&lk, &la.y, 4);
                    memcpy(&rk, &ra.y, 4);
                    kinds |= (lk & 0xffu) << (8u * k);
                    r[1 + k] = make_float4(la.x, ra.x, la.w, rk == (uint32_t)FYX_KEY_CUBIC ? ra.z : 0.0f);
                }
                float kinds_f = 0.f;
                memcpy(&kinds_f, &kinds, 4);
                r[0] = make_float4(key_location[hd[t].first_key[0] + i - 1], key_location[hd[t].first_key[0] + i], kinds_f, 0.f);
            }
        }
        // ... and once more as rows (TracksData::d_span_rows): the tracks whose key times are the first span-record track's
        std::vector<float4> rows;
        td.row_first.assign(n_tracks, kNoSpans);
        {
            int32_t lead = -1;
            uint32_t stride_row = 0;
            std::vector<uint32_t> members;
            for (uint32_t t = 0; t < n_tracks; ++t) {
                if (hot[t].span_first == kNoSpans) continue;
                if (lead < 0) lead = (int32_t)t;
                if (hot[t].n_keys != hot[lead].n_keys ||
                    memcmp(key_location + hd[t].first_key[0], key_location + hd[lead].first_key[0], (size_t)hot[t].n_keys * 4) != 0) continue;
                td.row_first[t] = stride_row;
                stride_row += span_stride(hot[t].kind == FYX_KIND_QUAT ? 4u : 3u);
                members.push_back(t);
            }
            const uint64_t total = lead >= 0 ? (uint64_t)(hot[lead].n_keys - 1u) * stride_row : 0;
            if (members.size() >= 2 && total <= 0x7fffffffull && stride_row < (1u << 23)) {
                rows.resize((size_t)total);
                for (uint32_t t : members) {
                    const uint32_t st = span_stride(hot[t].kind == FYX_KIND_QUAT ? 4u : 3u);
                    for (uint32_t i = 0; i + 1 < hot[t].n_keys; ++i)
                        memcpy(rows.data() + (size_t)i * stride_row + td.row_first[t], spans.data() + hot[t].span_first + (size_t)i * st, (size_t)st * 16);
                }
                td.row_stride = stride_row;
            } else {
                td.row_first.assign(n_tracks, kNoSpans);
            }
        }
        td.hot = hot;
        int rc = upload(c, &td.d_tracks, hd.data(), hd.size());
        if (!rc) rc = upload(c, &td.d_hot, hot.data(), hot.size());
        if (!rc && !spans.empty()) rc = upload(c, &td.d_spans, spans.data(), spans.size());
        if (!rc && !rows.empty()) rc = upload(c, &td.d_span_rows, rows.data(), rows.size());
        if (!rc) rc = upload(c, &td.d_loc, key_location, (size_t)n_keys);
        if (!rc) rc = upload(c, &td.d_aux, aux.data(), aux.size());
        if (!rc) rc = upload(c, &td.d_rec, recs.data(), recs.size());
        if (rc) { free_tracks(td); return rc; }
    }
    auto& m = store(c).tracks;
    auto it = m.find(tracks_id);
    if (it != m.end()) {
        for (auto& kv : store(c).animators)
            for (auto& an : kv.second->anims)
                if (an.td == &it->second) {
                    free_tracks(td);
                    return fail(c, FYX_ERR_INVALID_ARG, "tracks data %llu is in use by an animator", (unsigned long long)tracks_id);
                }
        if (has_device(c)) (void)hipStreamSynchronize(c->stream);
        free_tracks(it->second);
        m.erase(it);
    }
    m.emplace(tracks_id, std::move(td));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_tracks_data_free(fyx_ctx* c, uint64_t tracks_id) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    auto& m = store(c).tracks;
    auto it = m.find(tracks_id);
    if (it == m.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "tracks data %llu is not registered", (unsigned long long)tracks_id);
    for (auto& kv : store(c).animators)
        for (auto& an : kv.second->anims)
            if (an.td == &it->second)
                return fail(c, FYX_ERR_INVALID_ARG, "tracks data %llu is in use by an animator", (unsigned long long)tracks_id);
    if (has_device(c)) { if (int rc = enter_primary(c)) return rc; FYX_HIP(c, hipStreamSynchronize(c->stream)); }
    free_tracks(it->second);
    m.erase(it);
    return FYX_OK;
    FYX_GUARD_END(c)
}

// ---- rigs / bone lists ---------------------------------------------------------------------

int fyx_rig_create(fyx_ctx* c, uint64_t rig_id, uint32_t n_nodes, const int32_t* parent,
                   const fyx_transform* transforms, const float* inv_bind) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (n_nodes == 0 || n_nodes > (uint32_t)kMaxRigNodes)
        return fail(c, n_nodes ? FYX_ERR_UNSUPPORTED : FYX_ERR_INVALID_ARG, "n_nodes=%u outside 1..%d", n_nodes, kMaxRigNodes);
    if (!parent || !transforms) return fail(c, FYX_ERR_INVALID_ARG, "parent / transforms are null");
    if (store(c).rigs.count(rig_id)) return fail(c, FYX_ERR_INVALID_ARG, "rig %llu already exists", (unsigned long long)rig_id);
    Rig r;
    r.n_nodes = n_nodes;
    r.parent.assign(parent, parent + n_nodes);
    std::vector<uint32_t> depth(n_nodes, 0);
    uint32_t max_depth = 0;
    for (uint32_t i = 0; i < n_nodes; ++i) {
        if (parent[i] >= (int32_t)i) return fail(c, FYX_ERR_INVALID_ARG, "parent[%u] = %d is not an earlier node", i, parent[i]);
        depth[i] = parent[i] < 0 ? 0 : depth[parent[i]] + 1;
        max_depth = std::max(max_depth, depth[i]);
    }
    r.n_levels = max_depth + 1;
    std::vector<uint32_t> level_start(r.n_levels + 1, 0), level_nodes(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) ++level_start[depth[i] + 1];
    for (uint32_t l = 0; l < r.n_levels; ++l) level_start[l + 1] += level_start[l];
    {
        std::vector<uint32_t> cur(level_start.begin(), level_start.end() - 1);
        for (uint32_t i = 0; i < n_nodes; ++i) level_nodes[cur[depth[i]]++] = i;
    }
    static_assert(kMaxRigNodes <= 1024, "RigDev::walk packs node (10 bits), parent + 1 (11) and depth (11) into one word");
    // (ANY negative parent is a root, as everywhere else in the host code: only -1 may be incremented into the 11-bit field)
    for (uint32_t& e : level_nodes) e = e | (parent[e] < 0 ? 0u : (uint32_t)parent[e] + 1u) << 10 | depth[e] << 21;
    r.walk = level_nodes;
    // The same order in CHUNKS of sixteen for the wide walk (pose_update_body<.., WIDE>: a lane per matrix element, sixteen nodes
    // at a time): a level is padded to whole chunks, so that no lane of the walk ever tests whether it has a node.  Entry:
    // node | parent slot << 11 | (last chunk of its level) << 22, where slot n_nodes holds the identity (what a root is multiplied
    // by) and slot n_nodes + 1 is nobody's (the padding's node: its product goes there).
    static_assert(kMaxRigNodes + 2 <= 2048, "wide-walk entries pack node and parent slot into 11 bits each");
    r.chunks.clear();
    for (uint32_t l = 0; l < r.n_levels; ++l) {
        const uint32_t b = level_start[l], e = level_start[l + 1];
        for (uint32_t c0 = b; c0 < e; c0 += 16) {
            const uint32_t last = c0 + 16 >= e ? 1u << 22 : 0u;
            for (uint32_t g = 0; g < 16; ++g) {
                uint32_t node = n_nodes + 1, slot = n_nodes;
                if (c0 + g < e) {
                    node = r.walk[c0 + g] & 1023u;
                    slot = parent[node] < 0 ? n_nodes : (uint32_t)parent[node];
                }
                r.chunks.push_back(node | slot << 11 | last);
            }
        }
    }
    r.n_chunks = (uint32_t)(r.chunks.size() / 16);
    r.init_trs.assign((size_t)n_nodes * 12, 0.f);
    std::vector<float> statics((size_t)n_nodes * 28, 0.f);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        const fyx_transform& t = transforms[i];
        float* d = &r.init_trs[(size_t)i * 12];
        memcpy(d, t.local_position, 12);
        memcpy(d + 4, t.local_rotation, 16);
        memcpy(d + 8, t.local_scale, 12);
        float* s = &statics[(size_t)i * 28];
        memcpy(s, t.pre_rotation, 16);
        memcpy(s + 4, t.post_rotation_matrix, 36);
        memcpy(s + 13, t.rotation_offset, 12);
        memcpy(s + 16, t.rotation_pivot, 12);
        memcpy(s + 19, t.scaling_offset, 12);
        memcpy(s + 22, t.scaling_pivot, 12);
    }
    if (has_device(c)) {
        if (int rc = enter_primary(c)) return rc;
        std::vector<float> ib((size_t)n_nodes * 16, 0.f);
        if (inv_bind) {
            memcpy(ib.data(), inv_bind, ib.size() * 4);
        } else {
            for (uint32_t i = 0; i < n_nodes; ++i) ib[(size_t)i * 16] = ib[(size_t)i * 16 + 5] = ib[(size_t)i * 16 + 10] = ib[(size_t)i * 16 + 15] = 1.f;
        }
        int rc = upload(c, &r.d_statics, statics.data(), statics.size());
        if (!rc) {      // [n_nodes] walk words, then [n_chunks][16] chunk entries
            std::vector<uint32_t> both(level_nodes);
            both.insert(both.end(), r.chunks.begin(), r.chunks.end());
            rc = upload(c, &r.d_walk, both.data(), both.size());
        }
        if (!rc) rc = upload(c, &r.d_inv_bind, ib.data(), ib.size());
        if (rc) { free_rig(r); return rc; }
    }
    store(c).rigs.emplace(rig_id, std::move(r));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_rig_free(fyx_ctx* c, uint64_t rig_id) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    auto& m = store(c).rigs;
    auto it = m.find(rig_id);
    if (it == m.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "rig %llu is not registered", (unsigned long long)rig_id);
    for (auto& kv : store(c).animators)
        if (kv.second->rig == &it->second)
            return fail(c, FYX_ERR_INVALID_ARG, "rig %llu is in use by an animator", (unsigned long long)rig_id);
    if (has_device(c)) { if (int rc = enter_primary(c)) return rc; FYX_HIP(c, hipStreamSynchronize(c->stream)); }
    free_rig(it->second);
    m.erase(it);
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_bone_list_create(fyx_ctx* c, uint64_t bones_id, uint64_t rig_id, uint32_t n_bones, const int32_t* bone_nodes) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    auto rit = store(c).rigs.find(rig_id);
    if (rit == store(c).rigs.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "rig %llu is not registered", (unsigned long long)rig_id);
    if (n_bones == 0 || n_bones > 256 || !bone_nodes)
        return fail(c, FYX_ERR_INVALID_ARG, "n_bones=%u outside 1..256 (bone indices are u8)", n_bones);
    for (uint32_t b = 0; b < n_bones; ++b)
        if (bone_nodes[b] >= (int32_t)rit->second.n_nodes)
            return fail(c, FYX_ERR_INVALID_ARG, "bone %u refers to node %d of a %u-node rig", b, bone_nodes[b], rit->second.n_nodes);
    if (store(c).bones.count(bones_id)) return fail(c, FYX_ERR_INVALID_ARG, "bone list %llu already exists", (unsigned long long)bones_id);
    BoneList bl;
    bl.rig_id = rig_id;
    bl.n_bones = n_bones;
    if (has_device(c)) {
        if (int rc = enter_primary(c)) return rc;
        if (int rc = upload(c, &bl.d_bone_nodes, bone_nodes, (size_t)n_bones)) return rc;
    }
    store(c).bones.emplace(bones_id, bl);
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_bone_list_free(fyx_ctx* c, uint64_t bones_id) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    auto& m = store(c).bones;
    auto it = m.find(bones_id);
    if (it == m.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "bone list %llu is not registered", (unsigned long long)bones_id);
    for (auto& kv : store(c).animators)
        for (const Animator::PaletteOut& po : kv.second->palette_outputs)
            if (po.bones_id == bones_id)
                return fail(c, FYX_ERR_INVALID_ARG, "bone list %llu is a palette output of an animator", (unsigned long long)bones_id);
    if (has_device(c)) { if (int rc = enter_primary(c)) return rc; FYX_HIP(c, hipStreamSynchronize(c->stream)); }
    free_bones(it->second);
    m.erase(it);
    return FYX_OK;
    FYX_GUARD_END(c)
}

// ---- animators -----------------------------------------------------------------------------

int fyx_animator_create(fyx_ctx* c, uint64_t animator_id, uint64_t rig_id, uint32_t n_instances) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    auto rit = store(c).rigs.find(rig_id);
    if (rit == store(c).rigs.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "rig %llu is not registered", (unsigned long long)rig_id);
    if (n_instances == 0) return fail(c, FYX_ERR_INVALID_ARG, "n_instances is 0");
    if (n_instances > 65535u) return fail(c, FYX_ERR_UNSUPPORTED, "n_instances=%u: at most 65535 instances per animator (split the crowd)", n_instances);
    if (store(c).animators.count(animator_id))
        return fail(c, FYX_ERR_INVALID_ARG, "animator %llu already exists", (unsigned long long)animator_id);
    std::unique_ptr<Animator> a(new Animator());
    a->id = animator_id;
    a->rig_id = rig_id;
    a->rig = &rit->second;
    a->n_instances = n_instances;
    store(c).animators.emplace(animator_id, std::move(a));
    ++store(c).animators_gen;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_free(fyx_ctx* c, uint64_t animator_id) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    auto& m = store(c).animators;
    auto it = m.find(animator_id);
    if (it == m.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "animator %llu is not registered", (unsigned long long)animator_id);
    if (has_device(c)) { if (int rc = enter_primary(c)) return rc; FYX_HIP(c, hipStreamSynchronize(c->stream)); }
    free_animator(*it->second);
    m.erase(it);
    ++store(c).animators_gen;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_add_animation(fyx_ctx* c, uint64_t animator_id, uint64_t tracks_id, const int32_t* track_target,
                               const uint8_t* track_enabled, uint32_t* out_animation) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    auto tit = store(c).tracks.find(tracks_id);
    if (tit == store(c).tracks.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "tracks data %llu is not registered", (unsigned long long)tracks_id);
    const TracksData& td = tit->second;
    if (td.n_tracks && !track_target) return fail(c, FYX_ERR_INVALID_ARG, "track_target is null");
    AnimationDef an;
    an.tracks_id = tracks_id;
    an.td = &td;
    an.target.assign(td.n_tracks, -1);
    an.enabled.assign(td.n_tracks, 1);
    // Several tracks on one binding of one node, or a kind the binding cannot take: the node's pose is a list (pose.rs:107-121) and
    // such an animation keeps two views of it (AnimationDef::dup, built with the device state from the ENABLED tracks).
    std::vector<uint8_t> used((size_t)A->rig->n_nodes * 3, 0);
    std::vector<std::pair<int32_t, int32_t>> new_slots = A->prop_slots, seen_props;
    for (uint32_t t = 0; t < td.n_tracks; ++t) {
        an.target[t] = track_target[t];
        if (track_enabled) an.enabled[t] = track_enabled[t] ? 1 : 0;
        if (track_target[t] >= (int32_t)A->rig->n_nodes)
            return fail(c, FYX_ERR_INVALID_ARG, "track %u targets node %d of a %u-node rig", t, track_target[t], A->rig->n_nodes);
        if (track_target[t] >= 0 && td.tracks[t].binding >= FYX_BIND_PROPERTY0) {
            const std::pair<int32_t, int32_t> key(track_target[t], td.tracks[t].binding - FYX_BIND_PROPERTY0);
            if (std::find(seen_props.begin(), seen_props.end(), key) != seen_props.end()) an.maybe_dup = true;
            seen_props.push_back(key);
            if (std::find(new_slots.begin(), new_slots.end(), key) == new_slots.end()) new_slots.push_back(key);
        } else if (track_target[t] >= 0) {
            uint8_t& u = used[(size_t)track_target[t] * 3 + td.tracks[t].binding];
            if (u || !kind_fits(td.tracks[t].binding, td.tracks[t].kind)) an.maybe_dup = true;
            u = 1;
        }
    }
    if (new_slots.size() > 65535) return fail(c, FYX_ERR_UNSUPPORTED, "more than 65535 animated properties");
    if (new_slots.size() != A->prop_slots.size()) {
        A->prop_slots.swap(new_slots);
        for (AnimationDef& o : A->anims) o.slots_dirty = true;   // their slot tables grow
    }
    for (uint32_t t = 0; t < td.n_tracks; ++t) {  // lib.rs:507-534: the first track with the binding
        if (an.rm_pos_track < 0 && td.tracks[t].binding == FYX_BIND_POSITION) an.rm_pos_track = (int32_t)t;
        if (an.rm_rot_track < 0 && td.tracks[t].binding == FYX_BIND_ROTATION) an.rm_rot_track = (int32_t)t;
    }
    const uint32_t na = (uint32_t)A->anims.size();
    // re-layout [inst][anim] state for the new animation count
    std::vector<AnimState> ns((size_t)A->n_instances * (na + 1));
    for (uint32_t i = 0; i < A->n_instances; ++i)
        for (uint32_t a = 0; a < na; ++a) ns[(size_t)i * (na + 1) + a] = A->anim_state[(size_t)i * na + a];
    A->anim_state.swap(ns);
    A->anims.push_back(std::move(an));
    A->max_tracks = std::max(A->max_tracks, td.n_tracks);
    A->anims_dirty = true;
    if (out_animation) *out_animation = na;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animation_set_track_enabled(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t track, int enabled) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (animation >= A->anims.size() || A->anims[animation].removed) return fail(c, FYX_ERR_INVALID_ARG, "animation %u does not exist", animation);
    AnimationDef& an = A->anims[animation];
    if (track >= an.enabled.size()) return fail(c, FYX_ERR_INVALID_ARG, "track %u does not exist", track);
    an.enabled[track] = enabled ? 1 : 0;
    an.slots_dirty = true;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animation_set_time_slice(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance, float start, float end) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (!(start <= end)) return fail(c, FYX_ERR_INVALID_ARG, "time slice start > end (the reference asserts)");
    return for_instances(c, A, animation, instance, [&](AnimState& s) { s.start = start; s.end = end; set_time_position(s, s.time); });
    FYX_GUARD_END(c)
}
int fyx_animation_set_time_position(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance, float time) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    return for_instances(c, A, animation, instance, [&](AnimState& s) { set_time_position(s, time); });
    FYX_GUARD_END(c)
}
int fyx_animation_set_speed(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance, float speed) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    return for_instances(c, A, animation, instance, [&](AnimState& s) { s.speed = speed; });
    FYX_GUARD_END(c)
}
int fyx_animation_set_loop(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance, int looped) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    return for_instances(c, A, animation, instance, [&](AnimState& s) { s.looped = looped ? 1 : 0; });
    FYX_GUARD_END(c)
}
int fyx_animation_set_enabled(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance, int enabled) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    return for_instances(c, A, animation, instance, [&](AnimState& s) { s.enabled = enabled ? 1 : 0; });
    FYX_GUARD_END(c)
}
int fyx_animation_rewind(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    return for_instances(c, A, animation, instance, [&](AnimState& s) { set_time_position(s, s.start); });
    FYX_GUARD_END(c)
}
int fyx_animation_get_state(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance,
                            float* time_position, int* enabled, int* ended) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    return for_instances(c, A, animation, instance, [&](AnimState& s) {
        if (time_position) *time_position = s.time;
        if (enabled) *enabled = s.enabled;
        if (ended) *ended = has_ended(s) ? 1 : 0;
    });
    FYX_GUARD_END(c)
}

// ---- machine builder -----------------------------------------------------------------------

int fyx_machine_add_parameter(fyx_ctx* c, uint64_t animator_id, int kind, float f0, float f1, uint32_t u, uint32_t* out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (kind < FYX_PARAM_WEIGHT || kind > FYX_PARAM_SAMPLING_POINT) return fail(c, FYX_ERR_INVALID_ARG, "parameter kind %d", kind);
    Param p;
    p.kind = kind; p.f0 = f0; p.f1 = f1; p.u = u;
    A->param_defaults.push_back(p);
    sync_machine_state(*A);
    if (out) *out = (uint32_t)A->param_defaults.size() - 1;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_machine_set_parameter(fyx_ctx* c, uint64_t animator_id, uint32_t parameter, uint32_t instance, int kind,
                              float f0, float f1, uint32_t u) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    // games set parameters every frame, mostly to the value they already have: only an instance whose parameter really
    // changes loses its memoised fold program (MachineState), not the whole animator (hence the _RO form)
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (parameter >= A->param_defaults.size()) return fail(c, FYX_ERR_INVALID_ARG, "parameter %u does not exist", parameter);
    if (kind < FYX_PARAM_WEIGHT || kind > FYX_PARAM_SAMPLING_POINT) return fail(c, FYX_ERR_INVALID_ARG, "parameter kind %d", kind);
    Param p;
    p.kind = kind; p.f0 = f0; p.f1 = f1; p.u = u;
    auto assign = [&](MachineState& m) {
        Param& q = m.params[parameter];
        if (q.kind != p.kind || memcmp(&q.f0, &p.f0, 4) || memcmp(&q.f1, &p.f1, 4) || q.u != p.u) { m.memo_valid = false; A->steady_gen = 0; }
        q = p;
    };
    if (instance == FYX_ALL_INSTANCES) {
        A->param_defaults[parameter] = p;
        for (MachineState& m : A->mstate) assign(m);
        return FYX_OK;
    }
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    ensure_machine_state(*A);
    assign(A->mstate[instance]);
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_machine_add_layer(fyx_ctx* c, uint64_t animator_id, float weight, uint32_t* out_layer) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (A->layers.size() >= 255) return fail(c, FYX_ERR_UNSUPPORTED, "too many layers");
    LayerDef L;
    L.weight = weight;
    A->layers.push_back(std::move(L));
    sync_machine_state(*A);
    A->masks_dirty = true;
    if (out_layer) *out_layer = (uint32_t)A->layers.size() - 1;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_set_weight(fyx_ctx* c, uint64_t animator_id, uint32_t layer, float weight) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    L->weight = weight;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_set_mask(fyx_ctx* c, uint64_t animator_id, uint32_t layer, const int32_t* excluded, uint32_t n) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (n && !excluded) return fail(c, FYX_ERR_INVALID_ARG, "excluded_nodes is null");
    L->excluded.assign(excluded, excluded + n);
    A->masks_dirty = true;
    return FYX_OK;
    FYX_GUARD_END(c)
}

static int add_node(fyx_ctx* c, Animator* A, LayerDef* L, PoseNodeDef&& n, uint32_t* out_node) {
    for (const BlendInput& in : n.inputs)
        if (in.source >= (int32_t)L->nodes.size() + 1)
            return fail(c, FYX_ERR_INVALID_ARG, "pose source %d does not exist", in.source);
    L->nodes.push_back(std::move(n));
    std::vector<int> st(L->nodes.size(), -3);
    for (size_t h = 0; h < L->nodes.size(); ++h) {
        const int d = node_depth(*L, (int32_t)h, st);
        if (d < 0) { L->nodes.pop_back(); return fail(c, FYX_ERR_INVALID_ARG, "pose nodes form a cycle"); }
        if (d + 1 > kMaxFoldDepth - 1) {
            L->nodes.pop_back();
            return fail(c, FYX_ERR_UNSUPPORTED, "pose nodes nest deeper than %d blend levels", kMaxFoldDepth - 2);
        }
    }
    (void)A;
    if (out_node) *out_node = (uint32_t)L->nodes.size() - 1;
    return FYX_OK;
}

int fyx_layer_add_play_animation(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t animation, uint32_t* out_node) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (animation >= A->anims.size() || A->anims[animation].removed)
        return fail(c, FYX_ERR_INVALID_ARG, "animation %u does not exist (an invalid handle would leave a stale pose in the reference)", animation);
    if (animation >= (1u << 24)) return fail(c, FYX_ERR_UNSUPPORTED, "too many animations");
    PoseNodeDef n;
    n.type = NODE_PLAY;
    n.animation = animation;
    return add_node(c, A, L, std::move(n), out_node);
    FYX_GUARD_END(c)
}

int fyx_layer_add_blend_animations(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t n_inputs,
                                   const int32_t* pose_sources, const int32_t* weight_parameters,
                                   const float* weight_constants, uint32_t* out_node) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (n_inputs && !pose_sources) return fail(c, FYX_ERR_INVALID_ARG, "pose_sources is null");
    PoseNodeDef n;
    n.type = NODE_BLEND;
    n.inputs.resize(n_inputs);
    for (uint32_t i = 0; i < n_inputs; ++i) {
        n.inputs[i].source = pose_sources[i];
        n.inputs[i].weight_param = weight_parameters ? weight_parameters[i] : -1;
        n.inputs[i].weight_const = weight_constants ? weight_constants[i] : 0.f;
    }
    return add_node(c, A, L, std::move(n), out_node);
    FYX_GUARD_END(c)
}

int fyx_layer_add_blend_animations_by_index(fyx_ctx* c, uint64_t animator_id, uint32_t layer, int32_t index_parameter,
                                            uint32_t n_inputs, const int32_t* pose_sources,
                                            const float* blend_times, uint32_t* out_node) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (n_inputs && (!pose_sources || !blend_times)) return fail(c, FYX_ERR_INVALID_ARG, "inputs are null");
    PoseNodeDef n;
    n.type = NODE_BY_INDEX;
    n.param = index_parameter;
    n.inputs.resize(n_inputs);
    for (uint32_t i = 0; i < n_inputs; ++i) {
        n.inputs[i].source = pose_sources[i];
        n.inputs[i].blend_time = blend_times[i];
    }
    n.by_index_slot = L->by_index_count;
    int rc = add_node(c, A, L, std::move(n), out_node);
    if (rc) return rc;
    ++L->by_index_count;
    sync_machine_state(*A);
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_add_blend_space(fyx_ctx* c, uint64_t animator_id, uint32_t layer, int32_t sampling_parameter,
                              uint32_t n_points, const float* points_xy, const int32_t* pose_sources,
                              uint32_t n_triangles, const uint32_t* triangles, uint32_t* out_node) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (n_points && (!points_xy || !pose_sources)) return fail(c, FYX_ERR_INVALID_ARG, "points are null");
    if (n_triangles && !triangles) return fail(c, FYX_ERR_INVALID_ARG, "triangles is null");
    for (uint32_t i = 0; i < n_triangles * 3; ++i)
        if (triangles[i] >= n_points) return fail(c, FYX_ERR_INVALID_ARG, "triangle refers to point %u of %u", triangles[i], n_points);
    PoseNodeDef n;
    n.type = NODE_BLEND_SPACE;
    n.param = sampling_parameter;
    n.inputs.resize(n_points);
    for (uint32_t i = 0; i < n_points; ++i) n.inputs[i].source = pose_sources[i];
    n.points.assign(points_xy, points_xy + (size_t)n_points * 2);
    n.triangles.assign(triangles, triangles + (size_t)n_triangles * 3);
    return add_node(c, A, L, std::move(n), out_node);
    FYX_GUARD_END(c)
}

int fyx_layer_add_state(fyx_ctx* c, uint64_t animator_id, uint32_t layer, int32_t root_node, uint32_t* out_state) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    StateDef s;
    s.root = root_node;
    L->states.push_back(std::move(s));
    // layer.rs:229-235: `if self.active_state.is_none() { self.active_state = state }` -- the entry state is NOT touched, and
    // the test is on active_state alone: a state added while a transition is in flight (active_state is NONE then) becomes
    // the active one, as in the reference
    const int32_t idx = (int32_t)L->states.size() - 1;
    if (L->initial_active < 0) L->initial_active = idx;
    for (MachineState& m : A->mstate)
        if (layer < m.layers.size() && m.layers[layer].active_state < 0) { m.layers[layer].active_state = idx; m.memo_valid = false; A->steady_gen = 0; }
    if (out_state) *out_state = (uint32_t)idx;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_set_entry_state(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t state) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (state >= L->states.size()) return fail(c, FYX_ERR_INVALID_ARG, "state %u does not exist", state);
    L->entry_state = L->initial_active = (int32_t)state;
    for (MachineState& m : A->mstate) m.layers[layer].active_state = (int32_t)state;  // layer.rs:209-212
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_remove_animation(fyx_ctx* c, uint64_t animator_id, uint32_t animation) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (animation >= A->anims.size() || A->anims[animation].removed) return fail(c, FYX_ERR_INVALID_ARG, "animation %u does not exist", animation);
    A->anims[animation].removed = true;
    const uint32_t na = (uint32_t)A->anims.size();
    for (uint32_t i = 0; i < A->n_instances; ++i) {
        AnimState& st = A->anim_state[(size_t)i * na + animation];
        st.enabled = 0;          // nothing ticks it any more
        st.events.clear();
    }
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_state_add_action(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t state, int on_enter, int action, uint32_t animation) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (state >= L->states.size()) return fail(c, FYX_ERR_INVALID_ARG, "state %u does not exist", state);
    if (action < FYX_ACTION_NONE || action > FYX_ACTION_DISABLE_ANIMATION)
        return fail(c, FYX_ERR_INVALID_ARG, "state action %d (EnableRandomAnimation has its own call)", action);
    (on_enter ? L->states[state].on_enter : L->states[state].on_leave).push_back(Action{action, animation, {}});
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_state_add_random_action(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t state, int on_enter,
                                const uint32_t* animations, uint32_t n_animations) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (state >= L->states.size()) return fail(c, FYX_ERR_INVALID_ARG, "state %u does not exist", state);
    if (n_animations && !animations) return fail(c, FYX_ERR_INVALID_ARG, "animations is null");
    Action a{FYX_ACTION_ENABLE_RANDOM_ANIMATION, 0, {}};
    a.choices.assign(animations, animations + n_animations);
    (on_enter ? L->states[state].on_enter : L->states[state].on_leave).push_back(std::move(a));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_set_random_seed(fyx_ctx* c, uint64_t animator_id, uint32_t instance, uint64_t seed) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    ensure_rng(*A);
    if (instance == FYX_ALL_INSTANCES) {
        for (uint32_t i = 0; i < A->n_instances; ++i) A->rng[i] = seed + kGolden * (uint64_t)(i + 1);
        return FYX_OK;
    }
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    A->rng[instance] = seed;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_add_transition(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t source, uint32_t dest,
                             float transition_time, const int32_t* condition, uint32_t n_condition, uint32_t* out_transition) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (n_condition && !condition) return fail(c, FYX_ERR_INVALID_ARG, "condition is null");
    TransitionDef t;
    t.source = source;
    t.dest = dest;
    t.time = transition_time;
    t.logic.assign(condition, condition + n_condition);
    L->transitions.push_back(std::move(t));
    sync_machine_state(*A);
    if (out_transition) *out_transition = (uint32_t)L->transitions.size() - 1;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_get_state(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t instance, int32_t* active_state,
                        int32_t* active_transition) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    int32_t as = L->initial_active, at = -1;
    if (A->mstate.size() == A->n_instances) {
        as = A->mstate[instance].layers[layer].active_state;
        at = A->mstate[instance].layers[layer].active_transition;
    }
    if (active_state) *active_state = as;
    if (active_transition) *active_transition = at;
    return FYX_OK;
    FYX_GUARD_END(c)
}

// ---- run-time edits of a machine: the definition is re-sent, the run-time state carried over (fyrox_hip.h) ----

int fyx_machine_clear(fyx_ctx* c, uint64_t animator_id) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    // the root-motion records on the device stay: the next frame re-lays them out by position (anim_device.h)
    A->param_defaults.clear();
    A->layers.clear();
    for (MachineState& m : A->mstate) {
        m.params.clear();
        m.layers.clear();
        m.memo_valid = false;
    }
    A->steady_gen = 0;
    A->state_anims.clear();
    A->rm_layer_base.clear();
    A->n_rm_slots = 0;
    A->masks_dirty = true;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_machine_get_parameter(fyx_ctx* c, uint64_t animator_id, uint32_t parameter, uint32_t instance, int* kind, float* f0,
                              float* f1, uint32_t* u) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (parameter >= A->param_defaults.size()) return fail(c, FYX_ERR_INVALID_ARG, "parameter %u does not exist", parameter);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    const Param& p = A->mstate.size() == A->n_instances ? A->mstate[instance].params[parameter] : A->param_defaults[parameter];
    if (kind) *kind = p.kind;
    if (f0) *f0 = p.f0;
    if (f1) *f1 = p.f1;
    if (u) *u = p.u;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_set_state(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t instance, int32_t active_state,
                        int32_t active_transition) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (active_state < -1 || active_state >= (int32_t)L->states.size()) return fail(c, FYX_ERR_INVALID_ARG, "state %d does not exist", active_state);
    if (active_transition < -1 || active_transition >= (int32_t)L->transitions.size())
        return fail(c, FYX_ERR_INVALID_ARG, "transition %d does not exist", active_transition);
    return for_layer_states(c, A, layer, instance, [&](LayerState& S) {
        S.active_state = active_state;
        S.active_transition = active_transition;
    });
    FYX_GUARD_END(c)
}

int fyx_layer_get_transition_state(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t instance, uint32_t transition,
                                   float* elapsed_time, float* blend_factor) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (transition >= L->transitions.size()) return fail(c, FYX_ERR_INVALID_ARG, "transition %u does not exist", transition);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    TransitionState t;
    if (A->mstate.size() == A->n_instances) t = A->mstate[instance].layers[layer].transitions[transition];
    if (elapsed_time) *elapsed_time = t.elapsed;
    if (blend_factor) *blend_factor = t.blend_factor;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_set_transition_state(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t instance, uint32_t transition,
                                   float elapsed_time, float blend_factor) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (transition >= L->transitions.size()) return fail(c, FYX_ERR_INVALID_ARG, "transition %u does not exist", transition);
    return for_layer_states(c, A, layer, instance, [&](LayerState& S) {
        S.transitions[transition].elapsed = elapsed_time;
        S.transitions[transition].blend_factor = blend_factor;
    });
    FYX_GUARD_END(c)
}

int fyx_layer_get_node_state(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t instance, uint32_t node, int* has_prev,
                             uint32_t* prev_index, float* blend_time) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (node >= L->nodes.size() || L->nodes[node].type != NODE_BY_INDEX)
        return fail(c, FYX_ERR_INVALID_ARG, "node %u is not a BlendAnimationsByIndex node", node);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    ByIndexState b;
    if (A->mstate.size() == A->n_instances) b = A->mstate[instance].layers[layer].by_index[L->nodes[node].by_index_slot];
    if (has_prev) *has_prev = b.has_prev ? 1 : 0;
    if (prev_index) *prev_index = b.prev;
    if (blend_time) *blend_time = b.blend_time;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_set_node_state(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t instance, uint32_t node, int has_prev,
                             uint32_t prev_index, float blend_time) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (node >= L->nodes.size() || L->nodes[node].type != NODE_BY_INDEX)
        return fail(c, FYX_ERR_INVALID_ARG, "node %u is not a BlendAnimationsByIndex node", node);
    const uint32_t slot = L->nodes[node].by_index_slot;
    return for_layer_states(c, A, layer, instance, [&](LayerState& S) {
        S.by_index[slot].has_prev = has_prev != 0;
        S.by_index[slot].prev = prev_index;
        S.by_index[slot].blend_time = blend_time;
    });
    FYX_GUARD_END(c)
}

int fyx_layer_reset(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t instance) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    const int32_t entry = L->entry_state;
    return for_layer_states(c, A, layer, instance, [&](LayerState& S) {   // layer.rs:290-296
        for (TransitionState& t : S.transitions) t = TransitionState();
        S.active_state = entry;
    });
    FYX_GUARD_END(c)
}

// ---- per frame -----------------------------------------------------------------------------

static int update_common(fyx_ctx* c, uint64_t animator_id, int mode, float dt) {
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context: no GPU to run the pose kernels on");
    if (mode == 1 && A->layers.empty()) return fail(c, FYX_ERR_INVALID_ARG, "animator %llu has no machine layers", (unsigned long long)animator_id);
    if (int rc = plan_frame(c, *A, mode, dt)) return rc;
    return run_frame(c, *A, true);
}

int fyx_animation_player_update(fyx_ctx* c, uint64_t animator_id, float dt) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    return update_common(c, animator_id, 0, dt);
    FYX_GUARD_END(c)
}

int fyx_absm_update(fyx_ctx* c, uint64_t animator_id, float dt) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    return update_common(c, animator_id, 1, dt);
    FYX_GUARD_END(c)
}

static int scene_members(fyx_ctx* c, SceneBatch& S, const uint64_t* animator_ids, uint32_t n_animators) {
    if (n_animators && !animator_ids) return fail(c, FYX_ERR_INVALID_ARG, "null animator list");
    // the same list as last frame over the same set of animators: the same members (a scene of 256 characters spent 10 us per frame
    // on the duplicate check's hash set and the lookups)
    if (S.members_gen == store(c).animators_gen && S.member_ids.size() == n_animators && S.animators.size() == n_animators &&
        (n_animators == 0 || memcmp(S.member_ids.data(), animator_ids, (size_t)n_animators * 8) == 0))
        return FYX_OK;
    S.members_gen = 0;
    ++S.members_epoch;
    S.animators.clear();
    std::unordered_set<uint64_t> seen;
    for (uint32_t k = 0; k < n_animators; ++k) {
        auto it = store(c).animators.find(animator_ids[k]);
        if (it == store(c).animators.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "animator %llu", (unsigned long long)animator_ids[k]);
        if (!seen.insert(animator_ids[k]).second)
            return fail(c, FYX_ERR_INVALID_ARG, "animator %llu is listed twice", (unsigned long long)animator_ids[k]);
        S.animators.push_back(it->second.get());
    }
    S.member_ids.assign(animator_ids, animator_ids + n_animators);
    S.members_gen = store(c).animators_gen;
    return FYX_OK;
}

int fyx_scene_update(fyx_ctx* c, const uint64_t* animator_ids, uint32_t n_animators, float dt) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context: no GPU to run the pose kernels on");
    SceneBatch& S = store(c).scene;
    if (int rc = scene_members(c, S, animator_ids, n_animators)) return rc;
    if (n_animators == 0) return FYX_OK;
    return scene_frame(c, S, dt);
    FYX_GUARD_END(c)
}

int fyx_debug_scene_tables(fyx_ctx* c, const uint64_t* animator_ids, uint32_t n_animators, int stage, uint32_t* out_blocks,
                           uint32_t capacity, uint32_t* n_blocks) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    if (stage < 0 || stage >= kSceneStages) return fail(c, FYX_ERR_INVALID_ARG, "stage %d", stage);
    SceneBatch& S = store(c).scene;
    if (int rc = scene_members(c, S, animator_ids, n_animators)) return rc;
    std::vector<uint4> tables[kSceneStages];
    for (size_t k = 0; k < S.animators.size(); ++k)
        scene_blocks((uint32_t)k, scene_shape(c, *S.animators[k], (uint32_t)S.animators[k]->prop_slots.size()), tables);
    if (n_blocks) *n_blocks = (uint32_t)tables[stage].size();
    if (out_blocks) memcpy(out_blocks, tables[stage].data(), std::min<size_t>(tables[stage].size(), capacity) * sizeof(uint4));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_debug_rig_walk(fyx_ctx* c, uint64_t rig_id, uint32_t* out_words, uint32_t capacity, uint32_t* n_words) {
    if (!c || !n_words) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    auto it = store(c).rigs.find(rig_id);
    if (it == store(c).rigs.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "rig %llu", (unsigned long long)rig_id);
    const std::vector<uint32_t>& w = it->second.walk;
    *n_words = (uint32_t)w.size();
    if (out_words) memcpy(out_words, w.data(), std::min<size_t>(w.size(), capacity) * 4);
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_debug_rig_chunks(fyx_ctx* c, uint64_t rig_id, uint32_t* out_words, uint32_t capacity, uint32_t* n_words) {
    if (!c || !n_words) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    auto it = store(c).rigs.find(rig_id);
    if (it == store(c).rigs.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "rig %llu", (unsigned long long)rig_id);
    const std::vector<uint32_t>& w = it->second.chunks;
    *n_words = (uint32_t)w.size();
    if (out_words) memcpy(out_words, w.data(), std::min<size_t>(w.size(), capacity) * 4);
    return FYX_OK;
    FYX_GUARD_END(c)
}

// The kernels' decision-making leaves, compiled for the host (anim_leaves.h): what the CPU suite runs against the oracle.
int fyx_debug_span_value_at(const float* span_records, uint32_t n_keys, uint32_t need, float time, uint32_t hint, float out_values[4], uint32_t* out_hint) {
    if (!span_records || !out_values || !out_hint || n_keys < 2 || need < 1 || need > 4) return FYX_ERR_INVALID_ARG;
    const uint32_t stride = span_stride(need);              // f4 per span: header + one part per curve
    float val[4] = {0.f, 0.f, 0.f, 0.f};
    *out_hint = span_track_value_at(reinterpret_cast<const f4*>(span_records), n_keys, stride, (int)need, time, hint, val);
    memcpy(out_values, val, 16);
    return FYX_OK;
}

int fyx_debug_classify_fold_program(const uint32_t* ops_xy, uint32_t n_ops, uint32_t* out_d, uint32_t* out_k, int* out_mask, int* out_player, int* out_straight) {
    if ((n_ops && !ops_xy) || !out_d || !out_k || !out_mask || !out_player || !out_straight) return FYX_ERR_INVALID_ARG;
    const StraightShape s = classify_fold_program_host(ops_xy, n_ops);
    *out_d = s.d; *out_k = s.k; *out_mask = s.mask; *out_player = s.player; *out_straight = s.straight;
    return FYX_OK;
}

int fyx_scene_plan(fyx_ctx* c, const uint64_t* animator_ids, uint32_t n_animators, float dt) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    SceneBatch& S = store(c).scene;
    if (int rc = scene_members(c, S, animator_ids, n_animators)) return rc;
    return scene_plan(c, S, dt);
    FYX_GUARD_END(c)
}

int fyx_animator_update_transforms(fyx_ctx* c, uint64_t animator_id) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    return run_frame(c, *A, false);
    FYX_GUARD_END(c)
}

int fyx_animator_palette(fyx_ctx* c, uint64_t animator_id, uint64_t bones_id, float* d_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    auto bit = store(c).bones.find(bones_id);
    if (bit == store(c).bones.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "bone list %llu is not registered", (unsigned long long)bones_id);
    if (bit->second.rig_id != A->rig_id) return fail(c, FYX_ERR_INVALID_ARG, "bone list belongs to another rig");
    if (!d_out) return fail(c, FYX_ERR_INVALID_ARG, "d_out_palette is null");
    hipStream_t ps = nullptr;
    if (int rc = enter_skin(c, &ps)) return rc;      // behind the frame's pose update, on its stream
    if (int rc = ensure_device_state(c, *A)) return rc;
    FYX_HIP(c, launch_palette_gather(A->d_global, A->rig->d_inv_bind, bit->second.d_bone_nodes, A->rig->n_nodes,
                                     bit->second.n_bones, A->n_instances, d_out, ps));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_set_palette_output(fyx_ctx* c, uint64_t animator_id, uint64_t bones_id, float* d_out) {
    return fyx_animator_set_palette_output_pair(c, animator_id, bones_id, d_out, nullptr);
}

int fyx_animator_current_palette(fyx_ctx* c, uint64_t animator_id, uint64_t bones_id, float** d_palette) {
    if (!c || !d_palette) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    *d_palette = nullptr;
    FYX_ANIMATOR_RO(c, A, animator_id);
    for (const Animator::PaletteOut& p : A->palette_outputs)
        if (p.bones_id == bones_id) {
            *d_palette = palette_last_written(*A, p);
            return FYX_OK;
        }
    return fail(c, FYX_ERR_INVALID_ARG, "bone list %llu is not a palette output of animator %llu", (unsigned long long)bones_id, (unsigned long long)animator_id);
    FYX_GUARD_END(c)
}

int fyx_animator_set_palette_output_pair(fyx_ctx* c, uint64_t animator_id, uint64_t bones_id, float* d_out, float* d_out_alt) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!d_out && d_out_alt) return fail(c, FYX_ERR_INVALID_ARG, "a second palette buffer without a first");
    if (d_out && d_out == d_out_alt) return fail(c, FYX_ERR_INVALID_ARG, "the pair's two palette buffers are the same buffer");
    if (reinterpret_cast<uintptr_t>(d_out_alt) & 15u) return fail(c, FYX_ERR_INVALID_ARG, "palette output must be 16-byte aligned");
    auto bit = store(c).bones.find(bones_id);
    if (bit == store(c).bones.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "bone list %llu is not registered", (unsigned long long)bones_id);
    if (bit->second.rig_id != A->rig_id) return fail(c, FYX_ERR_INVALID_ARG, "bone list belongs to another rig");
    if (reinterpret_cast<uintptr_t>(d_out) & 15u) return fail(c, FYX_ERR_INVALID_ARG, "palette output must be 16-byte aligned");
    ++A->api_gen;
    auto& v = A->palette_outputs;
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i].bones_id == bones_id) {
            if (d_out) { v[i].d_out = d_out; v[i].d_out_alt = d_out_alt; return FYX_OK; }
            v.erase(v.begin() + (long)i);
            auto& so = A->skin_outputs;       // the skin outputs on this palette go with it
            for (size_t k = so.size(); k-- > 0;)
                if (so[k].bones_id == bones_id) so.erase(so.begin() + (long)k);
            return FYX_OK;
        }
    if (!d_out) return FYX_OK;
    if (v.size() >= (size_t)kMaxPaletteOutputs)
        return fail(c, FYX_ERR_UNSUPPORTED, "at most %d palette outputs per animator (use fyx_animator_palette for more)", kMaxPaletteOutputs);
    v.push_back(Animator::PaletteOut{bones_id, d_out, bit->second.d_bone_nodes, bit->second.n_bones, d_out_alt});
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_set_skin_output(fyx_ctx* c, uint64_t animator_id, uint64_t bones_id, uint64_t mesh_id, float* d_out_pos, float* d_out_normal,
                                 float* d_out_tangent) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    ++A->api_gen;
    auto& v = A->skin_outputs;
    size_t at = v.size();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i].bones_id == bones_id && v[i].mesh_id == mesh_id) at = i;
    if (!d_out_pos && !d_out_normal && !d_out_tangent) {      // remove
        if (at < v.size()) v.erase(v.begin() + (long)at);
        return FYX_OK;
    }
    const Animator::PaletteOut* po = nullptr;
    for (const Animator::PaletteOut& p : A->palette_outputs)
        if (p.bones_id == bones_id) po = &p;
    if (!po) return fail(c, FYX_ERR_INVALID_ARG, "bone list %llu is not a palette output of animator %llu (fyx_animator_set_palette_output first: the skin output uses that palette)",
                         (unsigned long long)bones_id, (unsigned long long)animator_id);
    if ((reinterpret_cast<uintptr_t>(d_out_pos) | reinterpret_cast<uintptr_t>(d_out_normal)) & 3u || reinterpret_cast<uintptr_t>(d_out_tangent) & 15u)
        return fail(c, FYX_ERR_INVALID_ARG, "skin outputs must be 4-byte (position, normal) / 16-byte (tangent) aligned");
    if (has_device(c)) {      // what fyx_lbs_skin_device would refuse is refused here, not in the middle of a frame
        LbsArgs a;
        if (int rc = skin_args_of(c, mesh_id, po->d_out, po->n_bones, A->n_instances, d_out_pos, d_out_normal, d_out_tangent, &a)) return rc;
    }
    if (at == v.size()) {
        if (v.size() >= (size_t)kMaxFrameSkins)
            return fail(c, FYX_ERR_UNSUPPORTED, "at most %d skin outputs per animator (skin the others with fyx_lbs_skin_device / fyx_lbs_skin_batch)", kMaxFrameSkins);
        v.push_back(Animator::SkinOut{bones_id, mesh_id, d_out_pos, d_out_normal, d_out_tangent});
    } else {
        v[at] = Animator::SkinOut{bones_id, mesh_id, d_out_pos, d_out_normal, d_out_tangent};
    }
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_debug_frame_counter_add(fyx_ctx* c, uint64_t animator_id, int32_t delta) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (int rc = sync_all(c)) return rc;
    if (int rc = ensure_device_state(c, *A)) return rc;
    if (!A->d_frame_counter) return fail(c, FYX_ERR_INVALID_ARG, "the animator has not run a one-launch frame yet");
    for (uint32_t r = 0; r < kFrameCounterReplicas; ++r) {
        uint32_t* w = A->d_frame_counter + (size_t)r * (kFrameCounterStride / 4u);
        uint32_t v = 0;
        FYX_HIP(c, hipMemcpy(&v, w, 4, hipMemcpyDeviceToHost));
        v += (uint32_t)delta;
        FYX_HIP(c, hipMemcpy(w, &v, 4, hipMemcpyHostToDevice));
    }
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_set_local_trs(fyx_ctx* c, uint64_t animator_id, uint32_t node, uint32_t first_instance,
                               uint32_t n_instances, const float* trs) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (node >= A->rig->n_nodes) return fail(c, FYX_ERR_INVALID_ARG, "node %u out of range", node);
    if ((uint64_t)first_instance + n_instances > A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance range out of bounds");
    if (n_instances == 0) return FYX_OK;
    if (!trs) return fail(c, FYX_ERR_INVALID_ARG, "trs is null");
    if (int rc = enter_primary(c)) return rc;
    if (int rc = ensure_device_state(c, *A)) return rc;
    std::vector<float> recs((size_t)n_instances * 12, 0.f);
    for (uint32_t i = 0; i < n_instances; ++i) {
        const float* s = trs + (size_t)i * 10;
        float* d = &recs[(size_t)i * 12];
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        d[4] = s[3]; d[5] = s[4]; d[6] = s[5]; d[7] = s[6];
        d[8] = s[7]; d[9] = s[8]; d[10] = s[9];
    }
    char* dst = reinterpret_cast<char*>(A->d_node_trs) + ((size_t)first_instance * A->rig->n_nodes + node) * 48;
    FYX_HIP(c, hipMemcpy2DAsync(dst, (size_t)A->rig->n_nodes * 48, recs.data(), 48, 48, n_instances,
                                hipMemcpyHostToDevice, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

static int locate(fyx_ctx* c, Animator* A, int what, void** ptr, size_t* bytes) {
    const size_t in = (size_t)A->n_instances * A->rig->n_nodes;
    if (int rc = ensure_device_state(c, *A)) return rc;
    if (what == FYX_READ_LOCAL_TRS) { *ptr = A->d_node_trs; *bytes = in * 48; return FYX_OK; }
    if (what == FYX_READ_LOCAL_MATRIX) { *ptr = A->d_local; *bytes = in * 64; return FYX_OK; }
    if (what == FYX_READ_GLOBAL_MATRIX) { *ptr = A->d_global; *bytes = in * 64; return FYX_OK; }
    // (an animator that keeps two device animations per animation: 2 a holds what the pose applies, 2 a + 1 what a blend reads of it)
    const size_t per_anim = A->shadows ? 2 : 1;
    if (what >= FYX_READ_ANIMATION_POSE && what < FYX_READ_ANIMATION_BLEND_VIEW && (size_t)(what - FYX_READ_ANIMATION_POSE) < A->anims.size()) {
        *ptr = reinterpret_cast<char*>(A->d_anim_pose) + (size_t)(what - FYX_READ_ANIMATION_POSE) * per_anim * in * 48;
        *bytes = in * 48;
        return FYX_OK;
    }
    if (what >= FYX_READ_ANIMATION_BLEND_VIEW && (size_t)(what - FYX_READ_ANIMATION_BLEND_VIEW) < A->anims.size()) {
        *ptr = reinterpret_cast<char*>(A->d_anim_pose) + ((size_t)(what - FYX_READ_ANIMATION_BLEND_VIEW) * per_anim + (per_anim - 1)) * in * 48;
        *bytes = in * 48;
        return FYX_OK;
    }
    return fail(c, FYX_ERR_INVALID_ARG, "unknown array selector %d", what);
}

int fyx_animator_read(fyx_ctx* c, uint64_t animator_id, int what, float* host_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (!host_out) return fail(c, FYX_ERR_INVALID_ARG, "host_out is null");
    if (int rc = enter_primary(c)) return rc;
    void* p = nullptr;
    size_t bytes = 0;
    if (int rc = locate(c, A, what, &p, &bytes)) return rc;
    FYX_HIP(c, hipMemcpyAsync(host_out, p, bytes, hipMemcpyDeviceToHost, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_device_ptr(fyx_ctx* c, uint64_t animator_id, int what, void** out) {
    if (!c || !out) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (int rc = enter_primary(c)) return rc;
    size_t bytes = 0;
    return locate(c, A, what, out, &bytes);
    FYX_GUARD_END(c)
}

int fyx_animator_plan(fyx_ctx* c, uint64_t animator_id, int mode, float dt, float* times, uint8_t* ticked,
                      uint32_t* program_offset, uint32_t* ops, uint32_t ops_capacity, uint32_t* n_ops) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (mode != 0 && mode != 1 && mode != -1) return fail(c, FYX_ERR_INVALID_ARG, "mode %d", mode);
    if (mode == 1 && A->layers.empty()) return fail(c, FYX_ERR_INVALID_ARG, "animator has no machine layers");
    if (mode >= 0) {
        if (int rc = plan_frame(c, *A, mode, dt)) return rc;
    }
    if (times) memcpy(times, A->times.data(), A->times.size() * 4);
    if (ticked) memcpy(ticked, A->ticked.data(), A->ticked.size());
    if (program_offset) memcpy(program_offset, A->prog_off.data(), A->prog_off.size() * 4);
    if (n_ops) *n_ops = (uint32_t)A->ops.size();
    if (ops) memcpy(ops, A->ops.data(), std::min<size_t>(A->ops.size(), ops_capacity) * 8);
    return FYX_OK;
    FYX_GUARD_END(c)
}

// ---- signals / events / root motion ----------------------------------------------------------

int fyx_animation_add_signal(fyx_ctx* c, uint64_t animator_id, uint32_t animation, float time, int enabled, uint32_t* out_signal) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (animation >= A->anims.size() || A->anims[animation].removed) return fail(c, FYX_ERR_INVALID_ARG, "animation %u does not exist", animation);
    A->anims[animation].signals.push_back(AnimationDef::Signal{time, (uint8_t)(enabled ? 1 : 0)});
    if (out_signal) *out_signal = (uint32_t)A->anims[animation].signals.size() - 1;
    return FYX_OK;
    FYX_GUARD_END(c)
}
int fyx_animation_set_signal_enabled(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t signal, int enabled) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (animation >= A->anims.size() || A->anims[animation].removed) return fail(c, FYX_ERR_INVALID_ARG, "animation %u does not exist", animation);
    if (signal >= A->anims[animation].signals.size()) return fail(c, FYX_ERR_INVALID_ARG, "signal %u does not exist", signal);
    A->anims[animation].signals[signal].enabled = enabled ? 1 : 0;
    return FYX_OK;
    FYX_GUARD_END(c)
}
int fyx_animation_set_max_event_capacity(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance, uint32_t capacity) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    return for_instances(c, A, animation, instance, [&](AnimState& s) { s.max_event_capacity = capacity; });
    FYX_GUARD_END(c)
}
int fyx_animation_pop_event(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance, int32_t* out_signal) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    if (!out_signal) return fail(c, FYX_ERR_INVALID_ARG, "out_signal is null");
    return for_instances(c, A, animation, instance, [&](AnimState& s) {
        if (s.events.empty()) { *out_signal = -1; return; }
        *out_signal = s.events.front();
        s.events.pop_front();
    });
    FYX_GUARD_END(c)
}
int fyx_animation_event_count(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance, uint32_t* out_count) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    if (!out_count) return fail(c, FYX_ERR_INVALID_ARG, "out_count is null");
    return for_instances(c, A, animation, instance, [&](AnimState& s) { *out_count = (uint32_t)s.events.size(); });
    FYX_GUARD_END(c)
}
int fyx_animation_clear_events(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    return for_instances(c, A, animation, instance, [&](AnimState& s) { s.events.clear(); });
    FYX_GUARD_END(c)
}

int fyx_animator_track_root_motion(fyx_ctx* c, uint64_t animator_id, int enabled) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (!enabled) {
        for (const AnimationDef& an : A->anims)
            if (an.rm_node >= 0) return fail(c, FYX_ERR_INVALID_ARG, "an animation still has root motion settings");
        if (has_device(c) && (A->d_rm_anim || A->d_rm_slots)) {
            if (int rc = enter_primary(c)) return rc;
            FYX_HIP(c, hipStreamSynchronize(c->stream));
        }
        dfree(A->d_rm_anim); dfree(A->d_rm_slots);
        A->d_rm_anim = nullptr; A->d_rm_slots = nullptr;
        A->dev_rm_anim_capacity = 0; A->dev_rm_slots = 0;
        A->dev_rm_layer_nodes.clear();
    }
    A->rm_enabled = enabled != 0;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animation_set_root_motion_settings(fyx_ctx* c, uint64_t animator_id, uint32_t animation, int32_t node,
                                           int ignore_x, int ignore_y, int ignore_z, int ignore_rotations) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (animation >= A->anims.size() || A->anims[animation].removed) return fail(c, FYX_ERR_INVALID_ARG, "animation %u does not exist", animation);
    if (node >= (int32_t)A->rig->n_nodes) return fail(c, FYX_ERR_INVALID_ARG, "root motion node %d of a %u-node rig", node, A->rig->n_nodes);
    AnimationDef& an = A->anims[animation];
    an.rm_node = node < 0 ? -1 : node;
    an.rm_ignore = (ignore_x ? 1u : 0u) | (ignore_y ? 2u : 0u) | (ignore_z ? 4u : 0u) | (ignore_rotations ? 8u : 0u);
    A->anims_dirty = true;
    if (node >= 0) A->rm_enabled = true;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animation_read_root_motion(fyx_ctx* c, uint64_t animator_id, uint32_t animation, fyx_root_motion* host_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (animation >= A->anims.size() || A->anims[animation].removed) return fail(c, FYX_ERR_INVALID_ARG, "animation %u does not exist", animation);
    if (!host_out) return fail(c, FYX_ERR_INVALID_ARG, "host_out is null");
    if (!A->rm_enabled) return fail(c, FYX_ERR_INVALID_ARG, "root motion is not tracked on this animator");
    if (int rc = enter_primary(c)) return rc;
    if (int rc = ensure_device_state(c, *A)) return rc;
    static_assert(sizeof(fyx_root_motion) == 32, "fyx_root_motion layout");
    // the first 32 bytes of a RootMotionDev are exactly a fyx_root_motion
    FYX_HIP(c, hipMemcpy2DAsync(host_out, sizeof(fyx_root_motion), A->d_rm_anim + (size_t)animation * (A->shadows ? 2 : 1) * A->n_instances,
                                sizeof(RootMotionDev), sizeof(fyx_root_motion), A->n_instances, hipMemcpyDeviceToHost, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < A->n_instances; ++i)
        if (!host_out[i].has) { memset(&host_out[i], 0, sizeof host_out[i]); host_out[i].delta_rotation[3] = 1.0f; }
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_absm_read_root_motion(fyx_ctx* c, uint64_t animator_id, int32_t layer, fyx_root_motion* host_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (!host_out) return fail(c, FYX_ERR_INVALID_ARG, "host_out is null");
    if (!A->rm_enabled) return fail(c, FYX_ERR_INVALID_ARG, "root motion is not tracked on this animator");
    if (layer >= (int32_t)A->layers.size()) return fail(c, FYX_ERR_INVALID_ARG, "layer %d does not exist", layer);
    if (int rc = enter_primary(c)) return rc;
    if (int rc = ensure_device_state(c, *A)) return rc;
    uint32_t slot = A->dev_rm_slots - 1;
    if (layer >= 0) {
        slot = 0;
        for (int32_t l = 0; l < layer; ++l) slot += (uint32_t)A->layers[l].nodes.size() + 1;
        slot += (uint32_t)A->layers[layer].nodes.size();
    }
    FYX_HIP(c, hipMemcpy2DAsync(host_out, 32, reinterpret_cast<const char*>(A->d_rm_slots) + (size_t)slot * 32,
                                (size_t)A->dev_rm_slots * 32, 32, A->n_instances, hipMemcpyDeviceToHost, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < A->n_instances; ++i)
        if (!host_out[i].has) { memset(&host_out[i], 0, sizeof host_out[i]); host_out[i].delta_rotation[3] = 1.0f; }
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_pop_event(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t instance, fyx_layer_event* out_event, int* out_has) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    (void)L;
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    if (!out_event || !out_has) return fail(c, FYX_ERR_INVALID_ARG, "out pointers are null");
    *out_has = 0;
    if (A->mstate.size() != A->n_instances) return FYX_OK;
    std::deque<fyx_layer_event>& q = A->mstate[instance].layers[layer].events;
    if (q.empty()) return FYX_OK;
    *out_event = q.front();
    q.pop_front();
    *out_has = 1;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_plan_root_motion(fyx_ctx* c, uint64_t animator_id, uint32_t* program_offset, uint32_t* ops,
                                  uint32_t ops_capacity, uint32_t* n_ops, uint32_t* n_slots, float* slices) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!A->rm_enabled) return fail(c, FYX_ERR_INVALID_ARG, "root motion is not tracked on this animator");
    if (A->rm_prog_off.size() != (size_t)A->n_instances + 1) return fail(c, FYX_ERR_INVALID_ARG, "no frame has been planned yet");
    if (program_offset) memcpy(program_offset, A->rm_prog_off.data(), A->rm_prog_off.size() * 4);
    if (n_ops) *n_ops = (uint32_t)A->rm_ops.size();
    if (n_slots) *n_slots = A->n_rm_slots;
    if (ops) memcpy(ops, A->rm_ops.data(), std::min<size_t>(A->rm_ops.size(), ops_capacity) * 16);
    if (slices) memcpy(slices, A->slices.data(), A->slices.size() * 8);
    return FYX_OK;
    FYX_GUARD_END(c)
}

// ---- Property{..} slots ----------------------------------------------------------------------

int fyx_animator_property_count(fyx_ctx* c, uint64_t animator_id, uint32_t* out_count) {
    if (!c || !out_count) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    *out_count = (uint32_t)A->prop_slots.size();
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_property_slot(fyx_ctx* c, uint64_t animator_id, int32_t node, int32_t property_id, int32_t* out_slot) {
    if (!c || !out_slot) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    const std::pair<int32_t, int32_t> key(node, property_id);
    const auto it = std::find(A->prop_slots.begin(), A->prop_slots.end(), key);
    *out_slot = it == A->prop_slots.end() ? -1 : (int32_t)(it - A->prop_slots.begin());
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_read_properties(fyx_ctx* c, uint64_t animator_id, int32_t animation, fyx_property_value* host_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (!host_out) return fail(c, FYX_ERR_INVALID_ARG, "host_out is null");
    if (animation >= (int32_t)A->anims.size() || (animation >= 0 && A->anims[animation].removed))
        return fail(c, FYX_ERR_INVALID_ARG, "animation %d does not exist", animation);
    if (A->prop_slots.empty()) return FYX_OK;
    if (int rc = enter_primary(c)) return rc;
    if (int rc = ensure_device_state(c, *A)) return rc;
    const size_t per = (size_t)A->n_instances * A->dev_prop_slots;
    static_assert(sizeof(fyx_property_value) == sizeof(PropRec), "same record on both sides of the boundary");
    const PropRec* src = animation < 0 ? A->d_prop_out : A->d_prop_pose + (size_t)animation * (A->shadows ? 2 : 1) * per;
    FYX_HIP(c, hipMemcpyAsync(host_out, src, per * sizeof(PropRec), hipMemcpyDeviceToHost, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_blend_shape_weights(fyx_ctx* c, uint64_t animator_id, uint32_t n_shapes, const int32_t* slots,
                                     const float* default_weights, float* d_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (n_shapes == 0) return FYX_OK;
    if (n_shapes > FYX_MAX_BLEND_SHAPES) return fail(c, FYX_ERR_UNSUPPORTED, "%u blend shapes", n_shapes);
    if (!slots || !default_weights || !d_out) return fail(c, FYX_ERR_INVALID_ARG, "null pointer");
    if (int rc = enter_primary(c)) return rc;
    if (int rc = ensure_device_state(c, *A)) return rc;
    // slots + defaults travel as one small block through the scratch buffer
    if (int rc = ensure_scratch(c, (size_t)n_shapes * 8 + 64)) return rc;
    int32_t* d_slots = static_cast<int32_t*>(c->scratch);
    float* d_def = reinterpret_cast<float*>(d_slots + n_shapes);
    FYX_HIP(c, hipMemcpyAsync(d_slots, slots, (size_t)n_shapes * 4, hipMemcpyHostToDevice, c->stream));
    FYX_HIP(c, hipMemcpyAsync(d_def, default_weights, (size_t)n_shapes * 4, hipMemcpyHostToDevice, c->stream));
    FYX_HIP(c, launch_blend_shape_weights(A->d_prop_out, A->dev_prop_slots, A->n_instances, d_slots, d_def, n_shapes, d_out,
                                          c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

// ---- MachineLayer::collect_active_animations_events (layer.rs:308-401) -----------------------

namespace {
struct EventCollector {
    const Animator& A;
    const LayerDef& L;
    const MachineState& ms;
    const LayerState& LS;
    const AnimState* as;
    int strategy;
    fyx_animation_event* out;
    uint32_t cap, n = 0;

    const Param* param(int32_t idx) const { return (idx >= 0 && (size_t)idx < ms.params.size()) ? &ms.params[idx] : nullptr; }
    void push(uint32_t anim, int32_t sig) {
        if (n < cap) { out[n].animation = anim; out[n].signal = sig; }
        ++n;
    }
    void node(int32_t h) {
        if (h < 0 || (size_t)h >= L.nodes.size()) return;
        const PoseNodeDef& nd = L.nodes[h];
        switch (nd.type) {
            case NODE_PLAY:  // play.rs:106-122: the animation's queued events, in order, not removed
                if (nd.animation < A.anims.size() && !A.anims[nd.animation].removed)
                    for (int32_t sgn : as[nd.animation].events) push(nd.animation, sgn);
                return;
            case NODE_BLEND: {  // blend.rs:172-222
                if (strategy == FYX_EVENTS_ALL) { for (const BlendInput& in : nd.inputs) node(in.source); return; }
                int best = -1;
                float bw = 0.f;
                for (size_t i = 0; i < nd.inputs.size(); ++i) {
                    float w;
                    if (nd.inputs[i].weight_param < 0) w = nd.inputs[i].weight_const;
                    else {
                        const Param* p = param(nd.inputs[i].weight_param);
                        if (!p || p->kind != FYX_PARAM_WEIGHT) continue;  // PoseWeight::value -> None
                        w = p->f0;
                    }
                    if (best < 0) { best = (int)i; bw = w; continue; }
                    // Iterator::max_by keeps the LAST of equal maxima, min_by the FIRST of equal minima
                    if (strategy == FYX_EVENTS_MAX_WEIGHT) { if (!(w < bw)) { best = (int)i; bw = w; } }
                    else if (w < bw) { best = (int)i; bw = w; }
                }
                if (best >= 0) node(nd.inputs[best].source);
                return;
            }
            case NODE_BY_INDEX: {  // blend.rs:370-438
                const Param* p = param(nd.param);
                const ByIndexState& st = LS.by_index[nd.by_index_slot];
                if (!p || p->kind != FYX_PARAM_INDEX || !st.has_prev) return;
                const uint32_t cur = p->u;
                if (st.prev != cur) {
                    if (st.prev < nd.inputs.size() && cur < nd.inputs.size()) {
                        const BlendInput& pi = nd.inputs[st.prev];
                        const BlendInput& ci = nd.inputs[cur];
                        const float interpolator = st.blend_time / ci.blend_time;
                        if (strategy == FYX_EVENTS_ALL) { node(pi.source); node(ci.source); }
                        else if (strategy == FYX_EVENTS_MAX_WEIGHT) node((interpolator < 0.5f ? pi : ci).source);
                        else node((interpolator < 0.5f ? ci : pi).source);
                    }
                } else if (cur < nd.inputs.size()) {
                    node(nd.inputs[cur].source);
                }
                return;
            }
            case NODE_BLEND_SPACE: {  // blendspace.rs:157-218
                const Param* p = param(nd.param);
                if (!p || p->kind != FYX_PARAM_SAMPLING_POINT) return;
                int idx[3];
                float w[3];
                const float sp[2] = {p->f0, p->f1};
                if (!Planner::blend_space_weights(nd, sp, idx, w)) return;
                const int32_t src[3] = {nd.inputs[idx[0]].source, nd.inputs[idx[1]].source, nd.inputs[idx[2]].source};
                for (int k = 0; k < 3; ++k)
                    if (src[k] < 0 || (size_t)src[k] >= L.nodes.size()) return;
                if (strategy == FYX_EVENTS_ALL) { for (int k = 0; k < 3; ++k) node(src[k]); return; }
                int best = 0;
                for (int k = 1; k < 3; ++k) {
                    if (strategy == FYX_EVENTS_MAX_WEIGHT) { if (!(w[k] < w[best])) best = k; }
                    else if (w[k] < w[best]) best = k;
                }
                node(src[best]);
                return;
            }
        }
    }
};
}  // namespace

int fyx_layer_collect_active_animations_events(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t instance,
                                               int strategy, fyx_animation_event* out_events, uint32_t capacity,
                                               uint32_t* n_events, fyx_events_source* out_source) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR_RO(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    if (strategy < FYX_EVENTS_ALL || strategy > FYX_EVENTS_MIN_WEIGHT) return fail(c, FYX_ERR_INVALID_ARG, "strategy %d", strategy);
    if (capacity && !out_events) return fail(c, FYX_ERR_INVALID_ARG, "out_events is null");
    ensure_machine_state(*A);
    const MachineState& ms = A->mstate[instance];
    const LayerState& LS = ms.layers[layer];
    EventCollector ec{*A, *L, ms, LS, A->anim_state.data() + (size_t)instance * A->anims.size(), strategy, out_events, capacity};
    fyx_events_source src{0, -1, -1, -1};
    if (LS.active_state >= 0 && (size_t)LS.active_state < L->states.size()) {
        src = fyx_events_source{1, LS.active_state, -1, -1};
        ec.node(L->states[LS.active_state].root);
    } else if (LS.active_transition >= 0 && (size_t)LS.active_transition < L->transitions.size()) {
        const TransitionDef& tr = L->transitions[LS.active_transition];
        if (tr.source < L->states.size() && tr.dest < L->states.size()) {
            src = fyx_events_source{2, LS.active_transition, (int32_t)tr.source, (int32_t)tr.dest};
            const float bf = LS.transitions[LS.active_transition].blend_factor;
            if (strategy == FYX_EVENTS_ALL) {
                ec.node(L->states[tr.source].root);
                ec.node(L->states[tr.dest].root);
            } else {
                const bool pick_source = strategy == FYX_EVENTS_MAX_WEIGHT ? bf < 0.5f : !(bf < 0.5f);
                ec.node(L->states[pick_source ? tr.source : tr.dest].root);
            }
        }
    }
    if (n_events) *n_events = ec.n;
    if (out_source) *out_source = src;
    return FYX_OK;
    FYX_GUARD_END(c)
}

}  // extern "C"
