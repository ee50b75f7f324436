// anim_device.h -- device side of an animator: persistent state, the per-frame control block and its upload, the
// per-animator frame (run_frame) and the scene frame (scene_plan / scene_frame).  Included by anim_api.hip only, after
// anim_planner.h.
#pragma once

namespace fyx {

namespace {

// A device allocation that is freed again unless released: the grow paths below allocate, fill and copy in several steps,
// any of which may fail and return early.
struct DevGuard {
    void* p = nullptr;
    DevGuard() = default;
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
    ~DevGuard() { if (p) (void)hipFree(p); }
    void* release() { void* r = p; p = nullptr; return r; }
};

// Curves TrackDataContainer::fetch reads of a kind (container.rs:182-297): with fewer it answers None and the node's list gets no value.
inline uint32_t kind_need(int kind) {
    switch (kind) {
        case FYX_KIND_REAL: return 1u;
        case FYX_KIND_VEC2: return 2u;
        case FYX_KIND_VEC3: case FYX_KIND_QUAT_EULER: return 3u;
        default: return 4u;      // Vector4, UnitQuaternion
    }
}
// BoundValueCollectionExt::apply's `if let` (scene/animation/mod.rs:147-186): the TrackValue variant a binding takes.
inline bool kind_fits(int binding, int kind) {
    return binding == FYX_BIND_ROTATION ? (kind == FYX_KIND_QUAT || kind == FYX_KIND_QUAT_EULER) : kind == FYX_KIND_VEC3;
}

// The two views of an animation's node lists (AnimationDef::slots / slots_f): Animation::update_pose pushes the enabled, bound tracks'
// values in track order (lib.rs:895-914).
void build_track_views(const Animator& A, AnimationDef& an, std::vector<int32_t>& ptrack_a, std::vector<int32_t>& ptrack_f) {
    const uint32_t n_nodes = A.rig->n_nodes;
    std::vector<int32_t> sa((size_t)n_nodes * 4, -1), sf((size_t)n_nodes * 4, -1);
    std::vector<uint8_t> bl(n_nodes, 0), mu(n_nodes, 0), first_seen((size_t)n_nodes * 3, 0);
    ptrack_a.assign(std::max<size_t>(A.prop_slots.size(), 1), -1);
    ptrack_f.assign(ptrack_a.size(), -1);
    for (uint32_t t = 0; t < an.td->n_tracks; ++t) {
        if (an.target[t] < 0 || !an.enabled[t]) continue;
        const fyx_track_desc& tr = an.td->tracks[t];
        if (tr.n_curves < kind_need(tr.kind)) continue;      // fetch() -> None: no value
        const size_t node = (size_t)an.target[t];
        if (tr.binding >= FYX_BIND_PROPERTY0) {
            const std::pair<int32_t, int32_t> key(an.target[t], tr.binding - FYX_BIND_PROPERTY0);
            const size_t sl = std::find(A.prop_slots.begin(), A.prop_slots.end(), key) - A.prop_slots.begin();
            if (sl < ptrack_a.size()) {
                ptrack_a[sl] = (int32_t)t;                                     // applied in order: the last one stays
                if (ptrack_f[sl] < 0) ptrack_f[sl] = (int32_t)t;               // find(): the first one
            }
            sa[node * 4 + 3] = sf[node * 4 + 3] = (int32_t)t;                  // the node's pose is not empty
            continue;
        }
        const int b = tr.binding;
        const bool fits = kind_fits(b, tr.kind);
        if (fits) {
            if (sa[node * 4 + b] >= 0) mu[node] |= (uint8_t)(1u << b);
            sa[node * 4 + b] = (int32_t)t;
        } else {
            bl[node] |= 8u;
        }
        if (!first_seen[node * 3 + b]) {
            first_seen[node * 3 + b] = 1;
            if (fits) sf[node * 4 + b] = (int32_t)t;
            else bl[node] |= (uint8_t)(1u << b);
        }
    }
    an.dup = sa != sf || ptrack_a != ptrack_f;
    an.slots = std::move(sa);
    an.slots_f = std::move(sf);
    an.blockers = std::move(bl);
    an.multi = std::move(mu);
}

// ------------------------------------------------------------------------------------------
// Device side of an animator
// ------------------------------------------------------------------------------------------
int ensure_device_state(fyx_ctx* c, Animator& A) {
    const Rig& rig = *A.rig;
    const size_t in = (size_t)A.n_instances * rig.n_nodes;
    if (!A.d_node_trs) {
        FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&A.d_node_trs), std::max<size_t>(in * 48, 16)));
        FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&A.d_local), std::max<size_t>(in * 64, 16)));
        FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&A.d_global), std::max<size_t>(in * 64, 16)));
        for (uint32_t i = 0; i < A.n_instances; ++i)  // every instance starts from the rig's transforms
            FYX_HIP(c, hipMemcpyAsync(reinterpret_cast<char*>(A.d_node_trs) + (size_t)i * rig.n_nodes * 48,
                                      rig.init_trs.data(), (size_t)rig.n_nodes * 48, hipMemcpyHostToDevice,
                                      c->stream));
        if (int rc_ = sync_all(c)) return rc_;
    }
    // the two views of every animation whose bindings changed (host side), and whether the animator needs two records per animation:
    // a machine blends poses, and a blend reads the READ view of the other pose (a player only applies: the APPLY view is all it needs)
    std::vector<std::vector<int32_t>> new_pa(A.anims.size()), new_pf(A.anims.size());
    bool any_dup = false;
    for (size_t a = 0; a < A.anims.size(); ++a) {
        if (A.anims[a].slots_dirty) build_track_views(A, A.anims[a], new_pa[a], new_pf[a]);
        any_dup = any_dup || (A.anims[a].dup && !A.anims[a].removed);
    }
    if (any_dup && !A.layers.empty() && !A.shadows) {
        // From one device animation per animation to two: device animation a becomes 2 a and 2 a + 1, both holding what a held (until
        // now the two views were the same record).  Pose records, sampled property values and root-motion state move; span hints start
        // over (advisory).
        if (int rc_ = sync_all(c)) return rc_;
        auto spread = [&](void** arr, size_t row_bytes, uint32_t rows) -> int {
            if (!*arr || !rows || !row_bytes) return FYX_OK;
            DevGuard n2;
            FYX_HIP(c, hipMalloc(&n2.p, std::max<size_t>(2 * row_bytes * rows, 16)));
            FYX_HIP(c, hipMemcpy2D(n2.p, 2 * row_bytes, *arr, row_bytes, row_bytes, rows, hipMemcpyDeviceToDevice));
            FYX_HIP(c, hipMemcpy2D(static_cast<char*>(n2.p) + row_bytes, 2 * row_bytes, *arr, row_bytes, row_bytes, rows, hipMemcpyDeviceToDevice));
            dfree(*arr);
            *arr = n2.release();
            return FYX_OK;
        };
        if (int rc = spread(reinterpret_cast<void**>(&A.d_anim_pose), in * 48, A.dev_anim_capacity)) return rc;
        if (int rc = spread(reinterpret_cast<void**>(&A.d_prop_pose), (size_t)A.n_instances * A.dev_prop_slots * sizeof(PropRec), A.dev_prop_anims)) return rc;
        if (int rc = spread(reinterpret_cast<void**>(&A.d_rm_anim), (size_t)A.n_instances * sizeof(RootMotionDev), A.dev_rm_anim_capacity)) return rc;
        if (A.d_hints && A.dev_anim_capacity && A.dev_track_capacity) {
            const size_t hb = (size_t)2 * A.dev_anim_capacity * A.n_instances * A.dev_track_capacity * 16;
            dfree(A.d_hints);
            A.d_hints = nullptr;
            FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&A.d_hints), std::max<size_t>(hb, 16)));
            FYX_HIP(c, hipMemset(A.d_hints, 0, std::max<size_t>(hb, 16)));
        }
        if (A.d_slot_hints) FYX_HIP(c, hipMemset(A.d_slot_hints, 0, A.slot_hint_words * 4));
        A.dev_anim_capacity *= 2;
        A.dev_prop_anims *= 2;
        A.dev_rm_anim_capacity *= 2;
        A.shadows = true;
        A.anims_dirty = true;
    }
    const uint32_t n_real = (uint32_t)A.anims.size();
    const uint32_t na = A.n_dev_anims();      // DEVICE animations from here on
    if (na > A.dev_anim_capacity || A.max_tracks > A.dev_track_capacity) {
        // grow pose records / hints; existing contents are preserved
        const uint32_t new_cap = std::max(na, A.dev_anim_capacity);
        const uint32_t new_tracks = std::max(A.max_tracks, A.dev_track_capacity);
        DevGuard np, nh;
        if (int rc_ = sync_all(c)) return rc_;
        FYX_HIP(c, hipMalloc(&np.p, std::max<size_t>((size_t)new_cap * in * 48, 16)));
        FYX_HIP(c, hipMemset(np.p, 0, std::max<size_t>((size_t)new_cap * in * 48, 16)));
        const size_t hb = std::max<size_t>((size_t)new_cap * A.n_instances * std::max(new_tracks, 1u) * 16, 16);
        FYX_HIP(c, hipMalloc(&nh.p, hb));
        FYX_HIP(c, hipMemset(nh.p, 0, hb));
        if (A.d_anim_pose && A.dev_anim_capacity)
            FYX_HIP(c, hipMemcpy(np.p, A.d_anim_pose, (size_t)A.dev_anim_capacity * in * 48, hipMemcpyDeviceToDevice));
        if (A.d_hints && A.dev_anim_capacity && A.dev_track_capacity) {
            // layout [anim][track][curve][instance]: one row per animation, old rows are a prefix of the new ones
            const size_t old_row = (size_t)A.dev_track_capacity * 16 * A.n_instances;
            const size_t new_row = (size_t)new_tracks * 16 * A.n_instances;
            FYX_HIP(c, hipMemcpy2D(nh.p, new_row, A.d_hints, old_row, old_row, A.dev_anim_capacity, hipMemcpyDeviceToDevice));
        }
        dfree(A.d_anim_pose);
        dfree(A.d_hints);
        A.d_anim_pose = static_cast<float4*>(np.release());
        A.d_hints = static_cast<uint32_t*>(nh.release());
        A.dev_anim_capacity = new_cap;
        A.dev_track_capacity = new_tracks;
        A.anims_dirty = true;
    }
    // The per-instance sampler's own hints -- one word per (animation, node, binding, curve, instance) -- only while the animator runs that
    // form (launch_pose_sample's rule: a crowd's form never reads them, and a 10 000-instance crowd would carry 120 MB for nothing: ADVICE r5).
    if (c->sample_form == 1 || (c->sample_form == 0 && A.n_instances < 32)) {
        const size_t want = (size_t)std::max(A.dev_anim_capacity, 1u) * rig.n_nodes * 12 * A.n_instances;
        if (want > A.slot_hint_words) {
            if (int rc_ = sync_all(c)) return rc_;
            dfree(A.d_slot_hints);
            A.d_slot_hints = nullptr;
            A.slot_hint_words = 0;
            FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&A.d_slot_hints), std::max<size_t>(want * 4, 16)));
            FYX_HIP(c, hipMemset(A.d_slot_hints, 0, std::max<size_t>(want * 4, 16)));
            A.slot_hint_words = want;
        }
    }
    // slot tables + animation descriptors (made for the sampler form the animator runs: launch_pose_sample's rule)
    const int inst_form = (c->sample_form == 1 || (c->sample_form == 0 && A.n_instances < 32)) ? 1 : 0;
    if (A.desc_form != inst_form) A.anims_dirty = true;
    bool any_slots = false;
    for (AnimationDef& an : A.anims) any_slots |= an.slots_dirty;
    if (any_slots || A.anims_dirty) {
        if (int rc_ = sync_all(c)) return rc_;
        std::vector<AnimDev> hd(na);
        for (uint32_t a = 0; a < n_real; ++a) {
            AnimationDef& an = A.anims[a];
            if (an.slots_dirty) {      // (the views were built above)
                const std::vector<int32_t>&pa = new_pa[a], &pf = new_pf[a];
                if (an.dev_prop_slots < pa.size()) {
                    dfree(an.d_prop_track);
                    dfree(an.d_prop_track_f);
                    an.d_prop_track = an.d_prop_track_f = nullptr;
                    FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&an.d_prop_track), std::max<size_t>(pa.size() * 4, 16)));
                    FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&an.d_prop_track_f), std::max<size_t>(pa.size() * 4, 16)));
                    an.dev_prop_slots = (uint32_t)pa.size();
                }
                FYX_HIP(c, hipMemcpy(an.d_prop_track, pa.data(), pa.size() * 4, hipMemcpyHostToDevice));
                FYX_HIP(c, hipMemcpy(an.d_prop_track_f, pf.data(), pf.size() * 4, hipMemcpyHostToDevice));
                if (!an.d_slot_track) {
                    FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&an.d_slot_track), std::max<size_t>(an.slots.size() * 4, 16)));
                    FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&an.d_slot_track_f), std::max<size_t>(an.slots.size() * 4, 16)));
                }
                FYX_HIP(c, hipMemcpy(an.d_slot_track, an.slots.data(), an.slots.size() * 4, hipMemcpyHostToDevice));
                FYX_HIP(c, hipMemcpy(an.d_slot_track_f, an.slots_f.data(), an.slots_f.size() * 4, hipMemcpyHostToDevice));
                an.slots_dirty = false;
            }
        }
        for (uint32_t a = 0; a < na; ++a) {
            const AnimationDef& an = A.anims[A.shadows ? a >> 1 : a];
            const bool read_view = A.shadows && (a & 1u);
            hd[a].tracks = an.td->d_tracks;
            hd[a].key_loc = an.td->d_loc;
            hd[a].key_aux = an.td->d_aux;
            hd[a].key_rec = an.td->d_rec;
            hd[a].hot = an.td->d_hot;
            hd[a].spans = an.td->d_spans;
            hd[a].slot_track = read_view ? an.d_slot_track_f : an.d_slot_track;
            hd[a].prop_track = read_view ? an.d_prop_track_f : an.d_prop_track;
            hd[a].n_tracks = an.td->n_tracks;
            hd[a].rm_node = an.rm_node;
            // (16 / 32: the root node's list holds two or more Vector3 Positions / UnitQuaternion Rotations.  The loop of lib.rs:558-657 walks
            //  them all; what stays of it is the LAST one's RootMotion -- the apply view's value -- computed with the remainders already taken
            //  by the first.  The read view's own RootMotion is never observed.)
            const uint8_t mu = an.rm_node >= 0 && (size_t)an.rm_node < an.multi.size() ? an.multi[(size_t)an.rm_node] : 0;
            hd[a].rm_ignore = an.rm_ignore | (!read_view && (mu & (1u << FYX_BIND_POSITION)) ? 16u : 0u)
                                           | (!read_view && (mu & (1u << FYX_BIND_ROTATION)) ? 32u : 0u);
            hd[a].rm_pos_track = an.rm_pos_track;
            hd[a].rm_rot_track = an.rm_rot_track;
            hd[a].pad = 0;
        }
        dfree(A.d_anims);
        A.d_anims = nullptr;
        if (int rc = upload(c, &A.d_anims, hd.data(), hd.size())) return rc;
        // the crowd sampler's descriptors: what pose_sample_crowd_body reads off slot table, track record and TrackHot, resolved
        std::vector<CrowdDesc> cd((size_t)na * rig.n_nodes * 3);
        for (uint32_t a = 0; a < na; ++a) {
            const AnimationDef& an = A.anims[A.shadows ? a >> 1 : a];
            const std::vector<int32_t>& view = (A.shadows && (a & 1u)) ? an.slots_f : an.slots;
            for (uint32_t node = 0; node < rig.n_nodes; ++node) {
                CrowdDesc* d = &cd[((size_t)a * rig.n_nodes + node) * 3];
                // (16: the node's list holds a value that is no operand of any blend and is never applied -- but the list is not empty)
                uint32_t present = node < an.blockers.size() && an.blockers[node] ? 16u : 0u;
                for (int b = 0; b < 3; ++b) {
                    d[b] = CrowdDesc{nullptr, 0u, 0u, 0u, -1, 0u, 0u};
                    const int32_t t = view.size() == (size_t)rig.n_nodes * 4 ? view[(size_t)node * 4 + b] : -1;
                    if (t < 0 || (size_t)t >= an.td->hot.size()) continue;
                    const TrackHot& th = an.td->hot[t];
                    const uint32_t need = th.kind == FYX_KIND_QUAT ? 4u : (th.kind == FYX_KIND_VEC3 || th.kind == FYX_KIND_QUAT_EULER) ? 3u : 0u;
                    const bool fits = b == FYX_BIND_ROTATION ? (th.kind == FYX_KIND_QUAT || th.kind == FYX_KIND_QUAT_EULER) : th.kind == FYX_KIND_VEC3;
                    d[b].track = (uint32_t)t;
                    d[b].kind = th.kind;
                    d[b].need = need;
                    d[b].valid = fits && need > 0 && th.n_curves >= need;
                    d[b].n_keys = th.n_keys;
                    if (d[b].valid && th.span_first != kNoSpans && an.td->d_spans) {
                        // the per-instance sampler reads one span of every track per frame: the [span][track] copy where the track is in it;
                        // the crowd form stages a track's whole table in LDS: the track's own
                        const bool row = inst_form && an.td->d_span_rows && (size_t)t < an.td->row_first.size() && an.td->row_first[t] != kNoSpans;
                        d[b].spans = row ? an.td->d_span_rows + an.td->row_first[t] : an.td->d_spans + th.span_first;
                        d[b].valid |= (row ? an.td->row_stride : span_stride(need)) << 8;
                    }
                    if (d[b].valid) present |= b == FYX_BIND_POSITION ? 1u : b == FYX_BIND_SCALE ? 2u : 4u;
                }
                if (view.size() == (size_t)rig.n_nodes * 4 && view[(size_t)node * 4 + 3] >= 0) present |= 8u;
                for (int b = 0; b < 3; ++b) d[b].present = present;
            }
        }
        dfree(A.d_crowd);
        A.d_crowd = nullptr;
        if (!cd.empty())
            if (int rc = upload(c, &A.d_crowd, cd.data(), cd.size())) return rc;
        A.anims_dirty = false;
        A.desc_form = inst_form;
    }
    const uint32_t nps = (uint32_t)A.prop_slots.size();
    if (nps && (A.dev_prop_slots != nps || A.dev_prop_anims < A.dev_anim_capacity)) {
        // property storage is re-created when slots or animations are added (values applied so far are kept per slot)
        if (int rc_ = sync_all(c)) return rc_;
        std::vector<int32_t> nodes(nps);
        for (uint32_t k = 0; k < nps; ++k) nodes[k] = A.prop_slots[k].first;
        dfree(A.d_prop_node);
        A.d_prop_node = nullptr;
        if (int rc = upload(c, &A.d_prop_node, nodes.data(), nodes.size())) return rc;
        DevGuard np, no;    // freed again if anything below fails
        const size_t pb = (size_t)A.dev_anim_capacity * A.n_instances * nps * sizeof(PropRec);
        const size_t ob = (size_t)A.n_instances * nps * sizeof(PropRec);
        FYX_HIP(c, hipMalloc(&np.p, std::max<size_t>(pb, 16)));
        FYX_HIP(c, hipMemset(np.p, 0, std::max<size_t>(pb, 16)));
        FYX_HIP(c, hipMalloc(&no.p, std::max<size_t>(ob, 16)));
        FYX_HIP(c, hipMemset(no.p, 0, std::max<size_t>(ob, 16)));
        // Slots only ever get appended (old slot k is new slot k) and animations too, so the old arrays are the top-left
        // corner of the new ones, row by row ((animation, instance) rows of dev_prop_slots records).  Both are kept: what
        // `apply` wrote last, AND every animation's sampled Property values -- in the reference an animation that does
        // not tick keeps its pose, and PlayAnimation nodes go on blending it (pose.rs:107-121).
        if (A.dev_prop_slots) {
            const size_t old_row = (size_t)A.dev_prop_slots * sizeof(PropRec), new_row = (size_t)nps * sizeof(PropRec);
            if (A.d_prop_out)
                FYX_HIP(c, hipMemcpy2D(no.p, new_row, A.d_prop_out, old_row, old_row, A.n_instances, hipMemcpyDeviceToDevice));
            if (A.d_prop_pose && A.dev_prop_anims)
                FYX_HIP(c, hipMemcpy2D(np.p, new_row, A.d_prop_pose, old_row, old_row, (size_t)A.dev_prop_anims * A.n_instances,
                                       hipMemcpyDeviceToDevice));
        }
        dfree(A.d_prop_pose);
        dfree(A.d_prop_out);
        A.d_prop_pose = static_cast<PropRec*>(np.release());
        A.d_prop_out = static_cast<PropRec*>(no.release());
        A.dev_prop_slots = nps;
        A.dev_prop_anims = A.dev_anim_capacity;
    }
    if (A.rm_enabled) {
        if (A.dev_rm_anim_capacity < A.dev_anim_capacity) {  // [anim][instance]: growing keeps the existing prefix
            DevGuard nr;
            const size_t nb = (size_t)A.dev_anim_capacity * A.n_instances * sizeof(RootMotionDev);
            if (int rc_ = sync_all(c)) return rc_;
            FYX_HIP(c, hipMalloc(&nr.p, std::max<size_t>(nb, 16)));
            FYX_HIP(c, hipMemset(nr.p, 0, std::max<size_t>(nb, 16)));
            if (A.d_rm_anim && A.dev_rm_anim_capacity)
                FYX_HIP(c, hipMemcpy(nr.p, A.d_rm_anim, (size_t)A.dev_rm_anim_capacity * A.n_instances * sizeof(RootMotionDev),
                                     hipMemcpyDeviceToDevice));
            dfree(A.d_rm_anim);
            A.d_rm_anim = static_cast<RootMotionDev*>(nr.release());
            A.dev_rm_anim_capacity = A.dev_anim_capacity;
        }
        // Slots: per layer its pose nodes then the layer's final pose; the machine's final pose last.  The records persist
        // from frame to frame (AnimationPose::reset keeps root_motion, pose.rs:125-129), so when the graph changes
        // (nodes / layers appended between frames, or the definition re-sent after an edit: fyx_machine_clear) every
        // record that still has a place keeps its value BY POSITION -- node n of layer l, layer l's final pose, the
        // machine's -- and the new ones start from None.
        uint32_t want = 1;
        std::vector<uint32_t> nodes_now(A.layers.size());
        for (size_t l = 0; l < A.layers.size(); ++l) { nodes_now[l] = (uint32_t)A.layers[l].nodes.size(); want += nodes_now[l] + 1; }
        if (want != A.dev_rm_slots || nodes_now != A.dev_rm_layer_nodes) {
            if (int rc_ = sync_all(c)) return rc_;
            DevGuard ns;
            const size_t nb = (size_t)A.n_instances * want * 32;
            FYX_HIP(c, hipMalloc(&ns.p, nb));
            FYX_HIP(c, hipMemset(ns.p, 0, nb));
            if (A.d_rm_slots && A.dev_rm_slots) {
                const size_t old_pitch = (size_t)A.dev_rm_slots * 32, new_pitch = (size_t)want * 32;
                auto keep = [&](uint32_t new_first, uint32_t old_first, uint32_t count) -> hipError_t {
                    if (!count) return hipSuccess;
                    return hipMemcpy2D(static_cast<char*>(ns.p) + (size_t)new_first * 32, new_pitch,
                                       reinterpret_cast<const char*>(A.d_rm_slots) + (size_t)old_first * 32, old_pitch,
                                       (size_t)count * 32, A.n_instances, hipMemcpyDeviceToDevice);
                };
                uint32_t ob = 0, nbase = 0;
                for (size_t l = 0; l < std::min(nodes_now.size(), A.dev_rm_layer_nodes.size()); ++l) {
                    const uint32_t on = A.dev_rm_layer_nodes[l], nn = nodes_now[l];
                    FYX_HIP(c, keep(nbase, ob, std::min(on, nn)));
                    FYX_HIP(c, keep(nbase + nn, ob + on, 1));      // the layer's final pose
                    ob += on + 1;
                    nbase += nn + 1;
                }
                FYX_HIP(c, keep(want - 1, A.dev_rm_slots - 1, 1));   // the machine's
            }
            dfree(A.d_rm_slots);
            A.d_rm_slots = static_cast<float4*>(ns.release());
            A.dev_rm_slots = want;
            A.dev_rm_layer_nodes = nodes_now;
        }
    }
    if (A.masks_dirty || A.dev_mask_layers != A.layers.size()) {
        if (int rc_ = sync_all(c)) return rc_;
        std::vector<uint8_t> m(std::max<size_t>(A.layers.size() * rig.n_nodes, 1), 0);
        for (size_t l = 0; l < A.layers.size(); ++l)
            for (int32_t n : A.layers[l].excluded)
                if (n >= 0 && (uint32_t)n < rig.n_nodes) m[l * rig.n_nodes + n] = 1;
        dfree(A.d_layer_masks);
        A.d_layer_masks = nullptr;
        if (int rc = upload(c, &A.d_layer_masks, m.data(), m.size())) return rc;
        A.masks_dirty = false;
        A.dev_mask_layers = (uint32_t)A.layers.size();
    }
    return FYX_OK;
}

RigDev rig_dev(const Rig& r) {
    RigDev d;
    memset(&d, 0, sizeof d);      // (unused palette entries and padding too: a scene compares its job array byte for byte)
    d.statics = r.d_statics;
    d.walk = r.d_walk;
    d.inv_bind = r.d_inv_bind;
    d.n_pal = 0;
    d.n_nodes = r.n_nodes;
    d.n_levels = r.n_levels;
    d.n_chunks = r.n_chunks;
    d.pad1 = 0;
    return d;
}

// The persistent part of an animator's kernel parameters.
void frame_static(const fyx_ctx* c, const Animator& A, PoseFrameDev& f) {
    memset(&f, 0, sizeof f);
    f.anims = A.d_anims;
    f.crowd = A.d_crowd;
    f.n_anims = A.n_dev_anims();
    f.shadows = A.shadows ? 1u : 0u;
    f.n_instances = A.n_instances;
    f.n_nodes = A.rig->n_nodes;
    f.layer_masks = A.d_layer_masks;
    f.hints = A.d_hints;
    f.slot_hints = A.d_slot_hints;
    f.max_tracks = A.dev_track_capacity;
    f.sample_form = (uint32_t)c->sample_form;
    f.anim_pose = A.d_anim_pose;
    f.node_trs = A.d_node_trs;
    f.local = A.d_local;
    f.global = A.d_global;
    f.n_prop_slots = A.dev_prop_slots;
    f.prop_node = A.d_prop_node;
    f.prop_pose = A.d_prop_pose;
    f.prop_out = A.d_prop_out;
}

// (sections 16-byte aligned: uint4 reads of the root-motion ops, whole uint4 copies; a scene of 256 characters is 256 such blocks)
constexpr size_t kCtrlAlign = 16;
// The frame's control sections as the kernels index them: per DEVICE animation.  With two device animations per animation
// (Animator::shadows) the planner's arrays are spread out -- clocks, tick flags and time slices once for each of the pair, and the
// programs' animation operands doubled (2 a: the apply view; the two-record folds read 2 a + 1 themselves).
void expand_ctrl(Animator& A) {
    if (!A.shadows) return;
    const size_t n = A.times.size();
    A.x_times.resize(2 * n);
    A.x_ticked.resize(2 * n);
    for (size_t k = 0; k < n; ++k) {
        A.x_times[2 * k] = A.x_times[2 * k + 1] = A.times[k];
        A.x_ticked[2 * k] = A.x_ticked[2 * k + 1] = A.ticked[k];
    }
    A.x_slices.resize(2 * A.slices.size());
    for (size_t k = 0; k < A.slices.size(); ++k) A.x_slices[2 * k] = A.x_slices[2 * k + 1] = A.slices[k];
    A.x_ops = A.ops;
    for (uint2& op : A.x_ops) {
        const uint32_t code = op.x & 0xffu;
        if (code == OP_BLEND_ANIM || code == OP_APPLY_ANIM) op.x = code | ((op.x >> 8) * 2u) << 8;
    }
    A.x_rm_ops = A.rm_ops;
    for (uint4& op : A.x_rm_ops)
        if (op.x == RM_SET_ANIM) op.z *= 2u;
}
inline const std::vector<float>& ctrl_times(const Animator& A) { return A.shadows ? A.x_times : A.times; }
inline const std::vector<uint8_t>& ctrl_ticked(const Animator& A) { return A.shadows ? A.x_ticked : A.ticked; }
inline const std::vector<float2>& ctrl_slices(const Animator& A) { return A.shadows ? A.x_slices : A.slices; }
inline const std::vector<uint2>& ctrl_ops(const Animator& A) { return A.shadows ? A.x_ops : A.ops; }
inline const std::vector<uint4>& ctrl_rm_ops(const Animator& A) { return A.shadows ? A.x_rm_ops : A.rm_ops; }

CtrlLayout ctrl_layout(const Animator& A) {
    CtrlLayout L;
    L.rm = A.rm_enabled;
    L.o_tick = align_up(ctrl_times(A).size() * 4, kCtrlAlign);
    L.o_off = L.o_tick + align_up(ctrl_ticked(A).size(), kCtrlAlign);
    L.o_ops = L.o_off + align_up(A.prog_off.size() * 4, kCtrlAlign);
    L.o_slices = L.o_ops + align_up(ctrl_ops(A).size() * 8, kCtrlAlign);
    L.o_rmoff = L.o_slices + (L.rm ? align_up(ctrl_slices(A).size() * 8, kCtrlAlign) : 0);
    L.o_rmops = L.o_rmoff + (L.rm ? align_up(A.rm_prog_off.size() * 4, kCtrlAlign) : 0);
    L.total = L.o_rmops + (L.rm ? align_up(ctrl_rm_ops(A).size() * 16, kCtrlAlign) : 0);
    return L;
}

// The same sections 16-byte aligned and behind the 16-byte header of a CtrlInline: what travels in the kernel arguments.
CtrlLayout ctrl_layout_inline(const Animator& A) {
    CtrlLayout L;
    L.rm = A.rm_enabled;
    L.o_times = 16;
    L.o_tick = L.o_times + align_up(ctrl_times(A).size() * 4, 16);
    L.o_off = L.o_tick + align_up(ctrl_ticked(A).size(), 16);
    L.o_ops = L.o_off + align_up(A.prog_off.size() * 4, 16);
    L.o_slices = L.o_ops + align_up(ctrl_ops(A).size() * 8, 16);
    L.o_rmoff = L.o_slices + (L.rm ? align_up(ctrl_slices(A).size() * 8, 16) : 0);
    L.o_rmops = L.o_rmoff + (L.rm ? align_up(A.rm_prog_off.size() * 4, 16) : 0);
    L.total = L.o_rmops + (L.rm ? align_up(ctrl_rm_ops(A).size() * 16, 16) : 0);
    return L;
}

// What changes from one steady frame to the next (anim_planner.h, steady_frame): the clocks and the tick flags.
void ctrl_write_clocks(const Animator& A, const CtrlLayout& L, char* h) {
    memcpy(h + L.o_times, ctrl_times(A).data(), ctrl_times(A).size() * 4);
    memcpy(h + L.o_tick, ctrl_ticked(A).data(), ctrl_ticked(A).size());
}

void ctrl_write(const Animator& A, const CtrlLayout& L, char* h) {
    memcpy(h + L.o_times, ctrl_times(A).data(), ctrl_times(A).size() * 4);
    memcpy(h + L.o_tick, ctrl_ticked(A).data(), ctrl_ticked(A).size());
    memcpy(h + L.o_off, A.prog_off.data(), A.prog_off.size() * 4);
    memcpy(h + L.o_ops, ctrl_ops(A).data(), ctrl_ops(A).size() * 8);
    if (L.rm) {
        memcpy(h + L.o_slices, ctrl_slices(A).data(), ctrl_slices(A).size() * 8);
        memcpy(h + L.o_rmoff, A.rm_prog_off.data(), A.rm_prog_off.size() * 4);
        memcpy(h + L.o_rmops, ctrl_rm_ops(A).data(), ctrl_rm_ops(A).size() * 16);
    }
}

// Point the frame's parameters at the device copy of the control block.
void ctrl_bind(const Animator& A, const CtrlLayout& L, const char* d, PoseFrameDev& f) {
    f.times = reinterpret_cast<const float*>(d + L.o_times);
    f.ticked = reinterpret_cast<const uint8_t*>(d + L.o_tick);
    f.prog_off = reinterpret_cast<const uint32_t*>(d + L.o_off);
    f.ops = reinterpret_cast<const uint2*>(d + L.o_ops);
    if (L.rm) {
        f.slices = reinterpret_cast<const float2*>(d + L.o_slices);
        f.rm_anim = A.d_rm_anim;
        f.rm_slots = A.d_rm_slots;
        f.n_rm_slots = A.dev_rm_slots;
        f.rm_prog_off = reinterpret_cast<const uint32_t*>(d + L.o_rmoff);
        f.rm_ops = reinterpret_cast<const uint4*>(d + L.o_rmops);
    }
}

// The buffer a palette output's CURRENT frame writes (and its skinning reads): the pair's second buffer for the frames of the second
// frame stream (anim.overlap; enter_pose has chosen the frame's stream before anything asks).
inline float* palette_of(const fyx_ctx* c, const Animator::PaletteOut& po) {
    return (po.d_out_alt && c->pose_overlap && c->frame_idx) ? po.d_out_alt : po.d_out;
}
// ... and the buffer the animator's most recent frame wrote, whatever frames of OTHER animators have started since
// (fyx_animator_current_palette).
inline float* palette_last_written(const Animator& A, const Animator::PaletteOut& po) {
    return (po.d_out_alt && A.last_frame_alt) ? po.d_out_alt : po.d_out;
}

// The rig's parameters plus the palettes the update kernel writes itself.
int rig_params(fyx_ctx* c, const Animator& A, RigDev& rd) {
    rd = rig_dev(*A.rig);
    for (const Animator::PaletteOut& po : A.palette_outputs) {      // (fyx_bone_list_free refuses a list that is registered here)
        PaletteOutDev& d = rd.pal[rd.n_pal++];
        d.bone_nodes = po.d_bone_nodes;
        d.out = palette_of(c, po);
        d.n_bones = po.n_bones;
        d.pad = 0;
    }
    return FYX_OK;
}

// The skinning launches of the animator's skin outputs (fyx_animator_set_skin_output): mesh and palette looked up now -- a mesh may
// have been uploaded again, a palette output moved -- and validated as fyx_lbs_skin_device validates its arguments.
int skin_output_args(fyx_ctx* c, const Animator& A, LbsArgs (&out)[kMaxFrameSkins]) {
    uint32_t k = 0;
    for (const Animator::SkinOut& so : A.skin_outputs) {
        const Animator::PaletteOut* po = nullptr;
        for (const Animator::PaletteOut& p : A.palette_outputs)
            if (p.bones_id == so.bones_id) po = &p;
        if (!po) return fail(c, FYX_ERR_INVALID_ARG, "skin output of mesh %llu: bone list %llu is no longer a palette output of the animator",
                             (unsigned long long)so.mesh_id, (unsigned long long)so.bones_id);
        if (int rc = skin_args_of(c, so.mesh_id, palette_of(c, *po), po->n_bones, A.n_instances, so.d_pos, so.d_nrm, so.d_tan, &out[k])) return rc;
        ++k;
    }
    return FYX_OK;
}

// Whether (and how) the one-launch frame takes the skinning along: every job's bone list, its mesh streams and its share of the
// launch's skinning workgroups.  false: the frame skins with launches of its own (a mesh too large for the resident grid).
bool frame_skin_plan(const fyx_ctx* c, const Animator& A, const LbsArgs* args, uint32_t n, uint32_t max_blocks, FrameSkin& sk, uint32_t auto_blocks = kFrameSkinAutoBlocks) {
    memset(&sk, 0, sizeof sk);
    if (n == 0 || n > (uint32_t)kMaxFrameSkins) return false;
    // Units per wave.  One is fastest while every skinning workgroup has a CU to itself (a lone wave issues an instruction every
    // ~4 cycles whatever it is: a unit costs it ~0.7 us, and the units of a wave come one after another); past ~one workgroup per CU
    // two workgroups share a CU's LDS pipeline and scheduler and the launch ends with the slowest: measured (tools/exp/r05_frame_skin.py)
    // C2, 782 units: 196 workgroups 8.9 us, 98: 9.2, 49: 10.5; C5, 1563 units: 391 workgroups 12.0 us, 196: 10.9, 98: 12.4.
    // anim.frame_skin_units = 0 (default): the smallest depth that keeps the launch's skinning workgroups within kFrameSkinAutoBlocks.
    auto blocks_at = [&](uint32_t per_wave) {
        uint64_t t = 0;
        for (uint32_t k = 0; k < n; ++k)
            t += (uint64_t)A.n_instances * std::max<uint32_t>(((args[k].n_verts + 63u) / 64u + 4u * per_wave - 1u) / (4u * per_wave), 1u);
        return t;
    };
    uint32_t per_wave = (uint32_t)std::max(c->frame_skin_units, 0);
    if (per_wave == 0)
        for (per_wave = 1; per_wave < 16u && blocks_at(per_wave) > std::min(auto_blocks, max_blocks); ++per_wave) {}
    const uint64_t want = blocks_at(per_wave), least = (uint64_t)n * A.n_instances;
    if (least > max_blocks) return false;
    // more work than the resident grid holds at the wanted depth: every job gets its share of the grid, waves loop over more units.
    // (Past ~16 units per wave the launch is a streaming kernel and the launch boundary it saves no longer matters: separate launches.)
    const double scale = want > max_blocks ? (double)max_blocks / (double)want : 1.0;
    uint32_t block0 = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const LbsArgs& a = args[k];
        const uint32_t upi = (a.n_verts + 63u) / 64u;
        uint32_t bpi = std::max<uint32_t>((upi + 4u * per_wave - 1u) / (4u * per_wave), 1u);
        bpi = std::max<uint32_t>((uint32_t)((double)bpi * scale), 1u);
        if (upi > (uint64_t)bpi * 4u * 16u) return false;
        const Animator::PaletteOut* po = nullptr;
        for (const Animator::PaletteOut& p : A.palette_outputs)
            if (p.bones_id == A.skin_outputs[k].bones_id) po = &p;
        if (!po) return false;
        FrameSkinJob& j = sk.job[k];
        j.pos = a.pos; j.nrm = a.out_nrm ? a.nrm : nullptr; j.tan = a.out_tan ? a.tan : nullptr; j.wgt = a.wgt; j.idx = a.idx;
        j.out_pos = a.out_pos; j.out_nrm = a.nrm ? a.out_nrm : nullptr; j.out_tan = a.tan ? a.out_tan : nullptr;
        j.bone_nodes = po->d_bone_nodes;
        j.n_verts = a.n_verts; j.n_bones = a.n_bones;
        j.block0 = block0;
        j.blocks_per_inst = bpi;
        block0 += bpi * A.n_instances;
    }
    if (block0 > max_blocks) return false;
    sk.n_jobs = n;
    sk.n_blocks = block0;
    return true;
}

// Send the planned frame to the GPU and run sample + update.
int run_frame(fyx_ctx* c, Animator& A, bool with_program) {
    hipStream_t ps = nullptr;     // the frame's stream (anim.overlap: frames alternate between two)
    if (int rc = enter_pose(c, &ps)) return rc;
    if (c->pose_overlap == 2) {      // a palette output with ONE buffer: this frame rewrites what the previous frame's skinning reads
        bool single = false;
        for (const Animator::PaletteOut& po : A.palette_outputs) single = single || !po.d_out_alt;
        if (single)
            if (int rc = pose_behind_all_skinning(c, ps)) return rc;
    }
    if (int rc = ensure_device_state(c, A)) return rc;
    expand_ctrl(A);      // (two device animations per animation: the control sections per device animation)
    A.last_frame_alt = (c->pose_overlap && c->frame_idx) ? 1 : 0;
    A.last_frame_kind = with_program ? 1 : 2;
    PoseFrameDev f;
    frame_static(c, A, f);
    int slot = 0;
    bool in_args = false, one_launch = false;
    CtrlInline inl;
    inl.bytes = 0;
    inl.first_ops = 0;
    inl.pad[0] = inl.pad[1] = 0;
    if (with_program) {
        // a small control block (one character, a handful of instances) rides in the kernel arguments: no copy, no event, no wait
        const CtrlLayout Li = ctrl_layout_inline(A);
        // (the kernels that read it there walk the hierarchy wide: a deep rig of ~1000 nodes, whose chunk table no longer fits
        // the LDS beside its matrices, takes the uploaded block and the narrow kernels)
        // (... and so does an animator whose folds keep two records per animation: kUpdDup is the plain launch's)
        // (+ 16: the one-launch frame's wait status word lies behind the walk's areas)
        in_args = c->inline_ctrl && !A.shadows && Li.total <= sizeof(CtrlInline) && wide_update_lds(A.rig->n_nodes, A.rig->n_chunks) + 16u <= kLdsPerWorkgroup;
        const CtrlLayout L = in_args ? Li : ctrl_layout(A);
        if (in_args) {
            ctrl_write(A, L, reinterpret_cast<char*>(&inl));    // the sections start behind the header (offset 16)
            inl.bytes = (uint32_t)L.total;
            inl.first_ops = A.prog_off.size() >= 2 && A.prog_off[0] == 0 ? A.prog_off[1] + 1u : 0u;
            ctrl_bind(A, L, nullptr, f);                        // the control pointers become offsets from the start of `inl`
        } else {
            char *h = nullptr, *d = nullptr;
            if (int rc = ctrl_acquire(c, A.ctrl, L.total, &slot, &h, &d)) return rc;
            ctrl_write(A, L, h);
            if (int rc = ctrl_upload(c, A.ctrl, slot, L.total, ps)) return rc;
            ctrl_bind(A, L, d, f);
        }
        // one character (a few instances) without root motion or property tracks: sampler and update are ONE launch
        one_launch = c->one_launch && in_args && !L.rm && f.n_prop_slots == 0 && f.n_anims > 0 &&
                     (f.sample_form == 1 || (f.sample_form == 0 && f.n_instances < 32));
        if (!one_launch) {
            if (int rc = timeline_arm(c, 1)) return rc;
            FYX_HIP(c, launch_pose_sample(f, ps, &inl));
            g_launch_events = LaunchEvents();
            FYX_HIP(c, launch_property_sample(f, ps, &inl));
            if (L.rm) FYX_HIP(c, launch_root_motion(f, !A.rm_ops.empty(), ps, &inl));
        }
    }
    RigDev rd;
    if (int rc = rig_params(c, A, rd)) return rc;
    const int upd_mode = !with_program ? kUpdNoProgram : A.shadows ? kUpdDup : (A.all_straight && c->upd_lean) ? kUpdStraight : kUpdGeneral;
    // the animator's skin outputs: the kernel arguments fyx_lbs_skin_device would launch with, on the palettes this frame writes
    LbsArgs skin_args[kMaxFrameSkins];
    const uint32_t n_skins = (uint32_t)A.skin_outputs.size();
    if (int rc = skin_output_args(c, A, skin_args)) return rc;
    if (int rc = timeline_arm(c, 2)) return rc;
    bool skinned = false;
    if (one_launch) {
        if (!A.d_frame_counter) {      // the counter of the animator's one-launch frames (FrameSync), on its first such frame
            const size_t cb = (size_t)kFrameCounterReplicas * kFrameCounterStride;
            FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&A.d_frame_counter), cb));
            A.frame_counter_total = 0;
            FYX_HIP(c, hipMemsetAsync(A.d_frame_counter, 0, cb, ps));
        }
        FrameSync wait;
        memset(&wait, 0, sizeof wait);
        wait.timeout_ticks = (uint32_t)c->wait_timeout_ms * 100000u;     // 100 MHz
        wait.err = reinterpret_cast<uint32_t*>(c->dev_err);
        wait.tag = A.id;
        FrameSkin sk;
        bool fused = n_skins && c->frame_skin && frame_skin_plan(c, A, skin_args, n_skins, upd_mode == kUpdStraight ? kFrameSkinMaxBlocks : kFrameSkinMaxBlocksGeneral, sk);
        if (fused) {      // the skinning workgroups' palette lies behind the walk's LDS areas: a deep rig whose walk nearly fills the LDS skins with launches of its own
            uint32_t max_bones = 0;
            for (uint32_t k = 0; k < sk.n_jobs; ++k) max_bones = std::max(max_bones, sk.job[k].n_bones);
            fused = wide_update_lds(A.rig->n_nodes, A.rig->n_chunks) + frame_skin_lds(max_bones) <= kLdsPerWorkgroup;
        }
        // (registered skin outputs are the same vertex buffers every frame: a launch that writes them lies behind the other frame stream's
        // skinning of them -- for a pose launch that skins, the whole launch)
        if (fused)
            if (int rc = skin_outputs_order(c, ps, true)) return rc;
        FYX_HIP(c, launch_pose_frame(f, rd, upd_mode, ps, inl, A.d_frame_counter, &A.frame_counter_total, wait, fused ? &sk : nullptr, c->lbs.exact != 0));
        skinned = fused;
    } else {
        FYX_HIP(c, launch_pose_update(f, rd, upd_mode, ps, &inl, c->upd_pack));
    }
    g_launch_events = LaunchEvents();
    if (with_program) {
        FYX_HIP(c, launch_property_update(f, ps, &inl));
        if (!in_args)
            if (int rc = ctrl_consumed(c, A.ctrl, slot, ps)) return rc;
    }
    if (int rc = exit_pose(c)) return rc;
    // a frame that could not take its skinning along (a crowd, root motion, property tracks, anim.frame_skin = 0): the same launches
    // fyx_lbs_skin_device would make, in order behind the update on the frame's stream
    if (!skinned && n_skins) {
        hipStream_t ss = nullptr;      // the stream of the frame's skinning (anim.overlap = 2: the second stream; else the frame's own)
        if (int rc = enter_skin(c, &ss)) return rc;
        if (int rc = skin_outputs_order(c, ss, false)) return rc;
        for (uint32_t k = 0; k < n_skins; ++k) FYX_HIP(c, launch_lbs(skin_args[k], c->lbs, ss));
        if (int rc = skin_outputs_issued(c, ss)) return rc;
    }
    return FYX_OK;
}

// One frame of MANY animators (fyx_scene_update): every animator is planned exactly as plan_frame does (different
// animators on different host threads), the control blocks travel in ONE upload, and each stage of the frame is ONE
// kernel launch over all of them.  Results are those of run_frame on each animator in turn: the animators share no
// device state, and the kernels' bodies are the same functions.
// Host control plane of a scene frame.  Crowds big enough to be split go first, one after another, each over the
// whole pool; the rest are dealt out to the pool in contiguous runs of about equal instance counts.
int scene_plan(fyx_ctx* c, SceneBatch& S, float dt) {
    const size_t n = S.animators.size();
    S.errors.assign(n, 0);
    std::vector<size_t> small;
    uint64_t small_instances = 0;
    for (size_t k = 0; k < n; ++k) {
        Animator& A = *S.animators[k];
        const unsigned nt = plan_tasks(c, A);
        if (nt > 1) S.errors[k] = plan_frame_core(A, A.layers.empty() ? 0 : 1, dt, nt, plan_pool(c, nt));
        else { small.push_back(k); small_instances += A.n_instances; }
    }
    unsigned n_tasks = 1;
    // Waking the pool costs tens of microseconds on the calling thread's clock (more on hosts with a CPU quota); an animator
    // costs ~0.15 us plus ~0.07 us per instance to plan, so the pool only pays for scenes whose serial planning takes a few
    // hundred microseconds (measured, 256 single-instance characters: 47 us of planning; frame pose path 0.080 ms on the
    // calling thread alone, 0.126 ms with eight planner threads).
    const uint64_t work = small_instances + 2 * (uint64_t)small.size();   // in units of one instance
    if (c->plan_threads > 1 && small.size() >= 32 && work >= 2 * (uint64_t)std::max(c->plan_split, 1))
        n_tasks = std::min<unsigned>((unsigned)c->plan_threads, (unsigned)(small.size() / 16));
    if (n_tasks > 1) {
        std::vector<size_t> cut(n_tasks + 1, small.size());   // task t plans small[cut[t] .. cut[t + 1])
        cut[0] = 0;
        uint64_t acc = 0;
        unsigned t = 1;
        for (size_t j = 0; j < small.size() && t < n_tasks; ++j) {
            acc += S.animators[small[j]]->n_instances;
            if (acc * n_tasks >= small_instances * t) cut[t++] = j + 1;
        }
        plan_pool(c, n_tasks)->run(n_tasks, [&](unsigned task) {
            for (size_t j = cut[task]; j < cut[task + 1]; ++j) {
                Animator& A = *S.animators[small[j]];
                S.errors[small[j]] = plan_frame_core(A, A.layers.empty() ? 0 : 1, dt, 1, nullptr);
            }
        });
    } else {
        for (size_t k : small) {
            Animator& A = *S.animators[k];
            S.errors[k] = plan_frame_core(A, A.layers.empty() ? 0 : 1, dt, 1, nullptr);
        }
    }
    for (size_t k = 0; k < n; ++k)
        if (S.errors[k]) return fail(c, S.errors[k], "animator %zu of the scene: pose nodes nest deeper than %d blend levels", k, kMaxFoldDepth - 2);
    return FYX_OK;
}

SceneJobShape scene_shape(const fyx_ctx* c, const Animator& A, uint32_t n_prop_slots) {
    SceneJobShape sh;
    sh.n_anims = (uint32_t)A.anims.size();
    sh.n_instances = A.n_instances;
    sh.n_nodes = A.rig->n_nodes;
    sh.n_prop_slots = n_prop_slots;
    sh.sample_form = (uint32_t)c->sample_form;
    sh.root_motion = sh.root_motion_program = A.rm_enabled;
    return sh;
}

// replay: the device side of the frame planned last, again (reissue_frame): the host control plane has run, its results are in the animators
int scene_frame(fyx_ctx* c, SceneBatch& S, float dt, bool replay = false) {
    const size_t n = S.animators.size();
    // (option debug.host_times: the sections' cost to the calling thread, fyx_debug_host_times)
    using HostClock = std::chrono::steady_clock;
    HostClock::time_point ht0;
    if (c->host_times_on) ht0 = HostClock::now();
    auto host_section = [&](int k) {
        if (!c->host_times_on) return;
        const HostClock::time_point t = HostClock::now();
        c->host_times[k] += std::chrono::duration<double, std::micro>(t - ht0).count();
        ht0 = t;
    };
    // 1. host control plane
    if (!replay)
        if (int rc = scene_plan(c, S, dt)) return rc;
    host_section(0);
    // a scene of ONE animator is that animator's own frame: the control block in the kernel arguments, sampler + update (+ skinning) in
    // one launch where the animator qualifies -- 8.6 us for a character where the scene's stages (copy kernel, sampler, update) take ~19
    if (n == 1) return run_frame(c, *S.animators[0], true);
    // An animator whose machine folds two records per animation (AnimationDef::dup) has no scene form: such a scene runs its members one
    // by one (same results: the scene's launches ARE the members' frames side by side).
    {
        bool one_by_one = false;
        for (size_t k = 0; k < n && !one_by_one; ++k) {
            const Animator& A = *S.animators[k];
            if (A.shadows) one_by_one = true;
            if (!A.layers.empty())
                for (const AnimationDef& an : A.anims) one_by_one = one_by_one || (an.maybe_dup && !an.removed);
        }
        if (one_by_one) {
            for (size_t k = 0; k < n; ++k) {
                if (int rc = run_frame(c, *S.animators[k], true)) return rc;
                S.animators[k]->last_frame_kind = 1;
            }
            return FYX_OK;
        }
    }

    // 2. device state, and the block tables if the scene's shape changed
    hipStream_t ps = nullptr;
    if (int rc = enter_pose(c, &ps)) return rc;
    const int par = c->pose_overlap ? c->frame_idx : 0;      // the frame's stream: its own resident job array
    for (size_t k = 0; k < n; ++k) { S.animators[k]->last_frame_alt = par; S.animators[k]->last_frame_kind = 3; }
    // Has anything changed that the launch plans, the job array or the control block's layout are made from (SceneBatch::static_gen)?
    bool unchanged = S.static_gen != 0 && S.seen.size() == n && S.seen_members_epoch == S.members_epoch && S.seen_options_gen == c->options_gen &&
                     S.seen_mesh_gen == c->mesh_gen;
    for (size_t k = 0; k < n && unchanged; ++k) unchanged = S.seen[k].api_gen == S.animators[k]->api_gen;
    bool fast = unchanged && S.fast_eligible && S.jobs_gen[par] == S.static_gen;
    // programs planned again since (a transition, a parameter): the section must still lie where the job array says and fit its place
    for (size_t k = 0; k < n && fast; ++k) {
        const Animator& A = *S.animators[k];
        if (S.seen[k].prog_gen == A.prog_gen) continue;
        const CtrlLayout L = ctrl_layout(A);
        const CtrlLayout& O = S.layouts[k];
        if (L.rm || L.o_times != O.o_times || L.o_tick != O.o_tick || L.o_off != O.o_off || L.o_ops != O.o_ops || L.total > S.caps[k]) {
            fast = unchanged = false;      // (laid out again below: a new state)
            break;
        }
        S.layouts[k] = L;
        S.seen[k].prog_gen = A.prog_gen;
    }
    SceneJobDev* jobs = nullptr;
    if (!fast) {
    // The animators' skin outputs ride in the scene's update launch when that is the 256-thread wide-walk stage for EVERY animator
    // (characters: few instances each, rigs whose walk tables fit the LDS) -- each animator's plan is cached until an API call, a
    // mesh upload or the options change it.  Share of the launch's skinning workgroups: what keeps the whole stage about resident.
    // Only SMALL scenes: the skinning workgroups recompute their character's pose, which is free on a chip that is mostly idle (one
    // character: 12.7 -> 8.9 us) and is not on one that is full -- measured (tools/exp/r05_scene.py, profiles/r05_scene_skin_outputs.jsonl):
    // 256 characters x 5 k vertices 64.6 us this way against 60.3 us with the update launch + fyx_lbs_skin_batch's launch, 64 x 4 x 20 k
    // 126 against 118.  anim.frame_skin = 2 forces it whatever the size (experiments).
    bool any_skin = false, stage256 = c->frame_skin != 0;
    size_t wide_all = 0;
    uint32_t max_skin_bones = 0;
    uint64_t skin_units = 0;
    for (size_t k = 0; k < n; ++k) {
        const Animator& A = *S.animators[k];
        any_skin = any_skin || !A.skin_outputs.empty();
        stage256 = stage256 && update_block_waves(A.rig->n_nodes, A.n_instances) == 4u;
        wide_all = std::max(wide_all, wide_update_lds(A.rig->n_nodes, A.rig->n_chunks));
        for (const Animator::SkinOut& so : A.skin_outputs) {
            auto mit = c->meshes.find(so.mesh_id);
            if (mit != c->meshes.end()) skin_units += (uint64_t)A.n_instances * ((mit->second.n_verts + 63u) / 64u);
        }
    }
    // The scene as ONE launch (scene_frame_kernel, kStageFrame): every animator a character of the per-instance sampler form without root
    // motion or property tracks -- sampler, update and skinning workgroups of all of them in one grid, job after job.  Built, tested
    // (2000 frames against the stage-by-stage form, a poisoned counter) and MEASURED SLOWER (profiles/r05_scene_one_launch.jsonl: 256 x 5 k
    // 73.0 us against 60.3, 64 x 4 x 20 k 138.6 against 107.8, 32 x 5 k 27.4 against 26.3 / 21.9 with the update stage skinning): the
    // samplers inherit the fused kernel's 157 registers (three waves per SIMD instead of eight) and waiting workgroups hold places.  It runs
    // only when asked for: anim.frame_skin = 3.
    bool one_frame = c->one_launch != 0 && c->frame_skin != 0 && any_skin && stage256 && wide_all + frame_skin_lds(256) <= kLdsPerWorkgroup;
    for (size_t k = 0; k < n && one_frame; ++k) {
        const Animator& A = *S.animators[k];
        one_frame = !A.rm_enabled && A.prop_slots.empty() && !A.anims.empty() && (c->sample_form == 1 || (c->sample_form == 0 && A.n_instances < 32));
    }
    one_frame = one_frame && c->frame_skin == 3;
    const bool skin_update = (any_skin && stage256 && wide_all + frame_skin_lds(256) <= kLdsPerWorkgroup &&
                              (skin_units <= kSceneSkinMaxUnits || c->frame_skin == 2) && c->frame_skin != 3) || one_frame;
    std::vector<uint64_t> sig;
    sig.reserve(n * 4 + 1);
    sig.push_back((uint64_t)c->sample_form | (skin_update ? 16u : 0u) | (one_frame ? 32u : 0u));
    for (size_t k = 0; k < n; ++k) {
        Animator& A = *S.animators[k];
        if (int rc = ensure_device_state(c, A)) return rc;
        uint32_t skin_blocks = 0;
        if (skin_update && !A.skin_outputs.empty()) {
            if (A.scene_skin_api_gen != A.api_gen || A.scene_skin_mesh_gen != c->mesh_gen || A.scene_skin_units != c->frame_skin_units + (one_frame ? 1000 : 0)) {
                LbsArgs args[kMaxFrameSkins];
                if (int rc = skin_output_args(c, A, args)) return rc;
                const uint32_t share = std::max<uint32_t>(1u, (one_frame ? 640u : 512u) / (uint32_t)n);
                A.scene_skin_ok = frame_skin_plan(c, A, args, (uint32_t)A.skin_outputs.size(), kFrameSkinMaxBlocks, A.scene_skin, share);
                if (!A.scene_skin_ok) memset(&A.scene_skin, 0, sizeof A.scene_skin);
                A.scene_skin_api_gen = A.api_gen;
                A.scene_skin_mesh_gen = c->mesh_gen;
                A.scene_skin_units = c->frame_skin_units + (one_frame ? 1000 : 0);
            }
            if (A.scene_skin_ok) {
                skin_blocks = A.scene_skin.n_blocks;
                for (uint32_t j = 0; j < A.scene_skin.n_jobs; ++j) max_skin_bones = std::max(max_skin_bones, A.scene_skin.job[j].n_bones);
            }
        }
        sig.push_back(((uint64_t)A.anims.size() << 32) | A.n_instances);
        sig.push_back(((uint64_t)A.rig->n_nodes << 32) | A.dev_prop_slots);
        sig.push_back(((uint64_t)A.rig->n_chunks << 1) | (A.rm_enabled ? 1 : 0));
        sig.push_back(((uint64_t)skin_blocks << 32) | max_skin_bones);
    }
    if (sig != S.signature) {
        std::vector<uint4> tables[kSceneStages];
        size_t lds[kSceneStages] = {};
        size_t wide256 = 0;
        for (size_t k = 0; k < n; ++k) {
            const Animator& A = *S.animators[k];
            SceneJobShape sh = scene_shape(c, A, A.dev_prop_slots);
            if (skin_update && A.scene_skin_ok && !A.skin_outputs.empty()) sh.skin_blocks = A.scene_skin.n_blocks;
            if (one_frame) {      // the whole scene in one launch: this job's samplers, then its update workgroups, then its skinning workgroups
                const uint32_t nsb = ((sh.n_nodes * 16u + 255u) / 256u) * sh.n_instances * sh.n_anims;
                for (uint32_t i = 0; i < nsb; ++i) tables[kStageFrame].push_back(make_uint4((uint32_t)k, i, 0, 0));
                for (uint32_t i = 0; i < sh.n_instances; ++i) tables[kStageFrame].push_back(make_uint4((uint32_t)k, i, 1, 0));
                for (uint32_t i = 0; i < sh.skin_blocks; ++i) tables[kStageFrame].push_back(make_uint4((uint32_t)k, i, 2, 0));
                continue;
            }
            scene_blocks((uint32_t)k, sh, tables);
            const int stage = kStageUpdate64 + (int)update_block_waves(sh.n_nodes, sh.n_instances) - 1;
            lds[stage] = std::max(lds[stage], (size_t)sh.n_nodes * 32 * sizeof(float));
            if (stage == kStageUpdate256) wide256 = std::max(wide256, wide_update_lds(sh.n_nodes, A.rig->n_chunks));
        }
        // the 256-thread updates walk the hierarchy wide (a lane per matrix element) when every rig's chunk table fits the LDS
        S.wide_update = wide256 > 0 && wide256 <= kLdsPerWorkgroup;
        if (S.wide_update) lds[kStageUpdate256] = wide256;
        S.skin_update = skin_update && (S.wide_update || one_frame) && max_skin_bones > 0;
        if (S.skin_update) lds[kStageUpdate256] = wide256 + frame_skin_lds(max_skin_bones);
        S.one_frame = one_frame && S.skin_update;
        if (one_frame) lds[kStageFrame] = wide_all + frame_skin_lds(std::max(max_skin_bones, 1u));
        size_t total = 0;
        for (int k = 0; k < kSceneStages; ++k) {
            if (tables[k].size() > 0x7fffffffull) return fail(c, FYX_ERR_UNSUPPORTED, "scene too large for one launch per stage");
            S.table_off[k] = total;
            S.n_blocks[k] = (uint32_t)tables[k].size();
            S.lds_bytes[k] = lds[k];
            total += tables[k].size();
        }
        if (int rc_ = sync_all(c)) return rc_;   // the previous scene's launches still read the old tables
        dfree(S.d_tables);
        S.d_tables = nullptr;
        S.signature.clear();
        FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&S.d_tables), std::max<size_t>(total, 1) * sizeof(uint4)));
        for (int k = 0; k < kSceneStages; ++k)
            if (!tables[k].empty())
                FYX_HIP(c, hipMemcpy(S.d_tables + S.table_off[k], tables[k].data(), tables[k].size() * sizeof(uint4), hipMemcpyHostToDevice));
        S.signature = sig;
    }

    // 3. one control block: every animator's sections.  The job array (352 bytes per animator, pointers into the animators' device
    // state and OFFSETS into this block) stays on the device and travels only when it changed: 256 one-instance characters upload
    // ~30 KB per frame instead of ~350 KB, which is what their copy took 11 us for.
    S.layouts.resize(n);
    S.offsets.resize(n);
    size_t total = 0;
    if (!unchanged || S.caps.size() != n) S.caps.assign(n, 0);
    for (size_t k = 0; k < n; ++k) {
        S.layouts[k] = ctrl_layout(*S.animators[k]);
        S.offsets[k] = total;
        // the section's place: its size and a quarter as much again (a transition's program is longer than a state's), KEPT while the state
        // lasts (the block travels whole every frame: 256 characters 30 KB of sections, 4.1 us of copy kernel; with half as much again 4.8).
        // While this path runs for another reason (the other frame stream's first frame of a state) a place moves only when its section
        // has outgrown it -- and that is a new state: the other stream's resident job array holds the old offsets.
        if (!(unchanged && S.layouts[k].total <= S.caps[k])) {
            unchanged = false;
            S.caps[k] = std::max(S.caps[k], align_up(S.layouts[k].total + std::max<size_t>(S.layouts[k].total / 4, 64), 16));
        }
        total += S.caps[k];
    }
    // (the one-launch frame: every job's counter target of THIS frame, behind the animators' sections)
    S.o_targets = align_up(total, 16);
    if (S.one_frame) total = S.o_targets + align_up(n * 4, 16);
    S.ctrl_total = total;
    S.h_jobs.resize(n * sizeof(SceneJobDev));      // (every byte of a job is written below: frame_static and rig_dev start from zeros)
    jobs = reinterpret_cast<SceneJobDev*>(S.h_jobs.data());
    for (size_t k = 0; k < n; ++k) {
        const Animator& A = *S.animators[k];
        frame_static(c, A, jobs[k].f);
        ctrl_bind(A, S.layouts[k], reinterpret_cast<const char*>(S.offsets[k]), jobs[k].f);     // offsets from the block's start
        if (int rc = rig_params(c, A, jobs[k].rig)) return rc;
        if (S.skin_update && A.scene_skin_ok && !A.skin_outputs.empty()) jobs[k].sk = A.scene_skin;
        else memset(&jobs[k].sk, 0, sizeof jobs[k].sk);
        jobs[k].counter = nullptr;
        jobs[k].tag = 0;
        jobs[k].n_sample_blocks = jobs[k].sx = 0;
        if (S.one_frame) {
            Animator& Aw = *S.animators[k];
            if (!Aw.d_frame_counter) {      // the animator's frame counter, on its first one-launch frame (as run_frame)
                const size_t cb = (size_t)kFrameCounterReplicas * kFrameCounterStride;
                FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&Aw.d_frame_counter), cb));
                Aw.frame_counter_total = 0;
                FYX_HIP(c, hipMemsetAsync(Aw.d_frame_counter, 0, cb, ps));
            }
            jobs[k].counter = Aw.d_frame_counter;
            jobs[k].tag = Aw.id;
            jobs[k].sx = (A.rig->n_nodes * 16u + 255u) / 256u;
            jobs[k].n_sample_blocks = jobs[k].sx * A.n_instances * (uint32_t)A.anims.size();
        }
    }
    bool eligible = !S.one_frame;
    S.single_palette = false;
    for (size_t k = 0; k < n; ++k) {
        const Animator& A = *S.animators[k];
        eligible = eligible && !A.rm_enabled && A.prop_slots.empty() && A.dev_prop_slots == 0;
        for (const Animator::PaletteOut& po : A.palette_outputs) S.single_palette = S.single_palette || !po.d_out_alt;
    }
    S.fast_eligible = eligible;
    if (unchanged)
        for (size_t k = 0; k < n; ++k) S.seen[k].prog_gen = S.animators[k]->prog_gen;
    else {                 // a new state: what the other frame stream holds and what the control slots hold belong to the old one
        ++S.static_gen;
        S.seen.resize(n);
        for (size_t k = 0; k < n; ++k) S.seen[k] = SceneBatch::Seen{S.animators[k]->api_gen, S.animators[k]->prog_gen};
        S.seen_members_epoch = S.members_epoch;
        S.seen_options_gen = c->options_gen;
        S.seen_mesh_gen = c->mesh_gen;
    }
    }      // (!fast)
    if (S.single_palette)      // (anim.overlap = 2, a palette output with ONE buffer: this frame rewrites what the previous frame's skinning reads)
        if (int rc = pose_behind_all_skinning(c, ps)) return rc;
    const size_t total = S.ctrl_total, o_targets = S.o_targets;
    bool all_straight = c->upd_lean != 0;      // (a property of the frame's programs)
    for (size_t k = 0; k < n; ++k) all_straight = all_straight && S.animators[k]->all_straight;
    host_section(1);
    // A job array that changed travels through the frame's PINNED staging block, behind the control sections (the block is not
    // rewritten before the event behind this frame's kernels: ctrl_consumed) -- not from the pageable vector, which the next frame
    // rewrites while a copy the runtime chose to make asynchronous might still read it.
    const bool send_jobs = !fast && S.h_jobs != S.sent_jobs[par];
    const size_t o_jobs = align_up(std::max<size_t>(total, 16), 256);
    int slot = 0;
    char *h = nullptr, *d = nullptr;
    if (int rc = ctrl_acquire(c, S.ctrl, send_jobs ? o_jobs + S.h_jobs.size() : std::max<size_t>(total, 16), &slot, &h, &d)) return rc;
    // a steady frame into a slot whose last full write was made in this state: programs and offsets are where that write left them
    // (staging block and device block: the copy below moves the whole block again), clocks and tick flags are this frame's
    std::vector<uint64_t>& in_slot = S.slot_prog[slot];
    const bool slot_laid_out = fast && S.slot_gen[slot] == S.static_gen && in_slot.size() == n;
    if (!slot_laid_out) in_slot.assign(n, 0);
    for (size_t k = 0; k < n; ++k) {
        const Animator& A = *S.animators[k];
        if (in_slot[k] == A.prog_gen) ctrl_write_clocks(A, S.layouts[k], h + S.offsets[k]);
        else {
            ctrl_write(A, S.layouts[k], h + S.offsets[k]);
            in_slot[k] = A.prog_gen;
        }
    }
    S.slot_gen[slot] = S.static_gen;
    if (S.one_frame) {
        uint32_t* tg = reinterpret_cast<uint32_t*>(h + o_targets);
        for (size_t k = 0; k < n; ++k) tg[k] = S.animators[k]->frame_counter_total + jobs[k].n_sample_blocks;
    }
    if (send_jobs) {
        if (S.h_jobs.size() > S.d_jobs_capacity[par]) {
            if (int rc_ = sync_all(c)) return rc_;       // launches in flight read the old array
            dfree(S.d_jobs[par]);
            S.d_jobs[par] = nullptr;
            S.d_jobs_capacity[par] = 0;
            S.sent_jobs[par].clear();
            FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&S.d_jobs[par]), S.h_jobs.size()));
            S.d_jobs_capacity[par] = S.h_jobs.size();
        }
        memcpy(h + o_jobs, S.h_jobs.data(), S.h_jobs.size());
        // in stream order behind the previous frame's pose kernels (enter_pose), whichever stream they ran on
        FYX_HIP(c, hipMemcpyAsync(S.d_jobs[par], h + o_jobs, S.h_jobs.size(), hipMemcpyHostToDevice, ps));
        S.sent_jobs[par] = S.h_jobs;
    }
    if (!fast) S.jobs_gen[par] = S.static_gen;
    if (int rc = ctrl_upload(c, S.ctrl, slot, std::max<size_t>(total, 16), ps)) return rc;
    if (send_jobs) S.ctrl.h_by_consumed[slot] = true;     // the staging block also fed a copy on `ps`: free when the event behind this frame's kernels is

    host_section(2);
    // 4. one launch per stage
    const uint4* tabs[kSceneStages];
    for (int k = 0; k < kSceneStages; ++k) tabs[k] = S.d_tables + S.table_off[k];
    if (S.skin_update)      // (the update launch skins: it lies behind the other frame stream's skinning of the same vertex buffers)
        if (int rc = skin_outputs_order(c, ps, true)) return rc;
    SceneWait sw;
    sw.o_targets = (uint32_t)o_targets;
    sw.timeout_ticks = (uint32_t)c->wait_timeout_ms * 100000u;
    sw.err = reinterpret_cast<uint32_t*>(c->dev_err);
    FYX_HIP(c, launch_scene(reinterpret_cast<const SceneJobDev*>(S.d_jobs[par]), d, tabs, S.n_blocks, S.lds_bytes, all_straight, S.wide_update, ps,
                            S.skin_update, c->lbs.exact != 0, S.one_frame ? &sw : nullptr));
    if (S.one_frame)      // (a launch that was refused has added nothing to the counters)
        for (size_t k = 0; k < n; ++k) S.animators[k]->frame_counter_total += jobs[k].n_sample_blocks;
    host_section(3);
    if (int rc = ctrl_consumed(c, S.ctrl, slot, ps)) return rc;
    if (int rc = exit_pose(c)) return rc;
    host_section(4);
    // the animators' skin outputs that did not ride in the update launch (a large scene, anim.frame_skin = 0, a stage that is not the
    // 256-thread one): ONE batched skinning launch for all of them, behind the scene's update launch on the frame's stream -- what
    // fyx_lbs_skin_batch does for the same list (its plan and device tables are cached from frame to frame)
    std::vector<fyx_skin_job>& skin_jobs = S.skin_jobs_of[par];
    const bool list_cached = fast && S.skin_jobs_gen[par] == S.static_gen;      // (the frame stream's list of this state)
    if (!list_cached) skin_jobs.clear();
    for (size_t k = 0; k < n && !list_cached; ++k) {
        const Animator& A = *S.animators[k];
        if (A.skin_outputs.empty() || (S.skin_update && A.scene_skin_ok)) continue;     // (skinned by the update launch itself)
        for (const Animator::SkinOut& so : A.skin_outputs) {
            const Animator::PaletteOut* po = nullptr;
            for (const Animator::PaletteOut& p : A.palette_outputs)
                if (p.bones_id == so.bones_id) po = &p;
            if (!po) return fail(c, FYX_ERR_INVALID_ARG, "skin output of mesh %llu: bone list %llu is no longer a palette output of the animator",
                                 (unsigned long long)so.mesh_id, (unsigned long long)so.bones_id);
            fyx_skin_job j;
            memset(&j, 0, sizeof j);
            j.mesh_id = so.mesh_id; j.d_palette = palette_of(c, *po); j.n_bones = po->n_bones; j.n_instances = A.n_instances;
            j.d_out_pos = so.d_pos; j.d_out_normal = so.d_nrm; j.d_out_tangent = so.d_tan;
            skin_jobs.push_back(j);
        }
    }
    S.skin_jobs_gen[par] = S.static_gen;
    if (!skin_jobs.empty()) {
        hipStream_t ss = nullptr;
        if (int rc = enter_skin(c, &ss)) return rc;
        if (int rc = skin_outputs_order(c, ss, false)) return rc;
        if (int rc = fyx_lbs_skin_batch(c, skin_jobs.data(), (uint32_t)skin_jobs.size())) return rc;
        if (int rc = skin_outputs_issued(c, ss)) return rc;
    }
    host_section(5);
    if (c->host_times_on) {
        c->host_times[6] += 1.0;
        if (fast) c->host_times[7] += 1.0;
    }
    return FYX_OK;
}

template <typename F>
int for_instances(fyx_ctx* c, Animator* A, uint32_t animation, uint32_t instance, F fn) {
    if (animation >= A->anims.size() || A->anims[animation].removed) return fail(c, FYX_ERR_INVALID_ARG, "animation %u does not exist", animation);
    const uint32_t na = (uint32_t)A->anims.size();
    if (instance == FYX_ALL_INSTANCES) {
        for (uint32_t i = 0; i < A->n_instances; ++i) fn(A->anim_state[(size_t)i * na + animation]);
        return FYX_OK;
    }
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    fn(A->anim_state[(size_t)instance * na + animation]);
    return FYX_OK;
}

LayerDef* find_layer(Animator* A, uint32_t layer) { return layer < A->layers.size() ? &A->layers[layer] : nullptr; }

// Runs fn(LayerState&) on one instance's state of `layer`, or on every instance's.
template <typename F>
int for_layer_states(fyx_ctx* c, Animator* A, uint32_t layer, uint32_t instance, F&& fn) {
    if (instance != FYX_ALL_INSTANCES && instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    ensure_machine_state(*A);
    if (instance == FYX_ALL_INSTANCES) {
        A->steady_gen = 0;
        for (MachineState& m : A->mstate) { fn(m.layers[layer]); m.memo_valid = false; }
    } else {
        fn(A->mstate[instance].layers[layer]);
        A->mstate[instance].memo_valid = false;
        A->steady_gen = 0;
    }
    return FYX_OK;
}

#define FYX_LAYER(c, A, L, layer)                                                                \
    LayerDef* L = find_layer((A), (layer));                                                      \
    if (!L) return fail((c), FYX_ERR_INVALID_ARG, "layer %u does not exist", (layer))

// Longest chain of nested blends below a node; -1 on a cycle (the reference would recurse forever).
int node_depth(const LayerDef& L, int32_t h, std::vector<int>& state) {
    if (h < 0 || (size_t)h >= L.nodes.size()) return 0;
    if (state[h] == -2) return -1;
    if (state[h] >= 0) return state[h];
    state[h] = -2;
    int d = 0;
    const PoseNodeDef& n = L.nodes[h];
    if (n.type != NODE_PLAY) {
        for (const BlendInput& in : n.inputs) {
            const int cd = node_depth(L, in.source, state);
            if (cd < 0) return -1;
            d = std::max(d, cd);
        }
        d += 1;
    }
    state[h] = d;
    return d;
}

}  // namespace

void anim_store_destroy(AnimStore* s) {
    if (!s) return;
    dfree(s->scene.d_tables);
    dfree(s->scene.d_jobs[0]);
    dfree(s->scene.d_jobs[1]);
    free_ctrl(s->scene.ctrl);
    for (auto& kv : s->animators) free_animator(*kv.second);
    for (auto& kv : s->bones) free_bones(kv.second);
    for (auto& kv : s->rigs) free_rig(kv.second);
    for (auto& kv : s->tracks) free_tracks(kv.second);
    delete s;
}

}  // namespace fyx
