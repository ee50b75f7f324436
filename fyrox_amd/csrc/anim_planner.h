// anim_planner.h -- the host control plane: Planner (one instance's frame: Animation::tick, Machine::evaluate_pose and
// everything below it, recorded as a fold program), the planner thread pool, plan_frame.  Included by anim_api.hip only,
// after anim_model.h.
#pragma once

namespace fyx {

namespace {

bool has_device(const fyx_ctx* c) { return c->device >= 0; }

AnimStore& store(fyx_ctx* c) {
    if (!c->anim) c->anim = new AnimStore();
    return *c->anim;
}

void dfree(void* p) { if (p) (void)hipFree(p); }

void free_tracks(TracksData& t) { dfree(t.d_tracks); dfree(t.d_loc); dfree(t.d_aux); dfree(t.d_rec); dfree(t.d_hot); dfree(t.d_spans); dfree(t.d_span_rows); t = TracksData(); }
void free_rig(Rig& r) {
    dfree(r.d_statics); dfree(r.d_walk); dfree(r.d_inv_bind);
    r = Rig();
}
void free_bones(BoneList& b) { dfree(b.d_bone_nodes); b = BoneList(); }
void free_animator(Animator& a) {
    for (auto& an : a.anims) { dfree(an.d_slot_track); dfree(an.d_prop_track); dfree(an.d_slot_track_f); dfree(an.d_prop_track_f); }
    dfree(a.d_prop_node); dfree(a.d_prop_pose); dfree(a.d_prop_out);
    dfree(a.d_anims); dfree(a.d_crowd); dfree(a.d_hints); dfree(a.d_anim_pose); dfree(a.d_node_trs); dfree(a.d_local);
    dfree(a.d_global); dfree(a.d_layer_masks); dfree(a.d_rm_anim); dfree(a.d_rm_slots); dfree(a.d_frame_counter); dfree(a.d_slot_hints);
    free_ctrl(a.ctrl);
}

template <typename T>
int upload(fyx_ctx* c, T** dst, const T* src, size_t count) {
    *dst = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(dst), bytes));
    if (count) FYX_HIP(c, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return FYX_OK;
}

Animator* find_animator(fyx_ctx* c, uint64_t id) {
    if (!c->anim) return nullptr;
    auto it = c->anim->animators.find(id);
    return it == c->anim->animators.end() ? nullptr : it->second.get();
}

// Every entry point that names an animator may change what its fold programs depend on (parameters, structure, clocks,
// removed clips ...): it invalidates the instances' memoised programs wholesale.  Only the per-frame calls (update, plan)
// use the _RO form.
#define FYX_ANIMATOR_RO(c, a, id)                                                                \
    Animator* a = find_animator((c), (id));                                                      \
    if (!a) return fail((c), FYX_ERR_UNKNOWN_ID, "animator %llu is not registered", (unsigned long long)(id))
#define FYX_ANIMATOR(c, a, id)                                                                   \
    FYX_ANIMATOR_RO(c, a, id);                                                                   \
    ++a->api_gen;                                                                                \
    ++a->edit_gen

// ------------------------------------------------------------------------------------------
// Animation scalars
// ------------------------------------------------------------------------------------------
// fyrox-math/src/lib.rs:179-203
float wrapf(float n, float min_limit, float max_limit) {
    if (n >= min_limit && n <= max_limit) return n;
    if (max_limit == 0.0f && min_limit == 0.0f) return 0.0f;
    max_limit -= min_limit;
    const float offset = min_limit;
    min_limit = 0.0f;
    n -= offset;
    const float num_of_max = floorf(fabsf(n / max_limit));
    if (n >= max_limit) {
        n -= num_of_max * max_limit;
    } else if (n < min_limit) {
        n += (num_of_max + 1.0f) * max_limit;
    }
    return n + offset;
}

// lib.rs:432-440
void set_time_position(AnimState& s, float time) {
    if (s.looped) {
        s.time = wrapf(time, s.start, s.end);
    } else {
        float t = time;  // f32::clamp
        if (t < s.start) t = s.start;
        if (t > s.end) t = s.end;
        s.time = t;
    }
}
// lib.rs:736-738
bool has_ended(const AnimState& s) { return !s.looped && fabsf(s.time - s.end) <= FLT_EPSILON; }

// ------------------------------------------------------------------------------------------
// Planner
// ------------------------------------------------------------------------------------------
// The generator behind StateAction::EnableRandomAnimation.  The reference draws from rand::thread_rng(), which no
// one can reproduce; here every instance owns a splitmix64 stream (documented in fyrox_hip.h, restated by the oracle)
// so that a run is repeatable and instances can be given the same or different streams.
constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
inline uint64_t splitmix64(uint64_t& state) {
    uint64_t z = (state += kGolden);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint32_t random_index(uint64_t& state, uint32_t n) {   // uniform in 0..n: the high word of draw * n
    return (uint32_t)(((unsigned __int128)splitmix64(state) * n) >> 64);
}
void ensure_rng(Animator& A) {
    if (A.rng.size() == A.n_instances) return;
    A.rng.resize(A.n_instances);
    for (uint32_t i = 0; i < A.n_instances; ++i) A.rng[i] = kGolden * (uint64_t)(i + 1);   // distinct default streams
}

struct Planner {
    Animator& A;
    PlanScratch& S;
    uint32_t inst;
    float dt;
    uint32_t n_anims;
    AnimState* as;       // this instance's animation states
    MachineState* ms;
    int error = 0;       // FYX_ERR_UNSUPPORTED when the fold nests too deep
    int depth = 0;
    bool touched = false;  // something has been blended into the pose the program is writing at this point (see emit_blend)
    bool ticks_done = false;   // the frame's ticks were made by the steady path before it found a transition firing (plan_frame_core)

    Planner(Animator& a, PlanScratch& sc, uint32_t i, float dt_) : A(a), S(sc), inst(i), dt(dt_) {
        n_anims = (uint32_t)a.anims.size();
        as = a.anim_state.data() + (size_t)i * n_anims;
        ms = a.mstate.empty() ? nullptr : &a.mstate[i];
    }

    void emit(uint32_t code, uint32_t arg, float w) {
        uint2 op;
        op.x = code | (arg << 8);
        memcpy(&op.y, &w, 4);
        S.ops.push_back(op);
    }

    // Animation::tick (lib.rs:471-496): the pose is sampled at the CURRENT time, then time advances.
    void tick(uint32_t a) {
        AnimState& s = as[a];
        const AnimationDef& def = A.anims[a];
        A.times[(size_t)inst * n_anims + a] = s.time;
        const float current = s.time, next = current + dt * s.speed;
        // signals (lib.rs:476-489).  Precedence exactly as written there: `a || b && cap`, so the
        // max_event_capacity cap guards only the negative-speed branch.
        for (size_t i = 0; i < def.signals.size(); ++i) {
            const AnimationDef::Signal& sg = def.signals[i];
            if (!sg.enabled) continue;
            if ((s.speed >= 0.0f && (current < sg.time && next >= sg.time)) ||
                (s.speed < 0.0f && (current > sg.time && next <= sg.time) && s.events.size() < s.max_event_capacity))
                s.events.push_back((int32_t)i);
        }
        set_time_position(s, next);
        // what update_root_motion needs besides the sampled pose (lib.rs:539-554)
        const bool new_loop = s.looped && ((s.speed > 0.0f && s.time < current) || (s.speed < 0.0f && s.time > current));
        A.ticked[(size_t)inst * n_anims + a] = (uint8_t)(1u | (new_loop ? 2u : 0u) | (s.speed > 0.0f ? 4u : 0u));
    }

    // ---- root-motion program (pose.rs:73,98-100; play.rs:97) ----
    bool rm() const { return A.rm_enabled; }
    uint32_t node_slot(uint32_t li, int32_t h) const { return A.rm_layer_base[li] + (uint32_t)h; }
    uint32_t layer_slot(uint32_t li) const { return A.rm_layer_base[li] + (uint32_t)A.layers[li].nodes.size(); }
    uint32_t machine_slot() const { return A.n_rm_slots - 1; }
    void rm_emit(uint32_t code, uint32_t dst, uint32_t src, float w) {
        uint4 op;
        op.x = code; op.y = dst; op.z = src;
        memcpy(&op.w, &w, 4);
        S.rm_ops.push_back(op);
    }
    uint32_t cur_layer = 0;
    bool unstable = false;   // this frame's program was planned inside a transition: not next frame's
    void layer_event(LayerState& LS, int32_t kind, int32_t a, int32_t b) {  // event.rs:79-83
        if (LS.events.size() < kLayerEventLimit) LS.events.push_back(fyx_layer_event{kind, a, b});
    }

    const Param* param(int32_t idx) const {
        return (idx >= 0 && (size_t)idx < ms->params.size()) ? &ms->params[idx] : nullptr;
    }

    uint32_t new_recipe_anim(uint32_t a) {
        Recipe r;
        r.anim = (int32_t)a;
        S.recipes.push_back(r);
        return (uint32_t)S.recipes.size() - 1;
    }
    uint32_t new_recipe_fold(const RecipeItem* it, uint32_t n) {
        Recipe r;
        r.first = (uint32_t)S.items.size();
        r.count = n;
        for (uint32_t i = 0; i < n; ++i) S.items.push_back(it[i]);
        S.recipes.push_back(r);
        return (uint32_t)S.recipes.size() - 1;
    }

    // transition.rs:141-173
    bool logic(const std::vector<int32_t>& code, size_t& pc) const {
        if (pc >= code.size()) return false;
        const int32_t op = code[pc++];
        switch (op) {
            case FYX_LOGIC_PARAMETER: {
                const int32_t idx = pc < code.size() ? code[pc++] : -1;
                const Param* p = param(idx);
                return p && p->kind == FYX_PARAM_RULE && p->u != 0;
            }
            case FYX_LOGIC_AND: { const bool l = logic(code, pc); const bool r = logic(code, pc); return l & r; }
            case FYX_LOGIC_OR: { const bool l = logic(code, pc); const bool r = logic(code, pc); return l | r; }
            case FYX_LOGIC_XOR: { const bool l = logic(code, pc); const bool r = logic(code, pc); return l ^ r; }
            case FYX_LOGIC_NOT: return !logic(code, pc);
            case FYX_LOGIC_IS_ANIMATION_ENDED: {
                const int32_t a = pc < code.size() ? code[pc++] : -1;
                if (a < 0 || (uint32_t)a >= n_anims || A.anims[a].removed) return true;  // invalid handle: is_none_or -> true
                return has_ended(as[a]);
            }
            default: return false;
        }
    }

    // AnimationPoseSource::eval_pose, control part.  Returns the recipe of the node's output pose
    // (-1: the handle does not resolve, nodes.try_borrow fails).
    int32_t eval_node(const LayerDef& L, LayerState& LS, int32_t handle, int32_t* node_recipe) {
        if (handle < 0 || (size_t)handle >= L.nodes.size()) return -1;
        const PoseNodeDef& n = L.nodes[handle];
        int32_t out = -1;
        switch (n.type) {
            case NODE_PLAY:  // play.rs:86-100
                out = (int32_t)new_recipe_anim(n.animation);   // of a removed animation: the pose it had last (see AnimationDef)
                if (rm() && !A.anims[n.animation].removed) rm_emit(RM_SET_ANIM, node_slot(cur_layer, handle), n.animation, 0.f);
                break;
            case NODE_BLEND: {  // blend.rs:136-164
                RecipeItem small[16];                      // no heap traffic for the usual fan-in
                std::vector<RecipeItem> big;
                RecipeItem* its = small;
                if (n.inputs.size() > 16) { big.resize(n.inputs.size()); its = big.data(); }
                uint32_t cnt = 0;
                for (const BlendInput& in : n.inputs) {
                    float w;
                    if (in.weight_param < 0) {
                        w = in.weight_const;
                    } else {
                        const Param* p = param(in.weight_param);
                        w = (p && p->kind == FYX_PARAM_WEIGHT) ? p->f0 : 0.0f;
                    }
                    const int32_t src = eval_node(L, LS, in.source, node_recipe);
                    if (src >= 0) {
                        its[cnt++] = {(uint32_t)src, w};
                        if (rm()) rm_emit(RM_BLEND, node_slot(cur_layer, handle), node_slot(cur_layer, in.source), w);
                    }
                }
                out = (int32_t)new_recipe_fold(its, cnt);
                break;
            }
            case NODE_BY_INDEX: {  // blend.rs:306-361
                ByIndexState& st = LS.by_index[n.by_index_slot];
                RecipeItem its[2];
                uint32_t cnt = 0;
                const Param* p = param(n.param);
                if (p && p->kind == FYX_PARAM_INDEX) {
                    const uint32_t current = p->u;
                    bool applied = false;
                    if (st.has_prev) {
                        if (st.prev != current && st.prev < n.inputs.size() && current < n.inputs.size()) {
                            const BlendInput& prev_in = n.inputs[st.prev];
                            const BlendInput& cur_in = n.inputs[current];
                            float bt = st.blend_time + dt;  // (blend_time + dt).min(current.blend_time)
                            if (cur_in.blend_time < bt) bt = cur_in.blend_time;
                            st.blend_time = bt;
                            const float interpolator = st.blend_time / cur_in.blend_time;
                            const int32_t pr = eval_node(L, LS, prev_in.source, node_recipe);
                            if (pr >= 0) {
                                its[cnt++] = {(uint32_t)pr, 1.0f - interpolator};
                                if (rm()) rm_emit(RM_BLEND, node_slot(cur_layer, handle), node_slot(cur_layer, prev_in.source), 1.0f - interpolator);
                            }
                            const int32_t cr = eval_node(L, LS, cur_in.source, node_recipe);
                            if (cr >= 0) {
                                its[cnt++] = {(uint32_t)cr, interpolator};
                                if (rm()) rm_emit(RM_BLEND, node_slot(cur_layer, handle), node_slot(cur_layer, cur_in.source), interpolator);
                            }
                            if (interpolator >= 1.0f) {
                                st.prev = current;
                                st.blend_time = 0.0f;
                            }
                            applied = true;
                        }
                    } else {
                        st.has_prev = true;
                        st.prev = current;
                    }
                    if (!applied) {
                        st.blend_time = 0.0f;
                        if (current < n.inputs.size()) {
                            const int32_t cr = eval_node(L, LS, n.inputs[current].source, node_recipe);
                            if (cr >= 0) {
                                its[cnt++] = {(uint32_t)cr, 1.0f};  // clone_into an empty pose
                                if (rm()) rm_emit(RM_COPY, node_slot(cur_layer, handle), node_slot(cur_layer, n.inputs[current].source), 0.f);
                            }
                        }
                    }
                }
                out = (int32_t)new_recipe_fold(its, cnt);
                break;
            }
            case NODE_BLEND_SPACE: {  // blendspace.rs:118-150
                RecipeItem its[3];
                uint32_t cnt = 0;
                const Param* p = param(n.param);
                if (p && p->kind == FYX_PARAM_SAMPLING_POINT) {
                    int idx[3];
                    float w[3];
                    const float sp[2] = {p->f0, p->f1};
                    if (blend_space_weights(n, sp, idx, w)) {
                        const int32_t sa = n.inputs[idx[0]].source, sb = n.inputs[idx[1]].source,
                                      sc = n.inputs[idx[2]].source;
                        auto ok = [&](int32_t h) { return h >= 0 && (size_t)h < L.nodes.size(); };
                        if (ok(sa) && ok(sb) && ok(sc)) {
                            // blendspace.rs:139-141: evaluate a, blend, evaluate b, blend, evaluate c, blend
                            const int32_t srcs[3] = {sa, sb, sc};
                            for (int k = 0; k < 3; ++k) {
                                its[cnt++] = {(uint32_t)eval_node(L, LS, srcs[k], node_recipe), w[k]};
                                if (rm()) rm_emit(RM_BLEND, node_slot(cur_layer, handle), node_slot(cur_layer, srcs[k]), w[k]);
                            }
                        }
                    }
                }
                out = (int32_t)new_recipe_fold(its, cnt);
                break;
            }
        }
        node_recipe[handle] = out;  // the node's cached output_pose now holds this
        return out;
    }

    // fyrox-math/src/lib.rs:291-313,326-328 and blendspace.rs:338-414 (fetch_weights)
    static bool blend_space_weights(const PoseNodeDef& n, const float sp[2], int idx[3], float w[3]) {
        const size_t np = n.inputs.size();
        const float* pts = n.points.data();
        if (np == 0) return false;
        if (np == 1) { idx[0] = idx[1] = idx[2] = 0; w[0] = 1.0f; w[1] = w[2] = 0.0f; return true; }
        if (np == 2) {
            const float e[2] = {pts[2] - pts[0], pts[3] - pts[1]};
            const float tp[2] = {sp[0] - pts[0], sp[1] - pts[1]};
            const float t = (tp[0] * e[0] + tp[1] * e[1]) / (e[0] * e[0] + e[1] * e[1]);
            if (t >= 0.0f && t <= 1.0f) {
                idx[0] = 0; idx[1] = 1; idx[2] = 0;
                w[0] = 1.0f - t; w[1] = t; w[2] = 0.0f;
                return true;
            }
        }
        const size_t nt = n.triangles.size() / 3;
        for (size_t k = 0; k < nt; ++k) {
            const uint32_t ia = n.triangles[k * 3], ib = n.triangles[k * 3 + 1], ic = n.triangles[k * 3 + 2];
            const float* a = pts + ia * 2;
            const float* b = pts + ib * 2;
            const float* c = pts + ic * 2;
            const float v0[2] = {b[0] - a[0], b[1] - a[1]}, v1[2] = {c[0] - a[0], c[1] - a[1]};
            const float v2[2] = {sp[0] - a[0], sp[1] - a[1]};
            const float d00 = v0[0] * v0[0] + v0[1] * v0[1], d01 = v0[0] * v1[0] + v0[1] * v1[1];
            const float d11 = v1[0] * v1[0] + v1[1] * v1[1], d20 = v2[0] * v0[0] + v2[1] * v0[1];
            const float d21 = v2[0] * v1[0] + v2[1] * v1[1];
            const float inv_denom = 1.0f / (d00 * d11 - d01 * d01);
            const float v = (d11 * d20 - d01 * d21) * inv_denom;
            const float ww = (d00 * d21 - d01 * d20) * inv_denom;
            const float u = 1.0f - v - ww;
            if (u >= 0.0f && v >= 0.0f && u + v < 1.0f) {
                idx[0] = (int)ia; idx[1] = (int)ib; idx[2] = (int)ic;
                w[0] = u; w[1] = v; w[2] = ww;
                return true;
            }
        }
        float min_distance = FLT_MAX;
        bool found = false;
        for (size_t k = 0; k < nt; ++k)
            for (int e = 0; e < 3; ++e) {
                const uint32_t a = n.triangles[k * 3 + e], b = n.triangles[k * 3 + (e + 1) % 3];
                const float* pa = pts + a * 2;
                const float* pb = pts + b * 2;
                const float edge[2] = {pb[0] - pa[0], pb[1] - pa[1]};
                const float tp[2] = {sp[0] - pa[0], sp[1] - pa[1]};
                const float t = (tp[0] * edge[0] + tp[1] * edge[1]) / (edge[0] * edge[0] + edge[1] * edge[1]);
                if (t >= 0.0f && t <= 1.0f) {
                    const float proj[2] = {pa[0] + edge[0] * t, pa[1] + edge[1] * t};
                    const float dx = sp[0] - proj[0], dy = sp[1] - proj[1];
                    const float distance = sqrtf(dx * dx + dy * dy);
                    if (distance < min_distance) {
                        min_distance = distance;
                        idx[0] = (int)a; idx[1] = (int)b; idx[2] = (int)b;
                        w[0] = 1.0f - t; w[1] = t; w[2] = 0.0f;
                        found = true;
                    }
                }
            }
        return found;
    }

    // acc.blend_with(<pose described by recipe r>, w).  A pose built from several others is evaluated into a pose of its own
    // (PUSH ... POP_BLEND w) only where that changes the result:
    //   * nothing has been blended into acc yet: an empty pose becomes a COPY of what is blended into it, weight ignored
    //     (NodePose::blend_with, pose.rs:41-47; a node the sub-tree leaves empty stays empty either way), so the sub-tree is
    //     written straight into acc;
    //   * the sub-tree is one clip: its pose would be a copy of the clip's, so the clip is blended with the outer weight.
    // `touched` is a property of the program, the same for every node: what the kernel's straight form (anim_kernels.hip,
    // pose_update_body) relies on is that the common machines come out of here without a single PUSH.
    void emit_blend(uint32_t r, float w) {
        const Recipe rc = S.recipes[r];
        if (rc.anim >= 0) { emit(OP_BLEND_ANIM, (uint32_t)rc.anim, w); touched = true; return; }
        if (rc.count == 0) return;  // blending with an empty pose changes nothing
        if (!touched) {
            for (uint32_t i = 0; i < rc.count; ++i) {
                const RecipeItem it = S.items[rc.first + i];
                emit_blend(it.recipe, it.w);
            }
            return;
        }
        if (rc.count == 1 && S.recipes[S.items[rc.first].recipe].anim >= 0) {
            emit(OP_BLEND_ANIM, (uint32_t)S.recipes[S.items[rc.first].recipe].anim, w);
            return;
        }
        if (depth + 1 >= kMaxFoldDepth) { error = FYX_ERR_UNSUPPORTED; return; }
        emit(OP_PUSH, 0, 0.f);
        ++depth;
        touched = false;
        for (uint32_t i = 0; i < rc.count; ++i) {
            const RecipeItem it = S.items[rc.first + i];
            emit_blend(it.recipe, it.w);
        }
        --depth;
        emit(OP_POP_BLEND, 0, w);
        touched = true;
    }

    void collect(const LayerDef& L, int32_t handle) {  // node/mod.rs:116-150
        if (handle < 0 || (size_t)handle >= L.nodes.size()) return;
        const PoseNodeDef& n = L.nodes[handle];
        if (n.type == NODE_PLAY) { S.seen[n.animation] = 1; return; }
        for (const BlendInput& in : n.inputs) collect(L, in.source);
    }

    void apply_actions(const std::vector<Action>& acts) {  // state.rs:48-80
        for (const Action& a : acts) {
            if (a.kind == FYX_ACTION_ENABLE_RANDOM_ANIMATION) {   // state.rs:108-114: handles.iter().choose(rng), then enable
                if (a.choices.empty()) continue;                  // choose() on an empty iterator: None, nothing drawn
                const uint32_t pick = a.choices[random_index(A.rng[inst], (uint32_t)a.choices.size())];
                if (pick < n_anims && !A.anims[pick].removed) as[pick].enabled = 1;
                continue;
            }
            if (a.animation >= n_anims || A.anims[a.animation].removed) continue;
            AnimState& s = as[a.animation];
            switch (a.kind) {
                case FYX_ACTION_REWIND_ANIMATION: set_time_position(s, s.start); break;
                case FYX_ACTION_ENABLE_ANIMATION: s.enabled = 1; break;
                case FYX_ACTION_DISABLE_ANIMATION: s.enabled = 0; break;
                default: break;
            }
        }
    }

    // MachineLayer::evaluate_pose (layer.rs:590-706); the layer's final_pose is the accumulator
    // the caller opened.
    void plan_layer(uint32_t li) {
        const LayerDef& L = A.layers[li];
        LayerState& LS = ms->layers[li];
        cur_layer = li;
        if (LS.active_state >= 0 || LS.active_transition >= 0) {
            S.node_recipe.assign(L.nodes.size(), -1);
            int32_t* nr = S.node_recipe.data();
            for (const StateDef& s : L.states) eval_node(L, LS, s.root, nr);  // state.update

            if (LS.active_transition < 0) {
                for (size_t t = 0; t < L.transitions.size(); ++t) {
                    const TransitionDef& tr = L.transitions[t];
                    if ((int32_t)tr.dest == LS.active_state || (int32_t)tr.source != LS.active_state) continue;
                    size_t pc = 0;
                    if (logic(tr.logic, pc)) {
                        if (LS.active_state >= 0 && (size_t)LS.active_state < L.states.size())
                            apply_actions(L.states[LS.active_state].on_leave);
                        layer_event(LS, FYX_EVENT_STATE_LEAVE, LS.active_state, -1);             // layer.rs:620
                        if (tr.dest < L.states.size()) apply_actions(L.states[tr.dest].on_enter);
                        layer_event(LS, FYX_EVENT_STATE_ENTER, (int32_t)tr.dest, -1);            // :634
                        LS.active_state = -1;
                        LS.active_transition = (int32_t)t;
                        layer_event(LS, FYX_EVENT_ACTIVE_TRANSITION_CHANGED, (int32_t)t, -1);    // :645
                        break;
                    }
                }
            }

            auto root_recipe = [&](uint32_t state) -> int32_t {
                if (state >= L.states.size()) return -1;
                const int32_t r = L.states[state].root;
                return (r >= 0 && (size_t)r < L.nodes.size()) ? nr[r] : -1;
            };

            if (LS.active_transition >= 0) {
                unstable = true;
                const TransitionDef& tr = L.transitions[LS.active_transition];
                TransitionState& ts = LS.transitions[LS.active_transition];
                const int32_t src = root_recipe(tr.source), dst = root_recipe(tr.dest);
                if (src >= 0) {
                    emit_blend((uint32_t)src, 1.0f - ts.blend_factor);
                    if (rm()) rm_emit(RM_BLEND, layer_slot(li), node_slot(li, L.states[tr.source].root), 1.0f - ts.blend_factor);
                }
                if (dst >= 0) {
                    emit_blend((uint32_t)dst, ts.blend_factor);
                    if (rm()) rm_emit(RM_BLEND, layer_slot(li), node_slot(li, L.states[tr.dest].root), ts.blend_factor);
                }
                ts.elapsed += dt;  // transition.rs:315-321
                if (ts.elapsed > tr.time) ts.elapsed = tr.time;
                ts.blend_factor = ts.elapsed / tr.time;
                if (fabsf(tr.time - ts.elapsed) <= FLT_EPSILON) {  // is_done
                    ts.elapsed = 0.0f;
                    ts.blend_factor = 0.0f;
                    LS.active_transition = -1;
                    layer_event(LS, FYX_EVENT_ACTIVE_TRANSITION_CHANGED, -1, -1);                 // :673
                    LS.active_state = (int32_t)tr.dest;
                    layer_event(LS, FYX_EVENT_ACTIVE_STATE_CHANGED, (int32_t)tr.source, (int32_t)tr.dest);  // :677
                }
            } else {
                const int32_t r = root_recipe((uint32_t)LS.active_state);
                if (r >= 0) {
                    emit_blend((uint32_t)r, 1.0f);  // clone_into the (reset) final pose
                    if (rm()) rm_emit(RM_COPY, layer_slot(li), node_slot(li, L.states[LS.active_state].root), 0.f);
                }
            }
        }
        if (!L.excluded.empty()) emit(OP_MASK, li, 0.f);
    }

    // The memo (MachineState): true when last frame's program of this instance was appended instead of planning a new one.
    bool try_reuse() {
        if (!A.memo_static_ok || !ms->memo_valid || ms->memo_gen != A.edit_gen) return false;
        if (A.prev_prog_off.size() != (size_t)A.n_instances + 1) return false;
        for (size_t li = 0; li < A.layers.size(); ++li) {
            const LayerDef& L = A.layers[li];
            const LayerState& LS = ms->layers[li];
            if (LS.active_transition >= 0 || LS.active_state != LS.memo_state) return false;
            if (LS.active_state < 0) continue;
            for (const TransitionDef& tr : L.transitions) {     // would one fire? (layer.rs:605-651; conditions have no side effects)
                if ((int32_t)tr.dest == LS.active_state || (int32_t)tr.source != LS.active_state) continue;
                size_t pc = 0;
                if (logic(tr.logic, pc)) return false;
            }
        }
        const uint32_t o0 = A.prev_prog_off[inst], o1 = A.prev_prog_off[inst + 1];
        if (o0 > o1 || o1 > A.prev_ops.size()) return false;      // (never with a completed last frame: see FailGuard)
        S.ops.insert(S.ops.end(), A.prev_ops.begin() + o0, A.prev_ops.begin() + o1);
        return true;
    }
    void remember() {
        bool stable = A.memo_static_ok && !unstable && !error;
        for (LayerState& LS : ms->layers) {
            stable = stable && LS.active_transition < 0;
            LS.memo_state = LS.active_state;
        }
        ms->memo_valid = stable;
        ms->memo_gen = A.edit_gen;
    }

    // Machine::evaluate_pose (machine/mod.rs:344-382) + apply
    void plan_absm() {
        std::fill(S.seen.begin(), S.seen.end(), 0);
        for (size_t li = 0; li < A.layers.size(); ++li) {
            const LayerDef& L = A.layers[li];
            const LayerState& LS = ms->layers[li];
            int32_t check[3] = {LS.active_state, -1, -1};
            if (LS.active_transition >= 0 && (size_t)LS.active_transition < L.transitions.size()) {
                check[1] = (int32_t)L.transitions[LS.active_transition].source;
                check[2] = (int32_t)L.transitions[LS.active_transition].dest;
            }
            for (int k = 0; k < 3; ++k)
                if (check[k] >= 0 && (size_t)check[k] < L.states.size())
                    for (uint32_t a : A.state_anims[li][check[k]]) S.seen[a] = 1;   // collect(), done once per frame and state (plan_frame_core)
        }
        if (!ticks_done)
            for (uint32_t a = 0; a < n_anims; ++a)
                if (S.seen[a] && as[a].enabled) tick(a);
        if (try_reuse()) return;
        S.recipes.clear();
        S.items.clear();
        touched = false;
        for (size_t li = 0; li < A.layers.size(); ++li) {
            // final_pose.blend_with(layer pose, layer.weight) (mod.rs:375-378): while the final pose is empty the layer is
            // written straight into it (emit_blend: the copy rule -- which is also why the first layer's weight never matters)
            const bool in_place = !touched;
            if (!in_place) {
                emit(OP_PUSH, 0, 0.f);
                depth = 1;
                touched = false;
            }
            plan_layer((uint32_t)li);
            if (!in_place) {
                depth = 0;
                emit(OP_POP_BLEND, 0, A.layers[li].weight);
                touched = true;
            }
            if (rm()) rm_emit(RM_BLEND, machine_slot(), layer_slot((uint32_t)li), A.layers[li].weight);  // mod.rs:375-378
        }
        emit(OP_APPLY, 0, 0.f);
        emit(OP_END, 0, 0.f);
        if (rm()) rm_emit(RM_END, 0, 0, 0.f);
        remember();
    }

    // AnimationContainerExt::update_animations (scene/animation/mod.rs:83-88)
    void plan_player() {
        for (uint32_t a = 0; a < n_anims; ++a)
            if (as[a].enabled) {
                tick(a);
                emit(OP_APPLY_ANIM, a, 0.f);
            }
        emit(OP_END, 0, 0.f);
    }
};

// The machine structure may still grow after some instance state exists (adding a parameter or
// a transition): keep every instance's state vectors in step.
void sync_machine_state(Animator& A) {
    for (MachineState& m : A.mstate) {
        while (m.params.size() < A.param_defaults.size()) m.params.push_back(A.param_defaults[m.params.size()]);
        const size_t had = m.layers.size();
        m.layers.resize(A.layers.size());
        for (size_t l = 0; l < A.layers.size(); ++l) {
            LayerState& LS = m.layers[l];
            if (l >= had) LS.active_state = A.layers[l].initial_active;   // a state of its own from here on
            LS.transitions.resize(A.layers[l].transitions.size());
            LS.by_index.resize(A.layers[l].by_index_count);
        }
    }
}

void ensure_machine_state(Animator& A) {
    ensure_rng(A);
    if (A.mstate.size() == A.n_instances) return;
    A.mstate.assign(A.n_instances, MachineState());
    sync_machine_state(A);
}

}  // namespace

// A small persistent pool for planning a crowd: instances are independent (own animation states, own machine
// state, own event queues), so a frame's planning splits into contiguous instance ranges.
class PlanPool {
public:
    explicit PlanPool(unsigned n) {
        for (unsigned i = 0; i < n; ++i) workers_.emplace_back([this, i] { loop(i); });
    }
    ~PlanPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    unsigned size() const { return (unsigned)workers_.size(); }
    // runs fn(k) for k in 0..n_tasks; the caller does task 0 itself, workers 1.. (n_tasks - 1 <= size()).
    // Workers sleep on a condition variable between frames (no spinning: measured, spinning workers starve the
    // calling thread on hosts with a CPU quota); waking them costs tens of microseconds, which is why plan_frame
    // only splits crowds whose planning takes much longer than that.
    void run(unsigned n_tasks, const std::function<void(unsigned)>& fn) {
        if (n_tasks <= 1) { if (n_tasks) fn(0); return; }
        { std::lock_guard<std::mutex> g(m_); fn_ = &fn; tasks_ = n_tasks; pending_ = n_tasks - 1; failed_ = false; ++gen_; }
        cv_.notify_all();
        bool threw = false;
        try { fn(0); } catch (...) { threw = true; }   // the workers still hold &fn: wait for them before unwinding
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
        fn_ = nullptr;
        if (threw || failed_) throw std::bad_alloc();   // the only thing planning throws; the C ABI maps it to FYX_ERR_OOM
    }
private:
    void loop(unsigned idx) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (idx + 1 < tasks_) fn = fn_;
            }
            if (fn) {
                bool threw = false;
                try { (*fn)(idx + 1); } catch (...) { threw = true; }   // nothing may unwind out of a worker thread
                std::lock_guard<std::mutex> g(m_);
                failed_ |= threw;
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)>* fn_ = nullptr;
    unsigned tasks_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false, failed_ = false;
};

void plan_pool_destroy(PlanPool* p) { delete p; }

namespace {

bool logic_reads_clock(const std::vector<int32_t>& code) {
    for (size_t pc = 0; pc < code.size();) {
        const int32_t op = code[pc++];
        if (op == FYX_LOGIC_IS_ANIMATION_ENDED) return true;
        if (op == FYX_LOGIC_PARAMETER) ++pc;
    }
    return false;
}

// After a machine frame in which every instance's memo came out valid: what a steady frame has to do.
void steady_prepare(Animator& A) {
    A.steady_gen = 0;
    if (!A.memo_static_ok || A.mstate.size() != A.n_instances || A.state_anims_gen != A.edit_gen) return;
    const uint32_t na = (uint32_t)A.anims.size();
    A.steady_ticks.assign((size_t)A.n_instances + 1, 0);
    A.steady_list.clear();
    A.steady_timed.clear();
    std::vector<uint8_t> seen(na ? na : 1);
    for (uint32_t i = 0; i < A.n_instances; ++i) {
        const MachineState& ms = A.mstate[i];
        if (!ms.memo_valid || ms.memo_gen != A.edit_gen || ms.layers.size() != A.layers.size()) return;
        std::fill(seen.begin(), seen.end(), 0);
        bool timed = false;
        for (size_t li = 0; li < A.layers.size(); ++li) {
            const LayerDef& L = A.layers[li];
            const LayerState& LS = ms.layers[li];
            if (LS.active_transition >= 0 || LS.active_state != LS.memo_state) return;
            if (LS.active_state < 0 || (size_t)LS.active_state >= L.states.size()) continue;
            for (uint32_t a : A.state_anims[li][LS.active_state]) seen[a] = 1;
            for (const TransitionDef& tr : L.transitions)
                if ((int32_t)tr.dest != LS.active_state && (int32_t)tr.source == LS.active_state && logic_reads_clock(tr.logic)) timed = true;
        }
        A.steady_ticks[i] = (uint32_t)A.steady_list.size();
        const AnimState* as = A.anim_state.data() + (size_t)i * na;
        for (uint32_t a = 0; a < na; ++a)
            if (seen[a] && as[a].enabled) A.steady_list.push_back(a);      // (enabled flags change only through API calls and actions: both end steadiness)
        if (timed) A.steady_timed.push_back(i);
    }
    A.steady_ticks[A.n_instances] = (uint32_t)A.steady_list.size();
    A.steady_all = A.steady_list.size() == (size_t)A.n_instances * na;
    for (const AnimationDef& d : A.anims) A.steady_all = A.steady_all && d.signals.empty();
    A.steady_gen = A.edit_gen;
}

// The steady frame's ticks (what Planner::tick does, for the animations each instance ticks), then the conditions that read a clock.
// Returns true when no transition fires: the frame is planned.  false: the ticks are made, the programs have to be planned (ticks_done).
bool steady_frame(Animator& A, float dt) {
    const uint32_t na = (uint32_t)A.anims.size();
    const uint32_t* list = A.steady_list.data();
    const uint32_t* off = A.steady_ticks.data();
    float* times = A.times.data();
    uint8_t* ticked = A.ticked.data();
    AnimState* as = A.anim_state.data();
    if (A.steady_all) {      // a crowd in one state: (instance, animation) records in memory order, no signals
        const size_t n = (size_t)A.n_instances * na;
        for (size_t k = 0; k < n; ++k) {
            AnimState& s = as[k];
            const float current = s.time, speed = s.speed, next = current + dt * speed;
            times[k] = current;
            if (s.looped && next >= s.start && next <= s.end) s.time = next;
            else set_time_position(s, next);
            const bool new_loop = s.looped && ((speed > 0.0f && s.time < current) || (speed < 0.0f && s.time > current));
            ticked[k] = (uint8_t)(1u | (new_loop ? 2u : 0u) | (speed > 0.0f ? 4u : 0u));
        }
    } else
    for (uint32_t i = 0; i < A.n_instances; ++i, as += na, times += na, ticked += na) {
        for (uint32_t t = off[i]; t < off[i + 1]; ++t) {
            const uint32_t a = list[t];
            AnimState& s = as[a];
            const float current = s.time, next = current + dt * s.speed;
            times[a] = current;
            const AnimationDef& def = A.anims[a];
            if (!def.signals.empty()) {        // lib.rs:476-489, as Planner::tick
                for (size_t k = 0; k < def.signals.size(); ++k) {
                    const AnimationDef::Signal& sg = def.signals[k];
                    if (!sg.enabled) continue;
                    if ((s.speed >= 0.0f && (current < sg.time && next >= sg.time)) ||
                        (s.speed < 0.0f && (current > sg.time && next <= sg.time) && s.events.size() < s.max_event_capacity))
                        s.events.push_back((int32_t)k);
                }
            }
            if (s.looped && next >= s.start && next <= s.end) s.time = next;      // wrapf's first line: the usual tick
            else set_time_position(s, next);
            const bool new_loop = s.looped && ((s.speed > 0.0f && s.time < current) || (s.speed < 0.0f && s.time > current));
            ticked[a] = (uint8_t)(1u | (new_loop ? 2u : 0u) | (s.speed > 0.0f ? 4u : 0u));
        }
    }
    if (!A.steady_timed.empty()) {
        static thread_local PlanScratch scratch;       // (Planner wants one; logic() does not touch it)
        for (uint32_t i : A.steady_timed) {
            Planner p(A, scratch, i, dt);
            for (size_t li = 0; li < A.layers.size(); ++li) {
                const LayerDef& L = A.layers[li];
                const LayerState& LS = p.ms->layers[li];
                for (const TransitionDef& tr : L.transitions) {
                    if ((int32_t)tr.dest == LS.active_state || (int32_t)tr.source != LS.active_state) continue;
                    size_t pc = 0;
                    if (p.logic(tr.logic, pc)) return false;
                }
            }
        }
    }
    return true;
}

// Plans one frame of every instance of A.  Touches nothing but A (fyx_scene_update plans different animators on
// different threads); n_tasks > 1 splits the instances over `pool`.  Returns 0 or the planner's error code.
int plan_frame_core(Animator& A, int mode, float dt, unsigned n_tasks, PlanPool* pool) {
    const uint32_t na = (uint32_t)A.anims.size();
    // The steady frame: nothing but clocks has moved since the last machine frame, in which every instance reused its program.  The
    // times / tick flags of animations that do not tick are last frame's zeros, the programs stay where they are.
    bool ticks_done = false;
    if (mode == 1 && A.prev_mode == 1 && A.steady_gen == A.edit_gen && A.steady_gen != 0 && A.times.size() == (size_t)A.n_instances * na) {
        if (steady_frame(A, dt)) return FYX_OK;
        ticks_done = true;        // a transition fires somewhere: plan the programs (the ticks are made)
    }
    A.steady_gen = 0;
    ++A.prog_gen;
    if (!ticks_done) {
        A.times.assign((size_t)A.n_instances * na, 0.f);
        A.ticked.assign((size_t)A.n_instances * na, 0);
    }
    // last frame's programs become the memo's source; a frame of another kind in between invalidates it
    A.ops.swap(A.prev_ops);
    A.prog_off.swap(A.prev_prog_off);
    if (mode != 1 || A.prev_mode != 1) ++A.edit_gen;
    A.prev_mode = mode;
    // A frame that does not complete (planner error, std::bad_alloc out of the pool or a vector) leaves A.ops / A.prog_off half
    // filled, and the instances planned before the failure have already stamped their memo with this generation: the next frame
    // would swap the torso in as the memo's source.  Whatever the exit, a failed frame invalidates every memo.
    struct FailGuard {
        Animator& a;
        bool ok = false;
        ~FailGuard() { if (!ok) { ++a.edit_gen; a.ops.clear(); a.prog_off.clear(); a.prev_ops.clear(); a.prev_prog_off.clear(); } }
    } guard{A};
    // which animations a state's pose tree plays (node/mod.rs:116-150): the same for every instance, and the same as last
    // frame unless an API call touched the animator in between (edit_gen)
    if (mode == 1 && A.state_anims_gen != A.edit_gen) {
        A.state_anims_gen = A.edit_gen;
        struct Walk {
            static void go(const LayerDef& L, int32_t h, std::vector<uint32_t>& out, int depth) {
                if (h < 0 || (size_t)h >= L.nodes.size() || depth > 64) return;
                const PoseNodeDef& n = L.nodes[h];
                if (n.type == NODE_PLAY) { out.push_back(n.animation); return; }
                for (const BlendInput& in : n.inputs) go(L, in.source, out, depth + 1);
            }
        };
        A.state_anims.resize(A.layers.size());
        for (size_t li = 0; li < A.layers.size(); ++li) {
            const LayerDef& L = A.layers[li];
            A.state_anims[li].resize(L.states.size());
            for (size_t st = 0; st < L.states.size(); ++st) {
                A.state_anims[li][st].clear();
                Walk::go(L, L.states[st].root, A.state_anims[li][st], 0);
            }
        }
    }
    A.memo_static_ok = mode == 1 && !A.rm_enabled;
    for (const LayerDef& L : A.layers) A.memo_static_ok = A.memo_static_ok && L.by_index_count == 0;
    A.ops.clear();
    A.prog_off.assign((size_t)A.n_instances + 1, 0);
    if (mode == 1) ensure_machine_state(A);  // instances get their machine state lazily
    A.rm_ops.clear();
    A.rm_prog_off.assign((size_t)A.n_instances + 1, 0);
    if (A.rm_enabled) {
        // slots: per layer its pose nodes then its final pose; the machine's final pose last
        A.rm_layer_base.assign(A.layers.size(), 0);
        uint32_t n = 0;
        for (size_t l = 0; l < A.layers.size(); ++l) { A.rm_layer_base[l] = n; n += (uint32_t)A.layers[l].nodes.size() + 1; }
        A.n_rm_slots = n + 1;
        A.slices.resize((size_t)A.n_instances * na);
        for (size_t k = 0; k < A.slices.size(); ++k) A.slices[k] = make_float2(A.anim_state[k].start, A.anim_state[k].end);
    }
    if (A.scratch.size() < n_tasks) A.scratch.resize(n_tasks);
    auto work = [&](unsigned k) {
        PlanScratch& S = A.scratch[k];
        const uint32_t i0 = (uint32_t)((uint64_t)A.n_instances * k / n_tasks);
        const uint32_t i1 = (uint32_t)((uint64_t)A.n_instances * (k + 1) / n_tasks);
        S.ops.clear(); S.rm_ops.clear(); S.prog_len.clear(); S.rm_prog_len.clear();
        S.seen.assign(na ? na : 1, 0);
        S.error = 0;
        S.all_straight = true;
        for (uint32_t i = i0; i < i1; ++i) {
            const size_t o0 = S.ops.size(), r0 = S.rm_ops.size();
            Planner p(A, S, i, dt);
            p.ticks_done = ticks_done;
            if (mode == 1) p.plan_absm(); else p.plan_player();
            if (p.error) S.error = p.error;
            // which form of the update kernel may run this frame: the one without the interpreter needs EVERY program straight
            static_assert(sizeof(uint2) == 8, "ops are {x, y} pairs");
            if (S.all_straight)
                S.all_straight = classify_fold_program_host(reinterpret_cast<const uint32_t*>(S.ops.data() + o0), (uint32_t)(S.ops.size() - o0)).straight;
            S.prog_len.push_back((uint32_t)(S.ops.size() - o0));
            S.rm_prog_len.push_back((uint32_t)(S.rm_ops.size() - r0));
        }
    };
    if (n_tasks > 1) pool->run(n_tasks, work); else work(0);
    uint32_t inst = 0;
    A.all_straight = A.n_instances > 0;
    for (unsigned k = 0; k < n_tasks; ++k) A.all_straight = A.all_straight && A.scratch[k].all_straight;
    for (unsigned k = 0; k < n_tasks; ++k) {  // merge in instance order
        const PlanScratch& S = A.scratch[k];
        if (S.error) return S.error;
        uint32_t o = (uint32_t)A.ops.size(), r = (uint32_t)A.rm_ops.size();
        for (size_t j = 0; j < S.prog_len.size(); ++j, ++inst) {
            A.prog_off[inst] = o;
            A.rm_prog_off[inst] = r;
            o += S.prog_len[j];
            r += S.rm_prog_len[j];
        }
        if (n_tasks == 1) {          // one task: its vectors ARE the result (no second copy of every program)
            A.ops.swap(A.scratch[k].ops);
            A.rm_ops.swap(A.scratch[k].rm_ops);
        } else {
            A.ops.insert(A.ops.end(), S.ops.begin(), S.ops.end());
            A.rm_ops.insert(A.rm_ops.end(), S.rm_ops.begin(), S.rm_ops.end());
        }
    }
    A.prog_off[A.n_instances] = (uint32_t)A.ops.size();
    A.rm_prog_off[A.n_instances] = (uint32_t)A.rm_ops.size();
    guard.ok = true;
    if (mode == 1) steady_prepare(A);
    return FYX_OK;
}

PlanPool* plan_pool(fyx_ctx* c, unsigned n_tasks) {
    if (!c->plan_pool || c->plan_pool->size() + 1 < n_tasks) {
        delete c->plan_pool;
        c->plan_pool = new PlanPool(n_tasks - 1);
    }
    return c->plan_pool;
}

// Planning costs ~0.1 us per instance and waking the pool tens of microseconds: split only big crowds, one task per
// `anim.split` instances (default 2048), at most anim.threads of them
unsigned plan_tasks(const fyx_ctx* c, const Animator& A) {
    const uint32_t split = (uint32_t)std::max(c->plan_split, 1);
    if (c->plan_threads > 1 && A.n_instances >= 2 * split) return std::min<unsigned>((unsigned)c->plan_threads, A.n_instances / split);
    return 1;
}

int plan_frame(fyx_ctx* c, Animator& A, int mode, float dt) {
    const unsigned n_tasks = plan_tasks(c, A);
    if (int e = plan_frame_core(A, mode, dt, n_tasks, n_tasks > 1 ? plan_pool(c, n_tasks) : nullptr))
        return fail(c, e, "pose nodes nest deeper than %d blend levels", kMaxFoldDepth - 2);
    return FYX_OK;
}

}  // namespace

}  // namespace fyx
