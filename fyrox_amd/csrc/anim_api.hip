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

namespace fyx {

namespace {

struct TracksData {
    uint32_t n_tracks = 0;
    std::vector<fyx_track_desc> tracks;
    TrackDev* d_tracks = nullptr;
    float* d_loc = nullptr;
    float4* d_aux = nullptr;
};

struct Rig {
    uint32_t n_nodes = 0, n_levels = 0;
    std::vector<int32_t> parent;
    std::vector<float> init_trs;  // [n_nodes][12]
    int32_t* d_parent = nullptr;
    float* d_statics = nullptr;
    uint32_t* d_level_nodes = nullptr;
    uint32_t* d_level_start = nullptr;
    uint32_t* d_node_level = nullptr;
    float* d_inv_bind = nullptr;
};

struct BoneList {
    uint64_t rig_id = 0;
    uint32_t n_bones = 0;
    int32_t* d_bone_nodes = nullptr;
};

// ---- shared structure of an animator ----
struct AnimationDef {
    uint64_t tracks_id = 0;
    const TracksData* td = nullptr;
    std::vector<int32_t> target;   // per track, <0: no TrackBinding
    std::vector<uint8_t> enabled;  // TrackBinding::enabled
    int32_t* d_slot_track = nullptr;
    int32_t* d_prop_track = nullptr;   // [animator's property slots]
    uint32_t dev_prop_slots = 0;
    bool slots_dirty = true;
    // AnimationSignal (signal.rs): the index stands for the {id, name} pair the shim keeps
    struct Signal { float time; uint8_t enabled; };
    std::vector<Signal> signals;
    // RootMotionSettings (lib.rs:307-319); node < 0: None
    int32_t rm_node = -1;
    uint32_t rm_ignore = 0;
    int32_t rm_pos_track = -1, rm_rot_track = -1;  // first Position / Rotation track of the tracks data
    // AnimationContainer::remove (lib.rs:1007): the handle is invalid from then on.  Nothing ticks it, conditions see
    // "ended" (is_none_or), actions skip it -- and a PlayAnimation node that still names it keeps the pose it copied
    // last (play.rs:93-99 only overwrites its output when the handle resolves), which is the record the device holds.
    bool removed = false;
};

struct AnimState {  // per instance, per animation (Animation's scalar fields)
    float time = 0.f, speed = 1.f, start = 0.f, end = 0.f;
    uint8_t enabled = 1, looped = 1;
    uint32_t max_event_capacity = 32;   // lib.rs:941
    std::deque<int32_t> events;         // VecDeque<AnimationEvent>, as signal indices
};

struct Param {
    int kind = FYX_PARAM_WEIGHT;
    float f0 = 0.f, f1 = 0.f;
    uint32_t u = 0;
};

struct BlendInput {
    int32_t source = -1;
    int32_t weight_param = -1;
    float weight_const = 0.f;
    float blend_time = 0.f;
};

enum NodeType { NODE_PLAY, NODE_BLEND, NODE_BY_INDEX, NODE_BLEND_SPACE };

struct PoseNodeDef {
    NodeType type = NODE_PLAY;
    uint32_t animation = 0;
    int32_t param = -1;
    std::vector<BlendInput> inputs;
    std::vector<float> points;       // BlendSpace xy
    std::vector<uint32_t> triangles;
    uint32_t by_index_slot = 0;      // index into per-instance ByIndex state
};

struct Action { int kind; uint32_t animation; std::vector<uint32_t> choices; };   // choices: EnableRandomAnimation's handles
struct StateDef { int32_t root = -1; std::vector<Action> on_enter, on_leave; };
struct TransitionDef { uint32_t source = 0, dest = 0; float time = 0.f; std::vector<int32_t> logic; };

struct LayerDef {
    float weight = 1.f;
    std::vector<PoseNodeDef> nodes;
    std::vector<StateDef> states;
    std::vector<TransitionDef> transitions;
    int32_t entry_state = -1;
    std::vector<int32_t> excluded;
    uint32_t by_index_count = 0;
};

// ---- per-instance machine state ----
struct TransitionState { float elapsed = 0.f, blend_factor = 0.f; };
struct ByIndexState { bool has_prev = false; uint32_t prev = 0; float blend_time = 0.f; };
struct LayerState {
    int32_t active_state = -1, active_transition = -1;
    std::vector<TransitionState> transitions;
    std::vector<ByIndexState> by_index;
    std::deque<fyx_layer_event> events;  // FixedEventQueue::new(2048), layer.rs:182
};
constexpr size_t kLayerEventLimit = 2048;
struct MachineState {
    std::vector<Param> params;
    std::vector<LayerState> layers;
};

// A recipe: what a pose node's output pose was made of at one evaluation.
struct Recipe {
    int32_t anim = -1;                  // >= 0: a copy of that animation's pose
    uint32_t first = 0, count = 0;      // else: fold of items[first .. first+count)
};
struct RecipeItem { uint32_t recipe; float w; };

// What planning a range of instances produces; one per planner thread, merged in instance order.
struct PlanScratch {
    std::vector<uint2> ops;
    std::vector<uint4> rm_ops;
    std::vector<uint32_t> prog_len, rm_prog_len;   // per instance of the range
    std::vector<Recipe> recipes;
    std::vector<RecipeItem> items;
    std::vector<int32_t> node_recipe;
    std::vector<uint8_t> seen;
    int error = 0;
};

struct Animator {
    uint64_t rig_id = 0;
    Rig* rig = nullptr;
    uint32_t n_instances = 0;
    std::vector<AnimationDef> anims;
    std::vector<AnimState> anim_state;  // [inst][anim]
    std::vector<Param> param_defaults;
    std::vector<LayerDef> layers;
    std::vector<MachineState> mstate;   // [inst]
    std::vector<uint64_t> rng;          // [inst] StateAction::EnableRandomAnimation's generator state (lazily sized)
    uint32_t max_tracks = 0;
    // device state
    AnimDev* d_anims = nullptr;
    bool anims_dirty = true;
    uint32_t* d_hints = nullptr;
    float4* d_anim_pose = nullptr;
    uint32_t dev_anim_capacity = 0, dev_track_capacity = 0;
    float4* d_node_trs = nullptr;
    float* d_local = nullptr;
    float* d_global = nullptr;
    uint8_t* d_layer_masks = nullptr;
    bool masks_dirty = true;
    uint32_t dev_mask_layers = 0;
    CtrlBuffers ctrl;   // per-frame control (device + pinned staging)
    // frame plan (host)
    std::vector<float> times;
    std::vector<uint8_t> ticked;
    std::vector<uint2> ops;
    std::vector<uint32_t> prog_off;
    // root motion (only when rm_enabled): per-frame slices + program, persistent device state
    bool rm_enabled = false;
    std::vector<float2> slices;
    std::vector<uint4> rm_ops;
    std::vector<uint32_t> rm_prog_off;
    // palettes the update kernel writes itself (fyx_animator_set_palette_output)
    struct PaletteOut { uint64_t bones_id; float* d_out; };
    std::vector<PaletteOut> palette_outputs;
    // Property{..} slots: one per distinct (node, property id) any animation of the animator drives
    std::vector<std::pair<int32_t, int32_t>> prop_slots;
    int32_t* d_prop_node = nullptr;
    PropRec* d_prop_pose = nullptr;    // [anim capacity][instance][slot]
    PropRec* d_prop_out = nullptr;     // [instance][slot]
    uint32_t dev_prop_slots = 0, dev_prop_anims = 0;
    std::vector<uint32_t> rm_layer_base;   // first slot of each layer; nodes, then the layer's final pose
    uint32_t n_rm_slots = 0;               // ... and the machine's final pose last
    RootMotionDev* d_rm_anim = nullptr;
    uint32_t dev_rm_anim_capacity = 0;
    float4* d_rm_slots = nullptr;
    uint32_t dev_rm_slots = 0;
    // scratch of the planner threads
    std::vector<PlanScratch> scratch;
};

}  // namespace

// The per-frame control block of an animator (what plan_frame produced), as it travels to the GPU: 256-byte aligned
// sections {times, ticked, prog_off, ops [, slices, rm_prog_off, rm_ops]}.
struct CtrlLayout {
    size_t o_tick = 0, o_off = 0, o_ops = 0, o_slices = 0, o_rmoff = 0, o_rmops = 0, total = 0;
    bool rm = false;
};

// fyx_scene_update's cached state: the block tables of the scene it last ran (they depend on the animators' shapes
// only) and the scene-wide control buffers.
struct SceneBatch {
    std::vector<uint64_t> signature;
    uint4* d_tables = nullptr;
    size_t table_off[kSceneStages] = {};
    uint32_t n_blocks[kSceneStages] = {};
    size_t lds_bytes[kSceneStages] = {};
    CtrlBuffers ctrl;
    std::vector<Animator*> animators;   // scratch of the current call
    std::vector<CtrlLayout> layouts;
    std::vector<size_t> offsets;
    std::vector<int> errors;
};

struct AnimStore {
    SceneBatch scene;
    std::unordered_map<uint64_t, TracksData> tracks;
    std::unordered_map<uint64_t, Rig> rigs;
    std::unordered_map<uint64_t, BoneList> bones;
    std::unordered_map<uint64_t, std::unique_ptr<Animator>> animators;
};

namespace {

bool has_device(const fyx_ctx* c) { return c->device >= 0; }

AnimStore& store(fyx_ctx* c) {
    if (!c->anim) c->anim = new AnimStore();
    return *c->anim;
}

void dfree(void* p) { if (p) (void)hipFree(p); }

void free_tracks(TracksData& t) { dfree(t.d_tracks); dfree(t.d_loc); dfree(t.d_aux); t = TracksData(); }
void free_rig(Rig& r) {
    dfree(r.d_parent); dfree(r.d_statics); dfree(r.d_level_nodes); dfree(r.d_level_start); dfree(r.d_node_level); dfree(r.d_inv_bind);
    r = Rig();
}
void free_bones(BoneList& b) { dfree(b.d_bone_nodes); b = BoneList(); }
void free_animator(Animator& a) {
    for (auto& an : a.anims) { dfree(an.d_slot_track); dfree(an.d_prop_track); }
    dfree(a.d_prop_node); dfree(a.d_prop_pose); dfree(a.d_prop_out);
    dfree(a.d_anims); dfree(a.d_hints); dfree(a.d_anim_pose); dfree(a.d_node_trs); dfree(a.d_local);
    dfree(a.d_global); dfree(a.d_layer_masks); dfree(a.d_rm_anim); dfree(a.d_rm_slots);
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

#define FYX_ANIMATOR(c, a, id)                                                                   \
    Animator* a = find_animator((c), (id));                                                      \
    if (!a) return fail((c), FYX_ERR_UNKNOWN_ID, "animator %llu is not registered", (unsigned long long)(id))

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

    // acc.blend_with(<pose described by recipe r>, w)
    void emit_blend(uint32_t r, float w) {
        const Recipe rc = S.recipes[r];
        if (rc.anim >= 0) { emit(OP_BLEND_ANIM, (uint32_t)rc.anim, w); return; }
        if (rc.count == 0) return;  // blending with an empty pose changes nothing
        if (depth + 1 >= kMaxFoldDepth) { error = FYX_ERR_UNSUPPORTED; return; }
        emit(OP_PUSH, 0, 0.f);
        ++depth;
        for (uint32_t i = 0; i < rc.count; ++i) {
            const RecipeItem it = S.items[rc.first + i];
            emit_blend(it.recipe, it.w);
        }
        --depth;
        emit(OP_POP_BLEND, 0, w);
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
                if (check[k] >= 0 && (size_t)check[k] < L.states.size()) collect(L, L.states[check[k]].root);
        }
        for (uint32_t a = 0; a < n_anims; ++a)
            if (S.seen[a] && as[a].enabled) tick(a);
        S.recipes.clear();
        S.items.clear();
        for (size_t li = 0; li < A.layers.size(); ++li) {
            emit(OP_PUSH, 0, 0.f);
            depth = 1;
            plan_layer((uint32_t)li);
            depth = 0;
            emit(OP_POP_BLEND, 0, A.layers[li].weight);
            if (rm()) rm_emit(RM_BLEND, machine_slot(), layer_slot((uint32_t)li), A.layers[li].weight);  // mod.rs:375-378
        }
        emit(OP_APPLY, 0, 0.f);
        emit(OP_END, 0, 0.f);
        if (rm()) rm_emit(RM_END, 0, 0, 0.f);
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
        m.layers.resize(A.layers.size());
        for (size_t l = 0; l < A.layers.size(); ++l) {
            LayerState& LS = m.layers[l];
            if (LS.active_state < 0 && LS.active_transition < 0) LS.active_state = A.layers[l].entry_state;
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

// Plans one frame of every instance of A.  Touches nothing but A (fyx_scene_update plans different animators on
// different threads); n_tasks > 1 splits the instances over `pool`.  Returns 0 or the planner's error code.
int plan_frame_core(Animator& A, int mode, float dt, unsigned n_tasks, PlanPool* pool) {
    const uint32_t na = (uint32_t)A.anims.size();
    A.times.assign((size_t)A.n_instances * na, 0.f);
    A.ticked.assign((size_t)A.n_instances * na, 0);
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
        for (uint32_t i = i0; i < i1; ++i) {
            const size_t o0 = S.ops.size(), r0 = S.rm_ops.size();
            Planner p(A, S, i, dt);
            if (mode == 1) p.plan_absm(); else p.plan_player();
            if (p.error) S.error = p.error;
            S.prog_len.push_back((uint32_t)(S.ops.size() - o0));
            S.rm_prog_len.push_back((uint32_t)(S.rm_ops.size() - r0));
        }
    };
    if (n_tasks > 1) pool->run(n_tasks, work); else work(0);
    uint32_t inst = 0;
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
        A.ops.insert(A.ops.end(), S.ops.begin(), S.ops.end());
        A.rm_ops.insert(A.rm_ops.end(), S.rm_ops.begin(), S.rm_ops.end());
    }
    A.prog_off[A.n_instances] = (uint32_t)A.ops.size();
    A.rm_prog_off[A.n_instances] = (uint32_t)A.rm_ops.size();
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
        FYX_HIP(c, hipStreamSynchronize(c->stream));
    }
    const uint32_t na = (uint32_t)A.anims.size();
    if (na > A.dev_anim_capacity || A.max_tracks > A.dev_track_capacity) {
        // grow pose records / hints; existing contents are preserved
        const uint32_t new_cap = std::max(na, A.dev_anim_capacity);
        const uint32_t new_tracks = std::max(A.max_tracks, A.dev_track_capacity);
        float4* np = nullptr;
        uint32_t* nh = nullptr;
        FYX_HIP(c, hipStreamSynchronize(c->stream));
        FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&np), std::max<size_t>((size_t)new_cap * in * 48, 16)));
        FYX_HIP(c, hipMemset(np, 0, std::max<size_t>((size_t)new_cap * in * 48, 16)));
        const size_t hb = std::max<size_t>((size_t)new_cap * A.n_instances * std::max(new_tracks, 1u) * 16, 16);
        FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&nh), hb));
        FYX_HIP(c, hipMemset(nh, 0, hb));
        if (A.d_anim_pose && A.dev_anim_capacity)
            FYX_HIP(c, hipMemcpy(np, A.d_anim_pose, (size_t)A.dev_anim_capacity * in * 48, hipMemcpyDeviceToDevice));
        if (A.d_hints && A.dev_anim_capacity && A.dev_track_capacity) {
            // layout [anim][track][curve][instance]: one row per animation, old rows are a prefix of the new ones
            const size_t old_row = (size_t)A.dev_track_capacity * 16 * A.n_instances;
            const size_t new_row = (size_t)new_tracks * 16 * A.n_instances;
            FYX_HIP(c, hipMemcpy2D(nh, new_row, A.d_hints, old_row, old_row, A.dev_anim_capacity, hipMemcpyDeviceToDevice));
        }
        dfree(A.d_anim_pose);
        dfree(A.d_hints);
        A.d_anim_pose = np;
        A.d_hints = nh;
        A.dev_anim_capacity = new_cap;
        A.dev_track_capacity = new_tracks;
        A.anims_dirty = true;
    }
    // slot tables + animation descriptors
    bool any_slots = false;
    for (AnimationDef& an : A.anims) any_slots |= an.slots_dirty;
    if (any_slots || A.anims_dirty) {
        FYX_HIP(c, hipStreamSynchronize(c->stream));
        std::vector<AnimDev> hd(na);
        for (uint32_t a = 0; a < na; ++a) {
            AnimationDef& an = A.anims[a];
            if (an.slots_dirty) {
                std::vector<int32_t> slots((size_t)rig.n_nodes * 4, -1);
                std::vector<int32_t> ptrack(std::max<size_t>(A.prop_slots.size(), 1), -1);
                for (uint32_t t = 0; t < an.td->n_tracks; ++t) {
                    if (an.target[t] < 0 || !an.enabled[t]) continue;
                    const int b = an.td->tracks[t].binding;
                    if (b >= FYX_BIND_PROPERTY0) {
                        if (an.td->tracks[t].n_curves < 1) continue;   // fetch() -> None
                        const std::pair<int32_t, int32_t> key(an.target[t], b - FYX_BIND_PROPERTY0);
                        const size_t sl = std::find(A.prop_slots.begin(), A.prop_slots.end(), key) - A.prop_slots.begin();
                        if (sl < ptrack.size()) ptrack[sl] = (int32_t)t;
                        slots[(size_t)an.target[t] * 4 + 3] = (int32_t)t;   // the node's pose is not empty
                        continue;
                    }
                    int32_t& s = slots[(size_t)an.target[t] * 4 + b];
                    if (s < 0) s = (int32_t)t;
                }
                if (an.dev_prop_slots < ptrack.size()) {
                    dfree(an.d_prop_track);
                    an.d_prop_track = nullptr;
                    FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&an.d_prop_track), std::max<size_t>(ptrack.size() * 4, 16)));
                    an.dev_prop_slots = (uint32_t)ptrack.size();
                }
                FYX_HIP(c, hipMemcpy(an.d_prop_track, ptrack.data(), ptrack.size() * 4, hipMemcpyHostToDevice));
                if (!an.d_slot_track)
                    FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&an.d_slot_track), std::max<size_t>(slots.size() * 4, 16)));
                FYX_HIP(c, hipMemcpy(an.d_slot_track, slots.data(), slots.size() * 4, hipMemcpyHostToDevice));
                an.slots_dirty = false;
            }
            hd[a].tracks = an.td->d_tracks;
            hd[a].key_loc = an.td->d_loc;
            hd[a].key_aux = an.td->d_aux;
            hd[a].slot_track = an.d_slot_track;
            hd[a].prop_track = an.d_prop_track;
            hd[a].n_tracks = an.td->n_tracks;
            hd[a].rm_node = an.rm_node;
            hd[a].rm_ignore = an.rm_ignore;
            hd[a].rm_pos_track = an.rm_pos_track;
            hd[a].rm_rot_track = an.rm_rot_track;
            hd[a].pad = 0;
        }
        dfree(A.d_anims);
        A.d_anims = nullptr;
        if (int rc = upload(c, &A.d_anims, hd.data(), hd.size())) return rc;
        A.anims_dirty = false;
    }
    const uint32_t nps = (uint32_t)A.prop_slots.size();
    if (nps && (A.dev_prop_slots != nps || A.dev_prop_anims < A.dev_anim_capacity)) {
        // property storage is re-created when slots or animations are added (values applied so far are kept per slot)
        FYX_HIP(c, hipStreamSynchronize(c->stream));
        std::vector<int32_t> nodes(nps);
        for (uint32_t k = 0; k < nps; ++k) nodes[k] = A.prop_slots[k].first;
        dfree(A.d_prop_node);
        A.d_prop_node = nullptr;
        if (int rc = upload(c, &A.d_prop_node, nodes.data(), nodes.size())) return rc;
        PropRec* np = nullptr;
        PropRec* no = nullptr;
        const size_t pb = (size_t)A.dev_anim_capacity * A.n_instances * nps * sizeof(PropRec);
        const size_t ob = (size_t)A.n_instances * nps * sizeof(PropRec);
        FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&np), std::max<size_t>(pb, 16)));
        FYX_HIP(c, hipMemset(np, 0, std::max<size_t>(pb, 16)));
        FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&no), std::max<size_t>(ob, 16)));
        FYX_HIP(c, hipMemset(no, 0, std::max<size_t>(ob, 16)));
        if (A.d_prop_out && A.dev_prop_slots)   // slots only ever get appended: old slot k is new slot k
            FYX_HIP(c, hipMemcpy2D(no, (size_t)nps * sizeof(PropRec), A.d_prop_out, (size_t)A.dev_prop_slots * sizeof(PropRec),
                                   (size_t)A.dev_prop_slots * sizeof(PropRec), A.n_instances, hipMemcpyDeviceToDevice));
        dfree(A.d_prop_pose);
        dfree(A.d_prop_out);
        A.d_prop_pose = np;
        A.d_prop_out = no;
        A.dev_prop_slots = nps;
        A.dev_prop_anims = A.dev_anim_capacity;
    }
    if (A.rm_enabled) {
        if (A.dev_rm_anim_capacity < A.dev_anim_capacity) {  // [anim][instance]: growing keeps the existing prefix
            RootMotionDev* nr = nullptr;
            const size_t nb = (size_t)A.dev_anim_capacity * A.n_instances * sizeof(RootMotionDev);
            FYX_HIP(c, hipStreamSynchronize(c->stream));
            FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&nr), std::max<size_t>(nb, 16)));
            FYX_HIP(c, hipMemset(nr, 0, std::max<size_t>(nb, 16)));
            if (A.d_rm_anim && A.dev_rm_anim_capacity)
                FYX_HIP(c, hipMemcpy(nr, A.d_rm_anim, (size_t)A.dev_rm_anim_capacity * A.n_instances * sizeof(RootMotionDev),
                                     hipMemcpyDeviceToDevice));
            dfree(A.d_rm_anim);
            A.d_rm_anim = nr;
            A.dev_rm_anim_capacity = A.dev_anim_capacity;
        }
        uint32_t want = 1;
        for (const LayerDef& L : A.layers) want += (uint32_t)L.nodes.size() + 1;
        if (want != A.dev_rm_slots) {  // the machine graph changed: every pose's root motion starts from None again
            FYX_HIP(c, hipStreamSynchronize(c->stream));
            dfree(A.d_rm_slots);
            A.d_rm_slots = nullptr;
            const size_t nb = (size_t)A.n_instances * want * 32;
            FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&A.d_rm_slots), nb));
            FYX_HIP(c, hipMemset(A.d_rm_slots, 0, nb));
            A.dev_rm_slots = want;
        }
    }
    if (A.masks_dirty || A.dev_mask_layers != A.layers.size()) {
        FYX_HIP(c, hipStreamSynchronize(c->stream));
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
    d.parent = r.d_parent;
    d.statics = r.d_statics;
    d.level_nodes = r.d_level_nodes;
    d.level_start = r.d_level_start;
    d.node_level = r.d_node_level;
    d.inv_bind = r.d_inv_bind;
    d.n_pal = 0;
    d.n_nodes = r.n_nodes;
    d.n_levels = r.n_levels;
    return d;
}

// The persistent part of an animator's kernel parameters.
void frame_static(const fyx_ctx* c, const Animator& A, PoseFrameDev& f) {
    memset(&f, 0, sizeof f);
    f.anims = A.d_anims;
    f.n_anims = (uint32_t)A.anims.size();
    f.n_instances = A.n_instances;
    f.n_nodes = A.rig->n_nodes;
    f.layer_masks = A.d_layer_masks;
    f.hints = A.d_hints;
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

CtrlLayout ctrl_layout(const Animator& A) {
    CtrlLayout L;
    L.rm = A.rm_enabled;
    L.o_tick = align_up(A.times.size() * 4, 256);
    L.o_off = L.o_tick + align_up(A.ticked.size(), 256);
    L.o_ops = L.o_off + align_up(A.prog_off.size() * 4, 256);
    L.o_slices = L.o_ops + align_up(A.ops.size() * 8, 256);
    L.o_rmoff = L.o_slices + (L.rm ? align_up(A.slices.size() * 8, 256) : 0);
    L.o_rmops = L.o_rmoff + (L.rm ? align_up(A.rm_prog_off.size() * 4, 256) : 0);
    L.total = L.o_rmops + (L.rm ? align_up(A.rm_ops.size() * 16, 256) : 0);
    return L;
}

void ctrl_write(const Animator& A, const CtrlLayout& L, char* h) {
    memcpy(h, A.times.data(), A.times.size() * 4);
    memcpy(h + L.o_tick, A.ticked.data(), A.ticked.size());
    memcpy(h + L.o_off, A.prog_off.data(), A.prog_off.size() * 4);
    memcpy(h + L.o_ops, A.ops.data(), A.ops.size() * 8);
    if (L.rm) {
        memcpy(h + L.o_slices, A.slices.data(), A.slices.size() * 8);
        memcpy(h + L.o_rmoff, A.rm_prog_off.data(), A.rm_prog_off.size() * 4);
        memcpy(h + L.o_rmops, A.rm_ops.data(), A.rm_ops.size() * 16);
    }
}

// Point the frame's parameters at the device copy of the control block.
void ctrl_bind(const Animator& A, const CtrlLayout& L, const char* d, PoseFrameDev& f) {
    f.times = reinterpret_cast<const float*>(d);
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

// The rig's parameters plus the palettes the update kernel writes itself.
int rig_params(fyx_ctx* c, const Animator& A, RigDev& rd) {
    rd = rig_dev(*A.rig);
    for (const Animator::PaletteOut& po : A.palette_outputs) {
        auto bit = store(c).bones.find(po.bones_id);
        if (bit == store(c).bones.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "bone list %llu of a palette output was freed", (unsigned long long)po.bones_id);
        PaletteOutDev& d = rd.pal[rd.n_pal++];
        d.bone_nodes = bit->second.d_bone_nodes;
        d.out = po.d_out;
        d.n_bones = bit->second.n_bones;
        d.pad = 0;
    }
    return FYX_OK;
}

// Send the planned frame to the GPU and run sample + update.
int run_frame(fyx_ctx* c, Animator& A, bool with_program) {
    if (int rc = enter_primary(c)) return rc;
    if (int rc = ensure_device_state(c, A)) return rc;
    PoseFrameDev f;
    frame_static(c, A, f);
    int slot = 0;
    if (with_program) {
        const CtrlLayout L = ctrl_layout(A);
        char *h = nullptr, *d = nullptr;
        if (int rc = ctrl_acquire(c, A.ctrl, L.total, &slot, &h, &d)) return rc;
        ctrl_write(A, L, h);
        if (int rc = ctrl_upload(c, A.ctrl, slot, L.total)) return rc;
        ctrl_bind(A, L, d, f);
        FYX_HIP(c, launch_pose_sample(f, c->stream));
        FYX_HIP(c, launch_property_sample(f, c->stream));
        if (L.rm) FYX_HIP(c, launch_root_motion(f, !A.rm_ops.empty(), c->stream));
    }
    RigDev rd;
    if (int rc = rig_params(c, A, rd)) return rc;
    FYX_HIP(c, launch_pose_update(f, rd, with_program, c->stream));
    if (with_program) {
        FYX_HIP(c, launch_property_update(f, c->stream));
        if (int rc = ctrl_consumed(c, A.ctrl, slot)) return rc;
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
    if (c->plan_threads > 1 && small.size() >= 32) n_tasks = std::min<unsigned>((unsigned)c->plan_threads, (unsigned)(small.size() / 16));
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

int scene_frame(fyx_ctx* c, SceneBatch& S, float dt) {
    const size_t n = S.animators.size();
    // 1. host control plane
    if (int rc = scene_plan(c, S, dt)) return rc;

    // 2. device state, and the block tables if the scene's shape changed
    if (int rc = enter_primary(c)) return rc;
    std::vector<uint64_t> sig;
    sig.reserve(n * 3 + 1);
    sig.push_back((uint64_t)c->sample_form);
    for (size_t k = 0; k < n; ++k) {
        Animator& A = *S.animators[k];
        if (int rc = ensure_device_state(c, A)) return rc;
        sig.push_back(((uint64_t)A.anims.size() << 32) | A.n_instances);
        sig.push_back(((uint64_t)A.rig->n_nodes << 32) | A.dev_prop_slots);
        sig.push_back(A.rm_enabled ? 1 : 0);
    }
    if (sig != S.signature) {
        std::vector<uint4> tables[kSceneStages];
        size_t lds[kSceneStages] = {};
        for (size_t k = 0; k < n; ++k) {
            const Animator& A = *S.animators[k];
            const SceneJobShape sh = scene_shape(c, A, A.dev_prop_slots);
            scene_blocks((uint32_t)k, sh, tables);
            const int stage = kStageUpdate64 + (int)std::min<uint32_t>((sh.n_nodes + 63) / 64, 4) - 1;
            lds[stage] = std::max(lds[stage], (size_t)sh.n_nodes * 32 * sizeof(float));
        }
        size_t total = 0;
        for (int k = 0; k < kSceneStages; ++k) {
            if (tables[k].size() > 0x7fffffffull) return fail(c, FYX_ERR_UNSUPPORTED, "scene too large for one launch per stage");
            S.table_off[k] = total;
            S.n_blocks[k] = (uint32_t)tables[k].size();
            S.lds_bytes[k] = lds[k];
            total += tables[k].size();
        }
        FYX_HIP(c, hipStreamSynchronize(c->stream));   // the previous scene's launches still read the old tables
        dfree(S.d_tables);
        S.d_tables = nullptr;
        S.signature.clear();
        FYX_HIP(c, hipMalloc(reinterpret_cast<void**>(&S.d_tables), std::max<size_t>(total, 1) * sizeof(uint4)));
        for (int k = 0; k < kSceneStages; ++k)
            if (!tables[k].empty())
                FYX_HIP(c, hipMemcpy(S.d_tables + S.table_off[k], tables[k].data(), tables[k].size() * sizeof(uint4), hipMemcpyHostToDevice));
        S.signature = sig;
    }

    // 3. one control block: the job array, then every animator's sections
    S.layouts.resize(n);
    S.offsets.resize(n);
    size_t total = align_up(n * sizeof(SceneJobDev), 256);
    for (size_t k = 0; k < n; ++k) {
        S.layouts[k] = ctrl_layout(*S.animators[k]);
        S.offsets[k] = total;
        total += S.layouts[k].total;
    }
    int slot = 0;
    char *h = nullptr, *d = nullptr;
    if (int rc = ctrl_acquire(c, S.ctrl, total, &slot, &h, &d)) return rc;
    SceneJobDev* jobs = reinterpret_cast<SceneJobDev*>(h);
    for (size_t k = 0; k < n; ++k) {
        const Animator& A = *S.animators[k];
        ctrl_write(A, S.layouts[k], h + S.offsets[k]);
        frame_static(c, A, jobs[k].f);
        ctrl_bind(A, S.layouts[k], d + S.offsets[k], jobs[k].f);
        if (int rc = rig_params(c, A, jobs[k].rig)) return rc;
    }
    if (int rc = ctrl_upload(c, S.ctrl, slot, total)) return rc;

    // 4. one launch per stage
    const uint4* tabs[kSceneStages];
    for (int k = 0; k < kSceneStages; ++k) tabs[k] = S.d_tables + S.table_off[k];
    FYX_HIP(c, launch_scene(reinterpret_cast<const SceneJobDev*>(d), tabs, S.n_blocks, S.lds_bytes, c->stream));
    return ctrl_consumed(c, S.ctrl, slot);
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
    free_ctrl(s->scene.ctrl);
    for (auto& kv : s->animators) free_animator(*kv.second);
    for (auto& kv : s->bones) free_bones(kv.second);
    for (auto& kv : s->rigs) free_rig(kv.second);
    for (auto& kv : s->tracks) free_tracks(kv.second);
    delete s;
}

}  // namespace fyx

using namespace fyx;

extern "C" {

int fyx_init_control_only(fyx_ctx** out_ctx) {
    if (!out_ctx) return FYX_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    FYX_GUARD_BEGIN
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
        if ((d.binding == FYX_BIND_ROTATION && !quat) || (d.binding != FYX_BIND_ROTATION && !vec3))
            return fail(c, FYX_ERR_UNSUPPORTED,
                        "track %u: value kind %d cannot be applied to binding %d (the reference logs an error and skips it)",
                        t, d.kind, d.binding);
        if (d.n_curves > 4) return fail(c, FYX_ERR_INVALID_ARG, "track %u has %u curves", t, d.n_curves);
        for (uint32_t k = 0; k < d.n_curves; ++k) total += d.curve_n_keys[k];
    }
    if (total != n_keys)
        return fail(c, FYX_ERR_INVALID_ARG, "tracks describe %llu keys but n_keys = %u", (unsigned long long)total, n_keys);
    for (uint32_t k = 0; k < n_keys; ++k)
        if (key_kind[k] > FYX_KEY_CUBIC) return fail(c, FYX_ERR_INVALID_ARG, "key %u has kind %u", k, key_kind[k]);
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
        int rc = upload(c, &td.d_tracks, hd.data(), hd.size());
        if (!rc) rc = upload(c, &td.d_loc, key_location, (size_t)n_keys);
        if (!rc) rc = upload(c, &td.d_aux, aux.data(), aux.size());
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
        int rc = upload(c, &r.d_parent, r.parent.data(), r.parent.size());
        if (!rc) rc = upload(c, &r.d_statics, statics.data(), statics.size());
        if (!rc) rc = upload(c, &r.d_level_nodes, level_nodes.data(), level_nodes.size());
        if (!rc) rc = upload(c, &r.d_level_start, level_start.data(), level_start.size());
        if (!rc) rc = upload(c, &r.d_node_level, depth.data(), depth.size());
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
    a->rig_id = rig_id;
    a->rig = &rit->second;
    a->n_instances = n_instances;
    store(c).animators.emplace(animator_id, std::move(a));
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
    std::vector<uint8_t> used((size_t)A->rig->n_nodes * 3, 0);
    std::vector<std::pair<int32_t, int32_t>> new_slots = A->prop_slots, seen_props;
    for (uint32_t t = 0; t < td.n_tracks; ++t) {
        an.target[t] = track_target[t];
        if (track_enabled) an.enabled[t] = track_enabled[t] ? 1 : 0;
        if (track_target[t] >= (int32_t)A->rig->n_nodes)
            return fail(c, FYX_ERR_INVALID_ARG, "track %u targets node %d of a %u-node rig", t, track_target[t], A->rig->n_nodes);
        if (track_target[t] >= 0 && td.tracks[t].binding >= FYX_BIND_PROPERTY0) {
            const std::pair<int32_t, int32_t> key(track_target[t], td.tracks[t].binding - FYX_BIND_PROPERTY0);
            if (std::find(seen_props.begin(), seen_props.end(), key) != seen_props.end())
                return fail(c, FYX_ERR_UNSUPPORTED, "two tracks drive the same property of node %d", track_target[t]);
            seen_props.push_back(key);
            if (std::find(new_slots.begin(), new_slots.end(), key) == new_slots.end()) new_slots.push_back(key);
        } else if (track_target[t] >= 0) {
            uint8_t& u = used[(size_t)track_target[t] * 3 + td.tracks[t].binding];
            if (u) return fail(c, FYX_ERR_UNSUPPORTED, "two tracks drive the same binding of node %d", track_target[t]);
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
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
    if (parameter >= A->param_defaults.size()) return fail(c, FYX_ERR_INVALID_ARG, "parameter %u does not exist", parameter);
    if (kind < FYX_PARAM_WEIGHT || kind > FYX_PARAM_SAMPLING_POINT) return fail(c, FYX_ERR_INVALID_ARG, "parameter kind %d", kind);
    Param p;
    p.kind = kind; p.f0 = f0; p.f1 = f1; p.u = u;
    if (instance == FYX_ALL_INSTANCES) {
        A->param_defaults[parameter] = p;
        for (MachineState& m : A->mstate) m.params[parameter] = p;
        return FYX_OK;
    }
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    ensure_machine_state(*A);
    A->mstate[instance].params[parameter] = p;
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
    if (L->entry_state < 0) {  // layer.rs:229-235
        L->entry_state = (int32_t)L->states.size() - 1;
        sync_machine_state(*A);
    }
    if (out_state) *out_state = (uint32_t)L->states.size() - 1;
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_layer_set_entry_state(fyx_ctx* c, uint64_t animator_id, uint32_t layer, uint32_t state) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (state >= L->states.size()) return fail(c, FYX_ERR_INVALID_ARG, "state %u does not exist", state);
    L->entry_state = (int32_t)state;
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
    FYX_ANIMATOR(c, A, animator_id);
    FYX_LAYER(c, A, L, layer);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    int32_t as = L->entry_state, at = -1;
    if (A->mstate.size() == A->n_instances) {
        as = A->mstate[instance].layers[layer].active_state;
        at = A->mstate[instance].layers[layer].active_transition;
    }
    if (active_state) *active_state = as;
    if (active_transition) *active_transition = at;
    return FYX_OK;
    FYX_GUARD_END(c)
}

// ---- per frame -----------------------------------------------------------------------------

static int update_common(fyx_ctx* c, uint64_t animator_id, int mode, float dt) {
    FYX_ANIMATOR(c, A, animator_id);
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
    S.animators.clear();
    std::unordered_set<uint64_t> seen;
    for (uint32_t k = 0; k < n_animators; ++k) {
        auto it = store(c).animators.find(animator_ids[k]);
        if (it == store(c).animators.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "animator %llu", (unsigned long long)animator_ids[k]);
        if (!seen.insert(animator_ids[k]).second)
            return fail(c, FYX_ERR_INVALID_ARG, "animator %llu is listed twice", (unsigned long long)animator_ids[k]);
        S.animators.push_back(it->second.get());
    }
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
    FYX_ANIMATOR(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    return run_frame(c, *A, false);
    FYX_GUARD_END(c)
}

int fyx_animator_palette(fyx_ctx* c, uint64_t animator_id, uint64_t bones_id, float* d_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    auto bit = store(c).bones.find(bones_id);
    if (bit == store(c).bones.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "bone list %llu is not registered", (unsigned long long)bones_id);
    if (bit->second.rig_id != A->rig_id) return fail(c, FYX_ERR_INVALID_ARG, "bone list belongs to another rig");
    if (!d_out) return fail(c, FYX_ERR_INVALID_ARG, "d_out_palette is null");
    if (int rc = enter_primary(c)) return rc;
    if (int rc = ensure_device_state(c, *A)) return rc;
    FYX_HIP(c, launch_palette_gather(A->d_global, A->rig->d_inv_bind, bit->second.d_bone_nodes, A->rig->n_nodes,
                                     bit->second.n_bones, A->n_instances, d_out, c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_set_palette_output(fyx_ctx* c, uint64_t animator_id, uint64_t bones_id, float* d_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    auto bit = store(c).bones.find(bones_id);
    if (bit == store(c).bones.end()) return fail(c, FYX_ERR_UNKNOWN_ID, "bone list %llu is not registered", (unsigned long long)bones_id);
    if (bit->second.rig_id != A->rig_id) return fail(c, FYX_ERR_INVALID_ARG, "bone list belongs to another rig");
    if (reinterpret_cast<uintptr_t>(d_out) & 15u) return fail(c, FYX_ERR_INVALID_ARG, "palette output must be 16-byte aligned");
    auto& v = A->palette_outputs;
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i].bones_id == bones_id) {
            if (d_out) v[i].d_out = d_out; else v.erase(v.begin() + (long)i);
            return FYX_OK;
        }
    if (!d_out) return FYX_OK;
    if (v.size() >= (size_t)kMaxPaletteOutputs)
        return fail(c, FYX_ERR_UNSUPPORTED, "at most %d palette outputs per animator (use fyx_animator_palette for more)", kMaxPaletteOutputs);
    v.push_back(Animator::PaletteOut{bones_id, d_out});
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_set_local_trs(fyx_ctx* c, uint64_t animator_id, uint32_t node, uint32_t first_instance,
                               uint32_t n_instances, const float* trs) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
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
    if (what >= FYX_READ_ANIMATION_POSE && (size_t)(what - FYX_READ_ANIMATION_POSE) < A->anims.size()) {
        *ptr = reinterpret_cast<char*>(A->d_anim_pose) + (size_t)(what - FYX_READ_ANIMATION_POSE) * in * 48;
        *bytes = in * 48;
        return FYX_OK;
    }
    return fail(c, FYX_ERR_INVALID_ARG, "unknown array selector %d", what);
}

int fyx_animator_read(fyx_ctx* c, uint64_t animator_id, int what, float* host_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
    if (instance >= A->n_instances) return fail(c, FYX_ERR_INVALID_ARG, "instance %u out of range", instance);
    if (!out_count) return fail(c, FYX_ERR_INVALID_ARG, "out_count is null");
    return for_instances(c, A, animation, instance, [&](AnimState& s) { *out_count = (uint32_t)s.events.size(); });
    FYX_GUARD_END(c)
}
int fyx_animation_clear_events(fyx_ctx* c, uint64_t animator_id, uint32_t animation, uint32_t instance) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (animation >= A->anims.size() || A->anims[animation].removed) return fail(c, FYX_ERR_INVALID_ARG, "animation %u does not exist", animation);
    if (!host_out) return fail(c, FYX_ERR_INVALID_ARG, "host_out is null");
    if (!A->rm_enabled) return fail(c, FYX_ERR_INVALID_ARG, "root motion is not tracked on this animator");
    if (int rc = enter_primary(c)) return rc;
    if (int rc = ensure_device_state(c, *A)) return rc;
    static_assert(sizeof(fyx_root_motion) == 32, "fyx_root_motion layout");
    // the first 32 bytes of a RootMotionDev are exactly a fyx_root_motion
    FYX_HIP(c, hipMemcpy2DAsync(host_out, sizeof(fyx_root_motion), A->d_rm_anim + (size_t)animation * A->n_instances,
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
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
    *out_count = (uint32_t)A->prop_slots.size();
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_property_slot(fyx_ctx* c, uint64_t animator_id, int32_t node, int32_t property_id, int32_t* out_slot) {
    if (!c || !out_slot) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    const std::pair<int32_t, int32_t> key(node, property_id);
    const auto it = std::find(A->prop_slots.begin(), A->prop_slots.end(), key);
    *out_slot = it == A->prop_slots.end() ? -1 : (int32_t)(it - A->prop_slots.begin());
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_read_properties(fyx_ctx* c, uint64_t animator_id, int32_t animation, fyx_property_value* host_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
    if (!has_device(c)) return fail(c, FYX_ERR_NO_DEVICE, "control-only context");
    if (!host_out) return fail(c, FYX_ERR_INVALID_ARG, "host_out is null");
    if (animation >= (int32_t)A->anims.size() || (animation >= 0 && A->anims[animation].removed))
        return fail(c, FYX_ERR_INVALID_ARG, "animation %d does not exist", animation);
    if (A->prop_slots.empty()) return FYX_OK;
    if (int rc = enter_primary(c)) return rc;
    if (int rc = ensure_device_state(c, *A)) return rc;
    const size_t per = (size_t)A->n_instances * A->dev_prop_slots;
    static_assert(sizeof(fyx_property_value) == sizeof(PropRec), "same record on both sides of the boundary");
    const PropRec* src = animation < 0 ? A->d_prop_out : A->d_prop_pose + (size_t)animation * per;
    FYX_HIP(c, hipMemcpyAsync(host_out, src, per * sizeof(PropRec), hipMemcpyDeviceToHost, c->stream));
    FYX_HIP(c, hipStreamSynchronize(c->stream));
    return FYX_OK;
    FYX_GUARD_END(c)
}

int fyx_animator_blend_shape_weights(fyx_ctx* c, uint64_t animator_id, uint32_t n_shapes, const int32_t* slots,
                                     const float* default_weights, float* d_out) {
    if (!c) return FYX_ERR_INVALID_ARG;
    FYX_GUARD_BEGIN
    FYX_ANIMATOR(c, A, animator_id);
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
    FYX_ANIMATOR(c, A, animator_id);
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
