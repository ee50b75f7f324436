// anim_model.h -- host-side model of the pose path: tracks data, rigs, the structure of an animator (animations, machine
// layers, pose nodes), per-instance state, and the context's registries.  Included by anim_api.hip only.
#pragma once

namespace fyx {

namespace {

struct TracksData {
    uint32_t n_tracks = 0;
    std::vector<fyx_track_desc> tracks;
    TrackDev* d_tracks = nullptr;
    float* d_loc = nullptr;
    float4* d_aux = nullptr;
    KeyRec* d_rec = nullptr;
    TrackHot* d_hot = nullptr;
    float4* d_spans = nullptr;
    // The same span records a second time, laid out [span][track] for the tracks that share their key times with the first such track
    // (every importer-made clip: one time grid for all bones): the per-instance sampler reads ONE span of every track of a clip per
    // frame, and in this order those are one dense run instead of a scattered line per track (round 6).  row_first[t]: track t's record
    // of span 0 in d_span_rows, or kNoSpans (the track keeps its own table); row_stride: f4 from one span's row to the next.
    float4* d_span_rows = nullptr;
    std::vector<uint32_t> row_first;
    uint32_t row_stride = 0;
    std::vector<TrackHot> hot;     // host copy of d_hot (the animators' crowd descriptors are made from it)
    std::vector<uint32_t> n_keys0; // keys of every track's first curve
};

struct Rig {
    uint32_t n_nodes = 0, n_levels = 0;
    std::vector<int32_t> parent;
    std::vector<float> init_trs;  // [n_nodes][12]
    float* d_statics = nullptr;
    uint32_t* d_walk = nullptr;          // RigDev::walk
    std::vector<uint32_t> walk;          // host copy of it (fyx_debug_rig_walk)
    std::vector<uint32_t> chunks;        // [n_chunks][16] the wide walk's entries (behind the walk words in d_walk; fyx_debug_rig_chunks)
    uint32_t n_chunks = 0;
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
    std::vector<int32_t> slots;    // host copy of d_slot_track
    int32_t* d_prop_track = nullptr;   // [animator's property slots]
    uint32_t dev_prop_slots = 0;
    bool slots_dirty = true;
    // A node's pose is a LIST of values (pose.rs:107-121): several enabled tracks may feed one binding of one node, and a track's
    // kind may fit no binding (a Real track bound to Position).  What can be observed of such a list is two values per binding:
    // the one the pose APPLIES -- the last value whose kind fits, scene/animation/mod.rs:147-186 -- and the one a blend READS when
    // this pose is the `other` -- the first value of the binding, whatever its kind (value.rs:438-444; a kind that differs blends
    // with nothing).  `slots` / d_slot_track / d_prop_track are the APPLY view; these are the READ view, and `dup` says the two
    // differ somewhere (then a machine's folds keep two records per animation: Animator::shadows).
    std::vector<int32_t> slots_f;      // [n_nodes][4]: first track of the binding if its kind fits, else -1 (entry 3 as in `slots`)
    std::vector<uint8_t> blockers;     // [n_nodes] bit b: the binding's FIRST value has a kind that does not fit; bit 3: some value of the node fits no binding
    std::vector<uint8_t> multi;        // [n_nodes] bit b: TWO OR MORE values of binding b fit (update_root_motion walks every one of them: the first
                                       //   takes the remainders of the last loop, lib.rs:575-578 / :634-637, the one that stays -- the last -- finds None)
    int32_t* d_slot_track_f = nullptr;
    int32_t* d_prop_track_f = nullptr;
    bool dup = false;
    bool maybe_dup = false;            // (host, set when tracks are bound: two tracks on one (node, binding) or a kind that fits no binding -- the
                                       //   animator then stays off the launches that have no two-record fold: one-launch frames, scenes)
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

// VecDeque<AnimationEvent> (signal indices) of one (instance, animation): almost always empty, so it is ONE pointer in the state record
// (a crowd's planning walks 4000 of these records per frame: 40 bytes each instead of 104 with the deque in place).
struct EventQueue {
    std::deque<int32_t>* q = nullptr;
    EventQueue() = default;
    EventQueue(const EventQueue& o) : q(o.q ? new std::deque<int32_t>(*o.q) : nullptr) {}
    EventQueue(EventQueue&& o) noexcept : q(o.q) { o.q = nullptr; }
    EventQueue& operator=(EventQueue o) noexcept { std::swap(q, o.q); return *this; }
    ~EventQueue() { delete q; }
    size_t size() const { return q ? q->size() : 0; }
    bool empty() const { return !q || q->empty(); }
    void push_back(int32_t v) { if (!q) q = new std::deque<int32_t>(); q->push_back(v); }
    int32_t front() const { return q->front(); }
    void pop_front() { q->pop_front(); }
    void clear() { if (q) q->clear(); }
    static const std::deque<int32_t>& none() { static const std::deque<int32_t> n; return n; }
    std::deque<int32_t>::const_iterator begin() const { return q ? q->begin() : none().begin(); }
    std::deque<int32_t>::const_iterator end() const { return q ? q->end() : none().end(); }
};

struct AnimState {  // per instance, per animation (Animation's scalar fields)
    float time = 0.f, speed = 1.f, start = 0.f, end = 0.f;
    uint8_t enabled = 1, looped = 1;
    uint32_t max_event_capacity = 32;   // lib.rs:941
    EventQueue events;                  // VecDeque<AnimationEvent>, as signal indices
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
    int32_t entry_state = -1;       // MachineLayer::entry_state: set by set_entry_state ONLY (layer.rs:209-212); reset() returns to it
    int32_t initial_active = -1;    // what active_state is on a machine that has only been built: add_state makes the first
                                    // state active when none is (layer.rs:229-235), set_entry_state overrides
    std::vector<int32_t> excluded;
    uint32_t by_index_count = 0;
};

// ---- per-instance machine state ----
struct TransitionState { float elapsed = 0.f, blend_factor = 0.f; };
struct ByIndexState { bool has_prev = false; uint32_t prev = 0; float blend_time = 0.f; };
struct LayerState {
    int32_t active_state = -1, active_transition = -1;
    int32_t memo_state = -2;             // active_state the instance's memoised fold program was planned in (see MachineState)
    std::vector<TransitionState> transitions;
    std::vector<ByIndexState> by_index;
    std::deque<fyx_layer_event> events;  // FixedEventQueue::new(2048), layer.rs:182
};
constexpr size_t kLayerEventLimit = 2048;
struct MachineState {
    std::vector<Param> params;
    std::vector<LayerState> layers;
    // Fold-program memo: the program planned last frame is this frame's too when nothing it depends on has changed --
    // same active states, no transition active then or now and none firing now, no API call on the animator in between
    // (Animator::edit_gen), no pose node with state of its own (BlendAnimationsByIndex).  Then planning the instance is
    // its animations' ticks, the transition conditions and a copy of last frame's ops.
    uint64_t memo_gen = 0;
    bool memo_valid = false;
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
    bool all_straight = false;      // every program of the range is straight (classify_fold_program_host)
};

struct Animator {
    uint64_t id = 0;
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
    // Two device animations per animation (AnimationDef::dup in an animator with a machine): 2 a the apply view, 2 a + 1 the read view.
    // Sticky once set (records, hints and root-motion state are laid out [device animation]...).
    bool shadows = false;
    uint32_t n_dev_anims() const { return (uint32_t)anims.size() * (shadows ? 2u : 1u); }
    std::vector<float> x_times;             // the frame's control sections per DEVICE animation (shadows only; run_frame fills them)
    std::vector<uint8_t> x_ticked;
    std::vector<float2> x_slices;
    std::vector<uint2> x_ops;
    std::vector<uint4> x_rm_ops;
    // device state
    AnimDev* d_anims = nullptr;
    CrowdDesc* d_crowd = nullptr;   // [anims][nodes][3]
    int desc_form = -1;             // the sampler form the descriptors were made for (1 per instance: span rows; 0 crowd: per-track tables)
    bool anims_dirty = true;
    uint32_t* d_hints = nullptr;
    uint32_t* d_slot_hints = nullptr;       // PoseFrameDev::slot_hints (hints are advisory: a fresh array of zeros is "no hint")
    size_t slot_hint_words = 0;
    float4* d_anim_pose = nullptr;
    uint32_t dev_anim_capacity = 0, dev_track_capacity = 0;
    float4* d_node_trs = nullptr;
    uint32_t* d_frame_counter = nullptr;   // FrameSync::counter: kFrameCounterReplicas words, kFrameCounterStride apart (own allocation)
    uint32_t frame_counter_total = 0;      // what it reaches when every launch issued so far has run
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
    bool all_straight = false;              // every instance's program of this frame is straight: the update kernel without the interpreter
    std::vector<uint2> prev_ops;            // last frame's programs (the memo's source)
    std::vector<uint32_t> prev_prog_off;
    std::vector<std::vector<std::vector<uint32_t>>> state_anims;   // [layer][state]: animations its pose tree plays
    uint64_t state_anims_gen = 0;           // edit_gen the lists were made at
    int prev_mode = -2;                     // mode of the frame planned last
    uint64_t edit_gen = 1;                  // bumped by every API call on the animator other than update / plan
    bool memo_static_ok = false;            // no layer has a BlendAnimationsByIndex node, no root motion (set per frame)
    // The STEADY frame (plan_frame_core): every instance's memo was valid when the last machine frame was planned and no API call
    // has touched the animator since -- then a frame is its animations' ticks and the transition conditions that can change with
    // time; the programs, their offsets and the kernel form are last frame's and stay where they are.
    uint64_t steady_gen = 0;                // edit_gen the lists below were made at (0: not steady)
    std::vector<uint32_t> steady_ticks;     // [instance]: offset into steady_list of the animations the instance ticks (+ 1 entry: the end)
    std::vector<uint32_t> steady_list;
    std::vector<uint32_t> steady_timed;     // instances with an outgoing transition whose condition reads an animation's clock
    bool steady_all = false;                // every instance ticks every animation and none has signals: no lists, a fixed inner loop
    // root motion (only when rm_enabled): per-frame slices + program, persistent device state
    bool rm_enabled = false;
    std::vector<float2> slices;
    std::vector<uint4> rm_ops;
    std::vector<uint32_t> rm_prog_off;
    // palettes the update kernel writes itself (fyx_animator_set_palette_output)
    // (d_bone_nodes / n_bones: the registered bone list's, kept here -- a bone list that is a palette output cannot be freed, and a
    // scene of 256 characters looked each of them up in the store every frame)
    struct PaletteOut {
        uint64_t bones_id; float* d_out; const int32_t* d_bone_nodes; uint32_t n_bones;
        // fyx_animator_set_palette_output_pair: the buffer of the frames that run on the SECOND frame stream under anim.overlap (frames
        // alternate between two streams, and so between the two buffers: frame n + 1's update does not write what frame n's skinning reads)
        float* d_out_alt = nullptr;
    };
    std::vector<PaletteOut> palette_outputs;
    // Which buffer of a palette pair the animator's MOST RECENT frame wrote (run_frame / scene_frame record it): 1 = the second.  The
    // context's frame_idx answers that for the frame in progress only -- every pose entry toggles it, other animators' updates included.
    int last_frame_alt = 0;
    // ... and how that frame was issued: 0 none yet, 1 run_frame with the planned programs, 2 run_frame without (update_transforms),
    // 3 as a member of the scene (reissue_frame runs it again the same way, minus the in-grid waits)
    int last_frame_kind = 0;
    // meshes every update of the animator skins itself (fyx_animator_set_skin_output): with the palette of `bones_id` -- which is
    // one of palette_outputs -- as fyx_lbs_skin_device(mesh_id, that palette, n_bones, n_instances, outputs) right behind the update
    struct SkinOut { uint64_t bones_id, mesh_id; float* d_pos; float* d_nrm; float* d_tan; };
    std::vector<SkinOut> skin_outputs;
    uint64_t prog_gen = 1;                  // bumped by every frame that plans programs (every frame but the steady ones): a scene's steady frames
                                            //   find the control sections other than clocks and tick flags where their last full write left them
    uint64_t api_gen = 1;                   // bumped by every API call that may change what the animator's device-side parameters hold
    // the skin outputs as a scene's update launch takes them along (FrameSkin): made when api_gen, the context's meshes or the launch
    // options changed, not every frame
    FrameSkin scene_skin;
    uint64_t scene_skin_api_gen = 0, scene_skin_mesh_gen = 0;
    int scene_skin_units = -1;
    bool scene_skin_ok = false;             // false: the animator's skin outputs are skinned by launches of their own behind the scene's
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
    std::vector<uint32_t> dev_rm_layer_nodes;   // pose nodes per layer of the layout d_rm_slots was made for
    // scratch of the planner threads
    std::vector<PlanScratch> scratch;
};

}  // namespace

// The per-frame control block of an animator (what plan_frame produced), as it travels to the GPU: 256-byte aligned
// sections {times, ticked, prog_off, ops [, slices, rm_prog_off, rm_ops]}.
struct CtrlLayout {
    size_t o_times = 0, o_tick = 0, o_off = 0, o_ops = 0, o_slices = 0, o_rmoff = 0, o_rmops = 0, total = 0;
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
    bool wide_update = false;              // the 256-thread update stage runs the wide hierarchy walk (its LDS is sized for it)
    bool one_frame = false;                // the scene runs as ONE launch (scene_frame_kernel)
    bool skin_update = false;              // ... and also holds the animators' skinning workgroups (pose_update_skin_scene_kernel)
    CtrlBuffers ctrl;
    // this frame's job array / the one the device holds -- one per frame stream: under anim.overlap the frames of the two streams
    // differ in the palette buffers they write (palette pairs), and each keeps its array resident
    std::vector<char> h_jobs, sent_jobs[2];
    char* d_jobs[2] = {nullptr, nullptr};
    size_t d_jobs_capacity[2] = {0, 0};
    // A frame in which nothing the job array and the launch plans are made from has changed since the last frame that made them -- same
    // members, no API call on any of them, same options and meshes -- goes straight to "write the control block, upload, launch": what it
    // skips cost a scene of 256 characters ~15 us of the calling thread per frame.  Every animator's control section has a CAPACITY
    // (its size plus a quarter) and the sections lie capacity after capacity, so an animator whose programs were planned again (a
    // transition: prog_gen) rewrites its own section where it is and moves nobody else's -- the job array holds offsets into the block;
    // the animators whose frame was a steady one write clocks and tick flags only (the rest of their section is where the slot's last
    // full write left it).  static_gen counts the states; an animator that tracks root motion or properties ends eligibility.
    struct Seen { uint64_t api_gen, prog_gen; };
    std::vector<Seen> seen;
    std::vector<size_t> caps;               // [member]: bytes of its control section's place
    uint64_t members_epoch = 1, seen_members_epoch = 0, seen_options_gen = 0, seen_mesh_gen = 0;
    uint64_t static_gen = 0;
    uint64_t jobs_gen[2] = {0, 0};          // static_gen d_jobs[frame stream] was made at
    uint64_t slot_gen[2] = {0, 0};          // static_gen control slot k's sections were laid out at ...
    std::vector<uint64_t> slot_prog[2];     // ... and [member]: the prog_gen of the programs its section holds there
    uint64_t skin_jobs_gen[2] = {0, 0};     // static_gen of skin_jobs_of[frame stream]
    std::vector<fyx_skin_job> skin_jobs_of[2];
    bool fast_eligible = false;             // no member tracks root motion or properties, and the scene is not the one-launch form
    bool single_palette = false;            // a member has a palette output without a second buffer (anim.overlap = 2: its frames wait for all earlier skinning)
    size_t ctrl_total = 0, o_targets = 0;
    std::vector<Animator*> animators;   // the members of the current call ...
    std::vector<uint64_t> member_ids;   // ... which are the previous call's when the id list and the store's set of animators are (members_gen)
    uint64_t members_gen = 0;
    std::vector<CtrlLayout> layouts;
    std::vector<size_t> offsets;
    std::vector<int> errors;
};

struct AnimStore {
    uint64_t animators_gen = 1;          // bumped when an animator is created or freed (a scene's cached member pointers)
    SceneBatch scene;
    std::unordered_map<uint64_t, TracksData> tracks;
    std::unordered_map<uint64_t, Rig> rigs;
    std::unordered_map<uint64_t, BoneList> bones;
    std::unordered_map<uint64_t, std::unique_ptr<Animator>> animators;
};

}  // namespace fyx
