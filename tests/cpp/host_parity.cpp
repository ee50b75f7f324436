// A compiled (C++) host driving libfyrox_hip.so through the C ABI only -- the way the engine's Rust side would --
// and checking it against the CPU oracle (test infrastructure, linked into THIS test binary only).
//
//   host_parity --control-only   no GPU needed: builds an AnimationPlayer-style animator, plans frames through the
//                                host control plane and compares the sample times / clocks with the oracle's
//                                Animation::tick; a two-state Machine re-sent four times (fyx_machine_clear + builder
//                                calls + run-time state put back), three of them inside a transition, stays in step
//                                with the oracle's untouched machine; every data-path call must answer
//                                FYX_ERR_NO_DEVICE.
//   host_parity                  BASELINE config C1 on the GPU: one SurfaceData of 1 k AnimatedVertex vertices /
//                                4 bones, fyx_mesh_upload + fyx_lbs_skin + fyx_skinned_aabb, bit-exact vs the oracle;
//                                then a 4-bone clip through fyx_animation_player_update -> fyx_animator_palette.
// Exit code 0 = pass, 1 = mismatch, 77 = no GPU (GPU mode only).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/fyrox_hip.h"
#include "../../oracle/fyrox_oracle.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int rc_ = (call);                                                                    \
        if (rc_ != 0) { std::printf("FAIL %s -> %d (%s)\n", #call, rc_, fyx_last_error(ctx)); return 1; } \
    } while (0)

static uint64_t g_state = 0x5EED0001ull;
static float urand() {  // splitmix64 -> [0, 1)
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)((z >> 40) * (1.0 / 16777216.0));
}

struct Clip {  // one Position + one Rotation (quaternion) + one Scale track per bone, Linear keys
    std::vector<fyx_track_desc> tracks;
    std::vector<float> loc, val;
    std::vector<uint8_t> kind;
    std::vector<int32_t> target;
};

static Clip make_clip(int n_bones, int n_keys) {
    Clip c;
    for (int b = 0; b < n_bones; ++b)
        for (int t = 0; t < 3; ++t) {
            fyx_track_desc d;
            std::memset(&d, 0, sizeof d);
            d.binding = t == 0 ? FYX_BIND_POSITION : t == 1 ? FYX_BIND_ROTATION : FYX_BIND_SCALE;
            d.kind = t == 1 ? FYX_KIND_QUAT : FYX_KIND_VEC3;
            d.n_curves = t == 1 ? 4 : 3;
            for (uint32_t k = 0; k < d.n_curves; ++k) {
                d.curve_n_keys[k] = (uint32_t)n_keys;
                for (int i = 0; i < n_keys; ++i) {
                    c.loc.push_back((float)i / (float)(n_keys - 1));
                    c.val.push_back(t == 2 ? 1.0f + 0.1f * urand() : urand() - 0.5f);
                    c.kind.push_back(FYX_KEY_LINEAR);
                }
            }
            c.tracks.push_back(d);
            c.target.push_back(b);
        }
    return c;
}

static fo_tracks* oracle_tracks(const Clip& c) {
    fo_tracks* td = fo_tracks_new();
    size_t key = 0;
    for (const fyx_track_desc& d : c.tracks) {
        fo_curve cv[4];
        std::vector<std::vector<float>> zeros(4);
        for (uint32_t k = 0; k < d.n_curves; ++k) {
            zeros[k].assign(d.curve_n_keys[k], 0.0f);
            cv[k].n_keys = d.curve_n_keys[k];
            cv[k].location = &c.loc[key];
            cv[k].value = &c.val[key];
            cv[k].kind = &c.kind[key];
            cv[k].left_tangent = zeros[k].data();
            cv[k].right_tangent = zeros[k].data();
            key += d.curve_n_keys[k];
        }
        fo_tracks_add_track(td, d.binding, d.kind, d.n_curves, cv);
    }
    return td;
}

static int build_animator(fyx_ctx* ctx, const Clip& clip, int n_bones, uint32_t n_instances) {
    std::vector<int32_t> parent(n_bones);
    std::vector<fyx_transform> tr(n_bones);
    for (int b = 0; b < n_bones; ++b) {
        parent[b] = b - 1;
        std::memset(&tr[b], 0, sizeof tr[b]);
        tr[b].local_rotation[3] = 1.0f;
        tr[b].pre_rotation[3] = 1.0f;
        tr[b].local_scale[0] = tr[b].local_scale[1] = tr[b].local_scale[2] = 1.0f;
        tr[b].post_rotation_matrix[0] = tr[b].post_rotation_matrix[4] = tr[b].post_rotation_matrix[8] = 1.0f;
    }
    CHECK(fyx_tracks_data_upload(ctx, 10, (uint32_t)clip.tracks.size(), clip.tracks.data(), (uint32_t)clip.loc.size(),
                                 clip.loc.data(), clip.val.data(), clip.kind.data(), nullptr, nullptr));
    CHECK(fyx_rig_create(ctx, 1, (uint32_t)n_bones, parent.data(), tr.data(), nullptr));
    CHECK(fyx_animator_create(ctx, 2, 1, n_instances));
    uint32_t anim = 0;
    CHECK(fyx_animator_add_animation(ctx, 2, 10, clip.target.data(), nullptr, &anim));
    CHECK(fyx_animation_set_time_slice(ctx, 2, anim, FYX_ALL_INSTANCES, 0.0f, 1.0f));
    CHECK(fyx_animation_set_speed(ctx, 2, anim, FYX_ALL_INSTANCES, 1.75f));
    return 0;
}

// idle <-> walk on a Rule parameter (0.2 s transitions), the way an engine-side shim sends a Machine -- and sends it AGAIN
// after the game has edited it (fyx_machine_clear + builder calls + the run-time state put back): here the definition is
// unchanged and re-sent in the middle of a transition, so the oracle's untouched machine must stay in step.
static int send_machine(fyx_ctx* ctx, uint64_t id) {
    uint32_t par = 0, layer = 0, n0 = 0, n1 = 0, s0 = 0, s1 = 0, t = 0;
    CHECK(fyx_machine_add_parameter(ctx, id, FYX_PARAM_RULE, 0.0f, 0.0f, 0, &par));
    CHECK(fyx_machine_add_layer(ctx, id, 1.0f, &layer));
    CHECK(fyx_layer_add_play_animation(ctx, id, layer, 0, &n0));
    CHECK(fyx_layer_add_play_animation(ctx, id, layer, 1, &n1));
    CHECK(fyx_layer_add_state(ctx, id, layer, (int32_t)n0, &s0));
    CHECK(fyx_layer_add_state(ctx, id, layer, (int32_t)n1, &s1));
    const int32_t go[2] = {FYX_LOGIC_PARAMETER, (int32_t)par}, back[3] = {FYX_LOGIC_NOT, FYX_LOGIC_PARAMETER, (int32_t)par};
    CHECK(fyx_layer_add_transition(ctx, id, layer, s0, s1, 0.2f, go, 2, &t));
    CHECK(fyx_layer_add_transition(ctx, id, layer, s1, s0, 0.2f, back, 3, &t));
    return 0;
}

static int machine_rebuilt_mid_transition(fyx_ctx* ctx, const Clip& clip, fo_tracks* td) {
    const uint64_t id = 3;
    const uint32_t n_inst = 2;
    CHECK(fyx_animator_create(ctx, id, 1, n_inst));
    fo_animation* oa[2];
    for (int a = 0; a < 2; ++a) {
        uint32_t index = 0;
        CHECK(fyx_animator_add_animation(ctx, id, 10, clip.target.data(), nullptr, &index));
        CHECK(fyx_animation_set_speed(ctx, id, index, FYX_ALL_INSTANCES, a ? 2.5f : 0.75f));
        oa[a] = fo_animation_new(td);
        for (size_t t = 0; t < clip.target.size(); ++t) fo_animation_bind(oa[a], (int)t, clip.target[t], 1);
        fo_animation_set_speed(oa[a], a ? 2.5f : 0.75f);
    }
    if (send_machine(ctx, id)) return 1;
    fo_machine* om = fo_machine_new();
    fo_machine_add_parameter(om, FYX_PARAM_RULE, 0.0f, 0.0f, 0);
    fo_machine_add_layer(om, 1.0f);
    fo_layer_add_play(om, 0, 0);
    fo_layer_add_play(om, 0, 1);
    fo_layer_add_state(om, 0, 0);
    fo_layer_add_state(om, 0, 1);
    const int go[2] = {FYX_LOGIC_PARAMETER, 0}, back[3] = {FYX_LOGIC_NOT, FYX_LOGIC_PARAMETER, 0};
    fo_layer_add_transition(om, 0, 0, 1, 0.2f, go, 2);
    fo_layer_add_transition(om, 0, 1, 0, 0.2f, back, 3);
    int rebuilt_in_transition = 0;
    for (int frame = 0; frame < 40; ++frame) {
        if (frame == 5 || frame == 22) {
            const uint32_t on = frame == 5 ? 1u : 0u;
            CHECK(fyx_machine_set_parameter(ctx, id, 0, FYX_ALL_INSTANCES, FYX_PARAM_RULE, 0.0f, 0.0f, on));
            fo_machine_set_parameter(om, 0, FYX_PARAM_RULE, 0.0f, 0.0f, on);
        }
        if (frame == 7 || frame == 8 || frame == 25 || frame == 31) {     // the game edited its Machine: send it again
            int32_t st[2][2];
            float tr[2][2][2];
            int kind[2];
            float f0[2], f1[2];
            uint32_t u[2];
            for (uint32_t i = 0; i < n_inst; ++i) {
                CHECK(fyx_layer_get_state(ctx, id, 0, i, &st[i][0], &st[i][1]));
                for (uint32_t t = 0; t < 2; ++t) CHECK(fyx_layer_get_transition_state(ctx, id, 0, i, t, &tr[i][t][0], &tr[i][t][1]));
                CHECK(fyx_machine_get_parameter(ctx, id, 0, i, &kind[i], &f0[i], &f1[i], &u[i]));
                fyx_layer_event ev;
                int has = 1;
                while (has) CHECK(fyx_layer_pop_event(ctx, id, 0, i, &ev, &has));      // they do not survive the clear
            }
            if (st[0][1] >= 0) ++rebuilt_in_transition;
            CHECK(fyx_machine_clear(ctx, id));
            int32_t dummy = 0;
            if (fyx_layer_get_state(ctx, id, 0, 0, &dummy, &dummy) == 0) { std::printf("FAIL: a layer survived fyx_machine_clear\n"); return 1; }
            if (send_machine(ctx, id)) return 1;
            for (uint32_t i = 0; i < n_inst; ++i) {
                CHECK(fyx_machine_set_parameter(ctx, id, 0, i, kind[i], f0[i], f1[i], u[i]));
                CHECK(fyx_layer_set_state(ctx, id, 0, i, st[i][0], st[i][1]));
                for (uint32_t t = 0; t < 2; ++t) CHECK(fyx_layer_set_transition_state(ctx, id, 0, i, t, tr[i][t][0], tr[i][t][1]));
            }
        }
        float times[4];
        uint8_t ticked[4];
        uint32_t off[3], n_ops = 0;
        std::vector<uint32_t> ops(2 * 256);
        CHECK(fyx_animator_plan(ctx, id, 1, 1.0f / 30.0f, times, ticked, off, ops.data(), 256, &n_ops));
        fo_machine_evaluate_pose(om, oa, 2, 1.0f / 30.0f);
        for (uint32_t i = 0; i < n_inst; ++i) {
            int32_t s = 0, t = 0;
            CHECK(fyx_layer_get_state(ctx, id, 0, i, &s, &t));
            float c0 = 0.0f, c1 = 0.0f;
            CHECK(fyx_animation_get_state(ctx, id, 0, i, &c0, nullptr, nullptr));
            CHECK(fyx_animation_get_state(ctx, id, 1, i, &c1, nullptr, nullptr));
            if (s != fo_layer_active_state(om, 0) || t != fo_layer_active_transition(om, 0) || c0 != fo_animation_time_position(oa[0]) ||
                c1 != fo_animation_time_position(oa[1])) {
                std::printf("FAIL machine frame %d instance %u: state %d / transition %d (oracle %d / %d), clocks %.9g %.9g (oracle %.9g %.9g)\n",
                            frame, i, s, t, fo_layer_active_state(om, 0), fo_layer_active_transition(om, 0), c0, c1,
                            fo_animation_time_position(oa[0]), fo_animation_time_position(oa[1]));
                return 1;
            }
        }
        if (std::memcmp(ops.data() + 2 * off[0], ops.data() + 2 * off[1], (size_t)(off[1] - off[0]) * 8) != 0) {
            std::printf("FAIL machine frame %d: the two instances planned different programs\n", frame);
            return 1;
        }
    }
    if (rebuilt_in_transition < 2) { std::printf("FAIL: the test no longer re-sends the machine inside a transition\n"); return 1; }
    fo_machine_free(om);
    fo_animation_free(oa[0]);
    fo_animation_free(oa[1]);
    std::printf("machine re-sent 4 times (%d inside a transition): states, transitions and clocks equal the oracle's\n", rebuilt_in_transition);
    return 0;
}

static int control_only() {
    fyx_ctx* ctx = nullptr;
    if (fyx_init_control_only(&ctx) != 0) { std::printf("FAIL fyx_init_control_only\n"); return 1; }
    const int n_bones = 4;
    const Clip clip = make_clip(n_bones, 7);
    if (build_animator(ctx, clip, n_bones, 3)) return 1;
    fo_tracks* td = oracle_tracks(clip);
    fo_animation* oa = fo_animation_new(td);
    for (size_t t = 0; t < clip.target.size(); ++t) fo_animation_bind(oa, (int)t, clip.target[t], 1);
    fo_animation_set_time_slice(oa, 0.0f, 1.0f);
    fo_animation_set_speed(oa, 1.75f);
    for (int frame = 0; frame < 50; ++frame) {
        const float before = fo_animation_time_position(oa);
        float times[3];
        uint8_t ticked[3];
        uint32_t n_ops = 0;
        CHECK(fyx_animator_plan(ctx, 2, 0, 1.0f / 30.0f, times, ticked, nullptr, nullptr, 0, &n_ops));
        fo_animation_tick(oa, 1.0f / 30.0f);
        float now = 0.0f;
        CHECK(fyx_animation_get_state(ctx, 2, 0, 2, &now, nullptr, nullptr));
        if (!(ticked[0] & 1) || times[0] != before || times[2] != before || now != fo_animation_time_position(oa)) {
            std::printf("FAIL frame %d: sampled at %.9g (oracle %.9g), clock %.9g (oracle %.9g)\n", frame, times[0], before, now,
                        fo_animation_time_position(oa));
            return 1;
        }
    }
    if (machine_rebuilt_mid_transition(ctx, clip, td)) return 1;
    // no GPU behind this context: the data path must refuse, not fall back
    if (fyx_animation_player_update(ctx, 2, 0.1f) != FYX_ERR_NO_DEVICE || fyx_sync(ctx) != FYX_ERR_NO_DEVICE) {
        std::printf("FAIL: a control-only context ran a data-path call\n");
        return 1;
    }
    fo_animation_free(oa);
    fo_tracks_free(td);
    fyx_shutdown(ctx);
    std::printf("control plane ok: 50 frames, sample times and clocks equal the oracle's Animation::tick\n");
    return 0;
}

static int gpu() {
    fyx_ctx* ctx = nullptr;
    const int rc = fyx_init(&ctx, 0);
    if (rc == FYX_ERR_NO_DEVICE) { std::printf("no GPU\n"); return 77; }
    if (rc != 0) { std::printf("FAIL fyx_init -> %d\n", rc); return 1; }
    // ---- C1: 1 k AnimatedVertex vertices (vertex.rs:139-155), 4 bones ----
    const uint32_t nv = 1000, nb = 4, stride = 68;
    std::vector<float> pos(nv * 3), nrm(nv * 3), tan(nv * 4), wgt(nv * 4), pal(nb * 16);
    std::vector<uint8_t> idx(nv * 4), aos((size_t)nv * stride, 0);
    for (uint32_t v = 0; v < nv; ++v) {
        float wsum = 0.0f;
        for (int k = 0; k < 3; ++k) { pos[v * 3 + k] = urand() * 2 - 1; nrm[v * 3 + k] = urand() - 0.5f; tan[v * 4 + k] = urand() - 0.5f; }
        tan[v * 4 + 3] = (v & 1) ? -1.0f : 1.0f;
        for (int k = 0; k < 4; ++k) { wgt[v * 4 + k] = urand(); wsum += wgt[v * 4 + k]; idx[v * 4 + k] = (uint8_t)(urand() * nb); }
        for (int k = 0; k < 4; ++k) wgt[v * 4 + k] /= wsum;
        uint8_t* r = &aos[(size_t)v * stride];
        std::memcpy(r + 0, &pos[v * 3], 12);
        std::memcpy(r + 20, &nrm[v * 3], 12);
        std::memcpy(r + 32, &tan[v * 4], 16);
        std::memcpy(r + 48, &wgt[v * 4], 16);
        std::memcpy(r + 64, &idx[v * 4], 4);
    }
    for (uint32_t b = 0; b < nb; ++b) {  // rotation about z + translation, column-major
        const float a = urand() * 6.28f, c = std::cos(a), s = std::sin(a);
        float* m = &pal[b * 16];
        std::memset(m, 0, 64);
        m[0] = c; m[1] = s; m[4] = -s; m[5] = c; m[10] = 1; m[15] = 1;
        m[12] = urand(); m[13] = urand(); m[14] = urand();
    }
    CHECK(fyx_mesh_upload(ctx, 7, aos.data(), nv, stride, 0, 20, 32, 48, 64));
    std::vector<float> op(nv * 3), on(nv * 3), ot(nv * 4), rp(nv * 3), rn(nv * 3), rt(nv * 4);
    float box[6], box2[6];
    CHECK(fyx_lbs_skin(ctx, 7, pal.data(), nb, 1, op.data(), on.data(), ot.data(), box));
    CHECK(fyx_skinned_aabb(ctx, 7, pal.data(), nb, box2));
    if (fo_lbs_skin(nv, pos.data(), nrm.data(), tan.data(), wgt.data(), idx.data(), pal.data(), nb, rp.data(), rn.data(), rt.data())) return 1;
    if (std::memcmp(op.data(), rp.data(), op.size() * 4) || std::memcmp(on.data(), rn.data(), on.size() * 4) ||
        std::memcmp(ot.data(), rt.data(), ot.size() * 4)) {
        std::printf("FAIL: skinned vertices differ from the oracle\n");
        return 1;
    }
    float rb[6] = {3.4e38f, 3.4e38f, 3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
    for (uint32_t v = 0; v < nv; ++v)
        for (int k = 0; k < 3; ++k) { rb[k] = std::fmin(rb[k], rp[v * 3 + k]); rb[3 + k] = std::fmax(rb[3 + k], rp[v * 3 + k]); }
    if (std::memcmp(box, rb, 24) || std::memcmp(box2, rb, 24)) { std::printf("FAIL: AABB differs\n"); return 1; }
    // an out-of-range bone index is an error code, not a panic (scene/mesh/mod.rs:514)
    if (fyx_lbs_skin(ctx, 7, pal.data(), 2, 1, op.data(), nullptr, nullptr, nullptr) != FYX_ERR_BONE_INDEX) {
        std::printf("FAIL: expected FYX_ERR_BONE_INDEX\n");
        return 1;
    }
    // ---- a clip through AnimationPlayer::update -> palette, against the oracle's tick / apply / hierarchy ----
    const int n_bones = 4;
    const Clip clip = make_clip(n_bones, 9);
    if (build_animator(ctx, clip, n_bones, 2)) return 1;
    std::vector<int32_t> bones = {0, 1, 2, 3};
    CHECK(fyx_bone_list_create(ctx, 3, 1, 4, bones.data()));
    fo_tracks* td = oracle_tracks(clip);
    fo_animation* oa = fo_animation_new(td);
    for (size_t t = 0; t < clip.target.size(); ++t) fo_animation_bind(oa, (int)t, clip.target[t], 1);
    fo_animation_set_time_slice(oa, 0.0f, 1.0f);
    fo_animation_set_speed(oa, 1.75f);
    std::vector<fo_transform> nodes(n_bones);
    for (auto& t : nodes) fo_transform_default(&t);
    void* d_pal = nullptr;
    CHECK(fyx_malloc(ctx, 2 * 4 * 64, &d_pal));
    for (int frame = 0; frame < 25; ++frame) {
        CHECK(fyx_animation_player_update(ctx, 2, 1.0f / 24.0f));
        fo_animation_tick(oa, 1.0f / 24.0f);
        fo_pose_apply(fo_animation_pose(oa), nodes.data(), n_bones);
    }
    CHECK(fyx_animator_palette(ctx, 2, 3, (float*)d_pal));
    std::vector<float> got(2 * 4 * 16), local(4 * 16), global(4 * 16), ident(4 * 16, 0.0f), ref(4 * 16);
    CHECK(fyx_memcpy_d2h(ctx, got.data(), d_pal, got.size() * 4));
    const int32_t parent[4] = {-1, 0, 1, 2};
    for (int b = 0; b < 4; ++b) { fo_calculate_local_transform(&nodes[b], &local[b * 16]); ident[b * 16] = ident[b * 16 + 5] = ident[b * 16 + 10] = ident[b * 16 + 15] = 1.0f; }
    fo_update_global_transforms(local.data(), parent, 4, global.data());
    fo_palette(global.data(), ident.data(), 4, ref.data());
    if (std::memcmp(got.data(), ref.data(), 256) || std::memcmp(got.data() + 64, ref.data(), 256)) {
        std::printf("FAIL: animated palette differs from the oracle\n");
        return 1;
    }
    // ---- a frame loop as the engine would drive it: scene update with registered (alternating) palette outputs, skinning
    // launches on the worker streams, a batch whose job table changes, pipelined and joined frames, per-launch timing.
    // Every batch result must equal the single-mesh call's bytes; under tools/asan_gpu.sh this is the part that walks
    // the stream / event / control-block bookkeeping of the library with AddressSanitizer watching.
    {
        const uint32_t sizes[3] = {700, 5003, 20000};
        std::vector<std::vector<float>> want(3);
        std::vector<void*> d_out(3, nullptr);
        void* d_static_pal = nullptr;
        CHECK(fyx_malloc(ctx, nb * 64, &d_static_pal));
        CHECK(fyx_memcpy_h2d(ctx, d_static_pal, pal.data(), nb * 64));
        for (int m = 0; m < 3; ++m) {
            const uint32_t n = sizes[m];
            std::vector<uint8_t> a((size_t)n * stride, 0);
            for (uint32_t v = 0; v < n; ++v) std::memcpy(&a[(size_t)v * stride], &aos[(size_t)(v % nv) * stride], stride);
            CHECK(fyx_mesh_upload(ctx, 100 + m, a.data(), n, stride, 0, 20, 32, 48, 64));
            want[m].resize((size_t)n * 3);
            CHECK(fyx_lbs_skin(ctx, 100 + m, pal.data(), nb, 1, want[m].data(), nullptr, nullptr, nullptr));
            CHECK(fyx_malloc(ctx, (size_t)n * 12 + 64, &d_out[m]));
        }
        void *d_pal_ring[2] = {nullptr, nullptr}, *d_inst_out = nullptr;
        for (auto& q : d_pal_ring) CHECK(fyx_malloc(ctx, 2 * 4 * 64, &q));
        CHECK(fyx_malloc(ctx, 2 * (size_t)sizes[1] * 12 + 64, &d_inst_out));
        const uint64_t ids[1] = {2};
        for (int frame = 0; frame < 16; ++frame) {
            CHECK(fyx_set_option(ctx, "lbs.streams", 1 + (frame / 4) % 2));
            CHECK(fyx_set_option(ctx, "anim.overlap", (frame / 2) % 2));
            CHECK(fyx_set_option(ctx, "lbs.timing", frame % 2));
            CHECK(fyx_animator_set_palette_output(ctx, 2, 3, (float*)d_pal_ring[frame & 1]));
            CHECK(fyx_scene_update(ctx, ids, 1, 1.0f / 24.0f));
            CHECK(fyx_lbs_skin_device(ctx, 101, (const float*)d_pal_ring[frame & 1], 4, 2, (float*)d_inst_out, nullptr, nullptr));
            fyx_skin_job jobs[3];
            uint32_t n_jobs = 0;
            for (int m = 0; m < 3; ++m) {
                if (frame % 3 == 1 && m == 0) continue;   // the job table changes from frame to frame
                CHECK(fyx_memcpy_h2d(ctx, d_out[m], std::vector<float>((size_t)sizes[m] * 3, -1.0f).data(), (size_t)sizes[m] * 12));
                jobs[n_jobs++] = fyx_skin_job{(uint64_t)(100 + m), (const float*)d_static_pal, nb, 1, (float*)d_out[m], nullptr, nullptr};
            }
            CHECK(fyx_lbs_skin_batch(ctx, jobs, n_jobs));
            if (frame % 2) CHECK(fyx_join(ctx));
            for (uint32_t j = 0; j < n_jobs; ++j) {
                const int m = (int)jobs[j].mesh_id - 100;
                std::vector<float> back((size_t)sizes[m] * 3);
                CHECK(fyx_memcpy_d2h(ctx, back.data(), d_out[m], back.size() * 4));
                if (std::memcmp(back.data(), want[m].data(), back.size() * 4)) { std::printf("FAIL: frame %d, batch job of mesh %d differs from the single-mesh call\n", frame, m); return 1; }
            }
            if (frame % 2) {
                double us = 0; uint32_t n = 0;
                CHECK(fyx_debug_kernel_time(ctx, &us, &n));
                if (n != 1 || !(us > 0.0)) { std::printf("FAIL: fyx_debug_kernel_time -> %u launches, %f us\n", n, us); return 1; }
            }
        }
        CHECK(fyx_set_option(ctx, "lbs.streams", 2)); CHECK(fyx_set_option(ctx, "anim.overlap", 0)); CHECK(fyx_set_option(ctx, "lbs.timing", 0));
        CHECK(fyx_animator_set_palette_output(ctx, 2, 3, nullptr));
        CHECK(fyx_sync(ctx));
        for (int m = 0; m < 3; ++m) { CHECK(fyx_mesh_free(ctx, 100 + m)); CHECK(fyx_free(ctx, d_out[m])); }
        for (auto& q : d_pal_ring) CHECK(fyx_free(ctx, q));
        CHECK(fyx_free(ctx, d_inst_out)); CHECK(fyx_free(ctx, d_static_pal));
    }
    CHECK(fyx_free(ctx, d_pal));
    fo_animation_free(oa);
    fo_tracks_free(td);
    fyx_shutdown(ctx);
    std::printf("gpu ok: C1 skinning + AABB bit-exact, 25-frame clip -> palette bit-exact, 16-frame engine-style loop (scene update, worker streams, batches, pipelining, timing) consistent\n");
    return 0;
}

int main(int argc, char** argv) { return (argc > 1 && !std::strcmp(argv[1], "--control-only")) ? control_only() : gpu(); }
