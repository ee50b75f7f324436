#!/usr/bin/env python3
"""Mutation check of the GPU parity suite: does `pytest -m gpu` notice a one-line change of the device code?

    python tools/mutants.py build        # here (no GPU): one library per mutant under fyrox_amd/mutants/ (git-ignored *.so, travels with gpurun)
    python tools/mutants.py run [--out gpurun_out/mutants.json]      # on the GPU box: each mutant under the suite, first failing test recorded

A mutant is a single textual replacement in fyrox_amd/csrc (arithmetic re-associated, a comparison relaxed, an exit dropped, a word not
written): the kind of slip that changes LAST BITS or one edge case.  A mutant that survives names a hole in the suite.  The sources are
restored after every build; the shipped library is never replaced here (the run step swaps libraries in a scratch copy of the tree)."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fyrox_amd", "csrc")
OUT = os.path.join(ROOT, "fyrox_amd", "mutants")
LIB = os.path.join(ROOT, "fyrox_amd", "libfyrox_hip.so")

# (name, file, old, new, what it is)
MUTANTS = [
    ("lerpf_other_form", "anim_leaves.h", "return a + (b - a) * t; }", "return a * (1.0f - t) + b * t; }",
     "fyrox-math lerpf written as a (1 - t) + b t: same value, other rounding"),
    ("constant_key_never_takes_right", "anim_leaves.h", "if (lk == FYX_KEY_CONSTANT) return t == 1.0f ? cv.y : cv.x;\n    if (lk == FYX_KEY_LINEAR) return lerpf_(cv.x, cv.y, t);",
     "if (lk == FYX_KEY_CONSTANT) return cv.x;\n    if (lk == FYX_KEY_LINEAR) return lerpf_(cv.x, cv.y, t);", "CurveKeyKind::Constant at t == 1 (span records)"),
    ("nlerp_without_the_flip", "anim_kernels.hip", "        if (dot4(a, o.r) < 0.0f) a = f4{-a.x, -a.y, -a.z, -a.w};\n        const f4 l = f4{a.x * omw + o.r.x * w",
     "        const f4 l = f4{a.x * omw + o.r.x * w", "value.rs:449-454: self is not negated when dot < 0"),
    ("position_lerp_other_form", "anim_kernels.hip", "        self.px = self.px * omw + o.px * w;", "        self.px = self.px + (o.px - self.px) * w;",
     "nalgebra lerp of one component as a + (b - a) w"),
    ("root_motion_ignore_bits_swapped", "anim_kernels.hip", "rm.delta_position[0] = (an.rm_ignore & 1u) ? 0.0f : delta[0];", "rm.delta_position[0] = (an.rm_ignore & 2u) ? 0.0f : delta[0];",
     "RootMotionSettings::ignore_x_movement read from ignore_y"),
    ("root_motion_remainder_not_taken", "anim_kernels.hip", "((prev.rem_flags & 1u) && !(an.rm_ignore & 16u))", "(prev.rem_flags & 1u)",
     "round 6's own bug: the second Position of the root node's list takes the remainder again"),
    ("first_span_exit_not_strict", "anim_kernels.hip", "if (lf.x < time && time < lf.y) {", "if (lf.x <= time && time <= lf.y) {",
     "the first-span exit of sample_curve takes times ON its keys"),
    ("hierarchy_element_chain_reordered", "anim_kernels.hip", "            float y = a0 * b.x;\n            y = a1 * b.y + y;\n            y = a2 * b.z + y;\n            y = a3 * b.w + y;",
     "            float y = a3 * b.w;\n            y = a2 * b.z + y;\n            y = a1 * b.y + y;\n            y = a0 * b.x + y;", "global = parent * local summed from the last column backwards (wide walk)"),
    ("skin_dot_reassociated", "lbs_leaves.h", "if constexpr (EXACT) return (a * x + b * y) + c * z;", "if constexpr (EXACT) return a * x + (b * y + c * z);",
     "M3x3 * v as a x + (b y + c z)"),
    ("skin_translation_added_first", "lbs_leaves.h", "z = ((C.x * px + C.y * py) + C.z * pz) + C.w;", "z = C.w + ((C.x * px + C.y * py) + C.z * pz);",
     "an EQUIVALENT mutant on purpose (IEEE addition commutes): must survive -- the harness's control"),
    ("tangent_w_not_passed_through", "lbs_leaves.h", "__builtin_bit_cast(u32x4, f32x4{o.tx, o.ty, o.tz, tw}), b.out_tan", "__builtin_bit_cast(u32x4, f32x4{o.tx, o.ty, o.tz, o.tz}), b.out_tan",
     "tangent.w of the buffer-resource kernels (lbs_skin_dyn, lbs_skin_batch_dyn)"),
    ("batch_ticket_skips_a_unit", "lbs_kernels.hip", "        if (tid == 0) *ticket = 2 * WPB;\n        bool pj = false;", "        if (tid == 0) *ticket = 2 * WPB + 1;\n        bool pj = false;",
     "lbs_skin_batch_dyn: one unit of every segment is never drawn"),
    ("aos_ticket_skips_a_unit", "lbs_kernels.hip", "    if (tid == 0) *ticket = (TWO ? 2 : 1) * WPB;", "    if (tid == 0) *ticket = (TWO ? 2 : 1) * WPB + 1;",
     "lbs_skin_aos: one unit of every segment is never drawn"),
    ("staged_output_tangent_w_masked", "lbs_kernels.hip", "wmask |= 15ull << (x.off_tan / 4);", "wmask |= 7ull << (x.off_tan / 4);",
     "lbs_skin_ex's staged interleaved output: tangent.w not written"),
    ("projective_divide_skipped", "lbs_leaves.h", "if (n != 0.0f) { xy.x = xy.x / n; xy.y = xy.y / n; z = z / n; }", "if (n != 0.0f && n != 1.0f) { xy.x = xy.x / n; xy.y = xy.y / n; z = z / n; }",
     "an EQUIVALENT mutant on purpose (x / 1 == x): must survive"),
    ("crowd_last_instance_of_a_run_dropped", "lbs_kernels.hip", "    const uint32_t i1 = (i0 + ipb < a.n_instances) ? i0 + ipb : a.n_instances;\n    if (i0 >= i1) return;\n    const uint32_t v = tile * BLOCK + tid;",
     "    const uint32_t i1 = (i0 + ipb < a.n_instances) ? i0 + ipb : a.n_instances - 1;\n    if (i0 >= i1) return;\n    const uint32_t v = tile * BLOCK + tid;", "the crowd kernel never skins the last instance"),
    # ---- second batch ----
    ("quat_normalize_by_reciprocal", "anim_kernels.hip", "    return f4{q.x / n, q.y / n, q.z / n, q.w / n};", "    const float r_ = 1.0f / n;\n    return f4{q.x * r_, q.y * r_, q.z * r_, q.w * r_};",
     "UnitQuaternion normalisation as a multiplication by 1 / n"),
    ("quat_to_matrix_term_order", "anim_kernels.hip", "o.m[0] = ww + ii - jj - kk;", "o.m[0] = ww - jj + ii - kk;", "rotation matrix diagonal summed in another order"),
    ("local_matrix_translation_reassociated", "anim_kernels.hip", "out[12 + row] = ro[row] + rp[row] + t[row] - rpx * f0", "out[12 + row] = ro[row] + (rp[row] + t[row]) - rpx * f0",
     "calculate_local_transform's translation row, first two additions re-associated"),
    ("mask_clears_nothing_on_node_0", "anim_kernels.hip", "                if (cx.layer_masks[(size_t)arg * cx.n_nodes + cx.node]) acc.mask = 0;\n                break;\n            case OP_APPLY:\n                apply_pose(cx, acc);",
     "                if (cx.node && cx.layer_masks[(size_t)arg * cx.n_nodes + cx.node]) acc.mask = 0;\n                break;\n            case OP_APPLY:\n                apply_pose(cx, acc);",
     "LayerMask::should_animate ignored for node 0 (general fold)"),
    ("list_not_empty_bit_dropped", "anim_kernels.hip", "| (has_prop ? 8u : 0u) | (d.present & 16u));", "| (has_prop ? 8u : 0u));", "a value that fits no binding no longer makes its node's list non-empty"),
    ("property_vec3_lerp_other_form", "anim_kernels.hip", "            self.v.x = self.v.x * omw + o.v.x * w; self.v.y = self.v.y * omw + o.v.y * w; self.v.z = self.v.z * omw + o.v.z * w;\n            break;\n        case FYX_VALUE_VEC4:",
     "            self.v.x = self.v.x + (o.v.x - self.v.x) * w; self.v.y = self.v.y * omw + o.v.y * w; self.v.z = self.v.z * omw + o.v.z * w;\n            break;\n        case FYX_VALUE_VEC4:",
     "a Vector3 property's first lane blended as a + (b - a) w"),
    ("crowd_span_clamp_at_first_key_strict", "anim_leaves.h", "    if (time <= l_first) {", "    if (time < l_first) {", "Curve::value_at's clamp at the first key (span-record form: crowd sampler)"),
    ("duplicate_key_locations_take_the_hint", "anim_leaves.h", "if (time >= locs.x && time <= locs.y && locs.x < locs.y) right = h;", "if (time >= locs.x && time <= locs.y) right = h;",
     "the hinted span accepted although its two keys share a location"),
    ("blend_shape_offsets_position_only_twice", "lbs_kernels.hip", "                        px = px + h2f(h[0]) * ws; py = py + h2f(h[1]) * ws; pz = pz + h2f(h[2]) * ws;\n                        nx = nx + h2f(h[3]) * ws; ny = ny + h2f(h[4]) * ws; nz = nz + h2f(h[5]) * ws;\n                        t.x = t.x + h2f(h[6]) * ws; t.y = t.y + h2f(h[7]) * ws; t.z = t.z + h2f(h[8]) * ws;\n                    } else {\n                        px = __builtin_fmaf(h2f(h[0]), ws, px); py = __builtin_fmaf(h2f(h[1]), ws, py);\n                        pz = __builtin_fmaf(h2f(h[2]), ws, pz); nx = __builtin_fmaf(h2f(h[3]), ws, nx);\n                        ny = __builtin_fmaf(h2f(h[4]), ws, ny); nz = __builtin_fmaf(h2f(h[5]), ws, nz);\n                        t.x = __builtin_fmaf(h2f(h[6]), ws, t.x); t.y = __builtin_fmaf(h2f(h[7]), ws, t.y);\n                        t.z = __builtin_fmaf(h2f(h[8]), ws, t.z);\n                    }\n                }\n            }\n            const Skinned o = skin_vertex<EXACT, 7>(rows, row3, projective, id, w, px, py, pz, nx, ny, nz, t.x, t.y, t.z);\n            if constexpr (AOS) {",
     "                        px = px + h2f(h[0]) * ws; py = py + h2f(h[1]) * ws; pz = pz + h2f(h[2]) * ws;\n                        nx = nx + h2f(h[3]) * ws; ny = ny + h2f(h[4]) * ws; nz = nz + h2f(h[5]) * ws;\n                        t.x = t.x + h2f(h[6]) * ws; t.y = t.y + h2f(h[7]) * ws; t.z = t.z + h2f(h[7]) * ws;\n                    } else {\n                        px = __builtin_fmaf(h2f(h[0]), ws, px); py = __builtin_fmaf(h2f(h[1]), ws, py);\n                        pz = __builtin_fmaf(h2f(h[2]), ws, pz); nx = __builtin_fmaf(h2f(h[3]), ws, nx);\n                        ny = __builtin_fmaf(h2f(h[4]), ws, ny); nz = __builtin_fmaf(h2f(h[5]), ws, nz);\n                        t.x = __builtin_fmaf(h2f(h[6]), ws, t.x); t.y = __builtin_fmaf(h2f(h[7]), ws, t.y);\n                        t.z = __builtin_fmaf(h2f(h[8]), ws, t.z);\n                    }\n                }\n            }\n            const Skinned o = skin_vertex<EXACT, 7>(rows, row3, projective, id, w, px, py, pz, nx, ny, nz, t.x, t.y, t.z);\n            if constexpr (AOS) {",
     "lbs_skin_ex (exact): the tangent's z offset of a blend shape read from its y plane"),
    ("palette_commit_row3_wrong_column", "lbs_kernels.hip", "            reinterpret_cast<float*>(row3 + b)[c] = col[i].w;\n            pj |= col[i].w != (c == 3 ? 1.0f : 0.0f);\n        }\n    }\n    const bool wave_pj = __any(pj) != 0;\n    if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;\n    __syncthreads();\n    bool projective = false;\n#pragma unroll\n    for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;\n    pin_vertex(A);",
     "            reinterpret_cast<float*>(row3 + b)[c] = col[i].z;\n            pj |= col[i].w != (c == 3 ? 1.0f : 0.0f);\n        }\n    }\n    const bool wave_pj = __any(pj) != 0;\n    if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;\n    __syncthreads();\n    bool projective = false;\n#pragma unroll\n    for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;\n    pin_vertex(A);",
     "lbs_skin_dyn stages the projective row from the wrong component"),
    # ---- third batch (third session)
    ("cubic_tangent_scale_signed", "anim_leaves.h", "    const float scale = absf_(p1 - p0);", "    const float scale = p1 - p0;", "cubicf: tangents scaled by |p1 - p0| (fyrox-math/src/lib.rs:212-221)"),
    ("euler_order_reversed", "anim_kernels.hip", "            q = quat_mul(quat_mul(qz, qy), qx);\n        }\n    }\n    if (has_r && rkind == FYX_KIND_QUAT) q = quat_normalize(f4{r0, r1, r2, r3});", "            q = quat_mul(quat_mul(qx, qy), qz);\n        }\n    }\n    if (has_r && rkind == FYX_KIND_QUAT) q = quat_normalize(f4{r0, r1, r2, r3});", "quat_from_euler XYZ = qz * qy * qx (fyrox-math/src/lib.rs:725-740)"),
    ("quat_mul_cross_term_sign", "anim_kernels.hip", "    const float i = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;", "    const float i = a.w * b.x + a.x * b.w - a.y * b.z + a.z * b.y;", "Hamilton product, i component"),
    ("max_bone_index_ignores_the_fourth_influence", "lbs_kernels.hip", "max((id >> 16) & 0xffu, id >> 24)));", "max((id >> 16) & 0xffu, (id >> 16) & 0xffu)));", "upload validation: the largest bone index of a mesh"),
    ("palette_product_terms_reordered", "lbs_kernels.hip", "    y = a[4 + i] * b[j * 4 + 1] + y;\n    y = a[8 + i] * b[j * 4 + 2] + y;\n    y = a[12 + i] * b[j * 4 + 3] + y;\n    out[e] = y;",
     "    y = a[8 + i] * b[j * 4 + 2] + y;\n    y = a[4 + i] * b[j * 4 + 1] + y;\n    y = a[12 + i] * b[j * 4 + 3] + y;\n    out[e] = y;", "fyx_palette: global * inv_bind, k ascending"),
    ("aabb_upper_bound_starts_at_zero", "lbs_kernels.hip", "    float mx[3] = {-__FLT_MAX__, -__FLT_MAX__, -__FLT_MAX__};\n    for (uint32_t v = blockIdx.x * kAabbBlock + threadIdx.x; v < a.n_verts;\n",
     "    float mx[3] = {0.0f, 0.0f, 0.0f};\n    for (uint32_t v = blockIdx.x * kAabbBlock + threadIdx.x; v < a.n_verts;\n", "accurate_world_bounding_box starts from (+MAX, -MAX) (fyrox-math/src/aabb.rs:33-40)"),
    ("unaligned_word_bytes_swapped", "lbs_kernels.hip", "return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);", "return (uint32_t)p[0] | ((uint32_t)p[2] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[3] << 24);",
     "little-endian field reads of a vertex buffer whose attributes are not 4-byte aligned (buffer.rs:1279-1321)"),
    ("root_motion_relative_rotation_other_side", "anim_kernels.hip", "const f4 current_relative_rotation = quat_mul(conj(pp), pose_rotation);", "const f4 current_relative_rotation = quat_mul(pose_rotation, conj(pp));",
     "lib.rs:610-640: prev_rotation.inverse() * pose_rotation"),
    ("root_motion_rotation_remainder_other_side", "anim_kernels.hip", "const f4 d = quat_mul(remainder, current_relative_rotation);", "const f4 d = quat_mul(current_relative_rotation, remainder);", "lib.rs:640-650: remainder * current_relative_rotation"),
    ("constant_key_right_value_early", "anim_leaves.h", "    if (lk == FYX_KEY_CONSTANT) return t == 1.0f ? ra.x : la.x;", "    if (lk == FYX_KEY_CONSTANT) return t >= 0.5f ? ra.x : la.x;", "stepf (curve.rs:25-31) in the key-record sampler"),
    # ---- fourth batch (third session): the homogeneous path's decisions, the crowd kernel's double buffer, the palette-length check
    ("crowd_projective_flag_into_the_current_buffer", "lbs_kernels.hip", "            if (lane == 0) flags[(cur ^ 1) * 4 + wave] = wave_pj ? 1u : 0u;", "            if (lane == 0) flags[cur * 4 + wave] = wave_pj ? 1u : 0u;",
     "lbs_skin_crowd: the next instance's projective flag belongs to the next buffer"),
    ("palette_commit_ignores_m32", "lbs_leaves.h", "    return !(r.c0.w == 0.0f && r.c1.w == 0.0f && r.c2.w == 0.0f && r.c3.w == 1.0f);", "    return !(r.c0.w == 0.0f && r.c1.w == 0.0f && r.c3.w == 1.0f);",
     "a matrix whose only non-affine entry is m32 (crowd kernel / lbs_skin staging)"),
    ("dyn_projective_test_looks_at_m33_only", "lbs_kernels.hip", "            pj |= col[i].w != (c == 3 ? 1.0f : 0.0f);\n        }\n    }\n    const bool wave_pj = __any(pj) != 0;\n    if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;\n    __syncthreads();\n    bool projective = false;\n#pragma unroll\n    for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;\n    pin_vertex(A);",
     "            pj |= c == 3 && col[i].w != 1.0f;\n        }\n    }\n    const bool wave_pj = __any(pj) != 0;\n    if (lane == 0) wave_flag[wave] = wave_pj ? 1u : 0u;\n    __syncthreads();\n    bool projective = false;\n#pragma unroll\n    for (uint32_t wv = 0; wv < WPB; ++wv) projective |= wave_flag[wv] != 0;\n    pin_vertex(A);",
     "lbs_skin_dyn: a matrix with m33 == 1 and a non-zero m30 / m31 / m32 is projective too"),
    ("stage_palette_ignores_m31", "lbs_leaves.h", "        projective |= !(c0.w == 0.0f && c1.w == 0.0f && c2.w == 0.0f && c3.w == 1.0f);", "        projective |= !(c0.w == 0.0f && c2.w == 0.0f && c3.w == 1.0f);",
     "stage_palette (the AABB kernels): a matrix whose only non-affine entry is m31"),
    ("projective_divides_by_zero_too", "lbs_leaves.h", "                if (n != 0.0f) { xy.x = xy.x / n; xy.y = xy.y / n; z = z / n; }", "                { xy.x = xy.x / n; xy.y = xy.y / n; z = z / n; }",
     "transform_point divides only when the homogeneous coordinate is not zero (nalgebra)"),
    ("palette_one_bone_short_is_accepted", "fyx_api.hip", "    if (m->n_verts > 0 && m->max_bone_index >= n_bones)", "    if (m->n_verts > 0 && m->max_bone_index > n_bones)",
     "a palette exactly one matrix short of the mesh's largest bone index is refused (the Rust loop would panic on the index)"),
    ("frame_skin_projective_test_looks_at_m33_only", "anim_kernels.hip", "            pj |= y.w != (c == 3u ? 1.0f : 0.0f);", "            pj |= c == 3u && y.w != 1.0f;",
     "the one-launch frame's skinning workgroups: their own copy of the affine test"),
    # ---- fifth batch: the copies of `global * inv_bind` / `parent * local` (nalgebra's k-ascending axpy chain) in the pose kernels
    ("mat4_mul_terms_reordered", "anim_kernels.hip", "            y = a[4 + i] * b[j * 4 + 1] + y;\n            y = a[8 + i] * b[j * 4 + 2] + y;\n            y = a[12 + i] * b[j * 4 + 3] + y;\n            out[j * 4 + i] = y;",
     "            y = a[8 + i] * b[j * 4 + 2] + y;\n            y = a[4 + i] * b[j * 4 + 1] + y;\n            y = a[12 + i] * b[j * 4 + 3] + y;\n            out[j * 4 + i] = y;", "mat4_mul (the hierarchy's narrow walk)"),
    ("frame_skin_palette_terms_reordered", "anim_kernels.hip", "                const f4 b4 = q_bb[k];\n                const f4* a = reinterpret_cast<const f4*>(l_global) + (size_t)q_node[k] * 4;\n                const f4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];\n                y.x = a0.x * b4.x; y.x = a1.x * b4.y + y.x; y.x = a2.x * b4.z + y.x; y.x = a3.x * b4.w + y.x;",
     "                const f4 b4 = q_bb[k];\n                const f4* a = reinterpret_cast<const f4*>(l_global) + (size_t)q_node[k] * 4;\n                const f4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];\n                y.x = a0.x * b4.x; y.x = a2.x * b4.z + y.x; y.x = a1.x * b4.y + y.x; y.x = a3.x * b4.w + y.x;", "the palette the one-launch frame's skinning workgroups form on chip"),
    ("update_epilogue_palette_terms_reordered", "anim_kernels.hip", "                    const f4 b4 = bb[k];\n                    const f4* a = reinterpret_cast<const f4*>(l_global) + (size_t)node[k] * 4;\n                    const f4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];\n                    y.x = a0.x * b4.x; y.x = a1.x * b4.y + y.x; y.x = a2.x * b4.z + y.x; y.x = a3.x * b4.w + y.x;",
     "                    const f4 b4 = bb[k];\n                    const f4* a = reinterpret_cast<const f4*>(l_global) + (size_t)node[k] * 4;\n                    const f4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];\n                    y.x = a0.x * b4.x; y.x = a2.x * b4.z + y.x; y.x = a1.x * b4.y + y.x; y.x = a3.x * b4.w + y.x;", "the palette the update kernel writes in its epilogue"),
    ("palette_gather_terms_reordered", "anim_kernels.hip", "            y = a[4 + i] * bb[j * 4 + 1] + y;\n            y = a[8 + i] * bb[j * 4 + 2] + y;", "            y = a[8 + i] * bb[j * 4 + 2] + y;\n            y = a[4 + i] * bb[j * 4 + 1] + y;", "palette_gather_kernel (fyx_animator_palette)"),
    ("palette_gather_invalid_bone_is_zero", "anim_kernels.hip", "            y = (i == j) ? 1.0f : 0.0f;", "            y = 0.0f;", "an invalid bone handle yields the identity (scene/mesh/mod.rs:789-791)"),
]


def build(only=None):
    os.makedirs(OUT, exist_ok=True)
    made, t0 = [], time.time()
    if only and os.path.exists(os.path.join(OUT, "index.json")):
        made = [m["name"] for m in json.load(open(os.path.join(OUT, "index.json"))) if m["name"] not in only]
    for name, fn, old, new, what in MUTANTS:
        if only and name not in only:
            continue
        path = os.path.join(SRC, fn)
        src = open(path).read()
        if src.count(old) != 1:
            print(f"{name}: the text to replace occurs {src.count(old)} times in {fn}: skipped", file=sys.stderr)
            continue
        keep = LIB + ".keep"
        shutil.copy2(LIB, keep)
        try:
            open(path, "w").write(src.replace(old, new))
            r = subprocess.run(["make", "-C", SRC], capture_output=True, text=True)
            if r.returncode != 0:
                print(f"{name}: does not build\n{r.stderr[-800:]}", file=sys.stderr)
                continue
            shutil.copy2(LIB, os.path.join(OUT, name + ".so"))
            made.append(name)
            print(f"{name}: built ({time.time() - t0:.0f} s)", flush=True)
        finally:
            open(path, "w").write(src)
            shutil.move(keep, LIB)
    # the objects of the last mutant are newer than the restored sources' library: rebuild what the tree ships
    subprocess.run(["touch"] + [os.path.join(SRC, f) for f in sorted({m[1] for m in MUTANTS})], check=True)
    r = subprocess.run(["make", "-C", SRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    json.dump([{"name": m[0], "file": m[1], "what": m[4]} for m in MUTANTS if m[0] in made], open(os.path.join(OUT, "index.json"), "w"), indent=1)
    print(f"{len(made)} mutants under {OUT}; the shipped library rebuilt from the restored sources")


def run(out, only=None, tests=None, k=None):
    idx = [m for m in json.load(open(os.path.join(OUT, "index.json"))) if not only or m["name"] in only]
    good = LIB + ".shipped"
    shutil.copy2(LIB, good)
    res = []
    try:
        for m in idx:
            shutil.copy2(os.path.join(OUT, m["name"] + ".so"), LIB)
            t0 = time.time()
            r = subprocess.run([sys.executable, "-m", "pytest", *(tests.split() if tests else ["tests"]), *(["-k", k] if k else []), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, capture_output=True, text=True)
            tail = [l for l in r.stdout.splitlines() if l.startswith("FAILED") or l.startswith("ERROR")]
            assert r.returncode in (0, 1), f"pytest did not run the tests (exit code {r.returncode}):\n{r.stdout[-600:]}"      # 5 = nothing collected: not a kill
            res.append(dict(m, killed=r.returncode != 0, by=(tail[0][:200] if tail else None), seconds=round(time.time() - t0, 1)))
            print(json.dumps(res[-1]), flush=True)
    finally:
        shutil.move(good, LIB)
    rec = {"what": "one-line mutants of the device code under `pytest -m gpu -x`", "mutants": len(res), "killed": sum(r["killed"] for r in res),
           "survived": [r["name"] for r in res if not r["killed"]], "results": res}
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("mutants", "killed", "survived")}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("step", choices=("build", "run"))
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="comma-separated mutant names")
    ap.add_argument("--tests", default=None, help="test files instead of the whole suite (to show that a NEW test kills a survivor)")
    ap.add_argument("--k", default=None, help="pytest -k expression to go with --tests")
    a = ap.parse_args()
    only = a.only.split(",") if a.only else None
    build(only) if a.step == "build" else run(a.out, only, a.tests, a.k)
