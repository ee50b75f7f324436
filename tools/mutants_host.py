#!/usr/bin/env python3
"""Mutation check of the CPU suite's hold on the HOST control plane (no GPU): one-line mutants of csrc/anim_planner.h / anim_api.hip, each built
into the library and run under the control-plane tests (tests/test_anim_control.py, test_machine_edits.py, test_device_leaves_on_host.py).

    python tools/mutants_host.py [--out profiles/r06_fuzz/mutants_host.json]

The sources and the shipped library are restored after every mutant."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fyrox_amd", "csrc")
LIB = os.path.join(ROOT, "fyrox_amd", "libfyrox_hip.so")
TESTS = ["tests/test_anim_control.py", "tests/test_machine_edits.py", "tests/test_device_leaves_on_host.py"]      # + BATCH2_TESTS (below) since the second batch

MUTANTS = [
    ("signal_on_the_current_time", "anim_planner.h", "(current < sg.time && next >= sg.time)", "(current <= sg.time && next >= sg.time)", "lib.rs:476-489: a signal AT the current time fires"),
    ("signal_cap_guards_both_branches", "anim_planner.h", "if ((s.speed >= 0.0f && (current < sg.time && next >= sg.time)) ||",
     "if ((s.speed >= 0.0f && (current < sg.time && next >= sg.time) && s.events.size() < s.max_event_capacity) ||", "the precedence quirk of lib.rs:478-482 'repaired'"),
    ("wrapf_upper_bound_strict", "anim_planner.h", "    if (n >= max_limit) {\n        n -= num_of_max * max_limit;", "    if (n > max_limit) {\n        n -= num_of_max * max_limit;", "fyrox-math wrapf"),
    ("looped_animations_end", "anim_planner.h", "return !s.looped && fabsf(s.time - s.end) <= FLT_EPSILON; }", "return fabsf(s.time - s.end) <= FLT_EPSILON; }", "Animation::has_ended"),
    ("xor_is_or", "anim_planner.h", "return l ^ r; }", "return l | r; }", "LogicNode::Xor"),
    ("ended_of_an_invalid_handle_is_false", "anim_planner.h", "return true;  // invalid handle: is_none_or -> true", "return false;", "LogicNode::IsAnimationEnded on a handle that does not resolve"),
    ("weight_parameter_of_any_kind", "anim_planner.h", "w = (p && p->kind == FYX_PARAM_WEIGHT) ? p->f0 : 0.0f;", "w = p ? p->f0 : 0.0f;", "BlendPose weight from a parameter of another kind"),
    ("rewind_to_the_end", "anim_planner.h", "case FYX_ACTION_REWIND_ANIMATION: set_time_position(s, s.start); break;", "case FYX_ACTION_REWIND_ANIMATION: set_time_position(s, s.end); break;", "StateAction::RewindAnimation"),
    ("transition_factor_after_the_clamp_only", "anim_planner.h", "                if (ts.elapsed > tr.time) ts.elapsed = tr.time;\n                ts.blend_factor = ts.elapsed / tr.time;",
     "                ts.blend_factor = ts.elapsed / tr.time;\n                if (ts.elapsed > tr.time) ts.elapsed = tr.time;", "transition.rs:315-321: the factor formed before the clamp"),
    ("by_index_weights_swapped", "anim_planner.h", "its[cnt++] = {(uint32_t)pr, 1.0f - interpolator};", "its[cnt++] = {(uint32_t)pr, interpolator};", "BlendAnimationsByIndex: the previous input's weight"),
    ("new_loop_flag_never_set", "anim_planner.h", "(uint8_t)(1u | (new_loop ? 2u : 0u) | (s.speed > 0.0f ? 4u : 0u));", "(uint8_t)(1u | (s.speed > 0.0f ? 4u : 0u));", "what update_root_motion is told about a wrapped loop"),
    ("max_weight_strategy_takes_the_first_of_equals", "anim_api.hip", "if (strategy == FYX_EVENTS_MAX_WEIGHT) { if (!(w < bw)) { best = (int)i; bw = w; } }",
     "if (strategy == FYX_EVENTS_MAX_WEIGHT) { if (w > bw) { best = (int)i; bw = w; } }", "collect_active_animations_events: max_by's tie-break (the last of equals)"),
    # ---- second batch: the rest of the host code the CPU suite reaches (blend-space weights, the curve simplifier, the shard cuts)
    ("blend_space_inside_test_closed", "anim_planner.h", "if (u >= 0.0f && v >= 0.0f && u + v < 1.0f) {", "if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f) {", "fyrox-math get_barycentric_coords / is_inside: u + v < 1 is strict"),
    ("blend_space_nearest_edge_takes_the_last_of_equals", "anim_planner.h", "if (distance < min_distance) {", "if (distance <= min_distance) {", "blendspace.rs:380-410: the first of equally near edges wins"),
    ("blend_space_two_points_weights_swapped", "anim_planner.h", "w[0] = 1.0f - t; w[1] = t; w[2] = 0.0f;\n                return true;\n            }\n        }\n        const size_t nt", "w[0] = t; w[1] = 1.0f - t; w[2] = 0.0f;\n                return true;\n            }\n        }\n        const size_t nt", "blendspace.rs:350-362: the segment's weights"),
    ("simplify_keeps_a_point_at_epsilon", "host_geom.hip", "if (far == 0 || far_dist < epsilon) continue;", "if (far == 0 || far_dist <= epsilon) continue;", "simplify.rs:128-131: a point exactly epsilon away is kept"),
    ("simplify_farthest_takes_the_last_of_equals", "host_geom.hip", "if (far_dist < dist) { far_dist = dist; far = i; }", "if (far_dist <= dist) { far_dist = dist; far = i; }", "simplify.rs:118-124: the first of equally far points"),
    ("simplify_two_equal_keys_stay_two", "host_geom.hip", "if (m == 2 && std::fabs(y[out_indices[0]] - y[out_indices[1]]) < epsilon) m = 1;", "if (false) m = 1;", "simplify.rs:62-64: a flat curve collapses to one key"),
    ("simplify_step_limit_off_by_one", "host_geom.hip", "next = std::max(k - 1, start + 1); break; }", "next = std::max(k, start + 1); break; }", "simplify.rs:85-101 find_step: the key BEFORE the one that oversteps"),
    ("shard_cut_rounds_the_groups_down", "comm_api.hip", "    const uint64_t groups = ((uint64_t)n_verts + kShardAlign - 1) / kShardAlign;\n    const uint64_t v = (g * groups / n_ranks) * kShardAlign;", "    const uint64_t groups = ((uint64_t)n_verts) / kShardAlign;\n    const uint64_t v = (g * groups / n_ranks) * kShardAlign;", "the ragged cut: a mesh's last partial group belongs to the last rank"),
    ("padded_shard_rounds_down", "comm_api.hip", "return (uint32_t)(((groups + n_ranks - 1) / n_ranks) * kShardAlign);", "return (uint32_t)((groups / n_ranks) * kShardAlign);", "the padded cut: ceil(groups / n_ranks)"),
    ("mesh_upload_attribute_may_overhang_the_stride", "fyx_api.hip", "if (a.off >= 0 && (uint64_t)a.off + a.size > stride)", "if (a.off >= 0 && (uint64_t)a.off > stride)", "fyx_mesh_upload: an attribute that does not fit the vertex is refused"),
    # ---- third batch: MachineLayer::evaluate_pose's order of things (layer.rs:590-706)
    ("transition_to_the_active_state_may_fire", "anim_planner.h", "if ((int32_t)tr.dest == LS.active_state || (int32_t)tr.source != LS.active_state) continue;", "if ((int32_t)tr.source != LS.active_state) continue;",
     "layer.rs:606-611: a transition whose destination is the active state is skipped"),
    ("leave_event_after_the_enter_event", "anim_planner.h", "                        layer_event(LS, FYX_EVENT_STATE_LEAVE, LS.active_state, -1);             // layer.rs:620\n                        if (tr.dest < L.states.size()) apply_actions(L.states[tr.dest].on_enter);\n                        layer_event(LS, FYX_EVENT_STATE_ENTER, (int32_t)tr.dest, -1);            // :634",
     "                        if (tr.dest < L.states.size()) apply_actions(L.states[tr.dest].on_enter);\n                        layer_event(LS, FYX_EVENT_STATE_ENTER, (int32_t)tr.dest, -1);            // :634\n                        layer_event(LS, FYX_EVENT_STATE_LEAVE, LS.active_state, -1);             // layer.rs:620", "the order of a firing transition's events"),
    ("later_transition_wins", "anim_planner.h", "                        layer_event(LS, FYX_EVENT_ACTIVE_TRANSITION_CHANGED, (int32_t)t, -1);    // :645\n                        break;", "                        layer_event(LS, FYX_EVENT_ACTIVE_TRANSITION_CHANGED, (int32_t)t, -1);    // :645\n                        continue;",
     "layer.rs:605-651: the FIRST transition whose condition holds fires, the search stops"),
    ("transition_done_without_epsilon", "anim_planner.h", "                if (fabsf(tr.time - ts.elapsed) <= FLT_EPSILON) {  // is_done", "                if (tr.time == ts.elapsed + 1.0f) {  // is_done",
     "a control: a transition that never finishes"),
    ("state_changed_event_before_transition_changed", "anim_planner.h", "                    LS.active_transition = -1;\n                    layer_event(LS, FYX_EVENT_ACTIVE_TRANSITION_CHANGED, -1, -1);                 // :673\n                    LS.active_state = (int32_t)tr.dest;\n                    layer_event(LS, FYX_EVENT_ACTIVE_STATE_CHANGED, (int32_t)tr.source, (int32_t)tr.dest);  // :677",
     "                    LS.active_transition = -1;\n                    LS.active_state = (int32_t)tr.dest;\n                    layer_event(LS, FYX_EVENT_ACTIVE_STATE_CHANGED, (int32_t)tr.source, (int32_t)tr.dest);  // :677\n                    layer_event(LS, FYX_EVENT_ACTIVE_TRANSITION_CHANGED, -1, -1);                 // :673", "the order of a finished transition's events"),
    ("every_states_root_is_not_evaluated", "anim_planner.h", "            for (const StateDef& s : L.states) eval_node(L, LS, s.root, nr);  // state.update", "            if (LS.active_state >= 0) eval_node(L, LS, L.states[LS.active_state].root, nr); else for (const StateDef& s : L.states) eval_node(L, LS, s.root, nr);  // state.update",
     "layer.rs:601-603: EVERY state's root is evaluated every frame (by-index nodes of inactive states keep their clocks running)"),
]

BATCH2_TESTS = ["tests/test_import_helpers.py", "tests/test_sharding.py", "tests/test_abi.py", "tests/test_oracle_golden.py", "tests/test_cpp_host.py"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    keep = LIB + ".shipped"
    shutil.copy2(LIB, keep)
    res = []
    try:
        for name, fn, old, new, what in MUTANTS:
            path = os.path.join(SRC, fn)
            src = open(path).read()
            if a.only and name not in a.only.split(","):
                continue
            if src.count(old) < 1:      # (a text that occurs twice -- the tick and the steady path's copy of it -- is replaced in both places)
                res.append({"name": name, "file": fn, "what": what, "built": False, "note": "the text does not occur"})
                print(json.dumps(res[-1]), flush=True)
                continue
            t0 = time.time()
            try:
                open(path, "w").write(src.replace(old, new))
                b = subprocess.run(["make", "-C", SRC], capture_output=True, text=True)
                if b.returncode != 0:
                    res.append({"name": name, "file": fn, "what": what, "built": False, "note": b.stderr[-300:]})
                    continue
                r = subprocess.run([sys.executable, "-m", "pytest", *TESTS, *BATCH2_TESTS, "-x", "-q", "-n", "6", "-p", "no:cacheprovider"], cwd=ROOT, capture_output=True, text=True)
                tail = [l for l in r.stdout.splitlines() if l.startswith("FAILED") or l.startswith("ERROR")]
                res.append({"name": name, "file": fn, "what": what, "built": True, "killed": r.returncode != 0, "by": tail[0][:160] if tail else None, "seconds": round(time.time() - t0, 1)})
            finally:
                open(path, "w").write(src)
            print(json.dumps(res[-1]), flush=True)
    finally:
        subprocess.run(["make", "-C", SRC], capture_output=True, text=True)     # the restored sources are newer than the last mutant's objects
        os.remove(keep)
    built = [r for r in res if r.get("built")]
    rec = {"what": "one-line mutants of the host control plane under the CPU control-plane tests", "mutants": len(built), "killed": sum(r["killed"] for r in built),
           "survived": [r["name"] for r in built if not r["killed"]], "results": res}
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("mutants", "killed", "survived")}))


if __name__ == "__main__":
    main()
