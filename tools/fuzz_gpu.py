#!/usr/bin/env python3
"""Bug hunt on the GPU box: random machines (tests/anim_cases.py::random_machine, plain and listy) far beyond the seeds the
test suite pins, every frame of every scenario against the oracle through tests/test_anim_gpu.py::run_scenario.

    python tools/fuzz_gpu.py --first 100 --count 300 [--listy] [--out gpurun_out/fuzz.json]

Prints one JSON record: seeds run, failures (seed, sampler form, instances, first line of the assertion).  Test infrastructure:
the oracle is the checker here exactly as in tests/."""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_with_edits(T, cases, orc, ctx, sc, n_instances, seed):
    """tests/test_anim_gpu.py::run_scenario with the setters of Animation called on both sides between frames (lib.rs:432-466, :664-666,
    track.rs TrackBinding::set_enabled, signal.rs)."""
    import numpy as np
    rng = np.random.default_rng(seed + 5 * 10 ** 6)
    lib = orc._alib()
    f32 = lambda x: float(np.float32(x))
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(ctx, sc, n_instances)
    n_nodes = sc.rig.n_nodes
    alive = [True] * len(sc.animations)
    for f in range(sc.n_frames):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par)
            p.set_parameter(idx, par)
        for a in sc.removals.get(f, []):
            o.remove_animation(a)
            p.remove_animation(a)
            alive[a] = False
        for _ in range(int(rng.integers(0, 3)) if rng.random() < 0.4 else 0):
            a = int(rng.integers(0, len(sc.animations)))
            if not alive[a]:
                continue
            h, spec = o.anims[a], sc.animations[a]
            td = sc.tracks_data[spec.tracks]
            kind = int(rng.integers(0, 11))
            if kind == 0 and td.tracks:
                t, on = int(rng.integers(0, len(td.tracks))), bool(rng.integers(2))
                lib.fo_animation_bind(h, t, int(spec.target[t]), int(on))
                p.set_track_enabled(a, t, on)
            elif kind == 1:
                v = f32(rng.choice([1.0, 0.5, 2.5, -1.0, -0.3, 0.0]))
                lib.fo_animation_set_speed(h, v); p.set_speed(a, v)
            elif kind == 2:
                v = bool(rng.integers(2))
                lib.fo_animation_set_loop(h, int(v)); p.set_loop(a, v)
            elif kind == 3:
                v = bool(rng.random() < 0.7)
                lib.fo_animation_set_enabled(h, int(v)); p.set_enabled(a, v)
            elif kind == 4:
                lib.fo_animation_rewind(h); p.rewind(a)
            elif kind == 5:
                v = f32(rng.random() * 1.4 - 0.2)
                lib.fo_animation_set_time_position(h, v); p.set_time_position(a, v)
            elif kind == 6:
                lo = f32(rng.random() * 0.4)
                hi = f32(lo + 0.05 + rng.random() * 0.6)
                lib.fo_animation_set_time_slice(h, lo, hi); p.set_time_slice(a, lo, hi)
            elif kind == 7 and sc.track_root_motion:
                node = int(rng.integers(-1, min(n_nodes, 4)))
                fl = [bool(rng.integers(2)) for _ in range(4)]
                lib.fo_animation_set_root_motion_settings(h, node, *[int(x) for x in fl])
                p.set_root_motion_settings(a, None if node < 0 else node, *fl)
            elif kind == 8 and spec.signals:
                sg, on = int(rng.integers(0, len(spec.signals))), bool(rng.integers(2))
                lib.fo_animation_set_signal_enabled(h, sg, int(on)); p.set_signal_enabled(a, sg, on)
            elif kind == 9:
                o.clear_events(a); p.clear_events(a)
            elif kind == 10:
                cap = int(rng.integers(0, 6))
                lib.fo_animation_set_max_event_capacity(h, cap); p.set_max_event_capacity(a, cap)
        if sc.machine is None:
            o.update_animations(sc.dt)
            p.update_animations(sc.dt)
        else:
            o.update_machine(sc.dt)
            p.update_machine(sc.dt)
        T.check_frame(p, o, sc, n_instances, f)
    return o, p


class InstanceView:
    """One instance of an animator behind the readers tests/test_anim_gpu.py::check_frame uses."""

    def __init__(self, p, i):
        self.p, self.i = p, i

    def read(self, what):
        return self.p.read(what)[self.i:self.i + 1]

    def animation_root_motion(self, a):
        return self.p.animation_root_motion(a)[self.i:self.i + 1]

    def machine_root_motion(self, layer=-1):
        return self.p.machine_root_motion(layer)[self.i:self.i + 1]

    def read_properties(self, animation=-1):
        return self.p.read_properties(animation)[self.i:self.i + 1]

    def layer_state(self, layer, instance=0):
        return self.p.layer_state(layer, self.i)

    def property_slot(self, node, prop):
        return self.p.property_slot(node, prop)

    def property_count(self):
        return self.p.property_count()


def run_diverge(T, cases, A, orc, ctx, sc, n_inst, seed):
    """Instances that go their own ways: parameters, speeds, time positions, enabled flags and rewinds set PER INSTANCE (the library keeps
    a memo of planned programs per instance and steady frames per animator), one oracle per instance."""
    import numpy as np
    rng = np.random.default_rng(seed + 11 * 10 ** 6)
    f32 = lambda x: float(np.float32(x))
    lib = orc._alib()
    os_ = [cases.build_oracle(orc, sc) for _ in range(n_inst)]
    p = cases.build_product(ctx, sc, n_inst)
    views = [InstanceView(p, i) for i in range(n_inst)]
    watch = sorted({0, n_inst - 1, int(rng.integers(0, n_inst)), int(rng.integers(0, n_inst))})
    try:
        for f in range(sc.n_frames):
            for idx, par in sc.script.get(f, []):
                who = range(n_inst) if rng.random() < 0.5 else [int(rng.integers(0, n_inst))]
                for i in who:
                    os_[i].set_parameter(idx, par)
                if len(who) == n_inst:
                    p.set_parameter(idx, par)
                else:
                    p.set_parameter(idx, par, instance=who[0])
            if rng.random() < 0.35:
                a, i = int(rng.integers(0, len(sc.animations))), int(rng.integers(0, n_inst))
                who = [i] if rng.random() < 0.7 else list(range(n_inst))
                inst = i if len(who) == 1 else A.ALL_INSTANCES
                kind = int(rng.integers(0, 5))
                v = f32(rng.choice([1.0, 0.5, 2.5, -1.0, 0.0])) if kind == 0 else f32(rng.random() * 1.2 - 0.1)
                on = bool(rng.integers(2))
                for j in who:
                    h = os_[j].anims[a]
                    if kind == 0: lib.fo_animation_set_speed(h, v)
                    elif kind == 1: lib.fo_animation_set_time_position(h, v)
                    elif kind == 2: lib.fo_animation_set_enabled(h, int(on))
                    elif kind == 3: lib.fo_animation_rewind(h)
                    else: lib.fo_animation_set_loop(h, int(on))
                if kind == 0: p.set_speed(a, v, instance=inst)
                elif kind == 1: p.set_time_position(a, v, instance=inst)
                elif kind == 2: p.set_enabled(a, on, instance=inst)
                elif kind == 3: p.rewind(a, instance=inst)
                else: p.set_loop(a, on, instance=inst)
            for o in os_:
                (o.update_machine if sc.machine is not None else o.update_animations)(sc.dt)
            (p.update_machine if sc.machine is not None else p.update_animations)(sc.dt)
            for i in watch:
                T.check_frame(views[i], os_[i], sc, 1, f)
        for i in watch:
            for a in range(len(sc.animations)):
                assert T._drain(lambda: p.pop_event(a, i)) == T._drain(lambda: os_[i].pop_event(a)), f"events of animation {a}, instance {i}"
    finally:
        for o in os_:
            o.close()
        p.free()


def run_scene(T, cases, A, orc, ctx, seeds, listy, bones, lattice=False):
    """fyx_scene_update over a changing list of random machines (order shuffled, a member left out of some frames, parameters scripted per
    member) against one oracle per member."""
    import numpy as np
    rng = np.random.default_rng(seeds[0] + 9 * 10 ** 6)
    scs = [cases.random_machine(s, n_bones=bones, listy=listy and k % 2 == 0, lattice=lattice) for k, s in enumerate(seeds)]
    n_inst = [1 + int(rng.integers(0, 3)) for _ in scs]
    dt = scs[0].dt
    os_ = [cases.build_oracle(orc, sc) for sc in scs]
    ps = [cases.build_product(ctx, sc, n) for sc, n in zip(scs, n_inst)]
    try:
        for f in range(28):
            for sc, o, p in zip(scs, os_, ps):
                for idx, par in sc.script.get(f, []):
                    o.set_parameter(idx, par)
                    p.set_parameter(idx, par)
            members = [k for k in rng.permutation(len(scs)) if rng.random() < 0.85]
            for k in members:
                os_[k].update_machine(dt)
            A.scene_update(ctx, [ps[k] for k in members], dt)
            for k in members:
                T.check_frame(ps[k], os_[k], scs[k], n_inst[k], f)
    finally:
        for o in os_:
            o.close()
        for p in ps:
            p.free()


def run_skin(T, cases, A, orc, ctx, sc, n_inst, seed):
    """The frame that goes through to the vertices (fyx_animator_set_skin_output): random bone lists (invalid handles among them), mesh
    sizes, launch forms switched from frame to frame; palette and vertices of every frame against the oracle's palette and loop."""
    import numpy as np
    from fyrox_amd import synth
    rng = np.random.default_rng(seed + 7 * 10 ** 6)
    nn = sc.rig.n_nodes
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(ctx, sc, n_inst)
    base = p.base_id
    bones = [int(b) if rng.random() > 0.1 else -1 for b in rng.integers(0, nn, int(rng.integers(1, 2 * nn)))]
    nb = len(bones)
    nv = int(rng.choice([1, 63, 64, 65, 777, 4097, int(rng.integers(1, 30_000))]))
    A.create_bone_list(ctx, base + 50, base, bones)
    d_pal = ctx.malloc(n_inst * nb * 64)
    p.set_palette_output(base + 50, d_pal.ptr)
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + seed)
    ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs = [ctx.malloc(n_inst * nv * w * 4 + 64) for w in (3, 3, 4)]
    p.set_skin_output(base + 50, base + 60, outs[0].ptr, outs[1].ptr, outs[2].ptr)
    keys = ("debug.frame_skin", "anim.one_launch", "anim.inline_ctrl", "anim.update_lean", "anim.frame_skin_units")
    try:
        for f in range(min(sc.n_frames, 24)):
            for idx, par in sc.script.get(f, []):
                o.set_parameter(idx, par)
                p.set_parameter(idx, par)
            form = (int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([0, 1, 2, 4, 16])))
            for k, v in zip(keys, form):
                ctx.set_option(k, v)
            if rng.random() < 0.3:
                (o.update_machine if sc.machine is not None else o.update_animations)(sc.dt)
                A.scene_update(ctx, [p], sc.dt)
            else:
                (o.update_machine if sc.machine is not None else o.update_animations)(sc.dt)
                (p.update_machine if sc.machine is not None else p.update_animations)(sc.dt)
            ctx.sync()
            ref_pal = o.palette(bones)
            pal = d_pal.download(np.float32, n_inst * nb * 16).reshape(n_inst, nb, 16)
            ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, ref_pal, mesh.normal, mesh.tangent)
            for i in range(n_inst):
                assert np.array_equal(pal[i].view(np.uint32), ref_pal.view(np.uint32)), f"{sc.name} frame {f} form {form}: palette of instance {i}"
            for buf, key, w in zip(outs, ("pos", "normal", "tangent"), (3, 3, 4)):
                got = buf.download(np.uint32, n_inst * nv * w).reshape(n_inst, nv * w)
                for i in range(n_inst):
                    assert np.array_equal(got[i], ref[key].view(np.uint32).reshape(-1)), f"{sc.name} frame {f} form {form}: {key} of instance {i} ({nv} vertices, {nb} bones)"
            T.check_frame(p, o, sc, n_inst, f)
    finally:
        for k, v in zip(keys, (1, 1, 1, 1, 0)):
            ctx.set_option(k, v)
        p.set_skin_output(base + 50, base + 60)
        o.close()
        p.free()
        for b in outs:
            b.free()
        d_pal.free()
        ctx.mesh_free(base + 60)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--listy", action="store_true")
    ap.add_argument("--lattice", action="store_true", help="blend-space points and sampling points on a coarse lattice (points ON edges and corners)")
    ap.add_argument("--subnormal", action="store_true", help="tests/anim_cases.py::with_subnormal_values over every scenario: Position keys x 1e-39, a third of the Scale keys x 1e-13")
    ap.add_argument("--bones", type=int, default=7)
    ap.add_argument("--curves", action="store_true", help="tests/anim_cases.py::random_curves instead of random_machine")
    ap.add_argument("--edits", action="store_true", help="random run-time edits between frames: track bindings switched, speeds, loops, "
                    "time positions and slices, enabled flags, rewinds, root-motion settings, signals, event queues")
    ap.add_argument("--diverge", action="store_true", help="per-instance parameters and clocks, one oracle per instance")
    ap.add_argument("--skin", action="store_true", help="with a registered palette and skin output: the frame through to the vertices")
    ap.add_argument("--scene", type=int, default=0, help="members per scene: the seeds run in groups through fyx_scene_update")
    ap.add_argument("--opt", action="append", default=[], help="key=value context options for the whole run")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import fyrox_amd
    import oracle
    import anim_cases as cases
    import test_anim_gpu as T
    from fyrox_amd import anim as A

    oracle.lib()
    ctx = fyrox_amd.Context(0)
    fails, t0 = [], time.time()
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    if args.scene:
        for first in range(args.first, args.first + args.count, args.scene):
            seeds = list(range(first, first + args.scene))
            ctx.set_option("anim.sample_form", first % 3)
            try:
                run_scene(T, cases, A, oracle, ctx, seeds, args.listy, args.bones, args.lattice)
            except Exception as e:   # noqa: BLE001
                fails.append({"seeds": [seeds[0], seeds[-1]], "listy": args.listy, "sample_form": first % 3,
                              "error": (str(e).strip().splitlines() or [repr(e)])[0][:300],
                              "where": traceback.format_exc().strip().splitlines()[-3][:200]})
    for seed in (range(args.first, args.first + args.count) if not args.scene else ()):
        form, n_inst = seed % 3, 1 + seed % 4 if seed % 5 else 70
        sc = cases.random_curves(seed, n_bones=args.bones) if args.curves else cases.random_machine(seed, n_bones=args.bones, listy=args.listy, lattice=args.lattice)
        if args.subnormal:
            sc = cases.with_subnormal_values(lambda sc=sc: sc)()
        ctx.set_option("anim.sample_form", form)
        o = p = None
        try:
            if args.skin:
                run_skin(T, cases, A, oracle, ctx, sc, min(n_inst, 3), seed)
                continue
            if args.diverge:
                run_diverge(T, cases, A, oracle, ctx, sc, max(n_inst, 2) if n_inst < 70 else 40, seed)
                continue
            o, p = (run_with_edits(T, cases, oracle, ctx, sc, n_inst, seed) if args.edits else T.run_scenario(ctx, oracle, sc, n_instances=n_inst))
            for a in range(len(sc.animations)):
                assert T._drain(lambda: p.pop_event(a, 0)) == T._drain(lambda: o.pop_event(a)), f"events of animation {a}"
        except Exception as e:   # noqa: BLE001 -- every failure is a finding
            fails.append({"seed": seed, "listy": args.listy, "sample_form": form, "instances": n_inst,
                          "error": (str(e).strip().splitlines() or [repr(e)])[0][:300],
                          "where": traceback.format_exc().strip().splitlines()[-3][:200]})
        finally:
            try:
                if p is not None:
                    p.free()
                if o is not None:
                    o.close()
            except Exception:   # noqa: BLE001
                pass
    ctx.set_option("anim.sample_form", 0)
    rec = {"what": "random machines on the GPU against the oracle, every frame", "listy": args.listy, "lattice": args.lattice, "subnormal": args.subnormal, "curves": args.curves, "edits": args.edits, "diverge": args.diverge, "skin": args.skin, "scene": args.scene, "options": args.opt, "bones": args.bones,
           "first_seed": args.first, "seeds": args.count, "failures": len(fails), "failed": fails[:40], "seconds": round(time.time() - t0, 1)}
    line = json.dumps(rec)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
