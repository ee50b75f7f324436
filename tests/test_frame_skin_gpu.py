"""One character's frame as ONE launch all the way to the vertices (fyx_animator_set_skin_output, FrameSkin in csrc/fyx_internal.h).

In the engine a character's frame is one dependent chain: Machine::evaluate_pose (fyrox-animation/src/machine/mod.rs:344-382) ->
hierarchy (scene/graph/mod.rs:1199-1241) -> bone matrices (scene/mesh/mod.rs:781-793) -> the skinning loop (mesh/mod.rs:501-522).
The launch that samples and updates a character also holds the workgroups that skin its meshes: they form the palette on chip.

Bar: the vertices are those of fyx_lbs_skin_device on the palette the same update wrote to memory, BIT FOR BIT (and that palette
and those vertices are checked against the oracle); every fallback (separate launches) gives the same bits; a wait that cannot be
satisfied is an ERROR, never a frame made of stale data.
"""
import numpy as np
import pytest

import fyrox_amd
from fyrox_amd import _native
from fyrox_amd import anim as A
from fyrox_amd import synth

import anim_cases as cases
from test_anim_gpu import check_frame

pytestmark = pytest.mark.gpu


class Outs:
    def __init__(self, ctx, n):
        self.n = n
        self.pos, self.nrm, self.tan = ctx.malloc(n * 12 + 64), ctx.malloc(n * 12 + 64), ctx.malloc(n * 16 + 64)
        for b, w in ((self.pos, 3), (self.nrm, 3), (self.tan, 4)):
            b.upload(np.full(n * w, np.nan, np.float32))

    def get(self):
        return (self.pos.download(np.uint32, self.n * 3), self.nrm.download(np.uint32, self.n * 3), self.tan.download(np.uint32, self.n * 4))

    def free(self):
        for b in (self.pos, self.nrm, self.tan):
            b.free()


def _update(p, sc):
    (p.update_machine if sc.machine is not None else p.update_animations)(sc.dt)


def _oupdate(o, sc):
    (o.update_machine if sc.machine is not None else o.update_animations)(sc.dt)


@pytest.mark.parametrize("make,n_inst,n_verts", [
    (cases.c5_blend_tree, 1, 100_000), (cases.player_only, 1, 50_000), (cases.transitions, 2, 4097), (cases.layered, 1, 20_001),
    (cases.by_index, 3, 777), (cases.program_forms, 1, 9000), (cases.blend_space, 1, 63), (cases.masked_transitions, 2, 12_345)],
    ids=lambda v: getattr(v, "__name__", str(v)))
def test_the_frame_that_skins_equals_update_then_lbs_skin(ctx, orc, make, n_inst, n_verts):
    """Every frame: outputs of the skin output == fyx_lbs_skin_device on the palette the update wrote; the pose side (poses, TRS,
    matrices) against the oracle as in test_anim_gpu; at the end the vertices against the oracle's loop on the oracle's palette (exact
    where no Euler track is involved)."""
    _frame_that_skins(ctx, orc, make(), n_inst, n_verts)


@pytest.mark.parametrize("entry", [3, 7, 11, 15], ids=["m30", "m31", "m32", "m33"])
def test_the_frame_that_skins_takes_the_homogeneous_path_for_any_entry_of_the_last_row(ctx, orc, entry):
    """The skinning workgroups of the one-launch frame form their palette on chip and decide THERE whether a matrix is affine (their own
    copy of the test, anim_kernels.hip frame_skin_body).  A bone whose inverse bind matrix has ONE non-affine entry in its last row gives a
    palette matrix with exactly that last row (the global matrix's is (0, 0, 0, 1)): each entry alone must take the divide path --
    bit for bit what update + lbs_skin gives, and the oracle."""
    sc = cases.transitions()
    ib = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (sc.rig.n_nodes, 1)) if sc.rig.inv_bind is None else np.array(sc.rig.inv_bind, np.float32).reshape(-1, 16)
    ib[5, entry] = {3: 0.125, 7: -0.25, 11: 0.5, 15: 2.0}[entry]
    sc.rig.inv_bind = ib
    _frame_that_skins(ctx, orc, sc, 2, 4097)


def _frame_that_skins(ctx, orc, sc, n_inst, n_verts):
    nb = sc.rig.n_nodes
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(ctx, sc, n_inst)
    base = p.base_id
    bones = list(range(nb))
    A.create_bone_list(ctx, base + 50, base, bones)
    d_pal = ctx.malloc(n_inst * nb * 64)
    p.set_palette_output(base + 50, d_pal.ptr)
    mesh = synth.make_mesh(n_verts, nb, synth.SEED_BASE + 21)
    ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    fused, sep = Outs(ctx, n_inst * n_verts), Outs(ctx, n_inst * n_verts)
    p.set_skin_output(base + 50, base + 60, fused.pos.ptr, fused.nrm.ptr, fused.tan.ptr)
    try:
        for f in range(14):
            for idx, par in sc.script.get(f, []):
                o.set_parameter(idx, par)
                p.set_parameter(idx, par)
            _oupdate(o, sc)
            _update(p, sc)
            ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_inst, sep.pos.ptr, sep.nrm.ptr, sep.tan.ptr)
            for a, b, what in zip(fused.get(), sep.get(), ("position", "normal", "tangent")):
                assert np.array_equal(a, b), f"{sc.name} frame {f}: {what} of the frame that skins differs from update + lbs_skin ({int((a != b).sum())} words)"
            check_frame(p, o, sc, n_inst, f)
        pal = d_pal.download(np.float32, n_inst * nb * 16).reshape(n_inst, nb, 16)
        ref_pal = o.palette(bones)
        got = fused.pos.download(np.float32, n_inst * n_verts * 3).reshape(n_inst, n_verts, 3)
        for i in range(n_inst):
            ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal[i], mesh.normal, mesh.tangent)
            assert np.array_equal(got[i].view(np.uint32), ref["pos"].view(np.uint32)), "vertices vs the oracle's loop on the GPU's palette"
            if not sc.has_euler:
                assert np.array_equal(pal[i].view(np.uint32), ref_pal.view(np.uint32)), "palette vs the oracle"
    finally:
        p.set_skin_output(base + 50, base + 60)
        o.close()
        p.free()
        for b in (fused, sep):
            b.free()
        d_pal.free()
        ctx.mesh_free(base + 60)


def test_forms_of_the_frame_and_every_fallback_give_the_same_vertices(ctx, orc):
    """anim.frame_skin / anim.one_launch / anim.inline_ctrl / anim.update_lean / anim.frame_skin_units switched from frame to frame; a mask
    of outputs (position only); two skin outputs on two bone lists (one with an invalid bone handle: identity matrix)."""
    sc = cases.c5_blend_tree(euler_every=10 ** 6)
    nb = sc.rig.n_nodes
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(ctx, sc, 2)
    base = p.base_id
    bones_a = list(range(nb))
    bones_b = [-1] + list(range(nb - 1, 0, -1))[: nb // 2]
    A.create_bone_list(ctx, base + 50, base, bones_a)
    A.create_bone_list(ctx, base + 51, base, bones_b)
    pal_a, pal_b = ctx.malloc(2 * len(bones_a) * 64), ctx.malloc(2 * len(bones_b) * 64)
    p.set_palette_output(base + 50, pal_a.ptr)
    p.set_palette_output(base + 51, pal_b.ptr)
    mesh_a = synth.make_mesh(30_000, len(bones_a), synth.SEED_BASE + 22)
    mesh_b = synth.make_mesh(5_001, len(bones_b), synth.SEED_BASE + 23)
    ctx.mesh_upload_soa(base + 60, mesh_a.pos, mesh_a.weights, mesh_a.indices, mesh_a.normal, mesh_a.tangent)
    ctx.mesh_upload_soa(base + 61, mesh_b.pos, mesh_b.weights, mesh_b.indices, mesh_b.normal, mesh_b.tangent)
    fa, sa = Outs(ctx, 2 * 30_000), Outs(ctx, 2 * 30_000)
    fb, sb = Outs(ctx, 2 * 5_001), Outs(ctx, 2 * 5_001)
    p.set_skin_output(base + 50, base + 60, fa.pos.ptr, fa.nrm.ptr, fa.tan.ptr)
    p.set_skin_output(base + 51, base + 61, fb.pos.ptr, 0, 0)          # position only
    forms = [(1, 1, 1, 1, 0), (0, 1, 1, 1, 1), (1, 0, 1, 1, 2), (1, 1, 0, 1, 1), (1, 1, 1, 0, 4), (0, 0, 0, 0, 1), (1, 1, 1, 1, 16), (1, 1, 1, 1, 2)]
    try:
        for f in range(24):
            fs, one, inl, lean, units = forms[f % len(forms)]
            for k, v in (("debug.frame_skin", fs), ("anim.one_launch", one), ("anim.inline_ctrl", inl), ("anim.update_lean", lean), ("anim.frame_skin_units", units)):
                ctx.set_option(k, v)
            _oupdate(o, sc)
            _update(p, sc)
            ctx.lbs_skin_device(base + 60, pal_a.ptr, len(bones_a), 2, sa.pos.ptr, sa.nrm.ptr, sa.tan.ptr)
            ctx.lbs_skin_device(base + 61, pal_b.ptr, len(bones_b), 2, sb.pos.ptr, 0, 0)
            for x, y in zip(fa.get(), sa.get()):
                assert np.array_equal(x, y), f"frame {f} form {forms[f % len(forms)]}: mesh a"
            assert np.array_equal(fb.get()[0], sb.get()[0]), f"frame {f} form {forms[f % len(forms)]}: mesh b"
            assert np.isnan(fb.nrm.download(np.float32, 16)).all(), "an output that is not wanted is not written"
            check_frame(p, o, sc, 2, f)
        ref_b = o.palette(bones_b)
        got_b = pal_b.download(np.float32, 2 * len(bones_b) * 16).reshape(2, len(bones_b), 16)
        assert np.array_equal(got_b[0].view(np.uint32), ref_b.view(np.uint32))
        assert np.array_equal(got_b[0, 0], np.eye(4, dtype=np.float32).reshape(16)), "invalid bone handle -> identity"
    finally:
        for k, v in (("anim.frame_skin", 1), ("anim.one_launch", 1), ("anim.inline_ctrl", 1), ("anim.update_lean", 1), ("anim.frame_skin_units", 0)):
            ctx.set_option(k, v)
        o.close()
        p.free()
        for b in (fa, sa, fb, sb):
            b.free()
        pal_a.free()
        pal_b.free()
        ctx.mesh_free(base + 60)
        ctx.mesh_free(base + 61)


@pytest.mark.parametrize("what", ["crowd", "root_motion", "scene", "big_scene", "update_transforms", "fused_arithmetic"])
def test_frames_that_cannot_take_the_skinning_along(ctx, orc, what):
    """A crowd (the frame is not one launch), root motion (two more kernels), fyx_scene_update, fyx_animator_update_transforms, and
    lbs.exact = 0: the skin outputs are still written by the update call, with fyx_lbs_skin_device's bits."""
    if what == "root_motion":
        sc = cases.with_root_motion_and_signals(cases.transitions)()
    else:
        sc = cases.c5_blend_tree(euler_every=10 ** 6)
    n_inst = 40 if what == "crowd" else 1
    nb = sc.rig.n_nodes
    p = cases.build_product(ctx, sc, n_inst)
    base = p.base_id
    A.create_bone_list(ctx, base + 50, base, list(range(nb)))
    d_pal = ctx.malloc(n_inst * nb * 64)
    p.set_palette_output(base + 50, d_pal.ptr)
    nv = 300_000 if what == "big_scene" else 3000      # big_scene: past kSceneSkinMaxUnits -- ONE batched launch behind the scene's update launch
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 24)
    ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    fused, sep = Outs(ctx, n_inst * nv), Outs(ctx, n_inst * nv)
    p.set_skin_output(base + 50, base + 60, fused.pos.ptr, fused.nrm.ptr, fused.tan.ptr)
    if what == "fused_arithmetic":
        ctx.set_option("lbs.exact", 0)
    try:
        for f in range(6):
            if what in ("scene", "big_scene"):
                A.scene_update(ctx, [p], sc.dt)
            elif what == "update_transforms":
                p.update_transforms()
            else:
                _update(p, sc)
            ctx.lbs_skin_device(base + 60, d_pal.ptr, nb, n_inst, sep.pos.ptr, sep.nrm.ptr, sep.tan.ptr)
            for a, b in zip(fused.get(), sep.get()):
                assert np.array_equal(a, b), f"{what}: frame {f}"
    finally:
        ctx.set_option("lbs.exact", 1)
        p.free()
        for b in (fused, sep):
            b.free()
        d_pal.free()
        ctx.mesh_free(base + 60)


def test_skin_output_arguments_are_checked(ctx):
    sc = cases.player_only()
    nb = sc.rig.n_nodes
    p = cases.build_product(ctx, sc, 1)
    base = p.base_id
    A.create_bone_list(ctx, base + 50, base, list(range(nb)))
    A.create_bone_list(ctx, base + 51, base, list(range(4)))
    d_pal, d_out = ctx.malloc(nb * 64), ctx.malloc(1000 * 16 + 64)
    mesh = synth.make_mesh(1000, nb, synth.SEED_BASE + 25)
    ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices)          # no normals, no tangents
    try:
        with pytest.raises(fyrox_amd.FyxError) as e:
            p.set_skin_output(base + 50, base + 60, d_out.ptr)                    # not a palette output yet
        assert e.value.code == _native.FYX_ERR_INVALID_ARG
        p.set_palette_output(base + 50, d_pal.ptr)
        p.set_palette_output(base + 51, d_pal.ptr)
        with pytest.raises(fyrox_amd.FyxError) as e:
            p.set_skin_output(base + 50, base + 61, d_out.ptr)                    # unknown mesh
        assert e.value.code == _native.FYX_ERR_UNKNOWN_ID
        with pytest.raises(fyrox_amd.FyxError) as e:
            p.set_skin_output(base + 50, base + 60, d_out.ptr, d_out.ptr)         # the mesh has no normals
        assert e.value.code == _native.FYX_ERR_MISSING_ATTRIBUTE
        with pytest.raises(fyrox_amd.FyxError) as e:
            p.set_skin_output(base + 51, base + 60, d_out.ptr)                    # the mesh references bones the list does not have
        assert e.value.code == _native.FYX_ERR_BONE_INDEX
        p.set_skin_output(base + 50, base + 60, d_out.ptr)
        p.update_animations(sc.dt)
        ctx.mesh_free(base + 60)                                                  # freed while registered: the next update says so
        with pytest.raises(fyrox_amd.FyxError) as e:
            p.update_animations(sc.dt)
        assert e.value.code == _native.FYX_ERR_UNKNOWN_ID
        p.set_palette_output(base + 50, 0)                                        # removing the palette output removes its skin outputs
        p.update_animations(sc.dt)
        ctx.sync()
    finally:
        p.free()
        d_pal.free()
        d_out.free()


def test_a_wait_that_cannot_be_satisfied_costs_no_frame(ctx, orc):
    """The device counter of the one-launch frame poisoned from outside (fyx_debug_frame_counter_add): the frame's update and skinning
    workgroups give up after anim.wait_timeout_ms and compute NOTHING; the next fyx_sync sees their report, switches the context to
    separate launches and runs THAT frame again (the engine's chain never skips a frame) -- it returns FYX_OK, the
    palette and the vertices are the poisoned frame's own, bit for bit against the oracle, and fyx_last_error / debug.frames_reissued say
    what happened; the context goes on bit-exact, and with the counter repaired the one-launch form works again."""
    sc = cases.c5_blend_tree(euler_every=10 ** 6)
    nb = sc.rig.n_nodes
    o = cases.build_oracle(orc, sc)
    p = cases.build_product(ctx, sc, 1)
    base = p.base_id
    A.create_bone_list(ctx, base + 50, base, list(range(nb)))
    d_pal = ctx.malloc(nb * 64)
    p.set_palette_output(base + 50, d_pal.ptr)
    mesh = synth.make_mesh(4000, nb, synth.SEED_BASE + 26)
    ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    outs = Outs(ctx, 4000)
    p.set_skin_output(base + 50, base + 60, outs.pos.ptr, outs.nrm.ptr, outs.tan.ptr)
    ctx.set_option("anim.wait_timeout_ms", 20)
    try:
        for f in range(3):
            _oupdate(o, sc)
            _update(p, sc)
        ctx.sync()
        pal_before, pos_before = d_pal.download(np.uint32, nb * 16), outs.pos.download(np.uint32, 4000 * 3)
        l = _native.lib()
        assert l.fyx_debug_frame_counter_add(ctx._h, p.id, -100000) == 0
        _oupdate(o, sc)
        reissued = ctx.get_option("debug.frames_reissued")
        _update(p, sc)                       # the launch is issued; its workgroups will give up
        ctx.sync()                           # ... and the frame is run again as separate launches, inside this call
        msg = ctx.last_error()
        assert msg.startswith("warning") and "anim.one_launch" in msg and "AGAIN" in msg and str(p.id) in msg, msg
        assert ctx.get_option("anim.one_launch") == 0 and ctx.get_option("debug.frames_reissued") == reissued + 1
        ref_pal = o.palette(list(range(nb)))                  # the poisoned frame's own palette and vertices
        assert not np.array_equal(ref_pal.view(np.uint32).reshape(-1), pal_before)
        assert np.array_equal(d_pal.download(np.float32, nb * 16).reshape(nb, 16).view(np.uint32), ref_pal.view(np.uint32)), "the re-issued frame's palette"
        ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, ref_pal, mesh.normal, mesh.tangent)
        assert not np.array_equal(ref["pos"].view(np.uint32).reshape(-1), pos_before)
        assert np.array_equal(outs.pos.download(np.float32, 4000 * 3).reshape(-1, 3).view(np.uint32), ref["pos"].view(np.uint32)), "... and its vertices"
        # separate launches from here on
        for f in range(3):
            _oupdate(o, sc)
            _update(p, sc)
        ctx.sync()
        ref_pal = o.palette(list(range(nb)))
        assert np.array_equal(d_pal.download(np.float32, nb * 16).reshape(nb, 16).view(np.uint32), ref_pal.view(np.uint32))
        ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, ref_pal, mesh.normal, mesh.tangent)
        assert np.array_equal(outs.pos.download(np.float32, 4000 * 3).reshape(-1, 3).view(np.uint32), ref["pos"].view(np.uint32))
        # repaired: the one-launch form again
        assert l.fyx_debug_frame_counter_add(ctx._h, p.id, 100000) == 0
        ctx.set_option("anim.one_launch", 1)
        for f in range(3):
            _oupdate(o, sc)
            _update(p, sc)
        ctx.sync()
        ref_pal = o.palette(list(range(nb)))
        assert np.array_equal(d_pal.download(np.float32, nb * 16).reshape(nb, 16).view(np.uint32), ref_pal.view(np.uint32))
        ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, ref_pal, mesh.normal, mesh.tangent)
        assert np.array_equal(outs.pos.download(np.float32, 4000 * 3).reshape(-1, 3).view(np.uint32), ref["pos"].view(np.uint32))
    finally:
        ctx.set_option("anim.wait_timeout_ms", 500)
        ctx.set_option("anim.one_launch", 1)
        o.close()
        p.free()
        outs.free()
        d_pal.free()
        ctx.mesh_free(base + 60)


def test_frames_that_skin_over_many_frames_beside_a_chip_filling_co_runner(ctx):
    """10 000 one-launch frames that skin, with a chip-filling skinning launch of another mesh in flight on the library's launch streams
    all the time (the places the frame's workgroups wait in are contended), against an animator in the same state that runs as
    separate launches: palettes and vertices bit for bit, checked every 500 frames and at the end."""
    sc = cases.c5_blend_tree(euler_every=10 ** 6)
    nb = sc.rig.n_nodes
    ps, pals, outs = [], [], []
    nv = 20_000
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 27)
    big = synth.make_mesh(1_000_000, nb, synth.SEED_BASE + 28)
    for k in range(2):
        p = cases.build_product(ctx, sc, 1)
        A.create_bone_list(ctx, p.base_id + 50, p.base_id, list(range(nb)))
        d = ctx.malloc(nb * 64)
        p.set_palette_output(p.base_id + 50, d.ptr)
        ps.append(p)
        pals.append(d)
        outs.append(Outs(ctx, nv))
    ctx.mesh_upload_soa(9710, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    ctx.mesh_upload_soa(9711, big.pos, big.weights, big.indices, big.normal, big.tangent)
    big_out = Outs(ctx, 1_000_000)
    big_pal = ctx.malloc(nb * 64)
    big_pal.upload(synth.make_palette(nb, synth.SEED_BASE + 28))
    ps[0].set_skin_output(ps[0].base_id + 50, 9710, outs[0].pos.ptr, outs[0].nrm.ptr, outs[0].tan.ptr)
    try:
        for f in range(10_000):
            ctx.lbs_skin_device(9711, big_pal.ptr, nb, 1, big_out.pos.ptr, big_out.nrm.ptr, big_out.tan.ptr)      # a launch stream: beside the frames
            ctx.set_option("anim.one_launch", 1)
            ps[0].update_machine(sc.dt)
            ctx.set_option("anim.one_launch", 0)
            ps[1].update_machine(sc.dt)
            ctx.lbs_skin_device(9710, pals[1].ptr, nb, 1, outs[1].pos.ptr, outs[1].nrm.ptr, outs[1].tan.ptr)
            if f % 500 == 499 or f < 3:
                assert np.array_equal(pals[0].download(np.uint32, nb * 16), pals[1].download(np.uint32, nb * 16)), f"frame {f}: palettes"
                for a, b in zip(outs[0].get(), outs[1].get()):
                    assert np.array_equal(a, b), f"frame {f}: vertices"
        ctx.sync()
    finally:
        ctx.set_option("anim.one_launch", 1)
        for p in ps:
            p.free()
        for b in outs + [big_out]:
            b.free()
        for b in pals + [big_pal]:
            b.free()
        ctx.mesh_free(9710)
        ctx.mesh_free(9711)


def test_a_scene_whose_update_launch_skins(ctx, orc):
    """fyx_scene_update over several animators, some with skin outputs (one with two, one with two instances), one without: the
    scene's update stage holds their skinning workgroups (pose_update_skin_scene_kernel).  Every frame: the skin outputs against
    fyx_lbs_skin_device on the palettes the same update wrote; palettes of every animator against an oracle; options switched
    between frames (anim.frame_skin = 0: separate launches behind the scene's, 2: the update stage skins, 3: the WHOLE scene -- samplers,
    updates, skinning -- is one launch with per-character waits; anim.frame_skin_units; anim.update_lean)."""
    specs = [(cases.c5_blend_tree(euler_every=10 ** 6), 1, 6000), (cases.transitions(), 2, 1500), (cases.layered(), 1, 0),
             (cases.player_only(euler_every=10 ** 6), 1, 9001), (cases.by_index(), 1, 300)]
    chars = []
    for sc, n_inst, nv in specs:
        o = cases.build_oracle(orc, sc)
        p = cases.build_product(ctx, sc, n_inst)
        nb = sc.rig.n_nodes
        base = p.base_id
        A.create_bone_list(ctx, base + 50, base, list(range(nb)))
        d_pal = ctx.malloc(n_inst * nb * 64)
        p.set_palette_output(base + 50, d_pal.ptr)
        rec = {"sc": sc, "o": o, "p": p, "nb": nb, "n_inst": n_inst, "pal": d_pal, "meshes": []}
        if nv:
            for j, (verts, mask) in enumerate(((nv, 7), (nv // 3 + 65, 1)) if sc.name == "c5_blend_tree" else ((nv, 7),)):
                mesh = synth.make_mesh(verts, nb, synth.SEED_BASE + 30 + j)
                ctx.mesh_upload_soa(base + 60 + j, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
                f, s = Outs(ctx, n_inst * verts), Outs(ctx, n_inst * verts)
                p.set_skin_output(base + 50, base + 60 + j, f.pos.ptr, f.nrm.ptr if mask & 2 else 0, f.tan.ptr if mask & 4 else 0)
                rec["meshes"].append((base + 60 + j, verts, mask, f, s))
        chars.append(rec)
    forms = [(1, 0, 1), (0, 0, 1), (2, 4, 1), (3, 0, 1), (1, 1, 0), (3, 2, 0), (2, 16, 1), (3, 8, 1), (1, 0, 1)]
    try:
        for f in range(27):
            fs, units, lean = forms[f % len(forms)]
            ctx.set_option("debug.frame_skin", fs)
            ctx.set_option("anim.frame_skin_units", units)
            ctx.set_option("anim.update_lean", lean)
            for ch in chars:
                for idx, par in ch["sc"].script.get(f, []):
                    ch["o"].set_parameter(idx, par)
                    ch["p"].set_parameter(idx, par)
                _oupdate(ch["o"], ch["sc"])
            A.scene_update(ctx, [ch["p"] for ch in chars], chars[0]["sc"].dt)
            for ch in chars:
                for mid, verts, mask, fo, so in ch["meshes"]:
                    ctx.lbs_skin_device(mid, ch["pal"].ptr, ch["nb"], ch["n_inst"], so.pos.ptr, so.nrm.ptr if mask & 2 else 0, so.tan.ptr if mask & 4 else 0)
                    for k, (x, y) in enumerate(zip(fo.get(), so.get())):
                        if mask & (1, 2, 4)[k]:
                            assert np.array_equal(x, y), f"frame {f} form {forms[f % len(forms)]}: {ch['sc'].name} mesh {mid} stream {k}"
                pal = ch["pal"].download(np.float32, ch["n_inst"] * ch["nb"] * 16).reshape(ch["n_inst"], ch["nb"], 16)
                ref = ch["o"].palette(list(range(ch["nb"])))
                assert np.array_equal(pal[0].view(np.uint32), ref.view(np.uint32)), f"frame {f}: palette of {ch['sc'].name}"
    finally:
        for k, v in (("anim.frame_skin", 1), ("anim.frame_skin_units", 0), ("anim.update_lean", 1)):
            ctx.set_option(k, v)
        for ch in chars:
            ch["o"].close()
            ch["p"].free()
            ch["pal"].free()
            for mid, verts, mask, fo, so in ch["meshes"]:
                fo.free()
                so.free()
                ctx.mesh_free(mid)


def test_a_scene_frame_in_one_launch_over_many_frames_and_its_timeout(ctx, orc):
    """anim.frame_skin = 3: fyx_scene_update runs samplers, updates and skinning of every character in ONE launch (scene_frame_kernel).
    2000 frames of 24 characters (more workgroups than the chip holds at once: the waits really wait) against the same scene run
    stage by stage (anim.frame_skin = 0) in a second set of animators: palettes and vertices bit for bit; then one character's
    counter poisoned: the frame reports, writes nothing for that character, and the context goes on with separate launches."""
    sc = cases.c5_blend_tree(euler_every=10 ** 6)
    nb = sc.rig.n_nodes
    nv = 9000
    mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 31)
    ctx.mesh_upload_soa(9800, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    sets = []
    for s_ in range(2):
        chars = []
        for k in range(24):
            p = cases.build_product(ctx, sc, 1)
            for a in range(len(sc.animations)):
                p.set_time_position(a, (k * 0.37 + a * 0.11) % 1.0)
            A.create_bone_list(ctx, p.base_id + 50, p.base_id, list(range(nb)))
            d_pal = ctx.malloc(nb * 64)
            p.set_palette_output(p.base_id + 50, d_pal.ptr)
            o = Outs(ctx, nv)
            p.set_skin_output(p.base_id + 50, 9800, o.pos.ptr, o.nrm.ptr, o.tan.ptr)
            chars.append((p, d_pal, o))
        sets.append(chars)
    ctx.set_option("anim.wait_timeout_ms", 50)
    try:
        for f in range(2000):
            ctx.set_option("debug.frame_skin", 3)
            A.scene_update(ctx, [c_[0] for c_ in sets[0]], sc.dt)
            ctx.set_option("anim.frame_skin", 0)
            A.scene_update(ctx, [c_[0] for c_ in sets[1]], sc.dt)
            if f % 400 == 399 or f < 2:
                for (pa, da, oa), (pb, db, ob) in zip(sets[0], sets[1]):
                    assert np.array_equal(da.download(np.uint32, nb * 16), db.download(np.uint32, nb * 16)), f"frame {f}: palette"
                    for x, y in zip(oa.get(), ob.get()):
                        assert np.array_equal(x, y), f"frame {f}: vertices"
        ctx.sync()
        # a counter poisoned from outside: that frame reports, and the whole scene's frame is run again stage by stage inside fyx_sync --
        # the poisoned character's outputs are THIS frame's, equal to the other set's (same scene, separate launches)
        victim = sets[0][5]
        before = victim[2].pos.download(np.uint32, nv * 3)
        assert _native.lib().fyx_debug_frame_counter_add(ctx._h, victim[0].id, -100000) == 0
        ctx.set_option("debug.frame_skin", 3)
        A.scene_update(ctx, [c_[0] for c_ in sets[0]], sc.dt)
        ctx.sync()
        assert ctx.last_error().startswith("warning") and str(victim[0].id) in ctx.last_error()
        assert ctx.get_option("anim.one_launch") == 0
        ctx.set_option("anim.frame_skin", 0)
        A.scene_update(ctx, [c_[0] for c_ in sets[1]], sc.dt)
        ctx.sync()
        for (pa, da, oa), (pb, db, ob) in zip(sets[0], sets[1]):
            assert np.array_equal(da.download(np.uint32, nb * 16), db.download(np.uint32, nb * 16)), "palette of the re-issued scene frame"
            for x, y in zip(oa.get(), ob.get()):
                assert np.array_equal(x, y), "vertices of the re-issued scene frame"
        assert not np.array_equal(victim[2].pos.download(np.uint32, nv * 3), before)
        assert _native.lib().fyx_debug_frame_counter_add(ctx._h, victim[0].id, 100000) == 0
        A.scene_update(ctx, [c_[0] for c_ in sets[0]], sc.dt)       # stage by stage now (anim.one_launch = 0)
        ctx.sync()
    finally:
        ctx.set_option("anim.wait_timeout_ms", 500)
        ctx.set_option("anim.frame_skin", 1)
        ctx.set_option("anim.one_launch", 1)
        for chars in sets:
            for p, d, o in chars:
                p.free()
                d.free()
                o.free()
        ctx.mesh_free(9800)


@pytest.mark.parametrize("mode", [1, 2], ids=["streams_by_frame", "streams_by_kind"])
@pytest.mark.parametrize("frame_skin", [0, 1, 3], ids=["batch_behind_the_update", "update_stage_skins", "one_launch"])
def test_a_pipelined_scene_with_palette_pairs_equals_the_one_stream_scene(ctx, orc, frame_skin, mode):
    """anim.overlap with palette PAIRS (fyx_animator_set_palette_output_pair): registered once, the frames of the two frame streams
    write one buffer each, the registered skin outputs read the frame's own, and the library orders frame n + 1's skinning of the
    (same) vertex buffers behind frame n's.  N frames are issued WITHOUT a host wait in between; afterwards the vertices are those of
    the last frame and the two palette buffers those of the last two frames of the same scene run on one stream -- bit for bit --
    and the last frame's palettes are the oracle's."""
    specs = [(cases.c5_blend_tree(euler_every=10 ** 6), 1, 7000), (cases.transitions(), 2, 2500), (cases.player_only(euler_every=10 ** 6), 1, 9001),
             (cases.by_index(), 1, 3000), (cases.layered(), 1, 0), (cases.c5_blend_tree(seed=synth.SEED_BASE + 77, euler_every=10 ** 6), 1, 12_000)]
    n_frames = 23
    runs = []
    for overlap in (0, mode):
        chars = []
        for sc, n_inst, nv in specs:
            p = cases.build_product(ctx, sc, n_inst)
            nb, base = sc.rig.n_nodes, p.base_id
            A.create_bone_list(ctx, base + 50, base, list(range(nb)))
            pals = [ctx.malloc(n_inst * nb * 64) for _ in range(2)]
            for b in pals:
                b.upload(np.full(n_inst * nb * 16, np.nan, np.float32))
            p.set_palette_output_pair(base + 50, pals[0].ptr, pals[1].ptr)
            rec = {"sc": sc, "p": p, "nb": nb, "n_inst": n_inst, "pals": pals, "out": None, "mid": base + 60, "history": []}
            if nv:
                mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 41)
                ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
                rec["out"] = Outs(ctx, n_inst * nv)
                p.set_skin_output(base + 50, base + 60, rec["out"].pos.ptr, rec["out"].nrm.ptr, rec["out"].tan.ptr)
            chars.append(rec)
        ctx.set_option("debug.frame_skin", frame_skin)
        ctx.set_option("debug.overlap", overlap)
        try:
            for f in range(n_frames):
                for ch in chars:
                    for idx, par in ch["sc"].script.get(f, []):
                        ch["p"].set_parameter(idx, par)
                A.scene_update(ctx, [ch["p"] for ch in chars], chars[0]["sc"].dt)
                for ch in chars:
                    cur = ch["p"].current_palette(ch["p"].base_id + 50)
                    assert cur == ch["pals"][(f + 1) & 1 if overlap else 0].ptr, f"frame {f}: current palette of {ch['sc'].name}"
                if not overlap and f >= n_frames - 2:      # the one-stream run: the palettes of the last two frames
                    for ch in chars:
                        ch["history"].append(ch["pals"][0].download(np.uint32, ch["n_inst"] * ch["nb"] * 16))
            ctx.sync()
            if overlap:     # frame f ran on stream (f + 1) & 1 (the first frame of the mode starts on the second stream)
                for ch in chars:
                    last = (n_frames - 1 + 1) & 1
                    ch["history"] = [ch["pals"][last ^ 1].download(np.uint32, ch["n_inst"] * ch["nb"] * 16),
                                     ch["pals"][last].download(np.uint32, ch["n_inst"] * ch["nb"] * 16)]
            runs.append([(ch["history"], ch["out"].get() if ch["out"] else None) for ch in chars])
        finally:
            ctx.set_option("anim.overlap", 0)
            ctx.set_option("anim.frame_skin", 1)
            for ch in chars:
                ch["p"].free()
                for b in ch["pals"]:
                    b.free()
                if ch["out"]:
                    ch["out"].free()
                    ctx.mesh_free(ch["mid"])
    for k, ((h0, v0), (h1, v1)) in enumerate(zip(*runs)):
        name = specs[k][0].name
        assert np.array_equal(h0[0], h1[0]), f"{name}: palette of frame N - 2"
        assert np.array_equal(h0[1], h1[1]), f"{name}: palette of frame N - 1"
        if v0 is not None:
            for s, (x, y) in enumerate(zip(v0, v1)):
                assert np.array_equal(x, y), f"{name}: vertex stream {s} after the pipelined frames"
    # the last frame's palette of the first character against the oracle
    sc = specs[0][0]
    o = cases.build_oracle(orc, sc)
    for f in range(n_frames):
        for idx, par in sc.script.get(f, []):
            o.set_parameter(idx, par)
        _oupdate(o, sc)
    ref = o.palette(list(range(sc.rig.n_nodes)))
    assert np.array_equal(runs[1][0][0][1].reshape(-1, 16)[:sc.rig.n_nodes], ref.view(np.uint32).reshape(-1, 16))
    o.close()


@pytest.mark.parametrize("overlap", [0, 1, 2], ids=["one_stream", "streams_by_frame", "streams_by_kind"])
def test_steady_scene_frames_keep_their_plans_and_follow_every_change(ctx, orc, overlap):
    """A scene in which nothing changes but the clocks keeps its launch plans, its job array and the programs in its control block
    (SceneBatch::static_gen, csrc/anim_model.h): such frames write clocks and tick flags only.  Every frame here is checked against
    oracles (palettes) and against fyx_lbs_skin_device on the palette the frame wrote (vertices), through changes of every kind the
    cached state is made from: a parameter that fires a transition (programs replanned), an animation's speed (API call), an option,
    a palette output moved, a mesh uploaded again, the member list reordered and shortened.  debug.host_times counts the steady frames:
    most frames are, none of the frames right after a change is."""
    specs = [(cases.c5_blend_tree(euler_every=10 ** 6), 1, 5000), (cases.transitions(), 2, 1200), (cases.by_index(), 1, 0),
             (cases.c5_blend_tree(seed=synth.SEED_BASE + 91, euler_every=10 ** 6), 1, 7001), (cases.masked_transitions(), 1, 900)]
    chars = []
    for j, (sc, n_inst, nv) in enumerate(specs):
        o = cases.build_oracle(orc, sc)
        p = cases.build_product(ctx, sc, n_inst)
        nb, base = sc.rig.n_nodes, p.base_id
        A.create_bone_list(ctx, base + 50, base, list(range(nb)))
        pals = [ctx.malloc(n_inst * nb * 64) for _ in range(3)]
        p.set_palette_output_pair(base + 50, pals[0].ptr, pals[1].ptr)
        rec = {"sc": sc, "o": o, "p": p, "nb": nb, "n_inst": n_inst, "pals": pals, "nv": nv, "mid": base + 60, "stepped": 0, "alt": pals[1]}
        if nv:
            rec["mesh"] = synth.make_mesh(nv, nb, synth.SEED_BASE + 50 + j)
            m = rec["mesh"]
            ctx.mesh_upload_soa(base + 60, m.pos, m.weights, m.indices, m.normal, m.tangent)
            rec["out"], rec["ref"] = Outs(ctx, n_inst * nv), Outs(ctx, n_inst * nv)
            p.set_skin_output(base + 50, base + 60, rec["out"].pos.ptr, rec["out"].nrm.ptr, rec["out"].tan.ptr)
        chars.append(rec)
    order = list(range(len(chars)))
    changes = {}           # frame -> what is changed before it
    changes[22] = "speed"
    changes[30] = "option"
    changes[37] = "palette"
    changes[44] = "mesh"
    changes[51] = "members"
    changes[58] = "members_back"
    ctx.set_option("debug.overlap", overlap)
    ctx.set_option("debug.host_times", 1)
    ctx.host_times()
    steady = []
    try:
        for f in range(66):
            what = changes.get(f)
            if what == "speed":
                chars[0]["p"].set_speed(1, 0.5)
                orc._alib().fo_animation_set_speed(chars[0]["o"].anims[1], 0.5)
            elif what == "option":
                ctx.set_option("anim.update_lean", 0)
            elif what == "palette":
                ch = chars[3]
                ch["p"].set_palette_output_pair(ch["p"].base_id + 50, ch["pals"][0].ptr, ch["pals"][2].ptr)
                ch["alt"] = ch["pals"][2]
            elif what == "mesh":
                ch = chars[1]
                ch["mesh"] = synth.make_mesh(ch["nv"], ch["nb"], synth.SEED_BASE + 99)
                m = ch["mesh"]
                ctx.mesh_upload_soa(ch["mid"], m.pos, m.weights, m.indices, m.normal, m.tangent)
            elif what == "members":
                order = [4, 0, 3]
            elif what == "members_back":
                order = [0, 1, 2, 3, 4]
            for k in order:
                ch = chars[k]
                for idx, par in ch["sc"].script.get(ch["stepped"], []):
                    ch["o"].set_parameter(idx, par)
                    ch["p"].set_parameter(idx, par)
                _oupdate(ch["o"], ch["sc"])
                ch["stepped"] += 1
            A.scene_update(ctx, [chars[k]["p"] for k in order], chars[0]["sc"].dt)
            ht = ctx.host_times()
            assert ht[6] == 1.0
            steady.append(ht[7] == 1.0)
            for k in order:
                ch = chars[k]
                cur = ch["p"].current_palette(ch["p"].base_id + 50)
                buf = ch["pals"][0] if cur == ch["pals"][0].ptr else ch["alt"]
                assert cur == buf.ptr
                pal = buf.download(np.float32, ch["n_inst"] * ch["nb"] * 16).reshape(ch["n_inst"], ch["nb"], 16)
                ref = ch["o"].palette(list(range(ch["nb"])))
                assert np.array_equal(pal[0].view(np.uint32), ref.view(np.uint32)), f"frame {f} ({what}): palette of {ch['sc'].name}"
                if ch["nv"]:
                    ctx.lbs_skin_device(ch["mid"], cur, ch["nb"], ch["n_inst"], ch["ref"].pos.ptr, ch["ref"].nrm.ptr, ch["ref"].tan.ptr)
                    for s, (x, y) in enumerate(zip(ch["out"].get(), ch["ref"].get())):
                        assert np.array_equal(x, y), f"frame {f} ({what}): {ch['sc'].name} vertex stream {s}"
        for f in changes:
            assert not steady[f], f"frame {f} follows a change ({changes[f]}) and was treated as a steady frame"
        assert sum(steady) >= 20, steady
    finally:
        ctx.set_option("debug.host_times", 0)
        ctx.set_option("anim.overlap", 0)
        ctx.set_option("anim.update_lean", 1)
        for ch in chars:
            ch["o"].close()
            ch["p"].free()
            for b in ch["pals"]:
                b.free()
            if ch["nv"]:
                ch["out"].free()
                ch["ref"].free()
                ctx.mesh_free(ch["mid"])


@pytest.mark.parametrize("mode", [1, 2], ids=["streams_by_frame", "streams_by_kind"])
def test_current_palette_is_the_animators_own_last_frame_whatever_other_animators_did_since(ctx, orc, mode):
    """fyx_animator_current_palette answered from the CONTEXT's frame index, which every pose entry toggles.  Two animators
    updated one after the other each frame (A then B) under anim.overlap: asked AFTER B's update, A's current palette must still be the
    buffer A's update wrote -- and a caller that skins A from it gets this frame's vertices, every frame, against the oracle."""
    sa, sb = cases.c5_blend_tree(euler_every=10 ** 6), cases.player_only(euler_every=10 ** 6)
    chars = []
    for sc, nv in ((sa, 5000), (sb, 3000)):
        p = cases.build_product(ctx, sc, 1)
        nb, base = sc.rig.n_nodes, p.base_id
        A.create_bone_list(ctx, base + 50, base, list(range(nb)))
        pals = [ctx.malloc(nb * 64) for _ in range(2)]
        for b in pals:
            b.upload(np.full(nb * 16, np.nan, np.float32))
        p.set_palette_output_pair(base + 50, pals[0].ptr, pals[1].ptr)
        mesh = synth.make_mesh(nv, nb, synth.SEED_BASE + 43)
        ctx.mesh_upload_soa(base + 60, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        chars.append({"sc": sc, "p": p, "nb": nb, "pals": pals, "mesh": mesh, "mid": base + 60, "out": Outs(ctx, nv), "o": cases.build_oracle(orc, sc), "wrote": []})
    ctx.set_option("debug.overlap", mode)
    try:
        for f in range(9):
            for ch in chars:
                _update(ch["p"], ch["sc"])
                ch["wrote"].append(ch["p"].current_palette(ch["p"].base_id + 50))      # right behind its own update
                _oupdate(ch["o"], ch["sc"])
            for ch in chars:      # ... and after the OTHER animator's update: the same buffer, holding this frame's palette
                cur = ch["p"].current_palette(ch["p"].base_id + 50)
                assert cur == ch["wrote"][-1], f"frame {f}: {ch['sc'].name}'s current palette moved when another animator was updated"
                assert cur in (ch["pals"][0].ptr, ch["pals"][1].ptr)
                ctx.lbs_skin_device(ch["mid"], cur, ch["nb"], 1, ch["out"].pos.ptr, ch["out"].nrm.ptr, ch["out"].tan.ptr)
            ctx.sync()
            for ch in chars:
                m, ref_pal = ch["mesh"], ch["o"].palette(list(range(ch["nb"])))
                which = 0 if ch["wrote"][-1] == ch["pals"][0].ptr else 1
                assert np.array_equal(ch["pals"][which].download(np.uint32, ch["nb"] * 16), ref_pal.view(np.uint32).reshape(-1)), f"frame {f}: palette of {ch['sc'].name}"
                ref = orc.lbs_skin(m.pos, m.weights, m.indices, ref_pal, m.normal, m.tangent)
                got = ch["out"].get()
                for s, k in enumerate(("pos", "normal", "tangent")):
                    assert np.array_equal(got[s], np.ascontiguousarray(ref[k]).view(np.uint32).reshape(-1)), f"frame {f}: {ch['sc'].name} {k}"
        # the two animators' frames alternate streams, so each animator's own frames all ran on ONE stream: one buffer of its pair
        assert len(set(chars[0]["wrote"])) == 1 and len(set(chars[1]["wrote"])) == 1
    finally:
        ctx.set_option("anim.overlap", 0)
        for ch in chars:
            ch["p"].free()
            ch["o"].close()
            for b in ch["pals"]:
                b.free()
            ch["out"].free()
            ctx.mesh_free(ch["mid"])
