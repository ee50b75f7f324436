"""Parity invariants that can be read off the compiled gfx950 code without a GPU (tools/isa_stats.py).

The reference computes in unfused IEEE f32; the kernels that promise bit-exact results are compiled with
-ffp-contract=off and written without fmaf.  A lost compiler flag (the Makefile, __graft_entry__.build) or a compiler
that starts contracting would show up as wrong LAST BITS on the GPU only -- this test shows it in the instruction
stream, wherever the library was built."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fyrox_amd", "libfyrox_hip.so")


@pytest.fixture(scope="module")
def stats():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_stats
    if not os.path.exists(isa_stats.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain is not here")
    s = isa_stats.kernel_stats(LIB)
    assert len(s) > 100, "the library's kernels were not found in the offload bundles"
    return s


def _targs(name):
    m = re.search(r"<(.*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else []


# position of the EXACT template argument in each skinning kernel
EXACT_ARG = {"fyx::lbs_skin": 0, "fyx::lbs_skin_dyn": 0, "fyx::lbs_skin_crowd": 1, "fyx::lbs_skin_batch": 0, "fyx::lbs_skin_batch_dyn": 0, "fyx::lbs_skin_ex": 0,
             "fyx::lbs_skin_aos": 0, "fyx::lbs_skin_aos_batch": 0}


def _no_contraction(name, c, sqrt_fma=2, slack_fma=0, slack_fmac=2):
    """every fused operation of the kernel is accounted for by the compiler's IEEE division / square-root expansions"""
    div, sq = c["v_div_fixup_f32"], c["v_sqrt_f32"]
    assert c["v_pk_fma_f32"] == 0 and c["v_mad_f32"] == 0 and c["v_mac_f32"] == 0 and c["v_fma_mix_f32"] == 0, (name, dict(c))
    assert c["v_fma_f32"] <= 3 * div + sqrt_fma * sq + slack_fma, (name, dict(c))
    assert c["v_fmac_f32"] <= 2 * div + slack_fmac, (name, dict(c))      # + integer-division expansions of the index math


def test_exact_skinning_kernels_contain_no_contracted_multiply_add(stats):
    seen = {"exact": 0, "fused": 0}
    for name, c in stats.items():
        family = name.split("<")[0]
        if family not in EXACT_ARG:
            continue
        exact = _targs(name)[EXACT_ARG[family]] == "true"
        if exact:
            _no_contraction(name, c)
            seen["exact"] += 1
        else:
            seen["fused"] += 1
            assert c["v_pk_fma_f32"] > 0, (name, "the fused variant is expected to use packed FMA")
    assert seen["exact"] >= 57 and seen["fused"] >= 47, seen


def test_pose_and_palette_kernels_contain_no_contracted_multiply_add(stats):
    """Always exact: sampling (sinf / cosf of Euler tracks bring their own polynomial FMAs: <= 1e-5 by contract, see
    DESIGN 2), the fold interpreter (nlerp: one sqrt and four divisions per blend), hierarchy, palettes, AABBs."""
    exact_everywhere = ("fyx::pose_update_kernel", "fyx::pose_update_scene_kernel", "fyx::pose_update_inl_kernel", "fyx::pose_update_pack_kernel", "fyx::property_update_kernel",
                        "fyx::property_update_scene_kernel", "fyx::root_motion_fold_kernel", "fyx::root_motion_fold_scene_kernel",
                        "fyx::palette_kernel", "fyx::palette_gather_kernel", "fyx::skinned_aabb_kernel", "fyx::skinned_aabb_inst_kernel",
                        "fyx::points_aabb_kernel", "fyx::aabb_final_kernel", "fyx::aabb_final_inst_kernel", "fyx::blend_shape_weights_kernel")
    found = 0
    for name, c in stats.items():
        if name.split("<")[0] in exact_everywhere:
            _no_contraction(name, c)
            found += 1
    assert found >= len(exact_everywhere)
    for name, c in stats.items():          # samplers: the only fused operations besides the expansions are sincosf's
        if "sample" in name or name.startswith("fyx::root_motion_kernel") or name.startswith("fyx::root_motion_scene_kernel"):
            assert c["v_pk_fma_f32"] == 0 and c["v_mad_f32"] == 0 and c["v_mac_f32"] == 0, (name, dict(c))
            assert c["v_fma_f32"] - 3 * c["v_div_fixup_f32"] - 2 * c["v_sqrt_f32"] <= 12, (name, dict(c))


def test_kernels_do_not_spill(stats):
    """No kernel keeps registers in scratch memory.  (Until round 4 the scene form of the update kernel did: it indexed a register
    copy of its job's palette outputs with a loop counter; it now reads them where they lie.)"""
    for name, c in stats.items():
        assert c["scratch"] == 0, (name, "register spills")


def test_register_budgets_behind_the_measured_occupancies():
    """What DESIGN's occupancy statements rest on (512 VGPRs per SIMD lane: waves per SIMD = 512 // allocated, allocation
    in steps of 8): four waves per SIMD for the streaming kernels (<= 128), six for the lean crowd form (<= 80) -- the
    launchers size their resident grids and LDS for these."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_stats
    if not os.path.exists(isa_stats.READELF):
        pytest.skip("llvm-readelf of the ROCm toolchain is not here")
    res = isa_stats.kernel_resources(LIB)
    budget = {"fyx::lbs_skin_dyn<true, 7>": 128, "fyx::lbs_skin<true, 7>": 128,
              "fyx::lbs_skin_batch<true, 7>": 128, "fyx::lbs_skin_batch_dyn<true, 7>": 128, "fyx::lbs_skin_batch_dyn<false, 7>": 128,
              "fyx::lbs_skin_crowd<512, true, 7, false>": 128,
              "fyx::lbs_skin_crowd<512, true, 7, true>": 80,
              # the update kernel without the interpreter (every program of the frame straight): three waves per SIMD, and one of
              # them fits into what ONE retiring workgroup of the crowd kernel frees on a SIMD (2 x 128) -- anim.overlap
              "fyx::pose_update_kernel<2>": 176, "fyx::pose_update_scene_kernel<2, false>": 176, "fyx::pose_update_scene_kernel<2, true>": 176, "fyx::pose_update_inl_kernel<2>": 176,
              "fyx::pose_update_pack_kernel<2, 4>": 176, "fyx::pose_frame_inl_kernel<2>": 176,
              # the one-launch frame that skins: two 256-thread workgroups per CU without the interpreter (kFrameSkinMaxBlocks = 448 rests on
              # it), one with it (kFrameSkinMaxBlocksGeneral = 192: whatever the compiler allocates within the SIMD's 512)
              "fyx::pose_frame_skin_kernel<2, true>": 256, "fyx::pose_frame_skin_kernel<2, false>": 256, "fyx::pose_sample_kernel": 64, "fyx::pose_sample_crowd_kernel<256u>": 64, "fyx::pose_sample_crowd_kernel<64u>": 64}
    for name, limit in budget.items():
        assert name in res, name
        assert res[name]["vgpr"] + res[name]["agpr"] <= limit, (name, res[name])
        assert res[name]["scratch_bytes"] == 0, (name, res[name])
    for name, r in res.items():        # every skinning kernel fits four waves per SIMD and uses no scratch ...
        if name.startswith("fyx::lbs_skin"):
            # ... except the vertex-buffer-out kernels that hold more per lane: 32- and 40-byte spans per lane (PL 8 / 10), and -- round 5 --
            # the plain (no blend shapes) narrow layouts, which keep TWO spans in flight per wave: three waves per SIMD (<= 168).  With
            # blend shapes the narrow layouts keep one span and four waves (<= 128)
            aos = name.startswith("fyx::lbs_skin_aos")
            wide = aos and (_targs(name)[2] in ("8u", "10u") or _targs(name)[1] == "false")
            assert r["vgpr"] + r["agpr"] <= (168 if wide else 128) and r["scratch_bytes"] == 0, (name, r)
