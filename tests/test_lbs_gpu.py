"""GPU parity: the HIP skinning path (through the C ABI) vs the CPU oracle on identical inputs.

Bar: BIT-EXACT in the default `lbs.exact=1` mode (the kernel keeps the reference's unfused
operation order); within 1e-5 relative (north_star tolerance) in the fused `lbs.exact=0` mode.
Configs follow BASELINE.json: C1 = 1k verts/4 bones, C2 = 50k/64, C3 = crowd, C4 = 1M/256.
"""
import numpy as np
import pytest

import fyrox_amd
from fyrox_amd import _native, synth

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north_star: "within 1e-5 relative f32"


def rel_err(got, ref):
    scale = max(float(np.abs(ref).max()), 1e-3)
    return float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()) / scale


def upload(ctx, mesh_id, m, aos=False):
    if aos:
        L = synth.ANIMATED_VERTEX
        ctx.mesh_upload(mesh_id, m.to_animated_vertex_aos(), m.n_verts, L["stride"], off_pos=L["off_pos"],
                        off_normal=L["off_normal"], off_tangent=L["off_tangent"], off_weights=L["off_weights"],
                        off_indices=L["off_indices"])
    else:
        ctx.mesh_upload_soa(mesh_id, m.pos, m.weights, m.indices, m.normal, m.tangent)


def oracle_skin(orc, m, pal, n_inst=1):
    nb = pal.shape[0] // n_inst
    outs = [orc.lbs_skin(m.pos, m.weights, m.indices, pal[i * nb:(i + 1) * nb], m.normal, m.tangent, threads=0)
            for i in range(n_inst)]
    return {k: np.concatenate([o[k] for o in outs]) for k in outs[0]}


def assert_bit_exact(got, ref):
    for k in ("pos", "normal", "tangent"):
        assert got[k].shape == ref[k].shape
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)) or np.array_equal(got[k], ref[k]), \
            f"{k}: max rel err {rel_err(got[k], ref[k]):.3e}"


def _set_defaults(ctx):
    for k, v in (("lbs.blocks_per_cu", 4), ("lbs.exact", 1), ("lbs.streams", 2), ("lbs.crowd", -1), ("lbs.crowd_ipb", 0),
                 ("lbs.crowd_lean", 0), ("lbs.dyn", 1), ("anim.inline_ctrl", 1), ("comm.form", 0)):
        ctx.set_option(k, v)


@pytest.fixture(autouse=True)
def _defaults(ctx):
    _set_defaults(ctx)
    yield
    _set_defaults(ctx)      # the context is shared by the whole session: leave it as it was found


# ---- BASELINE configs ---------------------------------------------------------------------

@pytest.mark.parametrize("aos", [False, True])
def test_c1_1k_verts_4_bones(ctx, orc, aos):
    m = synth.make_mesh(1000, 4, synth.SEED_BASE + 1)
    pal = synth.make_palette(4, synth.SEED_BASE + 1)
    upload(ctx, 1, m, aos)
    assert_bit_exact(ctx.lbs_skin(1, pal), oracle_skin(orc, m, pal))


@pytest.mark.parametrize("coherent", [True, False])
def test_c2_50k_verts_64_bones(ctx, orc, coherent):
    m = synth.make_mesh(50_000, 64, synth.SEED_BASE + 2, coherent)
    pal = synth.make_palette(64, synth.SEED_BASE + 2)
    upload(ctx, 2, m, aos=True)
    assert_bit_exact(ctx.lbs_skin(2, pal), oracle_skin(orc, m, pal))


def test_c3_crowd_instances_share_one_mesh(ctx, orc):
    # scaled crowd: 24 instances x 10k verts / 64 bones (full C3 = 1000 instances, see bench)
    n_inst = 24
    m = synth.make_mesh(10_000, 64, synth.SEED_BASE + 3)
    pal = synth.make_palette(64, synth.SEED_BASE + 3, n_instances=n_inst)
    upload(ctx, 3, m)
    assert_bit_exact(ctx.lbs_skin(3, pal, n_instances=n_inst), oracle_skin(orc, m, pal, n_inst))


def _download_at(ctx, buf, byte_offset, dtype, count):
    out = np.empty(count, dtype=dtype)
    ctx._check(ctx._l.fyx_memcpy_d2h(ctx._h, out.ctypes.data, buf.ptr + byte_offset, out.nbytes))
    return out


@pytest.mark.parametrize("exact", [1, 0])
def test_c3_full_size_crowd_1000_instances(ctx, orc, exact):
    """BASELINE C3 at its full size: 1000 instances x 10k verts / 64 bones in one instanced launch.  The crowd kernel's
    automatic instances-per-workgroup run differs with the crowd's size (16 here, 1-2 for a few dozen instances), so
    the shipped configuration is the one under test: instances at the start, across the first run boundary (15 | 16,
    17) and at the very end against the oracle, bit for bit (fused mode: 1e-5)."""
    n_inst, nv, nb = 1000, 10_000, 64
    m = synth.make_mesh(nv, nb, synth.SEED_BASE + 3)
    pal = synth.make_palette(nb, synth.SEED_BASE + 3, n_instances=n_inst)
    upload(ctx, 3, m)
    ctx.set_option("lbs.exact", exact)
    d_pal = ctx.to_device(pal)
    d_p, d_n, d_t = ctx.malloc(n_inst * nv * 12), ctx.malloc(n_inst * nv * 12), ctx.malloc(n_inst * nv * 16)
    ctx.lbs_skin_device(3, d_pal.ptr, nb, n_inst, d_p.ptr, d_n.ptr, d_t.ptr)
    ctx.sync()
    for i in (0, 1, 15, 16, 17, 500, 999):
        ref = orc.lbs_skin(m.pos, m.weights, m.indices, pal[i * nb:(i + 1) * nb], m.normal, m.tangent, threads=0)
        got = {"pos": _download_at(ctx, d_p, i * nv * 12, np.float32, nv * 3).reshape(-1, 3),
               "normal": _download_at(ctx, d_n, i * nv * 12, np.float32, nv * 3).reshape(-1, 3),
               "tangent": _download_at(ctx, d_t, i * nv * 16, np.float32, nv * 4).reshape(-1, 4)}
        if exact:
            assert_bit_exact(got, ref)
        else:
            for k in ref:
                assert rel_err(got[k], ref[k]) <= REL_TOL, (i, k)
    for d in (d_pal, d_p, d_n, d_t):
        d.free()
    ctx.mesh_free(3)


def test_c4_1m_verts_256_bones(ctx, orc):
    m = synth.make_mesh(1_000_000, 256, synth.SEED_BASE + 4)
    pal = synth.make_palette(256, synth.SEED_BASE + 4)
    upload(ctx, 4, m)
    got = ctx.lbs_skin(4, pal, aabb=True)
    ref = oracle_skin(orc, m, pal)
    assert_bit_exact(got, ref)
    assert np.array_equal(got["aabb"][:3], ref["pos"].min(axis=0))
    assert np.array_equal(got["aabb"][3:], ref["pos"].max(axis=0))
    # size-independent properties at full size
    ident = np.tile(np.eye(4, dtype=np.float32).reshape(1, 16), (256, 1))
    onehot = np.zeros_like(m.weights); onehot[:, 0] = 1
    ctx.mesh_upload_soa(40, m.pos, onehot, m.indices, m.normal, m.tangent)
    same = ctx.lbs_skin(40, ident)
    assert np.array_equal(same["pos"], m.pos) and np.array_equal(same["normal"], m.normal)
    assert np.array_equal(same["tangent"], m.tangent)
    ctx.mesh_free(40)
    ctx.mesh_free(4)


# ---- kernel variants ----------------------------------------------------------------------

@pytest.mark.parametrize("bpcu", [1, 2, 4, 16])
@pytest.mark.parametrize("n_verts,n_bones", [(70_001, 200), (63, 3), (4096, 256), (300_001, 64)])
def test_streaming_kernel_grid_sizes_are_bit_exact(ctx, orc, bpcu, n_verts, n_bones):
    """lbs_skin (lbs.dyn = 0 keeps it for large meshes too) under every grid size: who skins a unit changes, never the bytes.
    (The kernel-variant matrix of rounds 1 - 2 -- workgroup sizes, prefetch depths, cache policy, work splits -- left the product
    with its variants, tools/exp/README.md.)"""
    m = synth.make_mesh(n_verts, n_bones, 99, coherent=False)   # ragged: not a multiple of 4 or 256
    pal = synth.make_palette(n_bones, 99)
    upload(ctx, 5, m)
    ctx.set_option("lbs.dyn", 0); ctx.set_option("lbs.blocks_per_cu", bpcu)
    assert_bit_exact(ctx.lbs_skin(5, pal), oracle_skin(orc, m, pal))


# ---- lbs_skin_dyn: the large single-instance launch (units drawn from an LDS ticket counter, buffer-resource streams) ----

def _skin_device_masked(ctx, mesh_id, m, pal, want, guard=64):
    """fyx_lbs_skin_device into guarded device buffers; returns the outputs and whether the guard bytes survived."""
    nb = pal.shape[0]
    d_pal = ctx.to_device(pal)
    spec = {"pos": 3, "normal": 3, "tangent": 4}
    bufs = {}
    for k in want:
        n = m.n_verts * spec[k]
        b = ctx.malloc((n + guard) * 4)
        b.upload(np.full(n + guard, 0x7FC0DEAD, np.uint32))
        bufs[k] = b
    ctx.lbs_skin_device(mesh_id, d_pal.ptr, nb, 1, bufs["pos"].ptr if "pos" in bufs else 0,
                        bufs["normal"].ptr if "normal" in bufs else 0, bufs["tangent"].ptr if "tangent" in bufs else 0)
    ctx.sync()
    out, guards_ok = {}, True
    for k, b in bufs.items():
        n = m.n_verts * spec[k]
        raw = b.download(np.uint32, n + guard)
        guards_ok &= bool((raw[n:] == 0x7FC0DEAD).all())
        out[k] = raw[:n].view(np.float32).reshape(-1, spec[k])
        b.free()
    d_pal.free()
    return out, guards_ok


@pytest.mark.parametrize("n_bones", [200, 256, 5])
@pytest.mark.parametrize("n_verts", [1_048_576, 1_000_003, 530_001])
def test_drawn_kernel_ragged_sizes_bit_exact(ctx, orc, n_bones, n_verts):
    """The launch qualifies for lbs_skin_dyn from 2 units per wave and workgroup on; the last unit is ragged (the
    buffer resources drop what lies past the end: the guard words behind every output must survive)."""
    m = synth.make_mesh(n_verts, n_bones, 1234 + n_bones, coherent=False)
    pal = synth.make_palette(n_bones, 1234)
    upload(ctx, 6, m)
    ref = oracle_skin(orc, m, pal)
    got, guards_ok = _skin_device_masked(ctx, 6, m, pal, ("pos", "normal", "tangent"))
    assert guards_ok
    assert_bit_exact(got, ref)
    ctx.set_option("lbs.dyn", 0)                    # and the static kernel agrees
    got0, _ = _skin_device_masked(ctx, 6, m, pal, ("pos", "normal", "tangent"))
    assert_bit_exact(got0, ref)
    ctx.mesh_free(6)


@pytest.mark.parametrize("want", [("pos",), ("normal",), ("pos", "normal"), ("tangent",), ("pos", "tangent"),
                                  ("normal", "tangent"), ("pos", "normal", "tangent")])
@pytest.mark.parametrize("exact", [1, 0])
def test_drawn_kernel_every_output_set(ctx, orc, want, exact):
    m = synth.make_mesh(600_037, 64, 4321)
    pal = synth.make_palette(64, 4321)
    upload(ctx, 6, m)
    ctx.set_option("lbs.exact", exact)
    ref = oracle_skin(orc, m, pal)
    got, guards_ok = _skin_device_masked(ctx, 6, m, pal, want)
    assert guards_ok
    for k in want:
        if exact:
            assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k
        else:
            assert rel_err(got[k][:, :3], ref[k][:, :3]) <= REL_TOL, k      # tolerance: north_star's 1e-5 relative
    ctx.mesh_free(6)


def test_drawn_kernel_projective_palette_and_nan_propagation(ctx, orc):
    m = synth.make_mesh(700_001, 16, 777)
    pal = synth.make_palette(16, 777).copy()
    pal[3, 3] = 0.125; pal[3, 7] = -0.25; pal[3, 15] = 1.5      # bone 3 is projective: the divide path
    upload(ctx, 6, m)
    got, _ = _skin_device_masked(ctx, 6, m, pal, ("pos", "normal", "tangent"))
    assert_bit_exact(got, oracle_skin(orc, m, pal))
    pal2 = synth.make_palette(16, 777).copy()
    pal2[5, 12] = np.inf                                         # inf * 0 = NaN must come through
    got, _ = _skin_device_masked(ctx, 6, m, pal2, ("pos", "normal", "tangent"))
    ref = oracle_skin(orc, m, pal2)
    assert np.isnan(ref["pos"]).any()
    assert np.array_equal(np.isnan(got["pos"]), np.isnan(ref["pos"]))
    ok = ~np.isnan(ref["pos"])
    assert np.array_equal(got["pos"][ok], ref["pos"][ok]) and np.array_equal(got["normal"], ref["normal"])
    ctx.mesh_free(6)


@pytest.mark.parametrize("n_verts", [5000, 300_011], ids=["lbs_skin", "lbs_skin_dyn"])
def test_subnormal_inputs_products_and_results_bit_exact(ctx, orc, n_verts):
    """IEEE f32 all the way down: the reference's arithmetic (Rust f32 on the CPU) keeps subnormal numbers, so must the kernels -- packed
    multiplies and adds, the homogeneous divide (a projective bone), the accumulation.  Positions, normals and weights scaled so that
    inputs, products and results fall below 2^-126 (1.18e-38) for a third of the vertices each; every bit against the oracle."""
    m = synth.make_mesh(n_verts, 16, 909)
    rng = np.random.default_rng(909)
    pos, nrm, tan, w = m.pos.copy(), m.normal.copy(), m.tangent.copy(), m.weights.copy()
    third = rng.integers(0, 3, n_verts)
    pos[third == 0] *= np.float32(1e-39)                           # subnormal inputs
    nrm[third == 0] *= np.float32(3e-42)
    pos[third == 1] *= np.float32(1e-30)                           # normal inputs, subnormal products with the small weights below
    w[third == 1] *= np.float32(1e-9)
    tan[third == 2, :3] *= np.float32(1e-37)                       # results that cancel into the subnormal range
    pal = synth.make_palette(16, 909).copy()
    pal[:, 12:15] *= np.float32(1e-40)                             # translations that do not drown the small positions
    pal[3, 3] = 0.125; pal[3, 7] = -0.25; pal[3, 15] = 1.5         # bone 3 is projective: subnormal numerators through the divide
    assert (np.abs(pos[pos != 0]) < 1.17e-38).any()
    ctx.mesh_upload_soa(6, pos, w, m.indices, nrm, tan)
    try:
        ref = orc.lbs_skin(pos, w, m.indices, pal, nrm, tan, threads=0)
        sub = lambda a: int(((np.abs(a) < 1.17e-38) & (a != 0)).sum())
        assert sub(ref["pos"]) > n_verts // 10 and sub(ref["normal"]) > n_verts // 10, "the case produces subnormal results"
        assert_bit_exact(ctx.lbs_skin(6, pal), ref)
        got_box = ctx.lbs_skin(6, pal, want=("pos",), aabb=True)["aabb"]              # the AABB reduction keeps them too
        ok = ~np.isnan(ref["pos"]).any(axis=1)
        assert np.array_equal(got_box, np.concatenate([ref["pos"][ok].min(0), ref["pos"][ok].max(0)]))
        if n_verts <= 5000:
            # the crowd kernel (40 instances of the mesh, each with its own small palette), exact mode
            pals = np.concatenate([pal] + [pal * np.float32(0.5 ** k) for k in range(1, 40)])
            refc = {k: np.concatenate([orc.lbs_skin(pos, w, m.indices, pals[i * 16:(i + 1) * 16], nrm, tan, threads=0)[k] for i in range(40)]) for k in ref}
            assert_bit_exact(ctx.lbs_skin(6, pals, n_instances=40), refc)
    finally:
        ctx.mesh_free(6)


@pytest.mark.parametrize("n_verts", [5000, 300_011], ids=["lbs_skin", "lbs_skin_dyn"])
def test_overflow_to_infinity_and_nan_like_the_cpu(ctx, orc, n_verts):
    """The other end of the range: positions near f32::MAX whose products and sums overflow to +-inf, and infinities of opposite sign that
    meet in the accumulation (NaN).  Same infinities in the same places, NaN where the CPU has NaN, every finite value bit for bit."""
    m = synth.make_mesh(n_verts, 16, 910)
    rng = np.random.default_rng(910)
    pos, nrm = m.pos.copy(), m.normal.copy()
    big = rng.random(n_verts) < 0.5
    pos[big] *= np.float32(3e38)
    nrm[big] *= np.float32(2e38)
    pal = synth.make_palette(16, 910).copy()
    pal[:, :12] *= np.float32(1.5)                                 # a scale that pushes |m x| over the top
    ctx.mesh_upload_soa(6, pos, m.weights, m.indices, nrm, m.tangent)
    try:
        ref = orc.lbs_skin(pos, m.weights, m.indices, pal, nrm, m.tangent, threads=0)
        assert np.isinf(ref["pos"]).sum() > n_verts // 20 and np.isnan(ref["pos"]).sum() > 10, "the case overflows and cancels"
        got = ctx.lbs_skin(6, pal)
        for k in ("pos", "normal", "tangent"):
            assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), k
            ok = ~np.isnan(ref[k])
            assert np.array_equal(got[k][ok].view(np.uint32), ref[k][ok].view(np.uint32)), k
    finally:
        ctx.mesh_free(6)


def test_drawn_kernel_repeated_overlapping_launches(ctx, orc):
    """Launches dealt over the worker streams overlap; every one must produce the same bytes."""
    m = synth.make_mesh(1_000_000, 256, synth.SEED_BASE + 4)
    pal = synth.make_palette(256, synth.SEED_BASE + 4)
    upload(ctx, 6, m)
    ref = oracle_skin(orc, m, pal)
    d_pal = ctx.to_device(pal)
    outs = [(ctx.malloc(m.n_verts * 12), ctx.malloc(m.n_verts * 12), ctx.malloc(m.n_verts * 16)) for _ in range(4)]
    for rep in range(3):
        for o in outs:
            ctx.lbs_skin_device(6, d_pal.ptr, 256, 1, o[0].ptr, o[1].ptr, o[2].ptr)
    ctx.sync()
    for o in outs:
        assert np.array_equal(o[0].download(np.float32, m.n_verts * 3).reshape(-1, 3), ref["pos"])
        assert np.array_equal(o[1].download(np.float32, m.n_verts * 3).reshape(-1, 3), ref["normal"])
        assert np.array_equal(o[2].download(np.float32, m.n_verts * 4).reshape(-1, 4), ref["tangent"])
    for o in outs:
        for b in o:
            b.free()
    d_pal.free()
    ctx.mesh_free(6)


@pytest.mark.parametrize("n_inst,n_verts", [(1, 1), (1, 15), (1, 17), (3, 1001), (2, 4096), (1, 1_000_003), (5, 70)])
def test_streaming_kernel_covers_every_vertex_once(ctx, orc, n_inst, n_verts):
    """lbs_skin with the crowd kernel off: instance segments, ragged last units, one vertex."""
    m = synth.make_mesh(n_verts, 16, 7)
    pal = synth.make_palette(16, 7, n_instances=n_inst)
    upload(ctx, 9, m)
    ctx.set_option("lbs.crowd", 0); ctx.set_option("lbs.dyn", 0)
    assert_bit_exact(ctx.lbs_skin(9, pal, n_instances=n_inst), oracle_skin(orc, m, pal, n_inst))


@pytest.mark.parametrize("n_inst", [24, 7, 4])
def test_crowd_fused_mode_blends_matrices_first_within_1e5(ctx, orc, n_inst):
    """lbs.exact=0 on the crowd kernel: M = sum_k w_k M_k, then one transform (tolerance 1e-5 relative, north_star);
    a projective palette still takes the per-bone path with the homogeneous divide."""
    m = synth.make_mesh(10_007, 64, 321, coherent=False)
    pal = synth.make_palette(64, 321, n_instances=n_inst)
    upload(ctx, 7, m)
    ctx.set_option("lbs.exact", 0); ctx.set_option("lbs.crowd", 1)
    got, ref = ctx.lbs_skin(7, pal, n_instances=n_inst), oracle_skin(orc, m, pal, n_inst)
    for k in ("pos", "normal", "tangent"):
        assert rel_err(got[k], ref[k]) <= REL_TOL, k
    assert np.array_equal(got["tangent"][:, 3], np.tile(m.tangent[:, 3], n_inst))
    proj = pal.copy()
    proj[5::64, 3] = 0.25; proj[5::64, 15] = 1.5      # bone 5 of every instance: a projective row
    got, ref = ctx.lbs_skin(7, proj, n_instances=n_inst), oracle_skin(orc, m, proj, n_inst)
    for k in ("pos", "normal", "tangent"):
        assert rel_err(got[k], ref[k]) <= REL_TOL, k


@pytest.mark.parametrize("dyn,n_verts", [(0, 100_000), (1, 600_000)])
def test_fused_mode_within_1e5(ctx, orc, dyn, n_verts):
    m = synth.make_mesh(n_verts, 64, 123)
    pal = synth.make_palette(64, 123)
    upload(ctx, 6, m)
    ctx.set_option("lbs.exact", 0); ctx.set_option("lbs.dyn", dyn)
    got, ref = ctx.lbs_skin(6, pal), oracle_skin(orc, m, pal)
    for k in ("pos", "normal", "tangent"):
        assert rel_err(got[k], ref[k]) <= REL_TOL, k
    assert np.array_equal(got["tangent"][:, 3], m.tangent[:, 3])


@pytest.mark.parametrize("bpcu", [8, 1, 64])
@pytest.mark.parametrize("n_inst,n_verts", [(3, 1000), (5, 1001), (2, 4096), (7, 13), (300, 65), (2000, 3)])
def test_instanced_variants(ctx, orc, bpcu, n_inst, n_verts):
    m = synth.make_mesh(n_verts, 32, 7)
    pal = synth.make_palette(32, 7, n_instances=n_inst)
    upload(ctx, 7, m)
    ctx.set_option("lbs.crowd", 0)      # the streaming kernel's per-instance segments
    ctx.set_option("lbs.blocks_per_cu", bpcu)
    assert_bit_exact(ctx.lbs_skin(7, pal, n_instances=n_inst), oracle_skin(orc, m, pal, n_inst))


@pytest.mark.parametrize("ipb", [0, 1, 3, 16, 4096])
@pytest.mark.parametrize("n_inst,n_verts", [(1, 1000), (3, 1000), (5, 1001), (2, 4096), (7, 13), (300, 65), (2000, 3),
                                            (33, 10_000)])
def test_crowd_kernel_variants(ctx, orc, ipb, n_inst, n_verts):
    # vertices held in registers, palettes double-buffered in LDS, one barrier per instance
    m = synth.make_mesh(n_verts, 32, 7)
    pal = synth.make_palette(32, 7, n_instances=n_inst)
    upload(ctx, 7, m)
    ctx.set_option("lbs.crowd", 1); ctx.set_option("lbs.crowd_ipb", ipb)
    assert_bit_exact(ctx.lbs_skin(7, pal, n_instances=n_inst), oracle_skin(orc, m, pal, n_inst))


def test_crowd_kernel_edge_palettes(ctx, orc):
    # 256 bones (2 x 16 KiB of LDS), one projective matrix in ONE instance of the run (the flag is
    # per palette buffer), positions-only and normal-only launches, fused arithmetic within 1e-5
    n_inst = 9
    m = synth.make_mesh(3001, 256, 19, coherent=False)
    pal = synth.make_palette(256, 19, n_instances=n_inst).copy()
    pal[4 * 256 + 7, 3] = 0.125; pal[4 * 256 + 7, 7] = -0.25; pal[4 * 256 + 7, 15] = 1.5
    upload(ctx, 19, m)
    ctx.set_option("lbs.crowd", 1)
    ref = oracle_skin(orc, m, pal, n_inst)
    for ipb in (0, 2, 5):
        ctx.set_option("lbs.crowd_ipb", ipb)
        assert_bit_exact(ctx.lbs_skin(19, pal, n_instances=n_inst), ref)
    got = ctx.lbs_skin(19, pal, n_instances=n_inst, want=("pos",))
    assert set(got) == {"pos"} and np.array_equal(got["pos"], ref["pos"])
    got = ctx.lbs_skin(19, pal, n_instances=n_inst, want=("normal",))
    assert np.array_equal(got["normal"], ref["normal"])
    ctx.set_option("lbs.exact", 0)
    got = ctx.lbs_skin(19, pal, n_instances=n_inst)
    for k in ("pos", "normal", "tangent"):
        assert rel_err(got[k], ref[k]) <= REL_TOL, k
    assert np.array_equal(got["tangent"][:, 3], np.tile(m.tangent[:, 3], n_inst))


# ---- edge cases ---------------------------------------------------------------------------

@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4099])
def test_ragged_sizes(ctx, orc, n):
    m = synth.make_mesh(n, 8, 5)
    pal = synth.make_palette(8, 5)
    upload(ctx, 8, m)
    got = ctx.lbs_skin(8, pal, aabb=True)
    if n == 0:
        assert got["pos"].shape == (0, 3)
        fmax = np.finfo(np.float32).max
        assert got["aabb"].tolist() == [fmax] * 3 + [-fmax] * 3
        return
    ref = oracle_skin(orc, m, pal)
    assert_bit_exact(got, ref)
    assert np.array_equal(got["aabb"], np.concatenate([ref["pos"].min(0), ref["pos"].max(0)]))


def test_256_bones_maximum_palette(ctx, orc):
    m = synth.make_mesh(10_000, 256, 17, coherent=False)
    assert m.indices.max() == 255
    pal = synth.make_palette(256, 17)
    upload(ctx, 9, m)
    assert ctx.mesh_info(9)["max_bone_index"] == 255
    assert_bit_exact(ctx.lbs_skin(9, pal), oracle_skin(orc, m, pal))


def test_projective_palette_takes_the_divide_path(ctx, orc):
    m = synth.make_mesh(5000, 16, 31)
    pal = synth.make_palette(16, 31).copy()
    pal[3, 3] = 0.125; pal[3, 7] = -0.25; pal[3, 15] = 1.5      # only bone 3 is projective
    upload(ctx, 10, m)
    assert_bit_exact(ctx.lbs_skin(10, pal), oracle_skin(orc, m, pal))


def test_zero_weight_influences_still_multiply_through(ctx, orc):
    # all four slots are always evaluated (mesh/mod.rs:514-519): inf * 0 = NaN must propagate
    m = synth.make_mesh(256, 4, 41)
    w = m.weights.copy(); w[:, 3] = 0
    idx = m.indices.copy(); idx[:, 3] = 3
    pal = synth.make_palette(4, 41).copy()
    pal[3, 12] = np.inf
    ctx.mesh_upload_soa(11, m.pos, w, idx, m.normal, m.tangent)
    got = ctx.lbs_skin(11, pal)
    ref = orc.lbs_skin(m.pos, w, idx, pal, m.normal, m.tangent)
    assert np.isnan(ref["pos"][:, 0]).all()
    assert np.array_equal(np.isnan(got["pos"]), np.isnan(ref["pos"]))
    assert np.array_equal(got["normal"], ref["normal"])


def test_positions_only_and_missing_attributes(ctx, orc):
    m = synth.make_mesh(3000, 16, 51)
    pal = synth.make_palette(16, 51)
    ctx.mesh_upload_soa(12, m.pos, m.weights, m.indices)        # no normal / tangent streams
    got = ctx.lbs_skin(12, pal, want=("pos",))
    assert np.array_equal(got["pos"], orc.lbs_skin(m.pos, m.weights, m.indices, pal)["pos"])
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.lbs_skin(12, pal, want=("pos", "normal"))
    assert e.value.status == "FYX_ERR_MISSING_ATTRIBUTE"
    upload(ctx, 12, m)
    got = ctx.lbs_skin(12, pal, want=("normal",))
    assert set(got) == {"normal"}
    assert np.array_equal(got["normal"], oracle_skin(orc, m, pal)["normal"])


def _nan_aware_equal(got, ref, what):
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok].view(np.uint32), ref[ok].view(np.uint32)), what


@pytest.mark.parametrize("entry,value", [(3, 0.125), (7, -0.25), (11, 0.5), (15, 2.0), (15, 0.0)], ids=["m30", "m31", "m32", "m33", "last_row_zero"])
def test_every_entry_of_the_last_row_decides_the_homogeneous_path(ctx, orc, entry, value):
    """Matrix4::transform_point divides by n = m30 x + m31 y + m32 z + m33 whenever n != 0 (nalgebra; scene/mesh/mod.rs:515-517): a matrix is
    affine for the kernels only if its last row is EXACTLY (0, 0, 0, 1), and each of the four entries alone must send it down the divide
    path -- in every kernel's own copy of that test (tools/mutants.py, fourth batch: four of them ignored one entry each and the suite, whose
    projective palettes always changed m30, m31 and m33 together, did not notice).  A last row of zeros makes n == 0: then nothing is divided
    (a kernel that divides anyway writes inf / NaN).  lbs_skin, lbs_skin_dyn, the crowd kernel, both batched forms, the vertex-buffer kernel
    and the two AABB kernels; every bit against the oracle."""
    nb = 12
    pal = synth.make_palette(nb, 5150).copy()
    pal[5, entry] = value                                       # (m33 = 0 on an otherwise affine matrix: the whole last row is zero, n == 0)
    # lbs_skin (one persistent launch of a mid-size mesh) and its AABB forms
    m = synth.make_mesh(6000, nb, 5151, coherent=False)
    ref = orc.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal, m.tangent, threads=0)
    upload(ctx, 21, m)
    L = synth.ANIMATED_VERTEX
    try:
        got = ctx.lbs_skin(21, pal)
        for k in ("pos", "normal", "tangent"):
            _nan_aware_equal(got[k], ref[k], f"lbs_skin {k}")
        # the two AABB kernels (stage_palette): with an entry that sends the bone's vertices FAR out (n close to 0 for some of them), so that
        # the box itself -- all the AABB calls return -- cannot come out right by treating the matrix as affine
        pal_far = pal.copy()
        if value != 0.0:
            pal_far[5, entry] = 0.01 if entry == 15 else (0.99 if value > 0 else -0.99)
        far = orc.lbs_skin(m.pos, m.weights, m.indices, pal_far, threads=0)["pos"]
        finite = ~np.isnan(far).any(axis=1)
        box = np.concatenate([far[finite].min(0), far[finite].max(0)])
        if value != 0.0:
            aff = orc.lbs_skin(m.pos, m.weights, m.indices, synth.make_palette(nb, 5150), threads=0)["pos"]
            assert not np.array_equal(box, np.concatenate([aff.min(0), aff.max(0)])), "the case moves the box"
        assert np.array_equal(ctx.skinned_aabb(21, pal_far), box), "skinned_aabb_kernel (stage_palette)"
        d_pal, d_box = ctx.to_device(np.concatenate([pal_far, synth.make_palette(nb, 5152)])), ctx.malloc(48)
        ctx.skinned_aabb_device(21, d_pal.ptr, nb, 2, d_box.ptr)
        ctx.sync()
        assert np.array_equal(d_box.download(np.float32, 12)[:6], box), "skinned_aabb_inst_kernel"
        d_pal.free(); d_box.free()
        # the crowd kernel: the projective palette in the MIDDLE of a run of affine ones (the double buffer's flags)
        pals = np.concatenate([synth.make_palette(nb, 5160 + i) if i != 3 else pal for i in range(6)])
        refc = {k: np.concatenate([orc.lbs_skin(m.pos, m.weights, m.indices, pals[i * nb:(i + 1) * nb], m.normal, m.tangent, threads=0)[k]
                                   for i in range(6)]) for k in ref}
        gotc = ctx.lbs_skin(21, pals, n_instances=6)
        for k in ("pos", "normal", "tangent"):
            _nan_aware_equal(gotc[k], refc[k], f"lbs_skin_crowd {k}")
        # the vertex-buffer kernel (AnimatedVertex in, the same layout out)
        ctx.mesh_upload(22, m.to_animated_vertex_aos(), m.n_verts, L["stride"], off_pos=L["off_pos"], off_normal=L["off_normal"],
                        off_tangent=L["off_tangent"], off_weights=L["off_weights"], off_indices=L["off_indices"])
        d_pal, buf = ctx.to_device(pal), ctx.malloc(m.n_verts * L["stride"] + 64)
        ctx.lbs_skin_ex(22, d_pal.ptr, nb, 1, d_out_vertices=buf.ptr, out_stride=0)
        ctx.sync()
        raw = buf.download(np.uint8, m.n_verts * L["stride"]).reshape(m.n_verts, L["stride"])
        for k, off, size in (("pos", L["off_pos"], 12), ("normal", L["off_normal"], 12), ("tangent", L["off_tangent"], 12)):
            _nan_aware_equal(np.ascontiguousarray(raw[:, off:off + size]).view(np.float32), np.ascontiguousarray(ref[k][:, :3]), f"lbs_skin_aos {k}")
        d_pal.free(); buf.free()
    finally:
        ctx.mesh_free(21)
        ctx.mesh_free(22)
    # lbs_skin_dyn (a large single-instance launch)
    big = synth.make_mesh(530_011, nb, 5153)
    upload(ctx, 21, big)
    try:
        got, _ = _skin_device_masked(ctx, 21, big, pal, ("pos", "normal", "tangent"))
        refb = oracle_skin(orc, big, pal)
        for k in ("pos", "normal", "tangent"):
            _nan_aware_equal(got[k], refb[k], f"lbs_skin_dyn {k}")
    finally:
        ctx.mesh_free(21)
    # both batched forms: the projective palette belongs to ONE job of the batch
    for dyn in (1, 0):
        ctx.set_option("lbs.dyn", dyn)
        specs = [(9000, nb, 1, ("pos", "normal", "tangent")), (7000, nb, 1, ("pos", "normal", "tangent")), (5000, nb, 2, ("pos", "normal", "tangent"))]
        scene = _batch_scene(ctx, 7600, specs, 5170)
        try:
            mm, _, dp, o = scene[1]
            dp.upload(pal)
            scene[1] = (mm, pal, dp, o)
            ctx.lbs_skin_batch(_batch_jobs(7600, specs, scene))
            ctx.sync()
            for (nv, nbb, ni, want), (mm, pp, dp, o) in zip(specs, scene):
                refj = oracle_skin(orc, mm, pp, ni)
                for key, width in (("pos", 3), ("normal", 3), ("tangent", 4)):
                    _nan_aware_equal(o[key].download(np.float32, ni * nv * width).reshape(ni * nv, width), refj[key], f"batch dyn={dyn} {key}")
        finally:
            for k, (mm, pp, dp, o) in enumerate(scene):
                for b in list(o.values()) + [dp]:
                    b.free()
                ctx.mesh_free(7600 + k)
    ctx.set_option("lbs.dyn", 1)


def test_a_palette_one_matrix_short_of_the_largest_bone_index_is_refused(ctx):
    """The Rust loop indexes `bone_matrices[bone_index as usize]` (scene/mesh/mod.rs:514): a palette that ends exactly ON the mesh's largest
    bone index is one matrix short.  n_bones == max index is refused, max index + 1 accepted (the check's `>=`; tools/mutants.py: `>` survived
    a suite whose short palettes were all much shorter)."""
    m = synth.make_mesh(500, 8, 5180)
    idx = m.indices.copy()
    idx[7, 2] = 7                                               # the largest index there is, whatever the generator drew
    ctx.mesh_upload_soa(23, m.pos, m.weights, idx, m.normal, m.tangent)
    try:
        assert ctx.mesh_info(23)["max_bone_index"] == 7
        with pytest.raises(fyrox_amd.FyxError) as e:
            ctx.lbs_skin(23, synth.make_palette(7, 5180))
        assert e.value.status == "FYX_ERR_BONE_INDEX"
        with pytest.raises(fyrox_amd.FyxError) as e:
            ctx.skinned_aabb(23, synth.make_palette(7, 5180))
        assert e.value.status == "FYX_ERR_BONE_INDEX"
        assert ctx.lbs_skin(23, synth.make_palette(8, 5180), want=("pos",))["pos"].shape == (500, 3)
    finally:
        ctx.mesh_free(23)


def test_a_reported_failure_does_not_come_back_as_the_next_launch_error(ctx, orc):
    """The HIP runtime keeps a thread's last error until somebody asks for it, and the library asks after every kernel launch: an
    allocation that failed -- and was reported as FYX_ERR_OOM -- must not resurface as the launch error of the next frame's pose update or
    of the next skinning call (found when a GPU test file that updates poses first ran BEHIND this file's allocation-failure cases)."""
    from fyrox_amd import anim as A
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.malloc(1 << 50)
    assert e.value.status == "FYX_ERR_OOM"
    rig = synth.make_rig(5, 31)
    td, tgt = synth.make_clip(5, 31, 0, n_keys=4, fps=4.0, euler_every=10 ** 9)
    A.create_rig(ctx, 2_000_000_001, rig)
    A.upload_tracks_data(ctx, 2_000_000_002, td)
    an = A.Animator(ctx, 2_000_000_003, 2_000_000_001, rig, 1)
    an.add_animation(2_000_000_002, tgt)
    try:
        an.update_animations(1 / 60)                       # the one-launch frame: launch + hipGetLastError
        with pytest.raises(fyrox_amd.FyxError):
            ctx.malloc(1 << 50)
        m = synth.make_mesh(3000, 8, 32)
        pal = synth.make_palette(8, 32)
        upload(ctx, 15, m)
        assert_bit_exact(ctx.lbs_skin(15, pal), oracle_skin(orc, m, pal))
        with pytest.raises(fyrox_amd.FyxError):
            ctx.malloc(1 << 50)
        an.update_animations(1 / 60)
    finally:
        an.free()
        ctx.mesh_free(15)


def test_error_codes(ctx):
    m = synth.make_mesh(100, 8, 61)
    upload(ctx, 13, m)
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.lbs_skin(13, synth.make_palette(4, 61))          # palette shorter than max index + 1
    assert e.value.status == "FYX_ERR_BONE_INDEX"
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.lbs_skin(987654, synth.make_palette(8, 61))
    assert e.value.status == "FYX_ERR_UNKNOWN_ID"
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.mesh_upload(14, np.zeros(680, np.uint8), 10, 68, off_pos=0, off_weights=60, off_indices=64)
    assert e.value.status == "FYX_ERR_INVALID_ARG"           # weights @60 + 16 > 68
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.mesh_upload(14, np.zeros(680, np.uint8), 10, 68, off_pos=0, off_weights=-1, off_indices=64)
    assert e.value.status == "FYX_ERR_MISSING_ATTRIBUTE"
    with pytest.raises(fyrox_amd.FyxError):
        ctx.set_option("lbs.blocks_per_cu", 100)
    with pytest.raises(fyrox_amd.FyxError):
        ctx.set_option("lbs.block", 512)           # a kernel-variant switch of round 2: no longer an option
    ctx.mesh_free(13)
    with pytest.raises(fyrox_amd.FyxError):
        ctx.mesh_free(13)


def test_reference_vertex_buffer_fixture_through_aos_upload(ctx, orc, golden):
    # fyrox-impl/src/scene/mesh/buffer.rs:1688-1800: 76-byte test vertices, indices (1,2,3,4)
    g = golden["vertex_buffer_fixture"]
    vs = g["vertices"]
    aos = np.zeros((len(vs), g["stride"]), np.uint8)
    for i, v in enumerate(vs):
        rec = np.concatenate([np.float32(v["position"]), np.float32(v["tex_coord"]), np.float32(v["second_tex_coord"]),
                              np.float32(v["normal"]), np.float32(v["tangent"]), np.float32(v["bone_weights"])])
        aos[i, :72] = rec.view(np.uint8)
        aos[i, 72:76] = v["bone_indices"]
    ctx.mesh_upload(15, aos.reshape(-1), len(vs), g["stride"], off_pos=g["off_pos"], off_normal=g["off_normal"],
                    off_tangent=g["off_tangent"], off_weights=g["off_weights"], off_indices=g["off_indices"])
    assert ctx.mesh_info(15) == {"n_verts": 3, "max_bone_index": 4, "has_normal": True, "has_tangent": True}
    pal = synth.make_palette(5, 71)
    got = ctx.lbs_skin(15, pal)
    pos = np.float32([v["position"] for v in vs]); nrm = np.float32([v["normal"] for v in vs])
    tan = np.float32([v["tangent"] for v in vs]); w = np.float32([v["bone_weights"] for v in vs])
    idx = np.uint8([v["bone_indices"] for v in vs])
    ref = orc.lbs_skin(pos, w, idx, pal, nrm, tan)
    assert_bit_exact(got, ref)
    box = ctx.skinned_aabb(15, pal)
    assert np.array_equal(box, orc.accurate_world_bounding_box(aos.reshape(-1), 3, g["stride"], g["off_pos"],
                                                               g["off_weights"], g["off_indices"], pal))


def test_skinned_aabb_matches_accurate_world_bounding_box(ctx, orc):
    m = synth.make_mesh(123_457, 64, 81)
    pal = synth.make_palette(64, 81)
    upload(ctx, 16, m, aos=True)
    L = synth.ANIMATED_VERTEX
    ref = orc.accurate_world_bounding_box(m.to_animated_vertex_aos(), m.n_verts, L["stride"], L["off_pos"],
                                          L["off_weights"], L["off_indices"], pal)
    assert np.array_equal(ctx.skinned_aabb(16, pal), ref)


def test_palette_kernel_bit_exact(ctx, orc):
    for n in (1, 4, 64, 255, 256, 1000):
        g, ib = synth.make_bone_transforms(n, 91)
        assert np.array_equal(ctx.palette(g, ib), orc.palette(g, ib))


def test_device_resident_path_and_reupload(ctx, orc):
    m = synth.make_mesh(20_000, 64, 101)
    pal = synth.make_palette(64, 101)
    upload(ctx, 17, m)
    d_pal = ctx.to_device(pal)
    d_pos = ctx.malloc(m.n_verts * 12); d_nrm = ctx.malloc(m.n_verts * 12); d_tan = ctx.malloc(m.n_verts * 16)
    ctx.lbs_skin_device(17, d_pal.ptr, 64, 1, d_pos.ptr, d_nrm.ptr, d_tan.ptr)
    ctx.sync()
    ref = oracle_skin(orc, m, pal)
    assert np.array_equal(d_pos.download(np.float32, m.n_verts * 3).reshape(-1, 3), ref["pos"])
    assert np.array_equal(d_tan.download(np.float32, m.n_verts * 4).reshape(-1, 4), ref["tangent"])
    # raw-stream form on the registry's own streams
    s = ctx.mesh_streams(17)
    ctx.lbs_skin_streams(m.n_verts, s["pos"], s["normal"], s["tangent"], s["weights"], s["indices"],
                         d_pal.ptr, 64, 1, d_pos.ptr, d_nrm.ptr, 0)
    ctx.sync()
    assert np.array_equal(d_nrm.download(np.float32, m.n_verts * 3).reshape(-1, 3), ref["normal"])
    # re-upload under the same id (SurfaceData modified): old streams are replaced
    m2 = synth.make_mesh(777, 64, 102)
    upload(ctx, 17, m2)
    assert ctx.mesh_info(17)["n_verts"] == 777
    assert_bit_exact(ctx.lbs_skin(17, pal), oracle_skin(orc, m2, pal))
    for b in (d_pal, d_pos, d_nrm, d_tan):
        b.free()


@pytest.mark.parametrize("streams", [1, 2, 3, 4])
def test_overlapped_launch_streams_and_join(ctx, orc, streams):
    # independent launches are dealt onto `lbs.streams` worker streams; results are complete after
    # fyx_sync (implicit join); a dependent second pass over the same output needs fyx_join
    ctx.set_option("lbs.streams", streams)
    m = synth.make_mesh(30_011, 64, 131)
    upload(ctx, 18, m)
    pals = [synth.make_palette(64, 200 + i) for i in range(6)]
    d_pals = [ctx.to_device(p) for p in pals]
    outs = [(ctx.malloc(m.n_verts * 12), ctx.malloc(m.n_verts * 12), ctx.malloc(m.n_verts * 16)) for _ in pals]
    for _ in range(3):
        for dp, o in zip(d_pals, outs):
            ctx.lbs_skin_device(18, dp.ptr, 64, 1, o[0].ptr, o[1].ptr, o[2].ptr)
    ctx.sync()
    for p, o in zip(pals, outs):
        ref = oracle_skin(orc, m, p)
        assert np.array_equal(o[0].download(np.float32, m.n_verts * 3).reshape(-1, 3), ref["pos"])
        assert np.array_equal(o[1].download(np.float32, m.n_verts * 3).reshape(-1, 3), ref["normal"])
        assert np.array_equal(o[2].download(np.float32, m.n_verts * 4).reshape(-1, 4), ref["tangent"])
    # same output buffer reused by two launches: join in between -> the later palette wins
    ctx.lbs_skin_device(18, d_pals[0].ptr, 64, 1, outs[0][0].ptr, 0, 0)
    ctx.join()
    ctx.lbs_skin_device(18, d_pals[1].ptr, 64, 1, outs[0][0].ptr, 0, 0)
    ctx.sync()
    assert np.array_equal(outs[0][0].download(np.float32, m.n_verts * 3).reshape(-1, 3), oracle_skin(orc, m, pals[1])["pos"])
    # uploads are ordered before later launches (fork event): re-upload then skin immediately
    m2 = synth.make_mesh(30_011, 64, 132)
    upload(ctx, 18, m2)
    ctx.lbs_skin_device(18, d_pals[2].ptr, 64, 1, outs[2][0].ptr, 0, 0)
    ctx.sync()
    assert np.array_equal(outs[2][0].download(np.float32, m.n_verts * 3).reshape(-1, 3), oracle_skin(orc, m2, pals[2])["pos"])
    for b in d_pals + [x for o in outs for x in o]:
        b.free()


# ---- blend shapes ahead of skinning, interleaved (render-ready) output -------------------------------------

def oracle_skin_shapes(orc, m, pal, storage, plane, weights, n_inst=1):
    """standard.shader:157-200: per instance, offsets of every shape added first, then the four-bone blend."""
    nb = pal.shape[0] // n_inst
    w = np.asarray(weights, np.float32).reshape(n_inst, -1)
    outs = []
    for i in range(n_inst):
        p, n, t = orc.apply_blend_shapes(m.pos, m.normal, m.tangent, storage, plane, w[i])
        outs.append(orc.lbs_skin(p, m.weights, m.indices, pal[i * nb:(i + 1) * nb], n, t, threads=0))
    return {k: np.concatenate([o[k] for o in outs]) for k in outs[0]}


def run_ex(ctx, mesh_id, m, pal, n_inst, weights=None, aos_layout=None, init_bytes=None):
    nv = m.n_verts * n_inst
    d_pal = ctx.to_device(pal)
    d_w = ctx.to_device(np.asarray(weights, np.float32)) if weights is not None else None
    ns = 0 if weights is None else np.asarray(weights).size // n_inst
    try:
        if aos_layout is None:
            bufs = [ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64)]
            ctx.lbs_skin_ex(mesh_id, d_pal.ptr, pal.shape[0] // n_inst, n_inst, d_blend_shape_weights=d_w.ptr if d_w else 0,
                            n_blend_shapes=ns, d_out_pos=bufs[0].ptr, d_out_normal=bufs[1].ptr, d_out_tangent=bufs[2].ptr)
            ctx.sync()
            out = {"pos": bufs[0].download(np.float32, nv * 3).reshape(nv, 3),
                   "normal": bufs[1].download(np.float32, nv * 3).reshape(nv, 3),
                   "tangent": bufs[2].download(np.float32, nv * 4).reshape(nv, 4)}
            for b in bufs:
                b.free()
            return out
        stride, op, on, ot = aos_layout
        buf = ctx.to_device(init_bytes)
        ctx.lbs_skin_ex(mesh_id, d_pal.ptr, pal.shape[0] // n_inst, n_inst, d_blend_shape_weights=d_w.ptr if d_w else 0,
                        n_blend_shapes=ns, d_out_vertices=buf.ptr, out_stride=stride, out_off_pos=op, out_off_normal=on,
                        out_off_tangent=ot)
        ctx.sync()
        raw = buf.download(np.uint8, nv * stride).reshape(nv, stride)
        buf.free()
        return raw
    finally:
        d_pal.free()
        if d_w:
            d_w.free()


@pytest.mark.parametrize("n_verts,n_shapes,n_inst", [(1000, 1, 1), (4099, 5, 1), (50_000, 3, 2), (777, 128, 1)])
def test_blend_shapes_then_skinning_bit_exact(ctx, orc, n_verts, n_shapes, n_inst):
    m = synth.make_mesh(n_verts, 64, synth.SEED_BASE + 20)
    pal = synth.make_palette(64, synth.SEED_BASE + 20, n_instances=n_inst)
    storage, plane, w = synth.make_blend_shapes(n_verts, n_shapes, synth.SEED_BASE + 20)
    weights = np.stack([w * np.float32(1.0 - 0.3 * i) for i in range(n_inst)])
    upload(ctx, 60, m)
    ctx.mesh_set_blend_shapes(60, storage, n_shapes, plane)
    got = run_ex(ctx, 60, m, pal, n_inst, weights)
    assert_bit_exact(got, oracle_skin_shapes(orc, m, pal, storage, plane, weights, n_inst))
    # all-zero weights: x + offset*0 == x, so the plain kernel's result
    zero = run_ex(ctx, 60, m, pal, n_inst, np.zeros_like(weights))
    assert_bit_exact(zero, oracle_skin(orc, m, pal, n_inst))
    # removing the shapes again
    ctx.mesh_set_blend_shapes(60, None, 0, 0)
    with pytest.raises(fyrox_amd.FyxError):
        run_ex(ctx, 60, m, pal, n_inst, weights)
    ctx.mesh_free(60)


def test_blend_shapes_fused_mode_within_tolerance(ctx, orc):
    m = synth.make_mesh(30_011, 64, synth.SEED_BASE + 21)
    pal = synth.make_palette(64, synth.SEED_BASE + 21)
    storage, plane, w = synth.make_blend_shapes(m.n_verts, 4, synth.SEED_BASE + 21)
    upload(ctx, 61, m)
    ctx.mesh_set_blend_shapes(61, storage, 4, plane)
    ctx.set_option("lbs.exact", 0)
    got = run_ex(ctx, 61, m, pal, 1, w)
    ref = oracle_skin_shapes(orc, m, pal, storage, plane, w)
    for k in ("pos", "normal", "tangent"):
        assert rel_err(got[k], ref[k]) <= REL_TOL, k   # tolerance: 1e-5 relative (north_star)
    ctx.mesh_free(61)


# (the last two: long enough for every wave of lbs_skin_aos to go back to its workgroup's ticket -- twelve / six units per workgroup)
@pytest.mark.parametrize("n_verts,n_inst,shapes", [(9_001, 2, 0), (64, 1, 2), (63, 3, 0), (50_000, 1, 3), (1, 1, 1), (4096, 2, 0), (600_000, 1, 0), (300_000, 1, 2)])
@pytest.mark.parametrize("exact", [1, 0])
def test_vertex_buffer_in_vertex_buffer_out(ctx, orc, n_verts, n_inst, shapes, exact):
    """out_stride == 0: the mesh's own AnimatedVertex layout (vertex.rs:139-155).  Every output vertex is the input
    vertex with position / normal / tangent.xyz replaced; uv, tangent.w, bone weights and indices pass through."""
    L = synth.ANIMATED_VERTEX
    m = synth.make_mesh(n_verts, 48, synth.SEED_BASE + 23, coherent=False)
    pal = synth.make_palette(48, synth.SEED_BASE + 23, n_instances=n_inst)
    upload(ctx, 64, m, aos=True)
    weights = None
    if shapes:
        storage, plane, w = synth.make_blend_shapes(n_verts, shapes, synth.SEED_BASE + 23)
        ctx.mesh_set_blend_shapes(64, storage, shapes, plane)
        weights = np.stack([w * np.float32(1.0 + 0.25 * i) for i in range(n_inst)])
        ref = oracle_skin_shapes(orc, m, pal, storage, plane, weights, n_inst)
    else:
        ref = oracle_skin(orc, m, pal, n_inst)
    src = m.to_animated_vertex_aos().reshape(n_verts, L["stride"])
    ctx.set_option("lbs.exact", exact)
    guard = 256   # bytes after the last vertex must stay untouched (ragged last span)
    init = np.full(n_verts * n_inst * L["stride"] + guard, 0xEE, np.uint8)
    d_pal, buf = ctx.to_device(pal), ctx.to_device(init)
    d_w = ctx.to_device(weights) if shapes else None
    ctx.lbs_skin_ex(64, d_pal.ptr, 48, n_inst, d_blend_shape_weights=d_w.ptr if d_w else 0, n_blend_shapes=shapes,
                    d_out_vertices=buf.ptr, out_stride=0)
    ctx.sync()
    raw = buf.download(np.uint8, init.size)
    assert np.all(raw[-guard:] == 0xEE), "wrote past the last vertex"
    raw = raw[:-guard].reshape(n_verts * n_inst, L["stride"])
    expect = np.tile(src, (n_inst, 1))
    skinned = np.zeros(L["stride"], bool)
    for off, size, key in ((L["off_pos"], 12, "pos"), (L["off_normal"], 12, "normal"), (L["off_tangent"], 12, "tangent")):
        got = np.ascontiguousarray(raw[:, off:off + size]).view(np.float32)
        want = ref[key][:, :3]
        if exact:
            assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32)), key
        else:
            assert rel_err(got, want) <= REL_TOL, key
        skinned[off:off + size] = True
    assert np.array_equal(raw[:, ~skinned], expect[:, ~skinned]), "pass-through attributes changed"
    for b in (d_pal, buf, d_w):
        if b:
            b.free()
    ctx.mesh_free(64)


def test_vertex_buffer_output_needs_an_interleaved_upload(ctx):
    from fyrox_amd import _native
    m = synth.make_mesh(100, 8, 5)
    upload(ctx, 65, m, aos=False)
    pal, out = ctx.to_device(synth.make_palette(8, 5)), ctx.malloc(100 * 68)
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.lbs_skin_ex(65, pal.ptr, 8, 1, d_out_vertices=out.ptr, out_stride=0)
    assert e.value.code == _native.FYX_ERR_MISSING_ATTRIBUTE
    pal.free(); out.free()
    ctx.mesh_free(65)


@pytest.mark.parametrize("layout", ["animated_vertex", "static_vertex", "pos_only"])
@pytest.mark.parametrize("shapes", [0, 3])
def test_interleaved_output_is_a_render_ready_vertex_buffer(ctx, orc, layout, shapes):
    """Position / normal / tangent land at their offsets of the engine's vertex layout (vertex.rs:139-155 for
    AnimatedVertex: stride 68; StaticVertex: 48) and every other byte of the buffer is left alone."""
    n_inst = 2
    m = synth.make_mesh(9_001, 32, synth.SEED_BASE + 22, coherent=False)
    pal = synth.make_palette(32, synth.SEED_BASE + 22, n_instances=n_inst)
    upload(ctx, 62, m, aos=True)
    weights = None
    if shapes:
        storage, plane, w = synth.make_blend_shapes(m.n_verts, shapes, synth.SEED_BASE + 22)
        ctx.mesh_set_blend_shapes(62, storage, shapes, plane)
        weights = np.stack([w, w * np.float32(0.5)])
        ref = oracle_skin_shapes(orc, m, pal, storage, plane, weights, n_inst)
    else:
        ref = oracle_skin(orc, m, pal, n_inst)
    if layout == "animated_vertex":
        L = synth.ANIMATED_VERTEX
        stride, op, on, ot = L["stride"], L["off_pos"], L["off_normal"], L["off_tangent"]
        init = np.tile(m.to_animated_vertex_aos().reshape(m.n_verts, stride), (n_inst, 1))
    elif layout == "static_vertex":
        stride, op, on, ot = 48, 0, 20, 32      # position, tex_coord, normal, tangent (vertex.rs:34-44)
        init = np.full((m.n_verts * n_inst, stride), 0xA5, np.uint8)
    else:
        stride, op, on, ot = 16, 4, -1, -1
        init = np.full((m.n_verts * n_inst, stride), 0x3C, np.uint8)
    raw = run_ex(ctx, 62, m, pal, n_inst, weights, (stride, op, on, ot), init)
    touched = np.zeros(stride, bool)
    for off, size, key in ((op, 12, "pos"), (on, 12, "normal"), (ot, 16, "tangent")):
        if off < 0:
            continue
        got = np.ascontiguousarray(raw[:, off:off + size]).view(np.float32)
        assert np.array_equal(got.view(np.uint32), ref[key].view(np.uint32)), (layout, key)
        touched[off:off + size] = True
    assert np.array_equal(raw[:, ~touched], init[:, ~touched]), "bytes outside the written attributes changed"
    ctx.mesh_free(62)


def test_skin_ex_argument_errors(ctx):
    from fyrox_amd import _native
    m = synth.make_mesh(300, 8, 5)
    upload(ctx, 63, m)
    pal = ctx.to_device(synth.make_palette(8, 5))
    out = ctx.malloc(300 * 68)
    cases = [dict(d_out_vertices=out.ptr, out_stride=70, out_off_pos=0), dict(d_out_vertices=out.ptr, out_stride=68, out_off_pos=2),
             dict(d_out_vertices=out.ptr, out_stride=68, out_off_pos=60),
             dict(d_out_vertices=out.ptr, out_stride=68, out_off_pos=0, d_out_pos=out.ptr),
             dict(d_out_pos=out.ptr, n_blend_shapes=2, d_blend_shape_weights=out.ptr)]
    codes = []
    for kw in cases:
        with pytest.raises(fyrox_amd.FyxError) as e:
            ctx.lbs_skin_ex(63, pal.ptr, 8, 1, **kw)
        codes.append(e.value.code)
    assert codes == [_native.FYX_ERR_UNSUPPORTED, _native.FYX_ERR_UNSUPPORTED, _native.FYX_ERR_INVALID_ARG,
                     _native.FYX_ERR_INVALID_ARG, _native.FYX_ERR_INVALID_ARG]
    with pytest.raises(fyrox_amd.FyxError) as e:   # more shapes than the shader's weight array holds
        ctx.mesh_set_blend_shapes(63, np.zeros((129, 300, 9), np.uint16), 129, 300)
    assert e.value.code == _native.FYX_ERR_UNSUPPORTED
    pal.free(); out.free()
    ctx.mesh_free(63)


# ---- vertex-buffer-in / vertex-buffer-out on other vertex layouts -------------------------------------------

def _custom_aos(m, stride, offs, fill=0x5A):
    """Interleave a mesh into an arbitrary layout; bytes no attribute covers are `fill`."""
    n = m.n_verts
    aos = np.full((n, stride), fill, np.uint8)
    parts = {"pos": (m.pos, 12), "normal": (m.normal, 12), "tangent": (m.tangent, 16), "weights": (m.weights, 16)}
    for key, (arr, size) in parts.items():
        if offs.get(key, -1) >= 0:
            aos[:, offs[key]:offs[key] + size] = np.ascontiguousarray(arr).view(np.uint8).reshape(n, size)
    aos[:, offs["indices"]:offs["indices"] + 4] = m.indices
    return aos


@pytest.mark.parametrize("stride,offs", [
    (32, dict(pos=0, weights=12, indices=28)),                                   # position only: smallest skinnable vertex
    (48, dict(pos=4, weights=16, indices=32, normal=36)),                        # no tangent
    (64, dict(weights=0, indices=16, pos=20, normal=32, tangent=44)),            # attributes in another order
    (80, dict(pos=0, normal=12, tangent=24, weights=40, indices=56)),            # padded vertex
    (128, dict(pos=64, normal=76, tangent=88, weights=104, indices=120)),        # a second UV set etc. in front
    (160, dict(pos=0, normal=140, tangent=100, weights=60, indices=156)),        # the largest supported stride
], ids=lambda v: str(v) if isinstance(v, int) else "")
@pytest.mark.parametrize("n_verts", [1000, 129])
def test_vertex_buffer_path_handles_any_f32_layout(ctx, orc, stride, offs, n_verts):
    """out_stride == 0 for layouts other than AnimatedVertex: every per-lane span count the kernel is instantiated for
    (strides up to 64, 80, 128 and 160 bytes), attributes in any order, optional normal / tangent."""
    m = synth.make_mesh(n_verts, 24, synth.SEED_BASE + 40 + stride, coherent=False)
    pal = synth.make_palette(24, synth.SEED_BASE + 40)
    src = _custom_aos(m, stride, offs)
    ctx.mesh_upload(66, src.reshape(-1), n_verts, stride, off_pos=offs["pos"], off_normal=offs.get("normal", -1),
                    off_tangent=offs.get("tangent", -1), off_weights=offs["weights"], off_indices=offs["indices"])
    ref = orc.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal if "normal" in offs else None,
                       m.tangent if "tangent" in offs else None, threads=0)
    guard = 192
    d_pal, buf = ctx.to_device(pal), ctx.to_device(np.full(n_verts * stride + guard, 0xEE, np.uint8))
    ctx.lbs_skin_ex(66, d_pal.ptr, 24, 1, d_out_vertices=buf.ptr, out_stride=0)
    ctx.sync()
    raw = buf.download(np.uint8, n_verts * stride + guard)
    assert np.all(raw[-guard:] == 0xEE)
    raw = raw[:-guard].reshape(n_verts, stride)
    skinned = np.zeros(stride, bool)
    for key, size in (("pos", 12), ("normal", 12), ("tangent", 12)):
        if key in offs:
            got = np.ascontiguousarray(raw[:, offs[key]:offs[key] + size]).view(np.float32)
            assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(ref[key][:, :3]).view(np.uint32)), key
            skinned[offs[key]:offs[key] + size] = True
    assert np.array_equal(raw[:, ~skinned], src[:, ~skinned]), "pass-through bytes changed"
    d_pal.free(); buf.free()
    ctx.mesh_free(66)


def test_vertex_buffer_path_rejects_oversized_or_unaligned_layouts(ctx, orc):
    from fyrox_amd import _native
    m = synth.make_mesh(64, 8, 5)
    pal, out = ctx.to_device(synth.make_palette(8, 5)), ctx.malloc(64 * 200)
    for stride, offs in ((164, dict(pos=0, weights=12, indices=28)), (34, dict(pos=0, weights=14, indices=30))):
        src = _custom_aos(m, stride, offs)
        ctx.mesh_upload(67, src.reshape(-1), 64, stride, off_pos=offs["pos"], off_weights=offs["weights"], off_indices=offs["indices"])
        with pytest.raises(fyrox_amd.FyxError) as e:
            ctx.lbs_skin_ex(67, pal.ptr, 8, 1, d_out_vertices=out.ptr, out_stride=0)
        assert e.value.code == _native.FYX_ERR_UNSUPPORTED
        got = ctx.lbs_skin(67, synth.make_palette(8, 5), want=("pos",))["pos"]                       # the SoA path still serves it
        ref = orc.lbs_skin(m.pos, m.weights, m.indices, synth.make_palette(8, 5), threads=0)["pos"]
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), stride
        ctx.mesh_free(67)
    pal.free(); out.free()


@pytest.mark.parametrize("stride,offs", [
    (61, dict(pos=1, normal=13, tangent=25, weights=41, indices=57)),      # a byte in front: nothing is word-aligned in any vertex
    (70, dict(pos=2, normal=14, tangent=26, weights=42, indices=58)),      # half-word offsets and a stride that is no multiple of 4
    (69, dict(pos=0, normal=20, tangent=32, weights=48, indices=64)),      # AnimatedVertex + one byte: aligned in every fourth vertex only
], ids=lambda v: str(v) if isinstance(v, int) else "")
def test_upload_of_a_layout_whose_attributes_are_not_word_aligned(ctx, orc, stride, offs):
    """A VertexBuffer's layout is any list of attributes (buffer.rs:404-415: u8 attributes may precede f32 ones), and the reference reads
    every field byte-wise little-endian (buffer.rs:1279-1321).  fyx_mesh_upload's gather to the attribute streams reads such fields
    byte by byte too (deinterleave_kernel's unaligned road; tools/mutants.py: two of its bytes swapped survived the suite until this test)."""
    m = synth.make_mesh(777, 24, synth.SEED_BASE + 47, coherent=False)
    pal = synth.make_palette(24, synth.SEED_BASE + 47)
    src = _custom_aos(m, stride, offs)
    ctx.mesh_upload(68, src.reshape(-1), m.n_verts, stride, off_pos=offs["pos"], off_normal=offs["normal"], off_tangent=offs["tangent"],
                    off_weights=offs["weights"], off_indices=offs["indices"])
    try:
        assert_bit_exact(ctx.lbs_skin(68, pal), orc.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal, m.tangent, threads=0))
    finally:
        ctx.mesh_free(68)


def test_single_rank_communicator_all_gathers_a_skinned_shard(ctx, orc):
    """The exchange step behind the C ABI (fyx_comm_* / fyx_allgather_f32) with a communicator of one rank: the gathered
    buffer is this rank's skinned shard, bit for bit, and the collective is ordered after the skinning launches without
    a host sync in between.  The multi-rank layout (rank r's shard at r * count) is covered on CPU by
    tests/test_sharding.py with the same sharding arithmetic."""
    from fyrox_amd.sharding import vertex_range
    m = synth.make_mesh(20_000, 64, synth.SEED_BASE + 41)
    pal = synth.make_palette(64, synth.SEED_BASE + 41)
    b, e = vertex_range(m.n_verts, 1, 3)          # the middle shard of a three-way split
    ctx.mesh_upload_soa(7101, m.pos[b:e], m.weights[b:e], m.indices[b:e], m.normal[b:e], m.tangent[b:e])
    n = e - b
    d_pal = ctx.to_device(pal)
    d_p, d_n, d_t = ctx.malloc(n * 12), ctx.malloc(n * 12), ctx.malloc(n * 16)
    d_g = ctx.malloc(n * 12)
    with pytest.raises(fyrox_amd.FyxError):
        ctx.allgather_f32(d_p.ptr, n * 3, d_g.ptr)     # no communicator yet
    try:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    except fyrox_amd.FyxError as err:
        if err.code == fyrox_amd._native.FYX_ERR_UNSUPPORTED:
            pytest.skip("librccl.so not present")
        if "ncclGetUniqueId" in str(err) or "ncclCommInitRank" in str(err):
            # RCCL's bootstrap needs a usable network interface even for one rank: an environment matter, not a
            # property of the library under test (the all-gather itself is asserted below whenever RCCL comes up)
            pytest.skip(f"RCCL could not initialise here: {err}")
        raise
    with pytest.raises(fyrox_amd.FyxError):
        ctx.comm_init(bytes(128), 0, 1)                # one communicator per context
    ctx.lbs_skin_device(7101, d_pal.ptr, 64, 1, d_p.ptr, d_n.ptr, d_t.ptr)
    ctx.allgather_f32(d_p.ptr, n * 3, d_g.ptr)
    ctx.sync()
    ref = orc.lbs_skin(m.pos[b:e], m.weights[b:e], m.indices[b:e], pal, m.normal[b:e], m.tangent[b:e], threads=0)
    got = d_g.download(np.float32, n * 3).reshape(n, 3)
    assert np.array_equal(got.view(np.uint32), ref["pos"].view(np.uint32))
    # the in-place form (fyx_allgather_skinned): with one rank the shard is the whole mesh, written where it belongs;
    # the grouped broadcasts rooted at rank 0 leave every byte as the skinning launch wrote it
    assert ctx.comm_info() == (0, 1)
    for d in (d_p, d_n, d_t):
        d.upload(np.zeros(d.nbytes // 4, np.uint32))
    ctx.lbs_skin_device(7101, d_pal.ptr, 64, 1, d_p.ptr, d_n.ptr, d_t.ptr)
    ctx.allgather_skinned(n, d_p.ptr, d_n.ptr, d_t.ptr)
    ctx.allgather_skinned(n, d_p.ptr, 0, 0)            # any subset of the streams
    ctx.sync()
    assert np.array_equal(d_p.download(np.float32, n * 3).reshape(n, 3), ref["pos"])
    assert np.array_equal(d_n.download(np.float32, n * 3).reshape(n, 3), ref["normal"])
    assert np.array_equal(d_t.download(np.float32, n * 4).reshape(n, 4), ref["tangent"])
    ctx.comm_shutdown()
    ctx.comm_shutdown()                                # idempotent
    with pytest.raises(fyrox_amd.FyxError):
        ctx.allgather_skinned(n, d_p.ptr, d_n.ptr, d_t.ptr)   # the communicator is gone
    for d in (d_pal, d_p, d_n, d_t, d_g):
        d.free()
    ctx.mesh_free(7101)


@pytest.mark.parametrize("world,n_verts", [(8, 1_000_000), (3, 20_000), (2, 4097), (5, 1000)])
def test_shards_skinned_in_place_make_the_whole_mesh(ctx, orc, world, n_verts):
    """What every rank of a vertex-range split does before the exchange (bench.py's strong-scaling leg): its shard
    -- uploaded as a mesh of its own -- is skinned by the product kernel straight into its place of the FULL streams, i.e. at the
    non-zero offsets d_pos_all + 3 * begin, ... (16-byte aligned for the tangents, 4-byte aligned only for position / normal when
    `begin` is odd... it never is: the cut is 256-vertex aligned).  Here one process plays all ranks in turn on one GPU; the full
    buffers are then compared with the oracle's single pass, every byte, and the bytes past the last vertex must be untouched."""
    nb = 256 if n_verts >= 100_000 else 24
    m = synth.make_mesh(n_verts, nb, synth.SEED_BASE + 44)
    pal = synth.make_palette(nb, synth.SEED_BASE + 44)
    d_pal = ctx.to_device(pal)
    guard = 64
    full = [ctx.to_device(np.full(n_verts * w + guard, 0xFFFFFFFF, np.uint32)) for w in (3, 3, 4)]
    cuts = [sharding_range(n_verts, r, world) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == n_verts and all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
    for r, (b, e) in enumerate(cuts):
        if e == b:
            continue
        ctx.mesh_upload_soa(7300 + r, m.pos[b:e], m.weights[b:e], m.indices[b:e], m.normal[b:e], m.tangent[b:e])
        ctx.lbs_skin_device(7300 + r, d_pal.ptr, nb, 1, full[0].ptr + 12 * b, full[1].ptr + 12 * b, full[2].ptr + 16 * b)
    ctx.sync()
    ref = orc.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal, m.tangent, threads=0)
    for buf, w, key in zip(full, (3, 3, 4), ("pos", "normal", "tangent")):
        raw = buf.download(np.uint32, n_verts * w + guard)
        assert np.all(raw[n_verts * w:] == 0xFFFFFFFF), f"{key}: wrote past the last vertex"
        assert np.array_equal(raw[:n_verts * w].reshape(n_verts, w), ref[key].view(np.uint32)), key
    for r, (b, e) in enumerate(cuts):
        if e > b:
            ctx.mesh_free(7300 + r)
    for d in (d_pal, *full):
        d.free()


def sharding_range(n_verts, rank, world):
    from fyrox_amd.sharding import vertex_range_native
    return vertex_range_native(n_verts, rank, world)


def test_exchange_forms_agree_on_one_rank(ctx, orc):
    """Option comm.form: 0 = one broadcast per shard, 1 = grouped send / recv between every pair of ranks, 2 = one in-place all-gather
    per stream over the padded cut (buffers of n_ranks * shard vertices).  With a communicator of one rank form 1 has no peer (an empty
    group), form 0 one broadcast to itself, form 2 an all-gather of one shard onto itself: all must leave the skinned streams as they are."""
    m = synth.make_mesh(5000, 16, synth.SEED_BASE + 45)
    pal = synth.make_palette(16, synth.SEED_BASE + 45)
    ctx.mesh_upload_soa(7400, m.pos, m.weights, m.indices, m.normal, m.tangent)
    d_pal = ctx.to_device(pal)
    n = m.n_verts
    from fyrox_amd import sharding
    b0, e0, shard = sharding.vertex_range_padded(n, 0, 1)
    assert (b0, e0) == (0, n) and shard >= n and shard % 256 == 0
    d_p, d_n, d_t = ctx.malloc(shard * 12), ctx.malloc(shard * 12), ctx.malloc(shard * 16)     # room for the padded shard of form 2
    with pytest.raises(fyrox_amd.FyxError):
        ctx.set_option("comm.form", 3)
    try:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    except fyrox_amd.FyxError as err:
        if err.code == fyrox_amd._native.FYX_ERR_UNSUPPORTED or "ncclGetUniqueId" in str(err) or "ncclCommInitRank" in str(err):
            pytest.skip(f"RCCL could not initialise here: {err}")
        raise
    ref = orc.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal, m.tangent, threads=0)
    try:
        for form in (1, 2, 0):
            ctx.set_option("comm.form", form)
            assert ctx.get_option("comm.form") == form
            ctx.lbs_skin_device(7400, d_pal.ptr, 16, 1, d_p.ptr, d_n.ptr, d_t.ptr)
            if form == 2:
                # the padded form writes n_ranks * shard vertices per stream: only through the entry point that is told the capacity
                with pytest.raises(fyrox_amd.FyxError) as e:
                    ctx.allgather_skinned(n, d_p.ptr, d_n.ptr, d_t.ptr)
                assert e.value.code == fyrox_amd._native.FYX_ERR_INVALID_ARG and "fyx_allgather_skinned_padded" in str(e.value)
                with pytest.raises(fyrox_amd.FyxError) as e:
                    ctx.allgather_skinned_padded(n, n, d_p.ptr, d_n.ptr, d_t.ptr)          # buffers of n < shard vertices
                assert e.value.code == fyrox_amd._native.FYX_ERR_INVALID_ARG
                ctx.allgather_skinned_padded(n, shard, d_p.ptr, d_n.ptr, d_t.ptr)
            else:
                ctx.allgather_skinned(n, d_p.ptr, d_n.ptr, d_t.ptr)
            ctx.sync()
            assert np.array_equal(d_p.download(np.float32, n * 3).reshape(n, 3), ref["pos"])
            assert np.array_equal(d_t.download(np.float32, n * 4).reshape(n, 4), ref["tangent"])
    finally:
        ctx.set_option("comm.form", 0)
        ctx.comm_shutdown()
    for d in (d_pal, d_p, d_n, d_t):
        d.free()
    ctx.mesh_free(7400)


def test_one_process_form_of_the_exchange(ctx, orc):
    """fyx_comm_init_all / fyx_allgather_skinned_all -- one process (one thread) driving every GPU, the engine's shape --
    as far as one GPU can take them: a communicator of one context, the in-place exchange ordered after the skinning
    launches, and the refusals (two contexts on one GPU, contexts out of order, a stream missing on one GPU).  With more
    GPUs the same calls issue one grouped RCCL operation over all communicators; that has not run here."""
    m = synth.make_mesh(30_011, 32, synth.SEED_BASE + 43)
    pal = synth.make_palette(32, synth.SEED_BASE + 43)
    other = fyrox_amd.Context(0)                       # a second context on the same GPU
    try:
        with pytest.raises(fyrox_amd.FyxError) as e:
            fyrox_amd.Context.comm_init_all([ctx, other])
        assert "same GPU" in str(e.value)
        for c in (ctx, other):                         # and nobody keeps half a communicator
            with pytest.raises(fyrox_amd.FyxError):
                c.comm_info()
        try:
            fyrox_amd.Context.comm_init_all([ctx])
        except fyrox_amd.FyxError as err:
            if err.code == fyrox_amd._native.FYX_ERR_UNSUPPORTED or "ncclGetUniqueId" in str(err) or "ncclCommInitRank" in str(err):
                pytest.skip(f"RCCL could not initialise here: {err}")
            raise
        assert ctx.comm_info() == (0, 1)
        with pytest.raises(fyrox_amd.FyxError):
            fyrox_amd.Context.comm_init_all([ctx])     # one communicator per context
        ctx.mesh_upload_soa(7102, m.pos, m.weights, m.indices, m.normal, m.tangent)
        d_pal = ctx.to_device(pal)
        n = m.n_verts
        d_p, d_n, d_t = ctx.malloc(n * 12), ctx.malloc(n * 12), ctx.malloc(n * 16)
        ctx.lbs_skin_device(7102, d_pal.ptr, 32, 1, d_p.ptr, d_n.ptr, d_t.ptr)
        fyrox_amd.Context.allgather_skinned_all([ctx], n, [d_p.ptr], [d_n.ptr], [d_t.ptr])
        fyrox_amd.Context.allgather_skinned_all([ctx], n, [d_p.ptr], None, [d_t.ptr])     # any subset of the streams
        ctx.sync()
        ref = orc.lbs_skin(m.pos, m.weights, m.indices, pal, m.normal, m.tangent, threads=0)
        assert np.array_equal(d_p.download(np.float32, n * 3).reshape(n, 3), ref["pos"])
        assert np.array_equal(d_n.download(np.float32, n * 3).reshape(n, 3), ref["normal"])
        assert np.array_equal(d_t.download(np.float32, n * 4).reshape(n, 4), ref["tangent"])
        with pytest.raises(fyrox_amd.FyxError):
            fyrox_amd.Context.allgather_skinned_all([ctx], n, [0], None, None)            # a stream missing on a GPU
        with pytest.raises(fyrox_amd.FyxError):
            fyrox_amd.Context.allgather_skinned_all([ctx, other], n, [d_p.ptr, d_p.ptr], None, None)   # not the communicator's contexts
        with pytest.raises(fyrox_amd.FyxError):
            fyrox_amd.Context.allgather_skinned_all([other], n, [d_p.ptr], None, None)    # no communicator there
        ctx.comm_shutdown()
        for d in (d_pal, d_p, d_n, d_t):
            d.free()
        ctx.mesh_free(7102)
    finally:
        other.close()


# ---- fyx_lbs_skin_batch: many (mesh, palette) pairs, one launch -------------------------------------------------

def _batch_scene(ctx, base_id, specs, seed0):
    """specs: (n_verts, n_bones, n_instances, want) per job.  Uploads the meshes and returns per job
    (mesh, palette, device palette, outputs dict)."""
    scene = []
    for k, (nv, nb, ni, want) in enumerate(specs):
        m = synth.make_mesh(nv, nb, seed0 + k, coherent=bool(k % 2)) if nv else synth.make_mesh(0, nb, seed0 + k)
        pal = synth.make_palette(nb, seed0 + 100 + k, n_instances=ni)
        ctx.mesh_upload_soa(base_id + k, m.pos, m.weights, m.indices, m.normal, m.tangent)
        outs = {"pos": ctx.malloc(max(ni * nv * 12, 16)) if "pos" in want else None,
                "normal": ctx.malloc(max(ni * nv * 12, 16)) if "normal" in want else None,
                "tangent": ctx.malloc(max(ni * nv * 16, 16)) if "tangent" in want else None}
        scene.append((m, pal, ctx.to_device(pal), outs))
    return scene


def _batch_jobs(base_id, specs, scene):
    return [(base_id + k, dp.ptr, nb, ni, o["pos"].ptr if o["pos"] else 0, o["normal"].ptr if o["normal"] else 0,
             o["tangent"].ptr if o["tangent"] else 0)
            for k, ((nv, nb, ni, want), (m, pal, dp, o)) in enumerate(zip(specs, scene))]


def _check_batch(ctx, orc, specs, scene, exact):
    width = {"pos": 3, "normal": 3, "tangent": 4}
    for (nv, nb, ni, want), (m, pal, dp, o) in zip(specs, scene):
        if nv == 0:
            continue
        ref = oracle_skin(orc, m, pal, ni)
        for key in want:
            got = o[key].download(np.float32, ni * nv * width[key]).reshape(ni * nv, width[key])
            if exact:
                assert np.array_equal(got.view(np.uint32), ref[key].view(np.uint32)), f"{nv} verts / {nb} bones x {ni}: {key}"
            else:
                assert rel_err(got, ref[key]) <= REL_TOL


@pytest.mark.parametrize("exact", [1, 0])
@pytest.mark.parametrize("dyn", [1, 0], ids=["batch_dyn", "batch"])
def test_batch_of_many_meshes_is_bit_exact(ctx, orc, exact, dyn):
    """Ragged sizes (1 vertex, just under / over a unit, just over a workgroup's share), different bone counts,
    instanced jobs, an empty mesh, one job big enough to take the crowd launch -- all in one call; both forms of the batched
    kernel (lbs.dyn: units drawn from an LDS ticket, streams as buffer resources / lbs_skin's structure)."""
    ALL3 = ("pos", "normal", "tangent")
    specs = [(1, 4, 1, ALL3), (63, 8, 2, ALL3), (65, 64, 1, ALL3), (5000, 64, 1, ALL3), (4097, 256, 3, ALL3),
             (0, 16, 1, ALL3), (20_000, 96, 1, ALL3), (777, 33, 5, ALL3), (300, 12, 17, ALL3), (12_345, 200, 1, ALL3)]
    ctx.set_option("lbs.exact", exact)
    ctx.set_option("lbs.dyn", dyn)
    try:
        scene = _batch_scene(ctx, 7300, specs, synth.SEED_BASE + 300)
        ctx.lbs_skin_batch(_batch_jobs(7300, specs, scene))
        ctx.sync()
        _check_batch(ctx, orc, specs, scene, bool(exact))
    finally:
        ctx.set_option("lbs.exact", 1)


@pytest.mark.parametrize("dyn", [1, 0], ids=["batch_dyn", "batch"])
def test_batch_long_enough_for_every_wave_to_draw_again(ctx, orc, dyn):
    """A batch of ~15 000 units: a workgroup's range is ~15 units, so the waves of lbs_skin_batch_dyn go back to the ticket after their first two
    (tools/mutants.py: a ticket that skipped one unit of every segment survived the suite while no batch in it gave a workgroup more than
    eight).  Ragged meshes, two- and three-instance jobs (below the crowd launch's threshold), a workgroup range that crosses mesh boundaries
    more than once; every vertex against the oracle."""
    ALL3 = ("pos", "normal", "tangent")
    specs = [(60_001, 64, 1, ALL3), (90_000, 128, 1, ALL3), (33_333, 24, 3, ALL3), (120_007, 256, 1, ALL3), (70, 5, 1, ALL3), (45_000, 64, 2, ALL3),
             (200_000, 96, 1, ALL3), (1_000, 16, 1, ALL3), (80_129, 200, 1, ALL3)]
    ctx.set_option("lbs.dyn", dyn)
    scene = _batch_scene(ctx, 7350, specs, synth.SEED_BASE + 320)
    ctx.lbs_skin_batch(_batch_jobs(7350, specs, scene))
    ctx.sync()
    _check_batch(ctx, orc, specs, scene, True)
    for k, (m, pal, dp, o) in enumerate(scene):
        for b in list(o.values()) + [dp]:
            if b:
                b.free()
        ctx.mesh_free(7350 + k)


def test_batch_with_different_output_sets_and_repeated_calls(ctx, orc):
    """Jobs asking for different attribute sets are split into one launch per set; the same batch sent again re-uses the
    device table, a changed batch replaces it, and the single-job form still agrees."""
    specs = [(3000, 32, 1, ("pos",)), (3000, 32, 2, ("pos", "normal")), (900, 16, 1, ("tangent",)),
             (2500, 48, 1, ("pos", "normal", "tangent")), (70, 5, 3, ("normal", "tangent")), (4000, 64, 1, ("pos", "tangent"))]
    scene = _batch_scene(ctx, 7400, specs, synth.SEED_BASE + 340)
    jobs = _batch_jobs(7400, specs, scene)
    for rep in range(3):
        ctx.lbs_skin_batch(jobs)
    ctx.sync()
    _check_batch(ctx, orc, specs, scene, True)
    # another scene right behind it (other table), then the first again
    specs2 = [(1500, 20, 2, ("pos", "normal", "tangent")), (640, 64, 1, ("pos", "normal", "tangent"))]
    scene2 = _batch_scene(ctx, 7450, specs2, synth.SEED_BASE + 360)
    ctx.lbs_skin_batch(_batch_jobs(7450, specs2, scene2))
    ctx.lbs_skin_batch(jobs[::-1])
    ctx.sync()
    _check_batch(ctx, orc, specs2, scene2, True)
    _check_batch(ctx, orc, specs, scene, True)
    # and against the one-mesh entry point, bit for bit
    m, pal, dp, o = scene[3]
    one = ctx.malloc(2500 * 12)
    ctx.lbs_skin_device(7403, dp.ptr, 48, 1, one.ptr, 0, 0)
    ctx.sync()
    assert np.array_equal(one.download(np.uint32, 2500 * 3), o["pos"].download(np.uint32, 2500 * 3))


def test_batch_plan_is_reused_only_while_nothing_it_was_made_from_changed(ctx, orc):
    """The same job array again (a scene's frame) takes the previous call's plan and device tables.  What must still come out right:
    new palette CONTENTS behind the same pointers; a mesh uploaded again under the same id (other vertices, other device buffers);
    a launch option changed; another batch (its tables replace the cached ones) in between."""
    ALL3 = ("pos", "normal", "tangent")
    specs = [(3000, 32, 1, ALL3), (1200, 16, 2, ALL3), (65, 8, 1, ALL3), (4000, 64, 1, ALL3)]
    scene = _batch_scene(ctx, 7700, specs, synth.SEED_BASE + 420)
    jobs = _batch_jobs(7700, specs, scene)
    for _ in range(3):
        ctx.lbs_skin_batch(jobs)
    ctx.sync()
    _check_batch(ctx, orc, specs, scene, True)
    # 1. other palette values, same buffers
    scene1 = []
    for k, ((nv, nb, ni, want), (m, pal, dp, o)) in enumerate(zip(specs, scene)):
        pal2 = synth.make_palette(nb, synth.SEED_BASE + 900 + k, n_instances=ni)
        ctx.sync()
        dp.upload(pal2)
        scene1.append((m, pal2, dp, o))
    ctx.lbs_skin_batch(jobs)
    ctx.sync()
    _check_batch(ctx, orc, specs, scene1, True)
    # 2. mesh 7702 again with other vertices (same size): the cached plan holds the OLD buffers
    m2 = synth.make_mesh(65, 8, synth.SEED_BASE + 950)
    ctx.mesh_upload_soa(7702, m2.pos, m2.weights, m2.indices, m2.normal, m2.tangent)
    scene2 = list(scene1)
    scene2[2] = (m2, scene1[2][1], scene1[2][2], scene1[2][3])
    ctx.lbs_skin_batch(jobs)
    ctx.sync()
    _check_batch(ctx, orc, specs, scene2, True)
    # 3. another batch in between, and an option that changes the grid
    specs3 = [(500, 12, 1, ALL3)]
    scene3 = _batch_scene(ctx, 7750, specs3, synth.SEED_BASE + 470)
    ctx.lbs_skin_batch(_batch_jobs(7750, specs3, scene3))
    ctx.lbs_skin_batch(jobs)
    before = ctx.get_option("lbs.blocks_per_cu")
    ctx.set_option("lbs.blocks_per_cu", 1)
    try:
        ctx.lbs_skin_batch(jobs)
        ctx.lbs_skin_batch(jobs)
    finally:
        ctx.set_option("lbs.blocks_per_cu", before)
    ctx.sync()
    _check_batch(ctx, orc, specs3, scene3, True)
    _check_batch(ctx, orc, specs, scene2, True)


def test_batch_argument_errors_launch_nothing(ctx):
    specs = [(500, 16, 1, ("pos",)), (500, 16, 1, ("pos",))]
    scene = _batch_scene(ctx, 7500, specs, synth.SEED_BASE + 380)
    for _, _, _, o in scene:
        o["pos"].upload(np.full(500 * 3, 7.0, np.float32))
    jobs = _batch_jobs(7500, specs, scene)
    ctx.lbs_skin_batch([])                                   # an empty batch is fine
    bad = list(jobs)
    bad[1] = (9_999_999,) + jobs[1][1:]                       # unknown mesh in the LAST job
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.lbs_skin_batch(bad)
    assert e.value.code == fyrox_amd._native.FYX_ERR_UNKNOWN_ID
    bad[1] = jobs[1][:2] + (8,) + jobs[1][3:]                 # palette too small for the mesh's bone indices
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.lbs_skin_batch(bad)
    assert e.value.code == fyrox_amd._native.FYX_ERR_BONE_INDEX
    bad[1] = jobs[1][:1] + (0,) + jobs[1][2:]                 # null palette
    with pytest.raises(fyrox_amd.FyxError):
        ctx.lbs_skin_batch(bad)
    ctx.sync()
    for _, _, _, o in scene:                                  # the valid first job was not launched either
        assert np.all(o["pos"].download(np.float32, 500 * 3) == 7.0)


def _aos_expect(orc, m, pal, n_inst, layout_stride, offs, shapes=None):
    """Reference output vertex buffer bytes: input vertices with position / normal / tangent.xyz replaced."""
    if shapes:
        storage, plane, weights = shapes
        ref = oracle_skin_shapes(orc, m, pal, storage, plane, weights, n_inst)
    else:
        ref = oracle_skin(orc, m, pal, n_inst)
    src = m.to_animated_vertex_aos().reshape(m.n_verts, layout_stride)
    out = np.tile(src, (n_inst, 1)).copy()
    for off, key in zip(offs, ("pos", "normal", "tangent")):
        out[:, off:off + 12] = np.ascontiguousarray(ref[key][:, :3]).view(np.uint8).reshape(-1, 12)
    return out


def test_ex_batch_vertex_buffers_blend_shapes_and_plain_jobs_in_one_call(ctx, orc):
    """fyx_lbs_skin_ex_batch: vertex-buffer jobs without and with blend shapes (two launches), plain SoA jobs (the
    batched SoA launch), a blend-shapes-into-SoA job (its own launch) -- every output equals the oracle bit for bit
    and equals what fyx_lbs_skin_ex writes for the same job."""
    L = synth.ANIMATED_VERTEX
    offs = (L["off_pos"], L["off_normal"], L["off_tangent"])
    seed = synth.SEED_BASE + 400
    jobs, checks = [], []
    # (n_verts, n_bones, n_instances, n_shapes) vertex-buffer jobs
    for k, (nv, nb, ni, ns) in enumerate([(3000, 40, 1, 0), (65, 8, 3, 0), (10_000, 64, 2, 0), (2222, 30, 1, 2), (640, 16, 2, 3), (1, 4, 1, 0)]):
        mid = 7600 + k
        m = synth.make_mesh(nv, nb, seed + k, coherent=bool(k % 2))
        pal = synth.make_palette(nb, seed + 50 + k, n_instances=ni)
        upload(ctx, mid, m, aos=True)
        kw = dict(d_palette=ctx.to_device(pal).ptr, n_bones=nb, n_instances=ni)
        shapes = None
        if ns:
            storage, plane, w = synth.make_blend_shapes(nv, ns, seed + k)
            ctx.mesh_set_blend_shapes(mid, storage, ns, plane)
            weights = np.stack([w * np.float32(1.0 + 0.5 * i) for i in range(ni)])
            kw.update(d_blend_shape_weights=ctx.to_device(weights).ptr, n_blend_shapes=ns)
            shapes = (storage, plane, weights)
        buf = ctx.to_device(np.full(nv * ni * L["stride"] + 64, 0xEE, np.uint8))
        kw.update(d_out_vertices=buf.ptr, out_stride=0)
        jobs.append((mid, kw))
        checks.append(("aos", m, pal, ni, shapes, buf))
    # plain SoA jobs and one with blend shapes into SoA
    for k, (nv, nb, ni, ns) in enumerate([(4000, 32, 1, 0), (500, 12, 2, 0), (3000, 24, 1, 2)]):
        mid = 7650 + k
        m = synth.make_mesh(nv, nb, seed + 20 + k)
        pal = synth.make_palette(nb, seed + 70 + k, n_instances=ni)
        upload(ctx, mid, m)
        outs = (ctx.malloc(nv * ni * 12), ctx.malloc(nv * ni * 12), ctx.malloc(nv * ni * 16))
        kw = dict(d_palette=ctx.to_device(pal).ptr, n_bones=nb, n_instances=ni, d_out_pos=outs[0].ptr, d_out_normal=outs[1].ptr,
                  d_out_tangent=outs[2].ptr)
        shapes = None
        if ns:
            storage, plane, w = synth.make_blend_shapes(nv, ns, seed + 20 + k)
            ctx.mesh_set_blend_shapes(mid, storage, ns, plane)
            weights = w[None, :]
            kw.update(d_blend_shape_weights=ctx.to_device(weights).ptr, n_blend_shapes=ns)
            shapes = (storage, plane, weights)
        jobs.append((mid, kw))
        checks.append(("soa", m, pal, ni, shapes, outs))
    ctx.lbs_skin_ex_batch(jobs)
    ctx.lbs_skin_ex_batch(jobs)      # the same table again: no re-upload, same result
    ctx.sync()
    for (mid, kw), (kind, m, pal, ni, shapes, out) in zip(jobs, checks):
        nv = m.n_verts
        if kind == "aos":
            raw = out.download(np.uint8, nv * ni * L["stride"] + 64)
            assert np.all(raw[-64:] == 0xEE), "wrote past the last vertex"
            got = raw[:-64].reshape(nv * ni, L["stride"])
            assert np.array_equal(got, _aos_expect(orc, m, pal, ni, L["stride"], offs, shapes)), f"mesh {mid}"
            one = ctx.to_device(np.full(nv * ni * L["stride"] + 64, 0xEE, np.uint8))
            ctx.lbs_skin_ex(mid, **{**kw, "d_out_vertices": one.ptr})
            ctx.sync()
            assert np.array_equal(one.download(np.uint8, raw.size), raw), f"mesh {mid}: batch != single launch"
            one.free()
        else:
            ref = oracle_skin_shapes(orc, m, pal, *shapes, ni) if shapes else oracle_skin(orc, m, pal, ni)
            for o, key, wdt in zip(out, ("pos", "normal", "tangent"), (3, 3, 4)):
                got = o.download(np.float32, nv * ni * wdt).reshape(-1, wdt)
                assert np.array_equal(got.view(np.uint32), ref[key].view(np.uint32)), f"mesh {mid} {key}"


def test_ex_batch_validates_every_job_first(ctx):
    L = synth.ANIMATED_VERTEX
    m = synth.make_mesh(300, 8, 77)
    upload(ctx, 7700, m, aos=True)
    upload(ctx, 7701, m, aos=False)
    pal = ctx.to_device(synth.make_palette(8, 77))
    buf = ctx.to_device(np.full(300 * L["stride"], 0x11, np.uint8))
    good = (7700, dict(d_palette=pal.ptr, n_bones=8, d_out_vertices=buf.ptr, out_stride=0))
    bad = (7701, dict(d_palette=pal.ptr, n_bones=8, d_out_vertices=buf.ptr, out_stride=0))   # no interleaved upload
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.lbs_skin_ex_batch([good, bad])
    assert e.value.code == fyrox_amd._native.FYX_ERR_MISSING_ATTRIBUTE
    ctx.sync()
    assert np.all(buf.download(np.uint8, 300 * L["stride"]) == 0x11)      # the good job was not launched either
    ctx.lbs_skin_ex_batch([])


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("dyn", [1, 0], ids=["batch_dyn", "batch"])
def test_batch_of_hundreds_of_tiny_meshes_equals_per_mesh_launches(ctx, seed, dyn):
    """Workgroups whose unit range spans many segments (meshes of 1..200 vertices, so most segments are 1-3 units):
    both batched kernels against the per-mesh launches, bit for bit, guard bytes behind every output."""
    rng = np.random.default_rng(seed)
    ctx.set_option("lbs.dyn", dyn)      # (meshes this small never take lbs_skin_dyn on their own: the option picks the batched kernel's form)
    L = synth.ANIMATED_VERTEX
    n_jobs = 300
    jobs_soa, jobs_aos, singles = [], [], []
    for k in range(n_jobs):
        nv = int(rng.integers(1, 200)) if k % 17 else int(rng.integers(400, 3000))
        nb = int(rng.integers(1, 40))
        ni = int(rng.integers(1, 4))
        m = synth.make_mesh(nv, nb, 9000 + seed * 1000 + k, coherent=bool(k % 2))
        pal = ctx.to_device(synth.make_palette(nb, 9500 + k, n_instances=ni))
        mid = 20_000 + k
        upload(ctx, mid, m, aos=True)
        bytes_soa = [nv * ni * 12, nv * ni * 12, nv * ni * 16]
        a = [ctx.to_device(np.full(b + 32, 0xAB, np.uint8)) for b in bytes_soa]
        b_ = [ctx.to_device(np.full(b + 32, 0xAB, np.uint8)) for b in bytes_soa]
        va = ctx.to_device(np.full(nv * ni * L["stride"] + 32, 0xAB, np.uint8))
        vb = ctx.to_device(np.full(nv * ni * L["stride"] + 32, 0xAB, np.uint8))
        jobs_soa.append((mid, pal.ptr, nb, ni, a[0].ptr, a[1].ptr, a[2].ptr))
        jobs_aos.append((mid, dict(d_palette=pal.ptr, n_bones=nb, n_instances=ni, d_out_vertices=va.ptr, out_stride=0)))
        singles.append((mid, pal, nb, ni, a, b_, va, vb, bytes_soa, nv * ni * L["stride"]))
    ctx.lbs_skin_batch(jobs_soa)
    ctx.lbs_skin_ex_batch(jobs_aos)
    for mid, pal, nb, ni, a, b_, va, vb, bytes_soa, bytes_aos in singles:
        ctx.lbs_skin_device(mid, pal.ptr, nb, ni, b_[0].ptr, b_[1].ptr, b_[2].ptr)
        ctx.lbs_skin_ex(mid, pal.ptr, nb, ni, d_out_vertices=vb.ptr, out_stride=0)
    ctx.sync()
    for mid, pal, nb, ni, a, b_, va, vb, bytes_soa, bytes_aos in singles:
        for x, y, nbytes in zip(a, b_, bytes_soa):
            gx, gy = x.download(np.uint8, nbytes + 32), y.download(np.uint8, nbytes + 32)
            assert np.array_equal(gx, gy), f"mesh {mid}"
            assert np.all(gx[-32:] == 0xAB)
        gx, gy = va.download(np.uint8, bytes_aos + 32), vb.download(np.uint8, bytes_aos + 32)
        assert np.array_equal(gx, gy), f"mesh {mid} (vertex buffer)"
        assert np.all(gx[-32:] == 0xAB)
        for d in [pal, va, vb] + a + b_:
            d.free()
        ctx.mesh_free(mid)


@pytest.mark.gpu
def test_per_launch_timing_reports_every_launch_once(ctx):
    """Option lbs.timing: each fyx_lbs_skin_device launch carries its own start / stop events; fyx_debug_kernel_time returns
    the sum of the kernels' durations and their number since the last call and starts over.  Results are unaffected."""
    mesh = synth.make_mesh(30_000, 32, synth.SEED_BASE + 77)
    pal = synth.make_palette(32, synth.SEED_BASE + 77)
    ctx.mesh_upload_soa(9100, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    d_pal = ctx.to_device(pal)
    out = ctx.malloc(mesh.n_verts * 12 + 64)
    ctx.lbs_skin_device(9100, d_pal.ptr, 32, 1, out.ptr)
    ctx.sync()
    ref = out.download(np.uint32, mesh.n_verts * 3)
    ctx.set_option("lbs.timing", 1)
    try:
        ctx.kernel_time()
        for streams in (1, 2):
            ctx.set_option("lbs.streams", streams)
            for _ in range(7):
                ctx.lbs_skin_device(9100, d_pal.ptr, 32, 1, out.ptr)
            us, n = ctx.kernel_time()
            assert n == 7 and 7 * 0.5 < us < 7 * 500.0, (us, n)
        assert ctx.kernel_time() == (0.0, 0)
    finally:
        ctx.set_option("lbs.timing", 0)
        ctx.set_option("lbs.streams", 2)
    ctx.lbs_skin_device(9100, d_pal.ptr, 32, 1, out.ptr)
    ctx.sync()
    assert np.array_equal(out.download(np.uint32, mesh.n_verts * 3), ref)
    out.free(); d_pal.free(); ctx.mesh_free(9100)


@pytest.mark.gpu
@pytest.mark.parametrize("n_verts,n_bones,n_inst", [(5003, 64, 37), (300_017, 32, 2), (64, 4, 1), (10_000, 64, 1000)])
def test_instanced_aabb_on_the_device_matches_the_cpu_loop_per_instance(ctx, orc, n_verts, n_bones, n_inst):
    """fyx_skinned_aabb_device: Mesh::accurate_world_bounding_box (scene/mesh/mod.rs:470-526) of every instance of an
    instanced mesh in one call, palettes and boxes on the device: min / max of the oracle's skinned positions, bit for bit
    (single-slice and multi-slice launches, a crowd of C3's size on sampled instances)."""
    mesh = synth.make_mesh(n_verts, n_bones, synth.SEED_BASE + 91)
    pal = synth.make_palette(n_bones, synth.SEED_BASE + 91, n_instances=n_inst)
    ctx.mesh_upload_soa(9200, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    d_pal = ctx.to_device(pal)
    d_box = ctx.malloc(n_inst * 24)
    d_box.upload(np.full(n_inst * 6, 123.0, np.float32))
    ctx.skinned_aabb_device(9200, d_pal.ptr, n_bones, n_inst, d_box.ptr)
    ctx.sync()
    got = d_box.download(np.float32, n_inst * 6).reshape(n_inst, 6)
    pal = pal.reshape(n_inst, n_bones, 16)
    for i in sorted({0, min(1, n_inst - 1), n_inst // 2, n_inst - 1}):
        ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal[i], threads=0)["pos"]
        box = np.concatenate([ref.min(0), ref.max(0)])
        assert np.array_equal(got[i].view(np.uint32), box.view(np.uint32)), f"instance {i}"
    if n_inst == 1:   # and the host-pointer single-instance entry point gives the same box
        assert np.array_equal(ctx.skinned_aabb(9200, pal[0]), got[0])
    d_pal.free(); d_box.free(); ctx.mesh_free(9200)


@pytest.mark.gpu
def test_instanced_aabb_argument_errors_and_empty_mesh(ctx):
    mesh = synth.make_mesh(100, 8, synth.SEED_BASE + 92)
    ctx.mesh_upload_soa(9201, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    d_pal = ctx.to_device(synth.make_palette(8, synth.SEED_BASE + 92, n_instances=3))
    d_box = ctx.malloc(3 * 24)
    with pytest.raises(fyrox_amd.FyxError):
        ctx.skinned_aabb_device(9201, d_pal.ptr, 8, 3, 0)             # no output
    with pytest.raises(fyrox_amd.FyxError):
        ctx.skinned_aabb_device(9201, d_pal.ptr, 4, 3, d_box.ptr)     # the mesh references bone 7
    with pytest.raises(fyrox_amd.FyxError):
        ctx.skinned_aabb_device(424242, d_pal.ptr, 8, 3, d_box.ptr)   # unknown mesh
    ctx.mesh_upload_soa(9202, np.zeros((0, 3), np.float32), np.zeros((0, 4), np.float32), np.zeros((0, 4), np.uint8),
                        np.zeros((0, 3), np.float32), np.zeros((0, 4), np.float32))
    ctx.skinned_aabb_device(9202, d_pal.ptr, 8, 3, d_box.ptr)
    ctx.sync()
    got = d_box.download(np.float32, 18).reshape(3, 6)
    fmax = np.finfo(np.float32).max
    assert np.array_equal(got, np.tile(np.asarray([fmax] * 3 + [-fmax] * 3, np.float32), (3, 1)))   # AxisAlignedBoundingBox::default()
    d_pal.free(); d_box.free(); ctx.mesh_free(9201); ctx.mesh_free(9202)


@pytest.mark.gpu
@pytest.mark.parametrize("n_verts,n_bones,n_inst,projective", [(10_000, 64, 200, False), (4097, 256, 9, False), (1500, 32, 17, True)])
def test_lean_crowd_kernel_is_bit_identical(ctx, orc, n_verts, n_bones, n_inst, projective):
    """Option lbs.crowd_lean: the crowd kernel with the influences walked one by one (fewer registers) at two workgroups per
    CU -- the form that leaves room for the pose kernels of the next frame (anim.overlap).  Same operations in the same order:
    every byte equals the default crowd kernel's output, and the oracle's on sampled instances."""
    mesh = synth.make_mesh(n_verts, n_bones, synth.SEED_BASE + 95)
    pal = synth.make_palette(n_bones, synth.SEED_BASE + 95, n_instances=n_inst).reshape(n_inst, n_bones, 16).copy()
    if projective:
        pal[n_inst // 2, 3, 3] = 0.25      # one non-affine matrix: that instance takes the projective path
        pal[n_inst // 2, 3, 7] = -0.5
    ctx.mesh_upload_soa(9300, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
    d_pal = ctx.to_device(pal.reshape(-1, 16))
    nv = n_verts * n_inst
    outs = (ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 12 + 64), ctx.malloc(nv * 16 + 64))
    res = []
    try:
        for lean in (0, 1):
            ctx.set_option("lbs.crowd_lean", lean)
            ctx.set_option("lbs.crowd", 1)
            for b in outs:
                b.upload(np.zeros(16, np.uint32))
            ctx.lbs_skin_device(9300, d_pal.ptr, n_bones, n_inst, outs[0].ptr, outs[1].ptr, outs[2].ptr)
            ctx.join(); ctx.sync()
            res.append([b.download(np.uint32, nv * w) for b, w in zip(outs, (3, 3, 4))])
    finally:
        ctx.set_option("lbs.crowd_lean", 0)
        ctx.set_option("lbs.crowd", -1)
    for k in range(3):
        assert np.array_equal(res[0][k], res[1][k]), f"stream {k}: lean != default"
    for i in sorted({0, n_inst // 2, n_inst - 1}):
        ref = orc.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal[i], mesh.normal, mesh.tangent, threads=0)
        got = res[1][0].view(np.float32).reshape(n_inst, n_verts, 3)[i]
        assert np.array_equal(got.view(np.uint32), ref["pos"].view(np.uint32)), f"instance {i} vs oracle"
    for b in outs:
        b.free()
    d_pal.free(); ctx.mesh_free(9300)


def test_output_streams_are_separate_allocations(ctx, orc):
    """fyx_malloc_streams: the placement rule in code -- every output stream of a skinning launch is a device
    allocation of its own.  The pointers are distinct allocations (each can be freed on its own), a zero-size stream is NULL, a
    request that cannot be met leaves nothing behind, and a launch into the streams is the oracle's."""
    n_verts, n_bones = 70_001, 64
    bufs = ctx.malloc_streams([n_verts * 12 + 64, 0, n_verts * 16 + 64, n_verts * 12 + 64])
    assert bufs[1].ptr == 0 and len({b.ptr for b in bufs if b.ptr}) == 3
    m = synth.make_mesh(n_verts, n_bones, 4321)
    pal = synth.make_palette(n_bones, 4321)
    upload(ctx, 6, m)
    d_pal = ctx.to_device(pal)
    ctx.lbs_skin_device(6, d_pal.ptr, n_bones, 1, bufs[0].ptr, bufs[3].ptr, bufs[2].ptr)
    ctx.sync()
    ref = oracle_skin(orc, m, pal)
    got = {"pos": bufs[0].download(np.float32, n_verts * 3).reshape(-1, 3), "normal": bufs[3].download(np.float32, n_verts * 3).reshape(-1, 3),
           "tangent": bufs[2].download(np.float32, n_verts * 4).reshape(-1, 4)}
    assert_bit_exact(got, ref)
    bufs[0].free()          # one stream freed on its own: the others stay usable
    ctx.lbs_skin_device(6, d_pal.ptr, n_bones, 1, 0, bufs[3].ptr, bufs[2].ptr)
    ctx.sync()
    assert np.array_equal(bufs[3].download(np.float32, n_verts * 3).reshape(-1, 3), ref["normal"])
    with pytest.raises(fyrox_amd.FyxError) as e:
        ctx.malloc_streams([1 << 20, 1 << 50, 1 << 20])      # the second cannot be met: the first is released again, nothing is returned
    assert e.value.code in (_native.FYX_ERR_OOM, _native.FYX_ERR_HIP)
    for b in bufs[1:]:
        b.free()
    d_pal.free()
    ctx.mesh_free(6)
