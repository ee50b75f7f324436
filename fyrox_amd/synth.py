"""Deterministic synthetic inputs for the skinning path.

Pure host-side data generation (numpy): splitmix64 counter streams, seed 0x5EED0000 + config#.
Used by tests, bench.py and smoke(); it computes nothing on the hot path.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass
from typing import Optional

import numpy as np

SEED_BASE = 0x5EED0000
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """First n outputs of splitmix64(seed) as uint64 (vectorised: output i depends only on i)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (np.arange(1, n + 1, dtype=np.uint64) * _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _sub(seed: int, name: str) -> int:
    return (seed * 0x100000001B3 + zlib.crc32(name.encode())) & 0xFFFFFFFFFFFFFFFF


def uniform(seed: int, name: str, n: int) -> np.ndarray:
    """n doubles in [0, 1)."""
    return (splitmix64(_sub(seed, name), n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed: int, name: str, n: int) -> np.ndarray:
    u1 = 1.0 - uniform(seed, name + ".u1", n)
    u2 = uniform(seed, name + ".u2", n)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


@dataclass
class SkinnedMesh:
    pos: np.ndarray       # (N,3) f32
    normal: np.ndarray    # (N,3) f32
    tangent: np.ndarray   # (N,4) f32, w = +-1
    weights: np.ndarray   # (N,4) f32
    indices: np.ndarray   # (N,4) u8
    n_bones: int

    @property
    def n_verts(self) -> int:
        return self.pos.shape[0]

    def to_animated_vertex_aos(self) -> np.ndarray:
        """Bytes of `[AnimatedVertex]` (fyrox-impl/src/scene/mesh/vertex.rs:139-155): 68-byte stride,
        position 0, tex_coord 12, normal 20, tangent 32, bone_weights 48, bone_indices 64."""
        n = self.n_verts
        aos = np.zeros((n, 68), np.uint8)
        aos[:, 0:12] = self.pos.view(np.uint8).reshape(n, 12)
        uv = (self.pos[:, :2] * 0.5 + 0.5).astype(np.float32)
        aos[:, 12:20] = np.ascontiguousarray(uv).view(np.uint8).reshape(n, 8)
        aos[:, 20:32] = self.normal.view(np.uint8).reshape(n, 12)
        aos[:, 32:48] = self.tangent.view(np.uint8).reshape(n, 16)
        aos[:, 48:64] = self.weights.view(np.uint8).reshape(n, 16)
        aos[:, 64:68] = self.indices
        return aos.reshape(-1)


ANIMATED_VERTEX = dict(stride=68, off_pos=0, off_normal=20, off_tangent=32, off_weights=48, off_indices=64)


def make_mesh(n_verts: int, n_bones: int, seed: int, coherent: bool = True) -> SkinnedMesh:
    n = n_verts
    pos = (uniform(seed, "pos", 3 * n) * 2.0 - 1.0).reshape(n, 3).astype(np.float32)
    nrm = normal(seed, "nrm", 3 * n).reshape(n, 3)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
    aux = normal(seed, "tan", 3 * n).reshape(n, 3)
    tan = np.cross(nrm, aux)
    tan /= np.maximum(np.linalg.norm(tan, axis=1, keepdims=True), 1e-12)
    sign = np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
    tangent = np.concatenate([tan, sign[:, None]], axis=1).astype(np.float32)
    nrm = nrm.astype(np.float32)

    v = np.arange(n, dtype=np.int64)
    if coherent:
        base = (v * n_bones) // max(n, 1)
        off = 1 + (splitmix64(_sub(seed, "idx.off"), 3 * n) % np.uint64(3)).astype(np.int64).reshape(n, 3)
        idx = np.stack([base, base + off[:, 0], base - off[:, 1], base + off[:, 0] + off[:, 2]], axis=1)
    else:
        base = (splitmix64(_sub(seed, "idx.base"), n) % np.uint64(n_bones)).astype(np.int64)
        max_step = max(1, (n_bones - 1) // 3)
        step = 1 + (splitmix64(_sub(seed, "idx.step"), n) % np.uint64(max_step)).astype(np.int64)
        idx = np.stack([base, base + step, base + 2 * step, base + 3 * step], axis=1)
    idx = np.mod(idx, n_bones).astype(np.uint8)

    e = -np.log(1.0 - uniform(seed, "w", 4 * n)).reshape(n, 4)
    e = -np.sort(-e, axis=1)  # descending
    kind = uniform(seed, "w.kind", n)
    cnt = np.where(kind < 0.7, 4, 1 + (splitmix64(_sub(seed, "w.cnt"), n) % np.uint64(3)).astype(np.int64))
    mask = np.arange(4)[None, :] < cnt[:, None]
    e = np.where(mask, e, 0.0)
    w = e.astype(np.float32)
    w = w / w.sum(axis=1, keepdims=True, dtype=np.float32)
    return SkinnedMesh(pos, nrm, tangent, w.astype(np.float32), idx, n_bones)


def make_blend_shapes(n_verts: int, n_shapes: int, seed: int, density: float = 0.35):
    """The RGB16F volume BlendShapesContainer::from_lists packs (fyrox-impl/src/scene/mesh/surface.rs:116-217):
    width = min(n_verts, 512), height = ceil(n_verts / width), per shape one plane of width*height records of
    nine f16 {position, normal, tangent offset}; vertices a shape does not touch stay zero (the maps are
    sparse).  Returns (storage uint16 [n_shapes, plane, 9], plane_vertices, weights f32 [n_shapes]) with weights
    already divided by 100 as Mesh::collect_render_data does (scene/mesh/mod.rs:794-798)."""
    width = max(min(n_verts, 512), 1)
    height = -(-n_verts // width) if n_verts else 0
    plane = width * height
    st = np.zeros((n_shapes, plane, 9), np.float16)
    for sidx in range(n_shapes):
        tag = f"shape{sidx}"
        touched = uniform(seed, tag + ".mask", n_verts) < density
        off = ((uniform(seed, tag + ".off", n_verts * 9) - 0.5) * 0.25).reshape(n_verts, 9).astype(np.float16)
        off[~touched] = 0
        st[sidx, :n_verts] = off
    bits = st.view(np.uint16)
    if n_shapes and n_verts > 8:   # a subnormal, a negative zero and the largest finite half, to pin the f16 decode
        bits[0, 1, 0] = 0x0001
        bits[0, 2, 3] = 0x8000
        bits[0, 3, 6] = 0x7BFF
        bits[0, 4, 1] = 0x83FF
    weights = (np.float32(100.0) * (uniform(seed, "shape.w", n_shapes) * 1.2 - 0.1).astype(np.float32)) / np.float32(100.0)
    return bits.copy(), plane, weights.astype(np.float32)


def quat_to_mat3(q: np.ndarray) -> np.ndarray:
    """(n,4) quaternions (i,j,k,w), float64 -> (n,3,3) rotation matrices."""
    i, j, k, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    m = np.empty((q.shape[0], 3, 3))
    m[:, 0, 0] = w * w + i * i - j * j - k * k
    m[:, 0, 1] = 2 * (i * j - w * k)
    m[:, 0, 2] = 2 * (w * j + i * k)
    m[:, 1, 0] = 2 * (w * k + i * j)
    m[:, 1, 1] = w * w - i * i + j * j - k * k
    m[:, 1, 2] = 2 * (j * k - w * i)
    m[:, 2, 0] = 2 * (i * k - w * j)
    m[:, 2, 1] = 2 * (w * i + j * k)
    m[:, 2, 2] = w * w - i * i - j * j + k * k
    return m


def random_rigid(seed: int, name: str, n: int, t_range: float = 2.0) -> np.ndarray:
    """n rigid transforms as column-major mat4 rows (n,16) f32; last row exactly (0,0,0,1)."""
    q = normal(seed, name + ".q", 4 * n).reshape(n, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r = quat_to_mat3(q)
    t = (uniform(seed, name + ".t", 3 * n) * 2.0 - 1.0).reshape(n, 3) * t_range
    m = np.zeros((n, 4, 4))
    m[:, :3, :3] = r
    m[:, :3, 3] = t
    m[:, 3, 3] = 1.0
    return np.ascontiguousarray(m.transpose(0, 2, 1)).reshape(n, 16).astype(np.float32)  # column-major


def rigid_inverse(m_cm: np.ndarray) -> np.ndarray:
    m = m_cm.astype(np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)
    r = m[:, :3, :3]
    t = m[:, :3, 3]
    inv = np.zeros_like(m)
    inv[:, :3, :3] = r.transpose(0, 2, 1)
    inv[:, :3, 3] = -np.einsum("nij,nj->ni", r.transpose(0, 2, 1), t)
    inv[:, 3, 3] = 1.0
    return np.ascontiguousarray(inv.transpose(0, 2, 1)).reshape(-1, 16).astype(np.float32)


def mat4_mul_f32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(n,16)x(n,16) column-major mat4 product in f32 with nalgebra's operation order
    (y = a_col0*b_0j; y = a_colk*b_kj + y).  numpy f32 ops are single IEEE operations, so this is
    bit-identical to the reference order; used only to build *inputs* (bone palettes)."""
    a = a.reshape(-1, 4, 4)  # [n, col, row]
    b = b.reshape(-1, 4, 4)
    out = np.empty_like(a)
    for j in range(4):
        y = a[:, 0, :] * b[:, j, 0:1]
        for k in range(1, 4):
            y = a[:, k, :] * b[:, j, k:k + 1] + y
        out[:, j, :] = y
    return out.reshape(-1, 16)


def make_bone_transforms(n_bones: int, seed: int, instance: int = 0):
    """(global, inv_bind) per SURVEY 8(d): random rigid global pose and the inverse of a random
    rigid bind pose, so palette = global*inv_bind has last row exactly (0,0,0,1)."""
    g = random_rigid(seed, f"global.{instance}", n_bones)
    bind = random_rigid(seed, "bind", n_bones)
    return g, rigid_inverse(bind)


def make_palette(n_bones: int, seed: int, n_instances: int = 1) -> np.ndarray:
    pals = []
    for i in range(n_instances):
        g, ib = make_bone_transforms(n_bones, seed, i)
        pals.append(mat4_mul_f32(g, ib))
    return np.concatenate(pals, axis=0)


# ---------------------------------------------------------------------------------------------
# Rigs, clips and machines: host-side input generation only.
# ---------------------------------------------------------------------------------------------

def make_rig(n_bones: int, seed: int, chain_depth: int = 8, exotic: bool = False):
    """A skeleton of n_bones nodes: chains of `chain_depth` bones hanging off node 0.  Local
    transforms are random rigid (unit scale); inv_bind is the f64 inverse of the bind-pose global,
    rounded to f32.  exotic=True also fills pre-rotation, post-rotation, pivots and offsets so the
    whole Transform::calculate_local_transform expression is exercised."""
    from .anim import Rig, Transform
    parent = np.full(n_bones, -1, np.int32)
    for i in range(1, n_bones):
        parent[i] = 0 if (i - 1) % chain_depth == 0 else i - 1
    q = normal(seed, "rig.q", 4 * n_bones).reshape(n_bones, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = (uniform(seed, "rig.t", 3 * n_bones) - 0.5).reshape(n_bones, 3)
    qp = normal(seed, "rig.qpre", 4 * n_bones).reshape(n_bones, 4)
    qp /= np.linalg.norm(qp, axis=1, keepdims=True)
    qo = normal(seed, "rig.qpost", 4 * n_bones).reshape(n_bones, 4)
    qo /= np.linalg.norm(qo, axis=1, keepdims=True)
    piv = (uniform(seed, "rig.piv", 12 * n_bones) - 0.5).reshape(n_bones, 4, 3) * 0.2
    transforms = []
    for i in range(n_bones):
        tr = Transform.identity()
        tr.local_position[:] = t[i].astype(np.float32).tolist()
        tr.local_rotation[:] = q[i].astype(np.float32).tolist()
        if exotic and i % 3 == 1:
            tr.pre_rotation[:] = qp[i].astype(np.float32).tolist()
            # cached inverse of the post-rotation matrix (column-major 3x3), as set_post_rotation stores
            inv = quat_to_mat3(qo[i:i + 1])[0].T
            tr.post_rotation_matrix[:] = np.ascontiguousarray(inv.T).reshape(9).astype(np.float32).tolist()
            tr.rotation_offset[:] = piv[i, 0].astype(np.float32).tolist()
            tr.rotation_pivot[:] = piv[i, 1].astype(np.float32).tolist()
            tr.scaling_offset[:] = piv[i, 2].astype(np.float32).tolist()
            tr.scaling_pivot[:] = piv[i, 3].astype(np.float32).tolist()
        transforms.append(tr)
    # bind-pose globals in f64 (plain rigid chain; good enough for an inverse bind matrix input)
    loc = np.zeros((n_bones, 4, 4))
    loc[:, :3, :3] = quat_to_mat3(q.astype(np.float32).astype(np.float64))
    loc[:, :3, 3] = t.astype(np.float32)
    loc[:, 3, 3] = 1.0
    glob = np.zeros_like(loc)
    for i in range(n_bones):
        glob[i] = loc[i] if parent[i] < 0 else glob[parent[i]] @ loc[i]
    inv_bind = np.linalg.inv(glob)
    inv_bind[:, 3, :] = (0.0, 0.0, 0.0, 1.0)
    inv_bind_cm = np.ascontiguousarray(inv_bind.transpose(0, 2, 1)).reshape(n_bones, 16).astype(np.float32)
    return Rig(parent=parent, transforms=transforms, inv_bind=inv_bind_cm)


def make_clip(n_bones: int, seed: int, clip: int = 0, n_keys: int = 31, fps: float = 30.0,
              key_kind: int = 1, euler_every: int = 2):
    """AnimationTracksData with 3 tracks per bone (Position Vec3, Rotation, Scale Vec3), n_keys keys
    at 1/fps spacing.  Rotation tracks are UnitQuaternionEuler (FBX-like) for bones with
    i % euler_every == euler_every - 1 and UnitQuaternion (glTF-like) otherwise; a huge euler_every means none.  Returns (tracks_data, track_target)."""
    from . import anim as A
    tag = f"clip{clip}"
    nk = n_keys
    times = (np.arange(nk, dtype=np.float64) / fps).astype(np.float32)
    pos = ((uniform(seed, tag + ".pos", n_bones * nk * 3) - 0.5) * 0.6).reshape(n_bones, nk, 3)
    scl = (1.0 + (uniform(seed, tag + ".scl", n_bones * nk * 3) - 0.5) * 0.1).reshape(n_bones, nk, 3)
    quat = normal(seed, tag + ".q", n_bones * nk * 4).reshape(n_bones, nk, 4)
    # low-frequency drift so neighbouring keys are similar but not equal
    quat = np.cumsum(quat * 0.15, axis=1) + normal(seed, tag + ".q0", n_bones * 4).reshape(n_bones, 1, 4)
    quat /= np.linalg.norm(quat, axis=2, keepdims=True)
    eul = np.cumsum(normal(seed, tag + ".e", n_bones * nk * 3).reshape(n_bones, nk, 3) * 0.2, axis=1)
    tan = normal(seed, tag + ".tan", n_bones * nk * 10 * 2).reshape(n_bones, nk, 10, 2) * 0.5

    def curve(vals, b, comp):
        return A.Curve([A.CurveKey(float(times[k]), float(np.float32(vals[k])), key_kind,
                                   float(np.float32(tan[b, k, comp, 0])), float(np.float32(tan[b, k, comp, 1])))
                        for k in range(nk)])

    tracks, target = [], []
    for b in range(n_bones):
        tracks.append(A.Track(A.BIND_POSITION, A.KIND_VEC3, [curve(pos[b, :, c], b, c) for c in range(3)]))
        if euler_every <= 0 or b % euler_every != euler_every - 1:
            tracks.append(A.Track(A.BIND_ROTATION, A.KIND_QUAT, [curve(quat[b, :, c], b, 3 + c) for c in range(4)]))
        else:
            tracks.append(A.Track(A.BIND_ROTATION, A.KIND_QUAT_EULER, [curve(eul[b, :, c], b, 3 + c) for c in range(3)]))
        tracks.append(A.Track(A.BIND_SCALE, A.KIND_VEC3, [curve(scl[b, :, c], b, 7 + c) for c in range(3)]))
        target += [b, b, b]
    return A.AnimationTracksData(tracks), np.asarray(target, np.int32)


def make_c5_machine():
    """BASELINE config C5: one layer, one state whose root is BlendAnimations over four PlayAnimation
    nodes with constant weights (-, 0.5, 0.25, 0.75); animations 0..3."""
    from . import anim as A
    nodes = [A.PlayAnimation(a) for a in range(4)]
    nodes.append(A.BlendAnimations([A.BlendPose(0, 1.0), A.BlendPose(1, 0.5), A.BlendPose(2, 0.25), A.BlendPose(3, 0.75)]))
    return A.Machine(parameters=[], layers=[A.MachineLayer(nodes=nodes, states=[A.State(root=4)])])
