"""fyrox-impl scene side restated (test infrastructure, see __init__): Transform::calculate_local_transform, the
hierarchy walk, the bone palette and linear-blend skinning.  Matrices are numpy float32 (4, 4) arrays [row, col];
flat 16-float rows handed in / out are column-major, the bytes of nalgebra's Matrix4<f32>."""
import numpy as np

from . import na
from .na import F, ONE, ZERO


def local_matrix(t) -> np.ndarray:
    """Transform::calculate_local_transform (scene/transform.rs:421-540), expression for expression.
    `t`: dict of float32 tuples (position, rotation ijkw, scale, pre_rotation ijkw, post_rotation_matrix (9, column-major
    as Matrix3 stores it), rotation_offset, rotation_pivot, scaling_offset, scaling_pivot)."""
    por = t["post_rotation_matrix"]                       # por[i]: linear (column-major) index into the Matrix3
    prm = na.q_to_rotation_matrix(t["pre_rotation"])
    rm = na.q_to_rotation_matrix(t["rotation"])
    pr = [prm[i % 3, i // 3] for i in range(9)]           # pr[i], r[i]: the same linear indexing
    r = [rm[i % 3, i // 3] for i in range(9)]
    sx, sy, sz = t["scale"]
    tx, ty, tz = t["position"]
    rpx, rpy, rpz = t["rotation_pivot"]
    rox, roy, roz = t["rotation_offset"]
    spx, spy, spz = t["scaling_pivot"]
    sox, soy, soz = t["scaling_offset"]
    a0 = pr[0] * r[0] + pr[3] * r[1] + pr[6] * r[2]
    a1 = pr[1] * r[0] + pr[4] * r[1] + pr[7] * r[2]
    a2 = pr[2] * r[0] + pr[5] * r[1] + pr[8] * r[2]
    a3 = pr[0] * r[3] + pr[3] * r[4] + pr[6] * r[5]
    a4 = pr[1] * r[3] + pr[4] * r[4] + pr[7] * r[5]
    a5 = pr[2] * r[3] + pr[5] * r[4] + pr[8] * r[5]
    a6 = pr[0] * r[6] + pr[3] * r[7] + pr[6] * r[8]
    a7 = pr[1] * r[6] + pr[4] * r[7] + pr[7] * r[8]
    a8 = pr[2] * r[6] + pr[5] * r[7] + pr[8] * r[8]
    f0 = por[0] * a0 + por[1] * a3 + por[2] * a6
    f1 = por[0] * a1 + por[1] * a4 + por[2] * a7
    f2 = por[0] * a2 + por[1] * a5 + por[2] * a8
    f3 = por[3] * a0 + por[4] * a3 + por[5] * a6
    f4 = por[3] * a1 + por[4] * a4 + por[5] * a7
    f5 = por[3] * a2 + por[4] * a5 + por[5] * a8
    f6 = por[6] * a0 + por[7] * a3 + por[8] * a6
    f7 = por[6] * a1 + por[7] * a4 + por[8] * a7
    f8 = por[6] * a2 + por[7] * a5 + por[8] * a8
    m0, m1, m2 = sx * f0, sx * f1, sx * f2
    m4, m5, m6 = sy * f3, sy * f4, sy * f5
    m8, m9, m10 = sz * f6, sz * f7, sz * f8
    k0, k1, k2 = spx * f0, spy * f3, spz * f6
    m12 = rox + rpx + tx - rpx * f0 - rpy * f3 - rpz * f6 + sox * f0 + k0 + soy * f3 + k1 + soz * f6 + k2 - sx * k0 - sy * k1 - sz * k2
    k3, k4, k5 = spx * f1, spy * f4, spz * f7
    m13 = roy + rpy + ty - rpx * f1 - rpy * f4 - rpz * f7 + sox * f1 + k3 + soy * f4 + k4 + soz * f7 + k5 - sx * k3 - sy * k4 - sz * k5
    k6, k7, k8 = spx * f2, spy * f5, spz * f8
    m14 = roz + rpz + tz - rpx * f2 - rpy * f5 - rpz * f8 + sox * f2 + k6 + soy * f5 + k7 + soz * f8 + k8 - sx * k6 - sy * k7 - sz * k8
    # Matrix4::new takes its sixteen arguments row by row: (m0, m4, m8, m12, m1, ...)
    return np.array([[m0, m4, m8, m12], [m1, m5, m9, m13], [m2, m6, m10, m14], [ZERO, ZERO, ZERO, ONE]], np.float32)


def _from_flat(row16):
    return np.asarray(row16, np.float32).reshape(4, 4).T.copy()


def _to_flat(m):
    return m.T.reshape(16).copy()


def global_matrices(local_flat: np.ndarray, parent) -> np.ndarray:
    """Graph::update_global_transform_recursively (scene/graph/mod.rs:1199-1241): global = parent.global * local, the
    root's parent is the identity.  parent[i] < i."""
    n = len(parent)
    g = [None] * n
    for i in range(n):
        pg = g[parent[i]] if parent[i] >= 0 else na.mat_identity(4)
        g[i] = na.mat_mul(pg, _from_flat(local_flat[i]))
    return np.stack([_to_flat(m) for m in g]) if n else np.zeros((0, 16), np.float32)


def palette(global_flat: np.ndarray, inv_bind_flat: np.ndarray, bone_nodes) -> np.ndarray:
    """scene/mesh/mod.rs:781-793: global * inv_bind per bone, the identity for an invalid handle"""
    out = np.empty((len(bone_nodes), 16), np.float32)
    for b, node in enumerate(bone_nodes):
        if node < 0:
            out[b] = np.eye(4, dtype=np.float32).reshape(16)
        else:
            out[b] = _to_flat(na.mat_mul(_from_flat(global_flat[node]), _from_flat(inv_bind_flat[node])))
    return out


def lbs_skin(pos, weights, indices, palette_flat, normal=None, tangent=None) -> dict:
    """Linear-blend skinning, vectorised over the vertices (numpy float32 arrays: every elementwise operation rounds to
    f32, nothing fuses).
      position: sum over the four influences of M.transform_point(p).scale(w)       (scene/mesh/mod.rs:501-522)
      normal / tangent.xyz: sum of (mat3(M) * v) * w, tangent.w untouched           (standard.shader:187-200)
    `indices`: (n, 4) uint8 or packed uint32 (n,), influence 0 in the low byte."""
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    w = np.ascontiguousarray(weights, np.float32).reshape(-1, 4)
    idx = np.ascontiguousarray(indices)
    if idx.dtype == np.uint32 and idx.ndim == 1:
        idx = idx.view(np.uint8).reshape(-1, 4)
    idx = idx.reshape(-1, 4).astype(np.int64)
    pal = np.ascontiguousarray(palette_flat, np.float32).reshape(-1, 16)
    M = pal.reshape(-1, 4, 4).transpose(0, 2, 1)          # [bone][row][col]
    n = pos.shape[0]
    out = {"pos": np.zeros((n, 3), np.float32)}
    if normal is not None:
        out["normal"] = np.zeros((n, 3), np.float32)
    if tangent is not None:
        out["tangent"] = np.zeros((n, 4), np.float32)
        out["tangent"][:, 3] = np.ascontiguousarray(tangent, np.float32).reshape(-1, 4)[:, 3]
    px, py, pz = pos[:, 0], pos[:, 1], pos[:, 2]
    for k in range(4):
        m = M[idx[:, k]]                                  # (n, 4, 4)
        wk = w[:, k]
        # transform_point: M3 p (column after column), + t, then / n when n != 0
        lin = [(m[:, r, 0] * px + m[:, r, 1] * py) + m[:, r, 2] * pz for r in range(3)]
        res = [lin[r] + m[:, r, 3] for r in range(3)]
        nn = ((m[:, 3, 0] * px + m[:, 3, 1] * py) + m[:, 3, 2] * pz) + m[:, 3, 3]
        nz = nn != 0
        with np.errstate(all="ignore"):
            res = [np.where(nz, x / np.where(nz, nn, np.float32(1.0)), x) for x in res]
            for r in range(3):
                out["pos"][:, r] = out["pos"][:, r] + res[r] * wk
            for key, src in (("normal", normal), ("tangent", tangent)):
                if src is None:
                    continue
                v = np.ascontiguousarray(src, np.float32).reshape(n, -1)
                vx, vy, vz = v[:, 0], v[:, 1], v[:, 2]
                for r in range(3):
                    out[key][:, r] = out[key][:, r] + ((m[:, r, 0] * vx + m[:, r, 1] * vy) + m[:, r, 2] * vz) * wk
    return out
