"""nalgebra 0.35 leaves used by the path, restated on numpy float32 scalars (test infrastructure, see __init__).

Every function names the nalgebra source file and item it restates (from knowledge of the published source; the crate
is not vendored, so no line numbers) and the reference call site that reaches it.

| leaf                         | nalgebra item                                              | reached from (reference)                      |
|------------------------------|------------------------------------------------------------|-----------------------------------------------|
| dot2 / dot3 / dot4           | base/blas.rs  Matrix::dotx (2-, 3-, 4-element special cases) | value.rs:449-451, blendspace.rs:352-354, lib.rs:301-305 |
| mat_mul (4x4, 3x3 x vec)     | base/blas.rs  gemm -> gemv per column -> axcpy             | graph/mod.rs:1216, mesh/mod.rs:497, :787-788  |
| transform_point              | base/cg.rs    Matrix::transform_point                      | mesh/mod.rs:514-517                           |
| vlerp                        | base/matrix.rs Matrix::lerp = self * (1 - t) + rhs * t     | value.rs:224-226, lib.rs:341                  |
| q_lerp / q_nlerp             | geometry/quaternion.rs Quaternion::lerp, UnitQuaternion::nlerp | value.rs:453                              |
| norm / normalize             | base/norm.rs  norm_squared = dotc per column, sqrt; unscale | container.rs:277-279 (from_quaternion), nlerp |
| q_mul                        | geometry/quaternion_ops.rs  Quaternion * Quaternion         | fyrox-math lib.rs:733-739, lib.rs:616-636     |
| q_from_axis_angle            | geometry/quaternion_construction.rs from_axis_angle         | fyrox-math lib.rs:729-731                     |
| q_to_rotation_matrix         | geometry/quaternion.rs UnitQuaternion::to_rotation_matrix   | transform.rs:424-425                          |
| q_inverse                    | geometry/quaternion.rs UnitQuaternion::inverse = conjugate  | lib.rs:618, :630                              |
"""
import ctypes
import ctypes.util

import numpy as np

F = np.float32
ZERO, ONE, TWO = F(0.0), F(1.0), F(2.0)

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.sinf.restype = ctypes.c_float
_libm.sinf.argtypes = [ctypes.c_float]
_libm.cosf.restype = ctypes.c_float
_libm.cosf.argtypes = [ctypes.c_float]


def sin_cos(x):
    """f32::sin_cos: Rust lowers it to libm's sinf / cosf."""
    return F(_libm.sinf(float(x))), F(_libm.cosf(float(x)))


def sqrt(x):
    return F(np.sqrt(F(x)))      # IEEE: correctly rounded in every implementation


# ---- dot products: Matrix::dotx special cases for small vectors -----------------------------------------------------
def dot2(a, b):
    return a[0] * b[0] + a[1] * b[1]


def dot3(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def dot4(a, b):
    # 4 elements: two accumulators, (a0 b0 + a2 b2) + (a1 b1 + a3 b3)
    p0, p1, p2, p3 = a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]
    return (p0 + p2) + (p1 + p3)


# ---- matrices: numpy (n, n) float32 arrays indexed [row, col] --------------------------------------------------------
def mat_identity(n=4):
    return np.eye(n, dtype=np.float32)


def mat_vec(m, v):
    """gemv with beta = 0: the first column initialises (col_0 * v_0), every further column accumulates in order."""
    rows, cols = m.shape
    out = [m[r, 0] * v[0] for r in range(rows)]
    for c in range(1, cols):
        for r in range(rows):
            out[r] = out[r] + m[r, c] * v[c]
    return out


def mat_mul(a, b):
    """Matrix * Matrix = gemm: result column j = gemv(a, b column j)."""
    n = a.shape[0]
    out = np.empty((n, b.shape[1]), np.float32)
    for j in range(b.shape[1]):
        col = mat_vec(a, [b[k, j] for k in range(b.shape[0])])
        for r in range(n):
            out[r, j] = col[r]
    return out


def transform_point(m, p):
    """Matrix4::transform_point: (M3 p + t) / n with n = row3 . p + m33, the division skipped when n == 0."""
    lin = mat_vec(m[0:3, 0:3], p)
    res = [lin[r] + m[r, 3] for r in range(3)]
    n = dot3([m[3, 0], m[3, 1], m[3, 2]], p) + m[3, 3]
    if n != ZERO:
        res = [x / n for x in res]
    return res


# ---- vectors ----------------------------------------------------------------------------------------------------------
def vlerp(a, b, t):
    """Matrix::lerp: self * (1 - t) + rhs * t (NOT a + (b - a) t)."""
    omt = ONE - t
    return tuple(x * omt + y * t for x, y in zip(a, b))


def vsub(a, b):
    return tuple(x - y for x, y in zip(a, b))


def vadd(a, b):
    return tuple(x + y for x, y in zip(a, b))


def vscale(a, s):
    return tuple(x * s for x in a)


def norm(v):
    """norm = sqrt(norm_squared), norm_squared = column . column (dotx special cases)."""
    d = {2: dot2, 3: dot3, 4: dot4}[len(v)](v, v)
    return sqrt(ZERO + d)


# ---- quaternions: tuples (i, j, k, w), nalgebra's storage order ---------------------------------------------------------
Q_IDENTITY = (ZERO, ZERO, ZERO, ONE)


def q_mul(a, b):
    """Hamilton product, term order of Quaternion * Quaternion."""
    ai, aj, ak, aw = a
    bi, bj, bk, bw = b
    w = aw * bw - ai * bi - aj * bj - ak * bk
    i = aw * bi + ai * bw + aj * bk - ak * bj
    j = aw * bj - ai * bk + aj * bw + ak * bi
    k = aw * bk + ai * bj - aj * bi + ak * bw
    return (i, j, k, w)


def q_from_axis_angle(axis, angle):
    """(axis * sin(angle / 2), cos(angle / 2)); axis is a unit basis vector here."""
    s, c = sin_cos(angle / TWO)
    return (axis[0] * s, axis[1] * s, axis[2] * s, c)


def q_normalize(q):
    """Unit::new_normalize / normalize_mut: unscale by the norm (a division per component)."""
    n = norm(q)
    return tuple(x / n for x in q)


def q_neg(q):
    return tuple(-x for x in q)


def q_inverse(q):
    return (-q[0], -q[1], -q[2], q[3])


def q_nlerp(a, b, t):
    """UnitQuaternion::nlerp: normalize(Quaternion::lerp) with lerp = self * (1 - t) + other * t."""
    return q_normalize(vlerp(a, b, t))


def q_to_rotation_matrix(q):
    """UnitQuaternion::to_rotation_matrix -> 3x3 [row, col]."""
    i, j, k, w = q
    ww, ii, jj, kk = w * w, i * i, j * j, k * k
    ij, wk, wj = i * j * TWO, w * k * TWO, w * j * TWO
    ik, jk, wi = i * k * TWO, j * k * TWO, w * i * TWO
    m = np.empty((3, 3), np.float32)
    m[0, 0] = ww + ii - jj - kk; m[0, 1] = ij - wk; m[0, 2] = wj + ik
    m[1, 0] = wk + ij; m[1, 1] = ww - ii + jj - kk; m[1, 2] = jk - wi
    m[2, 0] = ik - wj; m[2, 1] = wi + jk; m[2, 2] = ww - ii - jj + kk
    return m
