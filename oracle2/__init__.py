"""oracle2 -- a SECOND, independent CPU restatement of the reference's animation / skinning path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product (fyrox_amd/, include/, bench.py's timed region) may import this
package; only tests/ does, as a checker of the checker.

Why it exists.  For most of the path the reference holds no test vector (LBS, the palette product,
Transform::calculate_local_transform, every blend_with, Animation::tick, Machine::evaluate_pose are "parity
unpinned"), so `oracle/` (plain C) is the only authority the GPU kernels are compared with -- and it was written by the
same hands as the kernels.  This package restates the same functions AGAIN, from the Rust sources only
(/root/reference, files and lines cited per function), in a different language (Python, numpy float32 scalars: every
operation rounds to f32, nothing fuses), with different data structures (poses are dicts of dicts, values are tagged
tuples, the machine is evaluated recursively with memoised pose objects exactly as the Rust does, matrices are numpy
(4, 4) arrays indexed [row, col]) and without looking at oracle/*.c.  tests/test_oracle2_differential.py runs both
oracles on random inputs (palettes incl. projective ones, meshes, rigs with pivots / offsets / pre- and post-rotation,
machines of all four pose-node kinds, transitions, masks, signals, root motion) and fails on ANY differing bit (Euler
tracks excepted only where libm's sinf / cosf are reached through different call paths -- both call glibc here, so
they agree bit for bit too).  It is also checked against the reference's own golden vectors
(tests/golden/fyrox_unit_vectors.json) on its own.

What it cannot settle.  The arithmetic leaves live in nalgebra 0.35, which is not vendored in the reference and
cannot be fetched here.  Their operation order is restated in `oracle2/na.py` from knowledge of nalgebra's published
source (file and function named per leaf; no line numbers -- they cannot be checked offline).  A misreading of
nalgebra shared by both restatements would still pass; the golden vectors the reference does hold
(quat_from_euler == from_euler_angles with exact f32 equality, the graph hierarchy test, fetch_weights) pin the
Hamilton product, from_axis_angle, the matrix product and the 2-element dot, and nothing else.

Not restated here (needs an unavailable crate or is outside rows a4-a13): BlendSpace triangulation (spade),
StateAction::EnableRandomAnimation (rand::thread_rng), blend shapes.
"""
from .na import F  # noqa: F401
from .curve import Curve  # noqa: F401
from .anim import AnimScene  # noqa: F401
from .scene import lbs_skin, local_matrix, global_matrices, palette  # noqa: F401
