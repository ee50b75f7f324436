"""Pythonic wrapper over the C ABI (include/fyrox_hip.h).  Host-side plumbing only: every
computation below happens in the HIP kernels of libfyrox_hip.so."""
from __future__ import annotations

import ctypes
from ctypes import byref, c_int, c_uint32, c_void_p
from typing import Optional, Sequence

import numpy as np

from . import _native
from ._native import FyxError


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _f32(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


class DeviceBuffer:
    """A raw HBM allocation owned by a Context (fyx_malloc / fyx_free)."""

    def __init__(self, ctx: "Context", nbytes: int, ptr: int | None = None):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        if ptr is not None:      # (an allocation made elsewhere: fyx_malloc_streams)
            self.ptr = ptr
            return
        p = c_void_p()
        ctx._check(ctx._l.fyx_malloc(ctx._h, self.nbytes, byref(p)))
        self.ptr = p.value or 0

    def upload(self, host: np.ndarray) -> "DeviceBuffer":
        host = np.ascontiguousarray(host)
        assert host.nbytes <= self.nbytes
        self.ctx._check(self.ctx._l.fyx_memcpy_h2d(self.ctx._h, self.ptr, _ptr(host), host.nbytes))
        return self

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._check(self.ctx._l.fyx_memcpy_d2h(self.ctx._h, _ptr(out), self.ptr, out.nbytes))
        return out

    def free(self) -> None:
        if self.ptr:
            self.ctx._l.fyx_free(self.ctx._h, self.ptr)
            self.ptr = 0


class Context:
    """One fyx_ctx: a GPU, a stream, the mesh registry.  Not thread-safe (as the reference's
    update/render thread)."""

    def __init__(self, device: int = 0, *, control_only: bool = False):
        self._l = _native.lib()
        h = c_void_p()
        if control_only:
            # no GPU: registry + control-plane calls only; every data-path call raises FYX_ERR_NO_DEVICE
            rc = self._l.fyx_init_control_only(byref(h))
            if rc != 0:
                raise FyxError(rc, "fyx_init_control_only failed")
            self._h = h
            self.device = -1
            return
        rc = self._l.fyx_init(byref(h), int(device))
        if rc != 0:
            raise FyxError(rc, f"fyx_init(device={device}) failed: no MI355X/HIP device visible; "
                               "fyrox_amd has no CPU fallback")
        self._h = h
        self.device = device

    # -- plumbing ------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise FyxError(rc, self._l.fyx_last_error(self._h).decode())

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._l.fyx_shutdown(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self) -> None:
        self._check(self._l.fyx_sync(self._h))

    def join(self) -> None:
        """GPU-side join: the context stream waits for every in-flight skinning launch."""
        self._check(self._l.fyx_join(self._h))

    def set_stream(self, hip_stream: int) -> None:
        self._check(self._l.fyx_set_stream(self._h, hip_stream))

    @property
    def stream(self) -> int:
        return self._l.fyx_get_stream(self._h) or 0

    def timer_begin(self) -> None:
        self._check(self._l.fyx_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = ctypes.c_float()
        self._check(self._l.fyx_timer_end(self._h, byref(ms)))
        return ms.value

    def kernel_time(self):
        """(sum of the kernel durations in us, number of launches) of the fyx_lbs_skin_device launches made under option
        lbs.timing = 1 since the last call."""
        us, n = ctypes.c_double(), ctypes.c_uint32()
        self._check(self._l.fyx_debug_kernel_time(self._h, byref(us), byref(n)))
        return us.value, n.value

    def timeline(self, capacity: int = 16384):
        """(kinds, start_us, stop_us) of the launches made under option debug.timeline = 1 since the last call: kind 0 skinning,
        1 pose_sample, 2 pose_update; times in microseconds after the first record's start."""
        kinds = np.zeros(capacity, np.int32)
        a, b = np.zeros(capacity, np.float64), np.zeros(capacity, np.float64)
        n = ctypes.c_uint32()
        self._check(self._l.fyx_debug_timeline(self._h, _ptr(kinds), _ptr(a), _ptr(b), capacity, byref(n)))
        k = min(n.value, capacity)
        return kinds[:k], a[:k], b[:k]

    def set_option(self, key: str, value: int) -> None:
        self._check(self._l.fyx_set_option(self._h, key.encode(), int(value)))

    def last_error(self) -> str:
        """fyx_last_error: the message of the last failure -- or warning (a one-launch frame that was run again)."""
        return self._l.fyx_last_error(self._h).decode()

    def get_option(self, key: str) -> int:
        v = c_int()
        self._check(self._l.fyx_get_option(self._h, key.encode(), byref(v)))
        return v.value

    def host_times(self) -> list:
        """Option debug.host_times: microseconds fyx_scene_update's sections cost the calling thread since the last call (fyx_debug_host_times)."""
        import numpy as np
        out = np.zeros(8, np.float64)
        self._check(self._l.fyx_debug_host_times(self._h, out.ctypes.data_as(c_void_p), 8))
        return out.tolist()

    def malloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def malloc_streams(self, sizes) -> list:
        """fyx_malloc_streams: one device allocation PER output stream (never ranges of one block: include/fyrox_hip.h)."""
        import ctypes
        n = len(sizes)
        arr = (ctypes.c_size_t * n)(*[int(s) for s in sizes])
        out = (c_void_p * n)()
        self._check(self._l.fyx_malloc_streams(self._h, n, arr, out))
        return [DeviceBuffer(self, int(s), out[i] or 0) for i, s in enumerate(sizes)]

    def free_ptr(self, ptr: int) -> None:
        self._check(self._l.fyx_free(self._h, ptr))

    def to_device(self, host: np.ndarray) -> DeviceBuffer:
        host = np.ascontiguousarray(host)
        return DeviceBuffer(self, max(host.nbytes, 16)).upload(host)

    # -- multi-GPU exchange (RCCL all-gather behind the C ABI) ----------------------------
    def comm_unique_id(self) -> bytes:
        buf = (ctypes.c_uint8 * 128)()
        self._check(self._l.fyx_comm_unique_id(self._h, buf))
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, n_ranks: int) -> None:
        assert len(unique_id) == 128
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._l.fyx_comm_init(self._h, buf, rank, n_ranks))

    def comm_shutdown(self) -> None:
        self._check(self._l.fyx_comm_shutdown(self._h))

    def allgather_f32(self, d_send: int, count: int, d_recv: int) -> None:
        self._check(self._l.fyx_allgather_f32(self._h, d_send, count, d_recv))

    def comm_info(self):
        """(rank, n_ranks) as RCCL reports them."""
        r, n = c_int(), c_int()
        self._check(self._l.fyx_comm_info(self._h, byref(r), byref(n)))
        return r.value, n.value

    def allgather_skinned(self, n_verts: int, d_pos_all: int = 0, d_normal_all: int = 0, d_tangent_all: int = 0) -> None:
        """fyx_allgather_skinned: every rank wrote its own (ragged) shard in place; afterwards all hold all."""
        self._check(self._l.fyx_allgather_skinned(self._h, n_verts, d_pos_all or None, d_normal_all or None,
                                                  d_tangent_all or None))

    def allgather_skinned_padded(self, n_verts: int, capacity_verts: int, d_pos_all: int = 0, d_normal_all: int = 0, d_tangent_all: int = 0) -> None:
        """fyx_allgather_skinned_padded: equal padded shards, ONE in-place all-gather per stream; every buffer holds capacity_verts vertices."""
        self._check(self._l.fyx_allgather_skinned_padded(self._h, n_verts, capacity_verts, d_pos_all or None, d_normal_all or None,
                                                         d_tangent_all or None))

    # -- one process, several GPUs (every call from one thread) ---------------------------
    @staticmethod
    def comm_init_all(contexts) -> None:
        """fyx_comm_init_all: contexts[i] (each on its own GPU) becomes rank i of len(contexts)."""
        arr = (ctypes.c_void_p * len(contexts))(*[c._h for c in contexts])
        contexts[0]._check(contexts[0]._l.fyx_comm_init_all(arr, len(contexts)))

    @staticmethod
    def allgather_skinned_all(contexts, n_verts: int, d_pos_all=None, d_normal_all=None, d_tangent_all=None) -> None:
        """fyx_allgather_skinned_all: d_*_all[i] = device address of context i's FULL stream (None: the stream is not
        exchanged); every context has written its own shard in place."""
        def arr(ptrs):
            if ptrs is None:
                return None
            assert len(ptrs) == len(contexts)
            return (ctypes.c_void_p * len(ptrs))(*[int(p) if p else None for p in ptrs])
        h = (ctypes.c_void_p * len(contexts))(*[c._h for c in contexts])
        contexts[0]._check(contexts[0]._l.fyx_allgather_skinned_all(h, len(contexts), n_verts, arr(d_pos_all), arr(d_normal_all),
                                                                    arr(d_tangent_all)))

    @staticmethod
    def allgather_skinned_padded_all(contexts, n_verts: int, capacity_verts: int, d_pos_all=None, d_normal_all=None, d_tangent_all=None) -> None:
        """fyx_allgather_skinned_padded_all: the padded form (one in-place all-gather per stream) for every GPU of the process."""
        def arr(ptrs):
            if ptrs is None:
                return None
            assert len(ptrs) == len(contexts)
            return (ctypes.c_void_p * len(ptrs))(*[int(p) if p else None for p in ptrs])
        h = (ctypes.c_void_p * len(contexts))(*[c._h for c in contexts])
        contexts[0]._check(contexts[0]._l.fyx_allgather_skinned_padded_all(h, len(contexts), n_verts, capacity_verts, arr(d_pos_all),
                                                                           arr(d_normal_all), arr(d_tangent_all)))

    # -- mesh registry -------------------------------------------------------------------
    def mesh_upload(self, mesh_id: int, aos: np.ndarray, n_verts: int, stride: int, *, off_pos: int,
                    off_normal: int = -1, off_tangent: int = -1, off_weights: int, off_indices: int) -> None:
        aos = np.ascontiguousarray(aos, dtype=np.uint8)
        assert aos.nbytes >= n_verts * stride
        self._check(self._l.fyx_mesh_upload(self._h, mesh_id, _ptr(aos), n_verts, stride, off_pos,
                                            off_normal, off_tangent, off_weights, off_indices))

    def mesh_upload_soa(self, mesh_id: int, pos, weights, indices, normal=None, tangent=None) -> None:
        pos = _f32(pos, (-1, 3))
        n = pos.shape[0]
        weights = _f32(weights, (n, 4))
        indices = np.ascontiguousarray(indices, dtype=np.uint8).reshape(n, 4)
        normal = None if normal is None else _f32(normal, (n, 3))
        tangent = None if tangent is None else _f32(tangent, (n, 4))
        self._check(self._l.fyx_mesh_upload_soa(self._h, mesh_id, n, _ptr(pos), _ptr(normal),
                                                _ptr(tangent), _ptr(weights), _ptr(indices)))

    def mesh_free(self, mesh_id: int) -> None:
        self._check(self._l.fyx_mesh_free(self._h, mesh_id))

    def mesh_info(self, mesh_id: int) -> dict:
        n, mb, am = c_uint32(), c_uint32(), c_uint32()
        self._check(self._l.fyx_mesh_info(self._h, mesh_id, byref(n), byref(mb), byref(am)))
        return {"n_verts": n.value, "max_bone_index": mb.value, "has_normal": bool(am.value & 1),
                "has_tangent": bool(am.value & 2)}

    def mesh_streams(self, mesh_id: int) -> dict:
        ps = [c_void_p() for _ in range(5)]
        self._check(self._l.fyx_mesh_streams(self._h, mesh_id, *[byref(p) for p in ps]))
        return dict(zip(("pos", "normal", "tangent", "weights", "indices"), [p.value or 0 for p in ps]))

    # -- skinning ------------------------------------------------------------------------
    def lbs_skin(self, mesh_id: int, palette, n_instances: int = 1, *, want: Sequence[str] = ("pos", "normal", "tangent"),
                 aabb: bool = False) -> dict:
        """Host palette in, host skinned vertices out (synchronous).  palette: (n_instances*n_bones, 16)
        column-major mat4 rows."""
        palette = _f32(palette, (-1, 16))
        assert palette.shape[0] % n_instances == 0
        n_bones = palette.shape[0] // n_instances
        info = self.mesh_info(mesh_id)
        nv = info["n_verts"] * n_instances
        out = {}
        if "pos" in want:
            out["pos"] = np.empty((nv, 3), np.float32)
        if "normal" in want:
            out["normal"] = np.empty((nv, 3), np.float32)
        if "tangent" in want:
            out["tangent"] = np.empty((nv, 4), np.float32)
        box = np.empty(6, np.float32) if aabb else None
        self._check(self._l.fyx_lbs_skin(self._h, mesh_id, _ptr(palette), n_bones, n_instances,
                                         _ptr(out.get("pos")), _ptr(out.get("normal")),
                                         _ptr(out.get("tangent")), _ptr(box)))
        if aabb:
            out["aabb"] = box
        return out

    def lbs_skin_device(self, mesh_id: int, d_palette: int, n_bones: int, n_instances: int,
                        d_out_pos: int = 0, d_out_normal: int = 0, d_out_tangent: int = 0) -> None:
        self._check(self._l.fyx_lbs_skin_device(self._h, mesh_id, d_palette, n_bones, n_instances,
                                                d_out_pos or None, d_out_normal or None, d_out_tangent or None))

    def lbs_skin_streams(self, n_verts: int, d_pos: int, d_normal: int, d_tangent: int, d_weights: int,
                         d_indices: int, d_palette: int, n_bones: int, n_instances: int,
                         d_out_pos: int = 0, d_out_normal: int = 0, d_out_tangent: int = 0) -> None:
        self._check(self._l.fyx_lbs_skin_streams(self._h, n_verts, d_pos or None, d_normal or None,
                                                 d_tangent or None, d_weights, d_indices, d_palette,
                                                 n_bones, n_instances, d_out_pos or None,
                                                 d_out_normal or None, d_out_tangent or None))

    def lbs_skin_batch(self, jobs) -> None:
        """jobs: sequence of (mesh_id, d_palette, n_bones, n_instances, d_out_pos, d_out_normal, d_out_tangent) or a
        prebuilt (SkinJob * n) array (fyx_lbs_skin_batch)."""
        from ._native import SkinJob
        if not isinstance(jobs, ctypes.Array):
            arr = (SkinJob * len(jobs))()
            for k, j in enumerate(jobs):
                arr[k] = SkinJob(*[(v or None) if i in (1, 4, 5, 6) else v for i, v in enumerate(j)])
            jobs = arr
        self._check(self._l.fyx_lbs_skin_batch(self._h, jobs, len(jobs)))

    def mesh_set_blend_shapes(self, mesh_id: int, storage: Optional[np.ndarray], n_shapes: int, plane_vertices: int) -> None:
        """storage: the RGB16F volume bytes of BlendShapesContainer (uint16 view), [shape][plane][9]."""
        st = None if storage is None else np.ascontiguousarray(storage).view(np.uint16)
        if st is not None:
            assert st.size >= n_shapes * plane_vertices * 9
        self._check(self._l.fyx_mesh_set_blend_shapes(self._h, mesh_id, n_shapes, _ptr(st), plane_vertices))

    def lbs_skin_ex(self, mesh_id: int, d_palette: int, n_bones: int, n_instances: int = 1, *,
                    d_blend_shape_weights: int = 0, n_blend_shapes: int = 0, d_out_pos: int = 0,
                    d_out_normal: int = 0, d_out_tangent: int = 0, d_out_vertices: int = 0, out_stride: int = 0,
                    out_off_pos: int = -1, out_off_normal: int = -1, out_off_tangent: int = -1) -> None:
        """fyx_lbs_skin_ex: blend shapes before skinning and / or interleaved output.  Asynchronous."""
        d = _native.SkinDesc(d_palette or None, n_bones, n_instances, d_blend_shape_weights or None, n_blend_shapes,
                             d_out_pos or None, d_out_normal or None, d_out_tangent or None, d_out_vertices or None,
                             out_stride, out_off_pos, out_off_normal, out_off_tangent)
        self._check(self._l.fyx_lbs_skin_ex(self._h, mesh_id, byref(d)))

    def lbs_skin_ex_batch(self, jobs) -> None:
        """fyx_lbs_skin_ex_batch.  jobs: sequence of (mesh_id, kwargs) with lbs_skin_ex's arguments
        (d_palette, n_bones, n_instances, d_blend_shape_weights, ...)."""
        n = len(jobs)
        ids = (ctypes.c_uint64 * max(n, 1))()
        descs = (_native.SkinDesc * max(n, 1))()
        for k, (mesh_id, kw) in enumerate(jobs):
            ids[k] = mesh_id
            descs[k] = _native.SkinDesc(kw.get("d_palette") or None, kw.get("n_bones", 0), kw.get("n_instances", 1),
                                        kw.get("d_blend_shape_weights") or None, kw.get("n_blend_shapes", 0),
                                        kw.get("d_out_pos") or None, kw.get("d_out_normal") or None, kw.get("d_out_tangent") or None,
                                        kw.get("d_out_vertices") or None, kw.get("out_stride", 0), kw.get("out_off_pos", -1),
                                        kw.get("out_off_normal", -1), kw.get("out_off_tangent", -1))
        self._check(self._l.fyx_lbs_skin_ex_batch(self._h, ids, descs, n))

    def skinned_aabb(self, mesh_id: int, palette) -> np.ndarray:
        palette = _f32(palette, (-1, 16))
        box = np.empty(6, np.float32)
        self._check(self._l.fyx_skinned_aabb(self._h, mesh_id, _ptr(palette), palette.shape[0], _ptr(box)))
        return box

    def skinned_aabb_device(self, mesh_id: int, d_palette: int, n_bones: int, n_instances: int, d_out_aabb: int) -> None:
        """Per-instance boxes of an instanced mesh, device to device (asynchronous)."""
        self._check(self._l.fyx_skinned_aabb_device(self._h, mesh_id, d_palette, n_bones, n_instances, d_out_aabb))

    def calib_stream_copy(self, d_src: int, d_dst: int, units: int) -> None:
        self._check(self._l.fyx_calib_stream_copy(self._h, d_src, d_dst, units))

    # -- palette -------------------------------------------------------------------------
    def palette(self, global_, inv_bind) -> np.ndarray:
        g = _f32(global_, (-1, 16))
        ib = _f32(inv_bind, (-1, 16))
        assert g.shape == ib.shape
        out = np.empty_like(g)
        self._check(self._l.fyx_palette(self._h, _ptr(g), _ptr(ib), g.shape[0], _ptr(out)))
        return out

    def palette_device(self, d_global: int, d_inv_bind: int, n: int, d_out: int) -> None:
        self._check(self._l.fyx_palette_device(self._h, d_global, d_inv_bind, n, d_out))
