"""Multi-GPU partitioning of the skinning path: one process per GPU.

Vertices are independent given the (replicated, <= 16 KiB) palette, so a mesh shards by contiguous
vertex range and a crowd by instance range; there is NO exchange step inside the path.  The only
collective is the optional all-gather of the skinned streams for a consumer that needs the whole
buffer on every GPU (RCCL over xGMI on the GPU box: backend "nccl"; "gloo" in the CPU tests).
Host-side plumbing only -- nothing here computes a vertex.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

VERTEX_ALIGN = 256   # shard boundaries fall on whole 256-vertex groups (4 wave-sized work units)


def vertex_range(n_verts: int, rank: int, world: int, align: int = VERTEX_ALIGN) -> Tuple[int, int]:
    """[begin, end) of rank's contiguous vertex shard: [g*N/G, (g+1)*N/G) rounded to `align`."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    groups = (n_verts + align - 1) // align

    def cut(g: int) -> int:
        return min(n_verts, (g * groups // world) * align)

    return cut(rank), cut(rank + 1)


def vertex_range_native(n_verts: int, rank: int, world: int) -> Tuple[int, int]:
    """The same cut computed by the library (fyx_shard_vertex_range), which fyx_allgather_skinned uses."""
    from ctypes import byref, c_uint32
    from . import _native
    b, e = c_uint32(), c_uint32()
    rc = _native.lib().fyx_shard_vertex_range(n_verts, rank, world, byref(b), byref(e))
    if rc:
        raise ValueError(f"fyx_shard_vertex_range({n_verts}, {rank}, {world}) -> {rc}")
    return b.value, e.value


def vertex_range_padded(n_verts: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(begin, end, shard_verts) of exchange form 2 (comm.form = 2, fyx_shard_vertex_range_padded): equal shards of shard_verts
    vertices, the full buffers hold world * shard_verts vertices, one in-place all-gather per stream."""
    from ctypes import byref, c_uint32
    from . import _native
    b, e, s = c_uint32(), c_uint32(), c_uint32()
    rc = _native.lib().fyx_shard_vertex_range_padded(n_verts, rank, world, byref(b), byref(e), byref(s))
    if rc:
        raise ValueError(f"fyx_shard_vertex_range_padded({n_verts}, {rank}, {world}) -> {rc}")
    return b.value, e.value, s.value


def instance_range(n_instances: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of rank's instances of a crowd (zero communication)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return rank * n_instances // world, (rank + 1) * n_instances // world


def shard_sizes(n_verts: int, world: int, align: int = VERTEX_ALIGN) -> List[int]:
    return [e - b for b, e in (vertex_range(n_verts, r, world, align) for r in range(world))]


def all_gather_stream(dist, local, n_verts: int, width: int, rank: int, world: int, align: int = VERTEX_ALIGN):
    """All-gather one skinned stream (`local`: this rank's (shard, width) tensor, CPU or GPU) into the
    full (n_verts, width) tensor on every rank.  Shards are padded to the largest shard so ONE
    all_gather_into_tensor moves everything (few large collectives beat many small ones on xGMI)."""
    import torch
    sizes = shard_sizes(n_verts, world, align)
    assert local.shape[0] == sizes[rank] and local.shape[1] == width, (local.shape, sizes[rank], width)
    mx = max(sizes)
    padded = local
    if sizes[rank] != mx:
        padded = torch.zeros((mx, width), dtype=local.dtype, device=local.device)
        padded[:sizes[rank]] = local
    out = torch.empty((world * mx, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous())
    if all(s == mx for s in sizes):
        return out[:n_verts]
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], dim=0)


def broadcast_palette(dist, palette, src: int = 0):
    """The per-frame palette is tiny; rank `src` owns the pose and everyone else receives it."""
    dist.broadcast(palette, src=src)
    return palette
