"""Multi-GPU sharding of the skinning path on CPU: world_size-2 (and 3) `gloo` process groups.

Each rank skins ITS vertex range with the oracle (the checker standing in for the per-GPU kernel; the
GPU kernel's own parity is tests/test_lbs_gpu.py), the ranks all-gather the skinned streams through
fyrox_amd.sharding, and every rank must hold exactly the single-process result."""
import os
import socket

import numpy as np
import pytest

from fyrox_amd import sharding, synth


def test_vertex_ranges_tile_the_mesh():
    for n in (0, 1, 255, 256, 257, 1000, 50_000, 1_000_000, 1_000_001):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                b, e = sharding.vertex_range(n, r, world)
                assert b == prev and e >= b
                assert b % sharding.VERTEX_ALIGN == 0 or b == n
                prev = e
            assert prev == n
    sizes = sharding.shard_sizes(1_000_000, 8)
    assert sum(sizes) == 1_000_000 and max(sizes) - min(sizes) <= sharding.VERTEX_ALIGN
    with pytest.raises(ValueError):
        sharding.vertex_range(10, 2, 2)


def test_library_cut_equals_the_python_cut():
    """fyx_shard_vertex_range (what fyx_allgather_skinned places the shards by) against fyrox_amd.sharding, and the
    BASELINE config-4 cut spelled out: 1 M vertices over 8 GPUs is ragged."""
    for n in (0, 1, 255, 256, 257, 1000, 4097, 50_000, 999_999, 1_000_000, 1_000_001, 4_000_000_000):
        for world in (1, 2, 3, 4, 5, 7, 8, 16):
            for r in range(world):
                assert sharding.vertex_range_native(n, r, world) == sharding.vertex_range(n, r, world), (n, r, world)
    cuts = [sharding.vertex_range_native(1_000_000, r, 8) for r in range(8)]
    assert [e - b for b, e in cuts] == [124928, 124928, 125184, 124928, 124928, 125184, 124928, 124992]
    assert [b for b, _ in cuts] == [0, 124928, 249856, 375040, 499968, 624896, 750080, 875008]
    assert [e - b for b, e in cuts] == sharding.shard_sizes(1_000_000, 8)
    with pytest.raises(ValueError):
        sharding.vertex_range_native(10, 2, 2)
    with pytest.raises(ValueError):
        sharding.vertex_range_native(10, 0, 0)


def test_padded_cut_of_exchange_form_2():
    """fyx_shard_vertex_range_padded (comm.form = 2): equal 256-aligned shards that tile the mesh, the buffers hold world * shard
    vertices, the BASELINE config-4 cut spelled out, and the refusals."""
    for n in (0, 1, 255, 256, 257, 1000, 4097, 50_000, 999_999, 1_000_000, 1_000_001):
        for world in (1, 2, 3, 4, 5, 7, 8, 16):
            prev, shard = 0, None
            for r in range(world):
                b, e, s = sharding.vertex_range_padded(n, r, world)
                shard = s if shard is None else shard
                assert s == shard and s % sharding.VERTEX_ALIGN == 0
                assert b == min(n, r * s) and e == min(n, (r + 1) * s) and b == prev
                prev = e
            assert prev == n and shard * world >= n and (n == 0 or shard * world - n < world * sharding.VERTEX_ALIGN + sharding.VERTEX_ALIGN)
    cuts = [sharding.vertex_range_padded(1_000_000, r, 8) for r in range(8)]
    assert {s for _, _, s in cuts} == {125_184} and 8 * 125_184 == 1_001_472
    assert [e - b for b, e, _ in cuts] == [125_184] * 7 + [123_712]
    with pytest.raises(ValueError):
        sharding.vertex_range_padded(10, 2, 2)
    with pytest.raises(ValueError):
        sharding.vertex_range_padded(4_294_967_295, 0, 7)       # 7 shards of 613 566 976 vertices do not fit 32 bits


def test_instance_ranges_tile_the_crowd():
    for n in (0, 1, 7, 1000):
        for world in (1, 2, 4, 8):
            rs = [sharding.instance_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_verts: int, n_bones: int, q):
    import torch
    import torch.distributed as dist
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seed = synth.SEED_BASE + 4
        mesh = synth.make_mesh(n_verts, n_bones, seed)          # every rank can build the inputs ...
        pal = torch.from_numpy(synth.make_palette(n_bones, seed) if rank == 0
                               else np.zeros((n_bones, 16), np.float32))
        sharding.broadcast_palette(dist, pal, src=0)             # ... but only rank 0 owns the pose
        b, e = sharding.vertex_range(n_verts, rank, world)       # this rank holds ONLY its slice
        out = oracle.lbs_skin(mesh.pos[b:e], mesh.weights[b:e], mesh.indices[b:e], pal.numpy(),
                              mesh.normal[b:e], mesh.tangent[b:e])
        full = {k: sharding.all_gather_stream(dist, torch.from_numpy(out[k]), n_verts, out[k].shape[1], rank, world).numpy()
                for k in ("pos", "normal", "tangent")}
        ref = oracle.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal.numpy(), mesh.normal, mesh.tangent)
        ok = all(np.array_equal(full[k], ref[k]) for k in ref)
        q.put((rank, ok, (b, e)))
    finally:
        dist.destroy_process_group()


def _worker_in_place(rank: int, world: int, port: int, n_verts: int, n_bones: int, q):
    """The schedule of fyx_allgather_skinned with gloo standing in for RCCL: every rank skins its shard INTO its place
    of the full streams (the library's own cut), then one broadcast per (stream, owner) over that owner's slice."""
    import torch
    import torch.distributed as dist
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seed = synth.SEED_BASE + 4
        mesh = synth.make_mesh(n_verts, n_bones, seed)
        pal = synth.make_palette(n_bones, seed)
        b, e = sharding.vertex_range_native(n_verts, rank, world)
        out = oracle.lbs_skin(mesh.pos[b:e], mesh.weights[b:e], mesh.indices[b:e], pal, mesh.normal[b:e], mesh.tangent[b:e])
        full = {k: torch.full((n_verts, w), float("nan")) for k, w in (("pos", 3), ("normal", 3), ("tangent", 4))}
        for k in full:
            full[k][b:e] = torch.from_numpy(out[k])
        for k in ("pos", "normal", "tangent"):
            for r in range(world):
                rb, re = sharding.vertex_range_native(n_verts, r, world)
                if re > rb:
                    piece = full[k][rb:re]            # a view: received in place
                    dist.broadcast(piece, src=r)
        ref = oracle.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal, mesh.normal, mesh.tangent)
        q.put((rank, all(np.array_equal(full[k].numpy(), ref[k]) for k in ref), (b, e)))
    finally:
        dist.destroy_process_group()


def _worker_send_recv(rank: int, world: int, port: int, n_verts: int, n_bones: int, q):
    """The schedule of fyx_allgather_skinned with comm.form = 1, gloo standing in for RCCL: every rank sends its own shard to each
    other rank and receives each other shard in place -- 2 (world - 1) point-to-point calls per stream, all posted before any is
    waited for (RCCL: inside one group)."""
    import torch
    import torch.distributed as dist
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seed = synth.SEED_BASE + 4
        mesh = synth.make_mesh(n_verts, n_bones, seed)
        pal = synth.make_palette(n_bones, seed)
        b, e = sharding.vertex_range_native(n_verts, rank, world)
        out = oracle.lbs_skin(mesh.pos[b:e], mesh.weights[b:e], mesh.indices[b:e], pal, mesh.normal[b:e], mesh.tangent[b:e])
        full = {k: torch.full((n_verts, w), float("nan")) for k, w in (("pos", 3), ("normal", 3), ("tangent", 4))}
        for k in full:
            full[k][b:e] = torch.from_numpy(out[k])
        reqs = []
        for k in ("pos", "normal", "tangent"):
            for r in range(world):
                if r == rank:
                    continue
                rb, re = sharding.vertex_range_native(n_verts, r, world)
                if e > b:
                    reqs.append(dist.isend(full[k][b:e], dst=r))
                if re > rb:
                    reqs.append(dist.irecv(full[k][rb:re], src=r))      # a view: received in place
        for rq in reqs:
            rq.wait()
        ref = oracle.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal, mesh.normal, mesh.tangent)
        q.put((rank, all(np.array_equal(full[k].numpy(), ref[k]) for k in ref), (b, e)))
    finally:
        dist.destroy_process_group()


def _worker_padded(rank: int, world: int, port: int, n_verts: int, n_bones: int, q):
    """The schedule of fyx_allgather_skinned with comm.form = 2, gloo standing in for RCCL: buffers of world * shard vertices, every
    rank skins [begin, end) of the padded cut in place and ONE in-place all_gather_into_tensor per stream (input = the rank's own
    slice of the output) moves everything."""
    import torch
    import torch.distributed as dist
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seed = synth.SEED_BASE + 4
        mesh = synth.make_mesh(n_verts, n_bones, seed)
        pal = synth.make_palette(n_bones, seed)
        b, e, shard = sharding.vertex_range_padded(n_verts, rank, world)
        out = oracle.lbs_skin(mesh.pos[b:e], mesh.weights[b:e], mesh.indices[b:e], pal, mesh.normal[b:e], mesh.tangent[b:e])
        full = {k: torch.full((world * shard, w), float("nan")) for k, w in (("pos", 3), ("normal", 3), ("tangent", 4))}
        for k in full:
            full[k][b:e] = torch.from_numpy(out[k])
            mine = full[k][rank * shard:(rank + 1) * shard].clone()      # (gloo wants a separate input; RCCL takes the slice itself)
            dist.all_gather_into_tensor(full[k], mine)
        ref = oracle.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal, mesh.normal, mesh.tangent)
        q.put((rank, all(np.array_equal(full[k][:n_verts].numpy(), ref[k]) for k in ref), (b, e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_verts", [(2, 4097), (3, 1000), (2, 300)])
def test_padded_all_gather_schedule(world, n_verts):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_padded, args=(r, world, port, n_verts, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    ranges = sorted(rg for _, _, rg in results)
    assert ranges[0][0] == 0 and ranges[-1][1] == n_verts


@pytest.mark.parametrize("world,n_verts", [(2, 4097), (3, 1000), (2, 300)])
def test_send_recv_gather_schedule(world, n_verts):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_send_recv, args=(r, world, port, n_verts, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)


@pytest.mark.parametrize("world,n_verts", [(2, 4097), (3, 1000), (2, 300)])
def test_in_place_ragged_gather_schedule(world, n_verts):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_in_place, args=(r, world, port, n_verts, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    ranges = sorted(rg for _, _, rg in results)
    assert ranges[0][0] == 0 and ranges[-1][1] == n_verts


@pytest.mark.parametrize("world,n_verts", [(2, 10_000), (2, 4097), (3, 1000)])
def test_sharded_skinning_all_gathers_to_the_single_process_result(world, n_verts):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_verts, 64, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in results) == list(range(world))
    assert all(ok for _, ok, _ in results)
    ranges = sorted(rg for _, _, rg in results)
    assert ranges[0][0] == 0 and ranges[-1][1] == n_verts
