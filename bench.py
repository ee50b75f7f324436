#!/usr/bin/env python3
"""bench.py -- skinned vertices/s of the HIP linear-blend-skinning hot path on MI355X.

A "step" = one pass of the hot path over one batch: ONE launch of the skinning kernel over the
C4 workload (1 M vertices / 256 bones; position + normal + tangent, 4 influences) through the
C ABI (fyx_lbs_skin_device), inputs resident in HBM.  Steps rotate through `--sets` disjoint
buffer sets (default 8 x 100 MB > 2 x the 256 MiB Infinity Cache) so the stream comes from HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; every rank skins its own 1 M-vertex shard of an N x 1 M-vertex scene
(vertex-range sharding, weak scaling, no data-path collective; `--allgather` adds the RCCL
all-gather of the skinned buffers that a consumer needing the whole scene on every GPU would pay).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_VERTS = 1_000_000
N_BONES = 256
BYTES_PER_VERTEX = 100          # 60 read + 40 written
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
KERNEL_NAME = "lbs_skin"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--sets", type=int, default=8, help="disjoint buffer sets rotated through")
    ap.add_argument("--verts", type=int, default=N_VERTS)
    ap.add_argument("--bones", type=int, default=N_BONES)
    ap.add_argument("--random-bones", action="store_true", help="fully random bone indices (worst-case LDS gather)")
    ap.add_argument("--allgather", action="store_true", help="N>1: add the RCCL all-gather of the skinned buffers (fyx_allgather_f32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--opt", action="append", default=[], help="kernel option key=value (e.g. lbs.prefetch=0)")
    ap.add_argument("--no-check", action="store_true", help="skip the parity spot-check before timing")
    ap.add_argument("--no-serialized", action="store_true",
                    help="skip the extra single-stream timing (roofline.serialized)")
    return ap.parse_args()


def cpu_baseline(mesh, pal, seconds: float) -> dict:
    """The oracle (C restatement of the reference's SERIAL loop, mesh/mod.rs:501-522 + the shader's
    normal/tangent math) timed on this host, 1 thread, on whole passes over the same workload."""
    import oracle
    oracle.lib()
    t0 = time.perf_counter()
    passes = 0
    while True:
        oracle.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal, mesh.normal, mesh.tangent, threads=1)
        passes += 1
        el = time.perf_counter() - t0
        if (el >= seconds and passes >= 2) or passes >= 1000:
            break
    serial = passes * mesh.n_verts / el
    # generous upper bound that does NOT exist in the reference: same arithmetic, OpenMP over vertices
    nthr = oracle.omp_max_threads()
    t0 = time.perf_counter()
    p2 = 0
    while True:
        oracle.lbs_skin(mesh.pos, mesh.weights, mesh.indices, pal, mesh.normal, mesh.tangent, threads=0)
        p2 += 1
        el2 = time.perf_counter() - t0
        if (el2 >= seconds / 2 and p2 >= 2) or p2 >= 5000:
            break
    return {"value": serial, "unit": "vertices/s", "cores": 1, "kind": "port",
            "sample": f"{passes} full passes over the same {mesh.n_verts}-vertex/{pal.shape[0]}-bone workload "
                      f"({el:.1f} s), C restatement of Fyrox's serial CPU loop (Rust toolchain unavailable)",
            "omp_value": p2 * mesh.n_verts / el2, "omp_cores": nthr,
            "omp_note": "OpenMP over vertices; not present in the reference (no rayon on this path)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import fyrox_amd
    from fyrox_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    ctx = fyrox_amd.Context(local_rank)    # owns its launch streams; torch is only used for RCCL + barriers
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    opts = {k: ctx.get_option(k) for k in ("lbs.block", "lbs.blocks_per_cu", "lbs.prefetch", "lbs.exact", "lbs.nt", "lbs.streams",
                                           "lbs.split")}
    from fyrox_amd import sharding
    shard = sharding.vertex_range(world * args.verts, rank, world)   # this rank's slice of the N x 1M scene

    # ---- synthetic inputs; each rank owns a different vertex-range shard --------
    seed = synth.SEED_BASE + 4
    mesh = synth.make_mesh(args.verts, args.bones, seed + 1000 * rank, coherent=not args.random_bones)
    pal = synth.make_palette(args.bones, seed)
    nv = mesh.n_verts
    d_pal = torch.from_numpy(pal).cuda()
    outs = []
    for s in range(args.sets):
        ctx.mesh_upload_soa(s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        outs.append((torch.empty(nv * 3 + 16, dtype=torch.float32, device="cuda"),
                     torch.empty(nv * 3 + 16, dtype=torch.float32, device="cuda"),
                     torch.empty(nv * 4 + 16, dtype=torch.float32, device="cuda")))
    gathered = None
    if world > 1 and args.allgather:
        gathered = [torch.empty(world * (nv * 3 + 16), dtype=torch.float32, device="cuda"),
                    torch.empty(world * (nv * 3 + 16), dtype=torch.float32, device="cuda"),
                    torch.empty(world * (nv * 4 + 16), dtype=torch.float32, device="cuda")]
        # the library's own communicator (fyx_comm_init): rank 0's unique id travels over the process group
        uid = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], rank, world)

    # One foreign call per step: fyx_lbs_skin_device(ctx, mesh_id, d_palette, n_bones, 1, out...) with
    # the ctypes arguments converted once, so the Python side costs ~1 us per launch.
    import ctypes
    from functools import partial
    fn = ctx._l.fyx_lbs_skin_device
    calls = [partial(fn, ctx._h, ctypes.c_uint64(s), ctypes.c_void_p(d_pal.data_ptr()), ctypes.c_uint32(args.bones),
                     ctypes.c_uint32(1), ctypes.c_void_p(o[0].data_ptr()), ctypes.c_void_p(o[1].data_ptr()),
                     ctypes.c_void_p(o[2].data_ptr())) for s, o in enumerate(outs)]
    n_sets = args.sets

    def step(i: int):
        rc = calls[i % n_sets]()
        if rc:
            ctx._check(rc)
        if gathered is not None:             # fyx_allgather_f32 is ordered after the launch on the GPU: no host sync
            for g, o in zip(gathered, outs[i % n_sets]):
                ctx.allgather_f32(o.data_ptr(), o.numel(), g.data_ptr())

    # ---- parity spot-check against the oracle before timing (checker only) -------------------
    parity = None
    if not args.no_check and rank == 0:
        import oracle
        step(0)
        ctx.sync()
        n_chk = min(nv, 50_000)
        ref = oracle.lbs_skin(mesh.pos[:n_chk], mesh.weights[:n_chk], mesh.indices[:n_chk], pal,
                              mesh.normal[:n_chk], mesh.tangent[:n_chk], threads=0)
        got_p = outs[0][0][:n_chk * 3].cpu().numpy().reshape(-1, 3)
        got_t = outs[0][2][:n_chk * 4].cpu().numpy().reshape(-1, 4)
        err = max(float(np.abs(got_p - ref["pos"]).max() / max(np.abs(ref["pos"]).max(), 1e-3)),
                  float(np.abs(got_t - ref["tangent"]).max() / max(np.abs(ref["tangent"]).max(), 1e-3)))
        parity = {"max_rel_err": err, "bit_exact": bool(np.array_equal(got_p, ref["pos"]) and np.array_equal(got_t, ref["tangent"])),
                  "checked_vertices": n_chk}
        if err > 1e-5:
            raise SystemExit(f"parity check failed before timing: max rel err {err:.3e} > 1e-5")

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_begin()                        # hipEvent on the context stream (joins the launch streams)
    for i in range(args.steps):
        step(args.warmup + i)
    gpu_ms = ctx.timer_end()                 # second hipEvent after a GPU-side join of all launch streams
    barrier()
    elapsed = time.perf_counter() - t0
    # Same launches serialized on ONE stream (outside the timed region above): there the HIP-event
    # average per launch IS the kernel's duration as rocprofv3 --kernel-trace reports it; with the
    # default two launch streams consecutive kernels overlap pairwise, so the trace shows ~2x longer
    # kernels while the device retires one launch every `avg_launch_us`.
    serial_us = None
    if not args.no_serialized and gathered is None:
        n_ser = max(200, min(args.steps, 1000))
        ctx.set_option("lbs.streams", 1)
        for i in range(50):
            step(i)
        ctx.sync()
        ctx.timer_begin()
        for i in range(n_ser):
            step(i)
        serial_us = ctx.timer_end() * 1e3 / n_ser
        ctx.set_option("lbs.streams", opts["lbs.streams"])
    if dist is not None:
        t = torch.tensor([elapsed, gpu_ms, serial_us or 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, gpu_ms = float(t[0]), float(t[1])
        serial_us = float(t[2]) or None

    if rank == 0:
        total_verts = float(world) * nv * args.steps
        value = total_verts / elapsed
        launch_us = gpu_ms * 1e3 / args.steps           # average per launch, HIP events
        achieved = BYTES_PER_VERTEX * nv / (launch_us * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):                        # PMC-derived HBM bytes per launch (see profiles/README.md)
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "skinned vertices/sec at 1M verts/256 bones; achieved HBM GB/s vs peak",
            "value": value, "unit": "vertices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4: {nv} verts / {args.bones} bones per GPU, 4-influence LBS of position+normal+tangent, "
                                   f"{args.sets} rotating 100 MB buffer sets, "
                                   f"{'random' if args.random_bones else 'spatially coherent'} bone indices",
                       "sharding": "contiguous vertex range per GPU, palette replicated" + (", + RCCL all-gather" if gathered else ""),
                       "rank0_vertex_range": list(shard),
                       "kernel_options": opts},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": KERNEL_NAME, "avg_launch_us": launch_us,
                         "algorithmic_bytes_per_launch": BYTES_PER_VERTEX * nv,
                         "launch_streams": opts["lbs.streams"],
                         "serialized": None if serial_us is None else {
                             "avg_launch_us": serial_us, "achieved": BYTES_PER_VERTEX * nv / (serial_us * 1e-6) / 1e9,
                             "frac": BYTES_PER_VERTEX * nv / (serial_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                             "note": "one launch stream: kernels do not overlap, avg_launch_us == rocprofv3 kernel duration"}},
            "parity": parity,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(mesh, pal, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
