#!/usr/bin/env python3
"""Timing of the extended skinning launches (blend shapes before skinning, interleaved output) next to the
plain kernel on the C4 workload.  One JSON line.  GPU only."""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--verts", type=int, default=1_000_000)
    ap.add_argument("--bones", type=int, default=256)
    ap.add_argument("--sets", type=int, default=6)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--shapes", type=int, default=4)
    ap.add_argument("--streams", type=int, default=2)
    args = ap.parse_args()
    import fyrox_amd
    from fyrox_amd import synth

    ctx = fyrox_amd.Context(0)
    ctx.set_option("lbs.streams", args.streams)
    seed = synth.SEED_BASE + 4
    mesh = synth.make_mesh(args.verts, args.bones, seed)
    pal = ctx.to_device(synth.make_palette(args.bones, seed))
    storage, plane, w = synth.make_blend_shapes(args.verts, args.shapes, seed)
    d_w = ctx.to_device(w)
    nv = args.verts
    L = synth.ANIMATED_VERTEX
    aos_init = mesh.to_animated_vertex_aos()
    sets = []
    for s in range(args.sets):
        ctx.mesh_upload_soa(100 + s, mesh.pos, mesh.weights, mesh.indices, mesh.normal, mesh.tangent)
        ctx.mesh_set_blend_shapes(100 + s, storage, args.shapes, plane)
        sets.append({"pos": ctx.malloc(nv * 12 + 64), "nrm": ctx.malloc(nv * 12 + 64), "tan": ctx.malloc(nv * 16 + 64),
                     "aos68": ctx.to_device(aos_init), "aos48": ctx.malloc(nv * 48)})

    def run(name, launch, bytes_per_vertex):
        for i in range(30):
            launch(i % args.sets)
        ctx.sync()
        ctx.timer_begin()
        for i in range(args.steps):
            launch(i % args.sets)
        us = ctx.timer_end() * 1e3 / args.steps
        return {"us_per_launch": us, "vertices_per_s": nv / (us * 1e-6), "algorithmic_bytes_per_vertex": bytes_per_vertex,
                "GBps": bytes_per_vertex * nv / (us * 1e-6) / 1e9, "frac_of_8TBps": bytes_per_vertex * nv / (us * 1e-6) / 8e12}

    res = {}
    res["plain_soa"] = run("plain", lambda s: ctx.lbs_skin_device(100 + s, pal.ptr, args.bones, 1, sets[s]["pos"].ptr,
                                                                  sets[s]["nrm"].ptr, sets[s]["tan"].ptr), 100)
    for s in range(args.sets):   # meshes 200+: uploaded interleaved, so the library keeps the vertex buffer resident
        ctx.mesh_upload(200 + s, aos_init, nv, L["stride"], off_pos=L["off_pos"], off_normal=L["off_normal"],
                        off_tangent=L["off_tangent"], off_weights=L["off_weights"], off_indices=L["off_indices"])
        ctx.mesh_set_blend_shapes(200 + s, storage, args.shapes, plane)
    res["vb_in_vb_out_68B"] = run("vb", lambda s: ctx.lbs_skin_ex(200 + s, pal.ptr, args.bones, 1,
                                                                 d_out_vertices=sets[s]["aos68"].ptr, out_stride=0), 136)
    res[f"vb_in_vb_out_68B_{args.shapes}_shapes"] = run("vb_shapes", lambda s: ctx.lbs_skin_ex(
        200 + s, pal.ptr, args.bones, 1, d_blend_shape_weights=d_w.ptr, n_blend_shapes=args.shapes,
        d_out_vertices=sets[s]["aos68"].ptr, out_stride=0), 136 + 18 * args.shapes)
    res["ex_aos_animated_vertex_68B"] = run("aos68", lambda s: ctx.lbs_skin_ex(
        100 + s, pal.ptr, args.bones, 1, d_out_vertices=sets[s]["aos68"].ptr, out_stride=L["stride"],
        out_off_pos=L["off_pos"], out_off_normal=L["off_normal"], out_off_tangent=L["off_tangent"]), 100)
    res["ex_aos_static_vertex_48B"] = run("aos48", lambda s: ctx.lbs_skin_ex(
        100 + s, pal.ptr, args.bones, 1, d_out_vertices=sets[s]["aos48"].ptr, out_stride=48, out_off_pos=0,
        out_off_normal=20, out_off_tangent=32), 100)
    res[f"ex_{args.shapes}_shapes_soa"] = run("shapes", lambda s: ctx.lbs_skin_ex(
        100 + s, pal.ptr, args.bones, 1, d_blend_shape_weights=d_w.ptr, n_blend_shapes=args.shapes,
        d_out_pos=sets[s]["pos"].ptr, d_out_normal=sets[s]["nrm"].ptr, d_out_tangent=sets[s]["tan"].ptr), 100 + 18 * args.shapes)
    res[f"ex_{args.shapes}_shapes_aos68"] = run("shapes_aos", lambda s: ctx.lbs_skin_ex(
        100 + s, pal.ptr, args.bones, 1, d_blend_shape_weights=d_w.ptr, n_blend_shapes=args.shapes,
        d_out_vertices=sets[s]["aos68"].ptr, out_stride=L["stride"], out_off_pos=L["off_pos"],
        out_off_normal=L["off_normal"], out_off_tangent=L["off_tangent"]), 100 + 18 * args.shapes)
    # host-pointer variant (fyx_lbs_skin): 16 KB palette H2D + 40 MB D2H per call, synchronous -- the PCIe-inclusive rate
    import time
    palh = synth.make_palette(args.bones, seed)
    ctx.lbs_skin(100, palh)
    t0 = time.perf_counter()
    n_host = 10
    for _ in range(n_host):
        ctx.lbs_skin(100, palh)
    dt = (time.perf_counter() - t0) / n_host
    res["host_pointer_variant_pcie_inclusive"] = {"ms_per_call": dt * 1e3, "vertices_per_s": nv / dt,
                                                  "note": "fyx_lbs_skin: palette H2D, kernel, 40 MB D2H into pageable host memory, sync"}
    print(json.dumps({"workload": f"{nv} verts / {args.bones} bones, {args.sets} rotating sets, {args.streams} launch streams",
                      "results": res}))
    ctx.close()


if __name__ == "__main__":
    main()
