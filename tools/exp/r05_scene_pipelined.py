#!/usr/bin/env python3
"""Round 5: the scene tick with pipelined frames (anim.overlap + palette pairs + registered skin outputs) against the one-stream
frame; bench.py's own scene record, one JSON line per scene.

    python tools/exp/r05_scene_pipelined.py [256x1x5000 64x4x20000 ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench            # noqa: E402
import fyrox_amd        # noqa: E402


def main():
    shapes = [a for a in sys.argv[1:] if "=" not in a] or ["256x1x5000", "64x4x20000", "32x1x5000"]
    opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
    with fyrox_amd.Context(0) as ctx:
        for k_, v_ in opts:
            ctx.set_option(k_, int(v_))
        for k, s in enumerate(shapes):
            nc, ni, nv = (int(x) for x in s.split("x"))
            rec = bench._scene_record(ctx, nc, ni, nv, 5_000_000 + k * 1_000_000)
            keep = {k_: rec[k_] for k_ in ("frame_ms", "frame_mode", "frame_ms_scene_update_then_skin_batch", "frame_ms_skin_outputs", "frame_ms_pipelined", "frame_ms_pipelined_streams_by_kind", "host_ms_pipelined_streams_by_kind", "host_sections_pipelined_streams_by_kind_us",
                                            "pipelined_bit_identical_to_skin_batch", "host_ms_pipelined", "host_ms_skin_outputs", "host_sections_pipelined_us", "host_sections_skin_outputs_us",
                                            "pose_ms", "skin_ms")}
            keep["scene"] = s
            keep["options"] = dict(opts)
            print(json.dumps(keep), flush=True)


if __name__ == "__main__":
    main()
