#!/usr/bin/env python3
"""Write-only bandwidth reference for the crowd kernel (its traffic is ~99 % writes): a 400 MB device fill and a
400 MB device-to-device copy, timed with HIP events through torch.  GPU only; one JSON line."""
import json
import torch

n = 100_000_000  # f32 -> 400 MB
bufs = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(4)]
src = torch.ones(n, dtype=torch.float32, device="cuda")
res = {}
for name, fn in (("fill_400MB", lambda b: b.fill_(1.5)), ("copy_400MB_read_plus_400MB_write", lambda b: b.copy_(src))):
    for i in range(8):
        fn(bufs[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    k = 40
    for i in range(k):
        fn(bufs[i % 4])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / k
    res[name] = {"us": us, "written_GBps": 400e6 / us / 1e3}
print(json.dumps(res))
