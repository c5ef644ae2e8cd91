#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this box: torch copy / scale of N MB (read + write counted), rotating buffers."""
import sys
import torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 105
dev = torch.device("cuda:0")
n = mb * (1 << 20) // 2
bufs = [(torch.randn(n, device=dev).to(torch.bfloat16), torch.empty(n, device=dev, dtype=torch.bfloat16)) for _ in range(6)]
for name, fn in (("copy_", lambda x, y: y.copy_(x)), ("mul", lambda x, y: torch.mul(x, 2.0, out=y)), ("silu", lambda x, y: torch.nn.functional.silu(x, inplace=False))):
    for x, y in bufs:
        fn(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        for x, y in bufs:
            fn(x, y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 60 * 1e3
    print(f"{name:6s} {mb} MB in + {mb} MB out: {us:7.1f} us  {2 * mb * 1.048576 / us:5.2f} TB/s")
