#!/usr/bin/env python3
"""Development aid: host time per step of dist.ShardedRunner (issue loop only) against the GPU time per step, one rank."""
import os
import socket
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[2] if len(sys.argv) > 2 else "21")
from yololite_amd import _lib

_lib.apply_runtime_knobs()
import torch
import torch.distributed as dist

from yololite_amd.dist import ShardedRunner, record_floats
from yololite_amd.nn import arch
from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.synth import synth_state_dict

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
sk = socket.socket()
sk.bind(("127.0.0.1", 0))
port = sk.getsockname()[1]
sk.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
specs, _ = arch.parse_model(arch.yolo11_dict("n"))
sd = synth_state_dict("n", 0, cls_bias=[-4.0, -4.0, -4.0], cls_gain=6.0)
B, H, W = 32, 640, 640
eng = HipEngine.from_model(specs, sd, B, H, W, dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
gb = [torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device=dev) for _ in range(3)]


def compute(im, slot, lane):
    return eng.detect_record(im, 0.25, 0.7, max_det=300, slot=slot, rec_key=lane)


r = ShardedRunner(compute, B, (3, H, W), torch.uint8, dev, record_floats(B, 300), slots=S, streams=streams)
r.run(gb, 4 * S)
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
r.run(gb, N)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"S={S}: host issue {1e3 * (t1 - t0) / N:.3f} ms/step, total {1e3 * (t2 - t0) / N:.3f} ms/step ({B * N / (t2 - t0):.0f} images/s)")
dist.destroy_process_group()
