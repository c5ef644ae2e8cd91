"""Development probe: YOLO11s batch 64 against the fp32 oracle and the bf16-storage emulation, one-sided boxes listed."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch
import bench
from oracle import yolo11_oracle as oracle
from oracle.parity_gate import parity_gate
dev = torch.device("cuda:0")
scale, batch, imgsz = "s", 64, 640
x = torch.rand(batch, 3, imgsz, imgsz, generator=torch.Generator().manual_seed(0)).to(dev)
specs, sd, eng = bench.build_model(scale, dev, batch, imgsz, imgsz, x)
x8 = (x * 255).to(torch.uint8)
og, cg = eng.detect(x8, 0.25, 0.7, max_det=3000, graph=False)
torch.cuda.synchronize()
torch.set_num_threads(8)
for n_or in (2, 6):
    with torch.inference_mode():
        y32, _ = oracle.forward(sd, x8[:n_or].float().cpu() / 255.0, scale)
        ref = oracle.non_max_suppression(y32, 0.25, 0.7, max_det=3000)
        yb, _ = oracle.forward(sd, x8[:n_or].float().cpu() / 255.0, scale, storage="bf16")
        refb = oracle.non_max_suppression(yb, 0.25, 0.7, max_det=3000)
    y_hip = eng.forward(x8, graph=False)[:n_or].float().cpu()
    g = parity_gate(ref, [og[i, : int(cg[i])].cpu() for i in range(n_or)], 0.25, 0.7, band=0.08, ref_y=y32, got_y=y_hip, verbose=True)
    print(n_or, {k: g[k] for k in ("ref", "det", "matched", "unexplained", "near_threshold", "nms_flip", "argmax_flip", "max_dconf", "ok")})
    gf = parity_gate(ref, refb, 0.25, 0.7, band=0.08, ref_y=y32, got_y=yb, verbose=True)
    print("format", {k: gf[k] for k in ("ref", "det", "matched", "unexplained", "near_threshold", "nms_flip", "argmax_flip", "max_dconf", "ok")})
