"""Lab: the personalization step's losses per step, direct fp32 engine vs the Winograd engine, from the same seeded networks and inputs (eager steps).
The first step's losses show the engines' numerical difference alone; later steps add the divergence of two training runs.
usage: lossdiv.py [--steps n] [--use-vgg] [--use-face]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--use-vgg", action="store_true")
ap.add_argument("--use-face", action="store_true")
args = ap.parse_args()
import torch
import bench_personalize as bp

dev = torch.device("cuda", 0)
rows = {}
for prec in ("fp32", "winograd", "fp32"):
    keep = {}
    bp.measure(dev, steps=1, warmup=0, size=512, use_vgg=args.use_vgg, use_face=args.use_face, precision=prec, graph=False, _keep=keep, _host_probe=False)
    tr = keep["trainer"]
    out = []
    for _ in range(args.steps):
        lg, ld = tr.optimize_parameters()
        out.append((float(lg), float(ld)))
    rows.setdefault(prec, []).append(out)
a, b, c = rows["fp32"][0], rows["winograd"][0], rows["fp32"][1]
print("step (after 2 updates)   loss_G fp32 / winograd (rel diff) [fp32 rerun rel diff]      loss_D fp32 / winograd")
for i in range(args.steps):
    print(f"{i:3d}  {a[i][0]:.6f} {b[i][0]:.6f} ({abs(a[i][0] - b[i][0]) / abs(a[i][0]):.1e}) [{abs(a[i][0] - c[i][0]) / abs(a[i][0]):.1e}]   {a[i][1]:.6f} {b[i][1]:.6f}")
