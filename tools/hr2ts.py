"""Lab: per-workgroup phase timeline of the bf16 halo-tile kernel (csrc/conv_igemm_bf16.hip, lwg_conv_bf16_hr2_kernel) on a -DLWG_HR2_TS variant library
(tools/labvariant.sh hr2ts conv_igemm_bf16.hip -DLWG_HR2_TS; python tools/hr2ts.py --lib tools/lab/liblwg_hr2ts.so [--shapes res64,shared64,skip0] [--batch-mul 3]):
wave 0 of every workgroup stamps kernel entry, first barrier passed, K-loop exit, end of its epilogue (cycles, medians over the workgroups)."""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--shapes", default="res64,shared64,skip0,skip1,shared256")
ap.add_argument("--batch-mul", type=int, default=3)
args = ap.parse_args()
import torch
from ipercore_amd import _lib
_lib.LIB_PATH = os.path.abspath(args.lib)
from ipercore_amd import ops
import tools.bf16lab as lab
lab.BATCH_MUL = args.batch_mul
BF = torch.bfloat16
for name in args.shapes.split(","):
    x0, x1, yshape, launches = lab.build(name)
    spec, kw = launches[0]
    y = torch.empty(yshape, device="cuda:0", dtype=BF)
    x0b, x1b = x0.to(BF), None if x1 is None else x1.to(BF)
    B, H, W, N = yshape
    nwg = ((H + 7) // 8) * ((W + 15) // 16) * B * max(1, N // 64)
    stamps = torch.zeros(nwg * 16 + 64, device="cuda:0", dtype=torch.float32)
    for _ in range(3):
        ops.conv2d(x0b, spec, y, x1=x1b, act=ops.ACT_RELU, res=stamps.view(BF))
    torch.cuda.synchronize()
    t = stamps.view(torch.int64)[: nwg * 8].view(nwg, 8).cpu().double()
    ok = (t[:, 3] > 0) & (t[:, 0] > 0)
    u = t[ok]
    d = [float((u[:, i + 1] - u[:, i]).median()) for i in range(3)]
    Cin = x0.shape[3] + (0 if x1 is None else x1.shape[3])
    print(f"[hr2ts] {name}: workgroups with stamps {int(ok.sum())} of {nwg}; chunks {Cin // 64}; medians: entry -> first barrier {d[0]:.0f}, K loop {d[1]:.0f} "
          f"({d[1] / (Cin // 64):.0f} per chunk), epilogue {d[2]:.0f}, total {float((u[:, 3] - u[:, 0]).median()):.0f} cycles")
