"""Lab: per-tile phase times of the fused bf16 transposed convolution (a -DLAB_TS build from tools/labvariant.sh: wave 0 of every workgroup
stamps s_memtime at start / halo in LDS / after each parity's K loop / after its stores are issued / after everything is acknowledged).
usage: up4ts.py LIB.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ipercore_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from ipercore_amd import ops
from ipercore_amd.networks import packing

dev, BF = "cuda:0", torch.bfloat16
B, H, Cin, N = 20, 512, 128, 64
g = torch.Generator().manual_seed(5)
w = (torch.randn(Cin, N, 4, 4, generator=g) * (Cin * 4) ** -0.5).to(BF).float()
specs = [packing.spec_to(s, dev) for s in packing.pack_conv_transpose(w, 0.1 * torch.randn(N, generator=g))]
x = torch.randn(B, H, H, Cin, generator=g).to(BF).to(dev)
y = torch.empty(B, 2 * H, 2 * H, N, device=dev, dtype=BF)
for _ in range(3):
    ops.conv_transpose2d(x, specs, y, act=ops.ACT_RELU)
a = ops.conv_args(x, specs[0], y, act=ops.ACT_RELU)
panel = torch.stack([ops._w16hr(s, False)[0] for s in specs]).contiguous()
a.w = ops._ptr(panel, BF)
tiles = B * (H // 8) * (H // 16)
ts = torch.zeros(tiles, 12, dtype=torch.int64, device=dev)
a.res = ts.data_ptr()
torch.cuda.synchronize()
_lib.check(_lib.lib().lwg_conv_transpose4_nhwc_bf16(a, ops._stream()), "up4")
torch.cuda.synchronize()
t = ts.cpu().numpy().astype(np.float64)
n = int((t[0] > 0).sum())
t = t[:, :n]
print("stamps per tile:", n, " tiles:", tiles, " (s_memtime is per XCD: only differences inside a tile mean something)")
names = ["halo in LDS"] + sum([[f"K loop p{k}", f"stores issued p{k}"] for k in range((n - 3) // 2)], []) + ["all acknowledged"]
d = np.diff(t, axis=1)
for k, nm in enumerate(names):
    print(f"  {nm:22s} mean {d[:, k].mean():9.0f}  p10 {np.percentile(d[:, k], 10):9.0f}  p50 {np.percentile(d[:, k], 50):9.0f}  p90 {np.percentile(d[:, k], 90):9.0f}")
tot = t[:, -1] - t[:, 0]
print(f"  {'tile total':22s} mean {tot.mean():9.0f}  p10 {np.percentile(tot, 10):9.0f}  p50 {np.percentile(tot, 50):9.0f}  p90 {np.percentile(tot, 90):9.0f}")
