"""lab (usage: headlab.py [LIB.so]): launch time of the fused fp32 head (two 5x5 convolutions + tanh/sigmoid + composite) at the benched shape (48 frames, 512x512,
64 channels) and for one frame.  Run on the GPU box: python tools/headlab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipercore_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])          # a variant library (tools/labvariant.sh)
from ipercore_amd import ops
from ipercore_amd.networks import packing

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
C, S = 64, 512
wpk = packing.pack_head(torch.randn(3, C, 5, 5, generator=g) * 0.03, torch.randn(1, C, 5, 5, generator=g) * 0.03).to(dev)
for B in (48, 8, 1):
    x = torch.randn(B, S, S, C, device=dev)
    bg = torch.randn(B, 3, S, S, device=dev)
    for _ in range(3):
        ops.head_compose(x, wpk, bg, want_pred=True, want_mask=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.head_compose(x, wpk, bg, want_pred=True, want_mask=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * S * S * 25 * C * 4
    print(f"B={B}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  {B * S * S * C * 4 / ms / 1e6:.0f} GB/s input")
    xq = x.view(B, S, S, C // 4, 4).permute(0, 3, 1, 2, 4).contiguous()          # the same values as channel-quad planes
    for _ in range(3):
        ops.head_compose(xq, wpk, bg, want_pred=True, want_mask=True, q4=True)
    e0.record()
    for _ in range(n):
        pq, mq, _ = ops.head_compose(xq, wpk, bg, want_pred=True, want_mask=True, q4=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    p0, m0, _ = ops.head_compose(x, wpk, bg, want_pred=True, want_mask=True)
    print(f"B={B} quad planes: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s   max |pred - NHWC form| = {float((pq - p0).abs().max()):.2e}")
