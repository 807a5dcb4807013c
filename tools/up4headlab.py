"""Lab: the fused bf16 up4 + head launch (csrc/up4_head_bf16.hip) at the benched shape (20 x 512^2 x 128 -> 1024^2 frames): launch time against the two
launches it replaces, and - with --ts on a -DUH_LAB_TS variant library (tools/labvariant.sh uhts up4_head_bf16.hip -DUH_LAB_TS; python tools/up4headlab.py
--lib tools/lab/liblwg_uhts.so --ts) - the per-tile phase timeline of waves 0 and 7 (cycles between the stamps, median over the workgroups)."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--ts", action="store_true")
ap.add_argument("--frames", type=int, default=20)
args = ap.parse_args()
import shutil
if args.lib:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copy(os.path.join(root, "ipercore_amd", "liblwg_hip.so"), "/tmp/liblwg_keep.so")
    shutil.copy(args.lib, os.path.join(root, "ipercore_amd", "liblwg_hip.so"))
import numpy as np
import torch
from ipercore_amd import ops
from ipercore_amd.networks import packing
DEV = "cuda:0"
g = torch.Generator().manual_seed(1)
B, H, W = args.frames, 512, 512
w = torch.randn(128, 64, 4, 4, generator=g) / np.sqrt(512)
bs = torch.randn(64, generator=g) * 0.1
specs = [packing.spec_to(s_, DEV) for s_ in packing.pack_conv_transpose(w, bs)]
head16 = packing.pack_head_bf16(torch.randn(3, 64, 5, 5, generator=g) * 0.05, torch.randn(1, 64, 5, 5, generator=g) * 0.05).to(DEV)
x = torch.randn(B, H, W, 128, generator=g).to(torch.bfloat16).to(DEV)
bg = torch.randn(1, 3, 2 * H, 2 * W, generator=g).to(DEV)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


y = torch.empty(B, 2 * H, 2 * W, 64, device=DEV, dtype=torch.bfloat16)
if not args.ts:
    t_up = timed(lambda: ops.conv_transpose2d(x, specs, y, act=ops.ACT_RELU))
    t_hd = timed(lambda: ops.head_compose(y, head16, bg, want_pred=True, want_mask=True))
    t_f = timed(lambda: ops.up4_head_compose_bf16(x, specs, head16, bg, want_pred=True, want_mask=True))
    print(f"{B} frames 512^2 x 128 -> 1024^2: up4 {t_up:.0f} us + head {t_hd:.0f} us = {t_up + t_hd:.0f} us; fused {t_f:.0f} us ({(t_up + t_hd) / t_f:.2f}x)")
else:
    pred, mask, _ = ops.up4_head_compose_bf16(x, specs, head16, bg, want_pred=True, want_mask=True)
    torch.cuda.synchronize()
    mask.zero_()
    pred, mask, _ = ops.up4_head_compose_bf16(x, specs, head16, bg, want_pred=True, want_mask=True)
    torch.cuda.synchronize()
    ts = mask.view(-1).view(torch.int64)[:256 * 16].view(256, 2, 8).cpu().numpy().astype(np.float64)
    names = ["(stamp 0)", "top barrier", "K loops (+ next halo requested)", "epilogue into T", "barrier (T complete)", "phase 2 MFMAs + partial stores", "wait next tile's loads", "barrier + final pass"]
    for wv, tag in ((0, "wave 0"), (1, "wave 7")):
        d = np.diff(ts[:, wv, :], axis=1)
        ok = (ts[:, wv, 0] > 0) & (d > 0).all(axis=1)
        med = np.median(d[ok], axis=0)
        print(tag, "tiles with stamps", int(ok.sum()), "total", int(med.sum()), "cycles;", ", ".join(f"{n}: {int(v)}" for n, v in zip(names[1:], med)))
    tile = np.median(ts[:, 0, 7] - ts[:, 0, 0])
    print("(stamp 0 = in front of the top barrier; cycle counter ticks at the shader clock)")
if args.lib:
    shutil.copy("/tmp/liblwg_keep.so", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ipercore_amd", "liblwg_hip.so"))
