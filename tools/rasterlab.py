#!/usr/bin/env python3
"""Time the rasterizer alone on the bench geometry (8 frames @512) - tuning aid for csrc/raster.hip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipercore_amd import ops, synthetic as syn
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
case = syn.build_case(image_size=S, n_frames=8, ns=2, num_filters=[64, 64, 128], n_res=1, bg_filters=[64, 64, 128])
im = syn.make_imitator(case, frame_batch=8)
tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
ref = im.body_rec.get_details(im.swap_params(im.src_info["cam"][0:1], im.src_info["shape"][0:1], tgt, "smooth").contiguous(), 0)
fv, _ = ops.project_faces(ref["verts"], ref["cam"].contiguous(), im.flow_comp.render.smpl_faces, want_f2pts=False)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fv = fv[:B].contiguous()
for _ in range(3):
    fim, wim = ops.rasterize_fim_wim(fv, S)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    fim, wim = ops.rasterize_fim_wim(fv, S)
b.record(); torch.cuda.synchronize()
# candidate statistics per 16x16 tile from the setup boxes (host side, frame 0)
import numpy as np
ws_rec = None
f = fv[0].cpu().numpy()
p = 0.5 * (f[:, :, 0:2] * S + S - 1)
front = ~(((f[:, 2, 1] - f[:, 0, 1]) * (f[:, 1, 0] - f[:, 0, 0])) < ((f[:, 1, 1] - f[:, 0, 1]) * (f[:, 2, 0] - f[:, 0, 0])))
x0 = np.floor(p[:, :, 0].min(1)) - 1; x1 = np.ceil(p[:, :, 0].max(1)) + 1
y0 = np.floor(p[:, :, 1].min(1)) - 1; y1 = np.ceil(p[:, :, 1].max(1)) + 1
T = S // 16
cnt = np.zeros((T, T), dtype=np.int64)
for i in np.nonzero(front)[0]:
    ta, tb = int(max(0, x0[i]) // 16), int(min(S - 1, x1[i]) // 16)
    c, d = int(max(0, y0[i]) // 16), int(min(S - 1, y1[i]) // 16)
    cnt[c:d + 1, ta:tb + 1] += 1
print("tiles", T * T, "nonempty", int((cnt > 0).sum()), "candidates/tile: mean(nonempty)", float(cnt[cnt > 0].mean()), "p90", float(np.percentile(cnt[cnt > 0], 90)), "max", int(cnt.max()), "sum", int(cnt.sum()))
print(f"S={S}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per {B}-frame rasterization, cover {(fim >= 0).float().mean().item():.3f}, "
      f"checksum {int(fim.long().sum())} {float(wim.double().sum()):.6f}")
