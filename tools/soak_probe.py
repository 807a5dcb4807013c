"""Lab: 3600 single-frame hipGraph replays (bitwise stable), then precision-mode switches (the graph is keyed on the mode and re-captured)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ipercore_amd import synthetic as syn  # noqa: E402
case = syn.build_case(image_size=512, n_frames=300, ns=2)
im = syn.make_imitator(case, frame_batch=1)
tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
ref = None
t0 = time.perf_counter()
for rep in range(12):
    v = im.synthesize(tgt, "smooth")
    torch.cuda.synchronize()
    if ref is None:
        ref = v.clone()
    assert torch.equal(v, ref), f"rep {rep}: graph replays diverged"
print(f"soak: 12 x 300 single-frame graph replays, bitwise stable, {(time.perf_counter()-t0):.1f} s", flush=True)
im.frame_batch = 32
big = im.synthesize(tgt, "smooth")
torch.cuda.synchronize()
print("batch-32 equals graph frames:", bool(torch.equal(big, ref)), flush=True)
# precision switch + back: graph must be re-captured (key includes the precision mode)
im.generator.conv_precision = "bf16"
im.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
im.frame_batch = 1
b = im.synthesize(tgt[:16], "smooth")
im.frame_batch = 8
b8 = im.synthesize(tgt[:16], "smooth")
torch.cuda.synchronize()
print("bf16 graph frames equal bf16 batch-8 frames:", bool(torch.equal(b, b8)), flush=True)
im.generator.conv_precision = "fp32"
im.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
im.frame_batch = 1
f = im.synthesize(tgt[:16], "smooth")
torch.cuda.synchronize()
print("fp32 again equals first run:", bool(torch.equal(f, ref[:16])), flush=True)
