#!/usr/bin/env python3
"""Lab: what one rank of the 8-GPU clip run does (38 of 300 frames at 512x512), on one GPU: time per shard for different chunk plans.
python tools/shard_probe.py [frames=38]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ipercore_amd import ops, synthetic as syn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 38
case = syn.build_case(image_size=512, n_frames=300, ns=2)
im = syn.make_imitator(case, frame_batch=32)
tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
for _ in range(3):
    im.synthesize(tgt[:64], "smooth")
torch.cuda.synchronize()
plans = {"32+6": [32, 6], "24+14": [24, 14], "38": [38], "16+16+6": [16, 16, 6], "19+19": [19, 19], "8x4+6": [8, 8, 8, 8, 6]}
if n != 38:
    plans = {"fb32": [32] * (n // 32) + ([n % 32] if n % 32 else []), "one": [n]}
for rep in range(2):
    for name, plan in plans.items():
        times = []
        for it in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            off = 0
            for m in plan:
                im.frame_batch = m
                u8 = ops.frames_to_u8(im.synthesize(tgt[off:off + m], "smooth", t0=off))
                off += m
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        print(f"rep {rep} plan {name:8s}: {min(times):7.2f} ms min, {sorted(times)[len(times) // 2]:7.2f} median  -> {n / min(times) * 1e3:7.1f} frames/s per rank", flush=True)
