"""Lab: the clip WITH the output stage (device uint8 + pinned D2H ring + PNG threads) at several frame batches / encoder thread counts.
usage: outlab.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shutil
import tempfile
import torch
from ipercore_amd import synthetic as syn
from ipercore_amd.output import FrameWriter

case = syn.build_case(image_size=512, n_frames=300, ns=2)
im = syn.make_imitator(case, frame_batch=300)
tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
for _ in range(2):
    im.synthesize(tgt, "smooth")
torch.cuda.synchronize()
for fb in (16, 30, 50, 100):
    for workers in (32, 64, 96):
        d = tempfile.mkdtemp(prefix="lwg_outlab_")
        try:
            w = FrameWriter(d, prefix="pred_", workers=workers, ring=max(4, 200 // fb))
            n = 300 // fb * fb
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(0, n, fb):
                tsf8, Tst, _ = im.make_inputs_for_tsf(im.src_info, tgt[s:s + fb], "smooth", t=s)
                w.submit(im.forward(tsf8, Tst)[0], s)
            torch.cuda.synchronize()
            tg = time.perf_counter() - t0
            w.close()
            dt = time.perf_counter() - t0
            print(f"fb {fb:3d} workers {workers:3d}: {n / dt:7.1f} frames/s with files, gpu side {n / tg:7.1f}", flush=True)
        finally:
            shutil.rmtree(d, ignore_errors=True)
