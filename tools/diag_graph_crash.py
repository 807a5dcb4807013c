#!/usr/bin/env python3
"""Which earlier stage of a long-lived process breaks the hipGraph capture of the personalization step?  (The default bench.py run
crashed in its personalize_step extra when that ran in-process after the synthesis extras.)
    python -X faulthandler tools/diag_graph_crash.py stage[,stage...]      stages: synth, pipelined, output, split, b1, bf16
then runs bench_personalize.measure() in the same process and prints its step time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler  # noqa: E402

faulthandler.enable()
import torch  # noqa: E402

import bench  # noqa: E402
import bench_personalize  # noqa: E402
from ipercore_amd import ops, synthetic as pu  # noqa: E402


def main():
    stages = [s for s in (sys.argv[1] if len(sys.argv) > 1 else "").split(",") if s]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if stages:
        case = pu.build_case(image_size=512, n_frames=48, ns=2)
        im = pu.make_imitator(case, frame_batch=16, device=dev)
        tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
        timer = bench.ConvTimer()
        ops.CONV_HOOK = lambda b, M, spec, epi=0, info=None: timer(b, M, spec, epi, info, 4)
        render = lambda: im.synthesize(tgt, "smooth")          # noqa: E731
        last = render()
        for st in stages:
            if st == "synth":
                render()
            elif st == "pipelined":
                print(st, bench.pipelined(im, render, 48, 1, 2, 3), flush=True)
            elif st == "output":
                print(st, bench.with_output(im, tgt, 16, 32, 0), flush=True)
            elif st == "split":
                print(st, bench.split_products(im, render, 48, 1, 2, last), flush=True)
            elif st == "b1":
                print(st, bench.b1_latency(im, tgt, timer, n_frames=16), flush=True)
            elif st == "bf16":
                print(st, str(bench.novel_view_1024_bf16(dev, timer, 1, 1))[:200], flush=True)
            torch.cuda.synchronize()
            print("stage done:", st, flush=True)
    r = bench_personalize.measure(dev, steps=3, warmup=1, size=512)
    print("personalize after", stages, "->", r["ms_per_step"], "ms,", r["config"]["step"], flush=True)


if __name__ == "__main__":
    main()
