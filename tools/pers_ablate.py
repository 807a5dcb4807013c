#!/usr/bin/env python3
"""Lab: upper bounds for what fusing a helper kernel away would buy the personalization step: the step with that helper replaced by a
no-op (WRONG values, timing only).  python tools/pers_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench_personalize as bp  # noqa: E402
from ipercore_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
orig = {k: getattr(ops, k) for k in ("colsum", "act_bwd")}


def run(tag):
    r = bp.measure(dev, steps=10, warmup=4)
    print(f"{tag:28s} {r['ms_per_step']:.2f} ms / step", flush=True)


run("product")
ops.colsum = lambda x: torch.zeros(x.shape[-1], device=x.device)
run("no colsum (bias gradients)")
ops.colsum = orig["colsum"]
ops.act_bwd = lambda dy, y, act: dy
run("no act_bwd (ReLU masks)")
ops.act_bwd = orig["act_bwd"]
run("product again")
