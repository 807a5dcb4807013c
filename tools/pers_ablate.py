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
names = ("colsum", "act_bwd", "conv2d_wgrad_unpacked", "conv2d_wgrad", "norm_bwd", "lwb_attention_bwd", "adam_step_dev", "norm_fwd")
orig = {k: getattr(ops, k) for k in names}


def run(tag):
    r = bp.measure(dev, steps=10, warmup=4)
    print(f"{tag:44s} {r['ms_per_step']:.2f} ms / step", flush=True)


def with_patch(tag, **patch):
    for k, v in patch.items():
        setattr(ops, k, v)
    try:
        run(tag)
    finally:
        for k in patch:
            setattr(ops, k, orig[k])


run("product")
with_patch("no colsum (bias gradients)", colsum=lambda x: torch.zeros(x.shape[-1], device=x.device))
with_patch("no weight-gradient kernels at all", conv2d_wgrad_unpacked=lambda x0, spec, dy, dw, *a, **k: dw,
           conv2d_wgrad=lambda x0, spec, dy, **k: torch.empty(spec.ntaps * spec.Cin, spec.N, device=dy.device))
with_patch("no norm backward", norm_bwd=lambda dy, y, x, mean, rstd, gamma=None, act=0: (dy, None if gamma is None else dy, None if gamma is None else dy))
with_patch("no attention backward", lwb_attention_bwd=lambda q, Ks, Vs, bk, bv, T, dout, src_batched=False: (dout, Ks, Vs))
with_patch("no Adam kernels", adam_step_dev=lambda *a, **k: None)
run("product again")
