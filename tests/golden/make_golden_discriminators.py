#!/usr/bin/env python3
"""Generate tests/golden/golden_discriminators_v1.npz with the REFERENCE's own discriminator classes
(models/networks/discriminators/multi_scale_dis.py: GlobalDiscriminator / GlobalLocalDiscriminator / GlobalBodyHeadDiscriminator over
patch_dis.py PatchDiscriminator; the two files only need torch) on seeded weights and inputs: every output map of the three composed
networks with the augmented-background branch on, the get_avg value, and crop_img on its own.  One head box is degenerate (the
reference drops it).

    python tests/golden/make_golden_discriminators.py
"""
import importlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("LWG_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ipercore_amd import synthetic  # noqa: E402

CFG = dict(cond_nc=6, bg_cond_nc=4, ndf=32, n_layers=3, max_nf_mult=8, norm_type="instance", use_sigmoid=False)
S = 128
BODY = [[20, 100, 10, 120], [30, 90, 5, 115]]
HEAD = [[50, 80, 10, 45], [60, 60, 15, 40]]                 # the second one is degenerate (min_x == max_x)
NAMES = ("patch_global", "patch_global_local", "patch_global_body_head")


def seeded_state_dict(module, seed):
    """Seeded parameters independent of torch's RNG (synthetic.fill_state_dict, numpy, keyed by parameter name)."""
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    return {k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=seed).items()}


def inputs():
    x = torch.tensor(synthetic.uniform_image((2, 6, S, S), 81, "dis_x"))
    bg_x = torch.tensor(synthetic.uniform_image((2, 4, S, S), 82, "dis_bg_x"))
    return x, bg_x, torch.tensor(BODY), torch.tensor(HEAD)


def main():
    # the two reference files import each other relatively: give them a package without running iPERCore/__init__
    pkg = types.ModuleType("refdis")
    pkg.__path__ = [os.path.join(REF, "iPERCore/models/networks/discriminators")]
    sys.modules["refdis"] = pkg
    m = importlib.import_module("refdis.multi_scale_dis")
    cfg = synthetic.AttrDict(**CFG)
    x, bg_x, body, head = inputs()
    out = {}
    for name, cls in zip(NAMES, (m.GlobalDiscriminator, m.GlobalLocalDiscriminator, m.GlobalBodyHeadDiscriminator)):
        D = cls(cfg, use_aug_bg=True).eval()
        D.load_state_dict(seeded_state_dict(D, 17), strict=True)
        with torch.no_grad():
            outs, avg = D({"x": x, "bg_x": bg_x, "body_rects": body, "head_rects": head, "get_avg": True})
        out[f"{name}/n"] = np.array(len(outs))
        out[f"{name}/avg"] = np.array(float(avg))
        for i, o in enumerate(outs):
            out[f"{name}/out{i}"] = o.numpy()
    out["crop_body"] = m.crop_img(x, body, fact=2).numpy()[:, ::2]
    out["crop_head"] = m.crop_img(x, head, fact=4).numpy()
    dst = os.path.join(ROOT, "tests/golden/golden_discriminators_v1.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
