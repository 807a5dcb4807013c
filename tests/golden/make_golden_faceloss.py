#!/usr/bin/env python3
"""Generate tests/golden/golden_faceloss_v1.npz with the REFERENCE's own Sphere20a (models/networks/criterions/faceloss.py:203-285,
loaded in isolation - the file only needs torch) on seeded weights and inputs: the five feature outputs (sub-sampled) and the
FaceLoss.compute_loss value (:362-379) between two inputs.

    python tests/golden/make_golden_faceloss.py
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("LWG_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ipercore_amd import synthetic  # noqa: E402
from ipercore_amd.trainers import Sphere20aFeatures  # noqa: E402


def face_state_dict():
    """Seeded Sphere20a parameters independent of torch's RNG: synthetic.fill_state_dict (numpy), PReLU slopes moved to
    0.25 + 0.1 * (that value) so that every channel has its own slope.  tests/gpu_checks.py rebuilds the same dict."""
    shapes = {k: tuple(v.shape) for k, v in Sphere20aFeatures(None, allow_seeded=True).state_dict().items()}
    sd = {k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=13).items()}
    for k in sd:
        if k.startswith("relu"):
            sd[k] = 0.25 + 0.1 * sd[k]
    return sd


def main():
    spec = importlib.util.spec_from_file_location("ref_faceloss", os.path.join(REF, "iPERCore/models/networks/criterions/faceloss.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    net = m.Sphere20a(feature=True).eval()
    sd = face_state_dict()
    net.load_state_dict(sd, strict=True)
    x = torch.tensor(synthetic.uniform_image((2, 3, 112, 96), 70, "face_x"))
    y = torch.tensor(synthetic.uniform_image((2, 3, 112, 96), 71, "face_y"))
    with torch.no_grad():
        fx, fy = net(x), net(y)
    w = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]
    loss = sum(wi * torch.nn.functional.l1_loss(a, b) for wi, a, b in zip(w, fx, fy))
    out = {"loss": np.array(float(loss))}
    for i, f in enumerate(fx):
        out[f"fx{i}"] = f.numpy()[:, ::8] if f.dim() == 4 else f.numpy()
    dst = os.path.join(ROOT, "tests/golden/golden_faceloss_v1.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes", {k: v.shape for k, v in out.items()}, "loss", float(loss))


if __name__ == "__main__":
    main()
