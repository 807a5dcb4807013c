#!/usr/bin/env python3
"""Generate tests/golden/golden_tsf256_v1.npz: the REFERENCE's own AttLWB-SPADE generator (imported from /root/reference, authoring container
only; stubs as in make_golden.py) at S = 256, full width ([64, 128, 256], 6 residual blocks, ns = 2) - forward_src + forward_tsf on seeded
synthetic inputs and a smooth synthetic flow with out-of-range samples.  Every other generator golden is S = 64 / 128: this one pins the
size-dependent behaviour (flow resizing to 128 / 64 / 32, grid_sample at the coarser levels, the decoder at four times the area).

    python tests/golden/make_golden_tsf256.py

Stored: the transferred image / mask sub-sampled 2x (exact values), their means, the last encoder / residual features sub-sampled.  Inputs are
regenerated in the test from ipercore_amd.synthetic (same seeds) and synthetic_flow() below (imported by the test)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S, NS = 256, 2


def synthetic_flow(S=S, ns=NS):
    """(1, ns, S, S, 2) fp32 grid_sample coordinates: the identity grid bent by low-frequency waves, a band pushed out of [-1, 1] and a block
    of the reference's 'no correspondence' value -2 (what cal_bc_transform writes outside the body)."""
    y, x = np.meshgrid(np.linspace(-1, 1, S, dtype=np.float64), np.linspace(-1, 1, S, dtype=np.float64), indexing="ij")
    out = np.zeros((1, ns, S, S, 2), dtype=np.float64)
    for s in range(ns):
        out[0, s, :, :, 0] = x + 0.15 * np.sin(3.1 * y + 0.7 * s) + 0.05 * np.cos(7.3 * x)
        out[0, s, :, :, 1] = y + 0.12 * np.cos(2.3 * x - 0.4 * s) - 0.04 * np.sin(5.9 * y)
        out[0, s, : S // 8, :, :] += 0.9                                   # partly outside: zeros padding
        out[0, s, S // 2: S // 2 + S // 6, S // 3: S // 3 + S // 5, :] = -2.0   # background pixels
    return out.astype(np.float32)


def main():
    from tests.golden import make_golden as mg
    mg.install_stubs()
    import torch
    from iPERCore.models.networks.generators.attlwb_spade_resunet import AttentionLWBGenerator
    from ipercore_amd import synthetic
    nf, nres, bgf = [64, 128, 256], 6, [64, 128, 128, 256]
    G = AttentionLWBGenerator(mg.gen_cfg(nf, nres, bgf), temporal=False).eval()
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    sd = synthetic.fill_state_dict(shapes, seed=7)
    G.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    src_inputs = torch.tensor(synthetic.uniform_image((1, NS, 6, S, S), 8, "src_inputs_256"))
    tsf_inputs = torch.tensor(synthetic.uniform_image((1, 6, S, S), 9, "tsf_inputs_256"))
    Tst = torch.tensor(synthetic_flow())
    with torch.no_grad():
        enc, res = G.forward_src(src_inputs, only_enc=True)
        img, mask = G.forward_tsf(tsf_inputs, enc, res, Tst)
    out = {"img_sub": img.numpy()[:, :, ::2, ::2], "mask_sub": mask.numpy()[:, :, ::2, ::2],
           "img_mean": np.array(img.double().mean().item()), "mask_mean": np.array(mask.double().mean().item()),
           "img_abs_mean": np.array(img.abs().double().mean().item()),
           "enc2_sub": enc[-1].numpy()[:, ::16, ::2, ::2], "res_last_sub": res[-1].numpy()[:, ::16, ::2, ::2]}
    dst = os.path.join(ROOT, "tests/golden/golden_tsf256_v1.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
