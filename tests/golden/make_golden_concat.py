#!/usr/bin/env python3
"""Generate tests/golden/golden_concat_v1.npz from the REFERENCE's own classes (authoring container only): the two
input-concatenation baselines of its factory (networks/__init__.py:38-44) - ``InputConcatGenerator``
(generators/input_concat_resunet.py:182-307) and ``TextureWarpingGenerator`` (generators/texture_warping_resunet.py:8-112) - and
``MultiScaleDiscriminator`` (discriminators/multi_scale_dis.py:287-332, norm_type="instance", get_avg=False: its get_avg=True path
calls an undefined method).  Reduced-width configs at S = 64, seeded weights and inputs from ipercore_amd.synthetic.

    python tests/golden/make_golden_concat.py
"""
import hashlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("LWG_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
for m in ("cv2", "torchvision", "neural_renderer"):
    sys.modules[m] = types.ModuleType(m)
sys.path.insert(0, REF)

import torch  # noqa: E402

from ipercore_amd import synthetic  # noqa: E402

S = 64
NF, NRES, BGF = [64, 64, 128], 2, [64, 64, 128]


def concat_cfg(name, cond_nc, num_source=None):
    tsf = synthetic.AttrDict(norm_type="instance", cond_nc=cond_nc, n_res_block=NRES, num_filters=list(NF))
    if num_source is not None:
        tsf["num_source"] = num_source
    return synthetic.AttrDict(name=name, BGNet=synthetic.AttrDict(norm_type="instance", cond_nc=4, n_res_block=NRES, num_filters=list(BGF)), TSFNet=tsf)


def inputs():
    return (torch.tensor(synthetic.uniform_image((1, 1, 4, S, S), 10, "bg_inputs")),
            torch.tensor(synthetic.uniform_image((1, 2, 6, S, S), 8, "src_inputs")),
            torch.tensor(synthetic.uniform_image((1, 2, 6, S, S), 9, "tsf_inputs2")))


def main():
    from iPERCore.models.networks.generators.input_concat_resunet import InputConcatGenerator
    from iPERCore.models.networks.generators.texture_warping_resunet import TextureWarpingGenerator
    from iPERCore.models.networks.discriminators import MultiScaleDiscriminator
    bg_in, src_in, tsf_in = inputs()
    out = {}
    for name, cls, cfg in (("InputConcat", InputConcatGenerator, concat_cfg("InputConcat", 27, 4)),
                           ("TextureWarping", TextureWarpingGenerator, concat_cfg("TextureWarping", 6))):
        G = cls(cfg, temporal=False).eval()
        shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
        G.load_state_dict({k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=13).items()}, strict=True)
        with torch.no_grad():
            enc, _ = G.forward_src(src_in, only_enc=True)
            img, mask = G.forward_tsf(tsf_in[:, 0], enc)
            bg, imgs, masks = G(bg_in, src_in, tsf_in)
        out[f"{name}/keys_sha"] = np.array(hashlib.sha256("\n".join(f"{k}:{shapes[k]}" for k in sorted(shapes)).encode()).hexdigest())
        out[f"{name}/src_enc_shape"] = np.array(enc.shape)
        out[f"{name}/img"], out[f"{name}/mask"] = img.numpy(), mask.numpy()
        out[f"{name}/bg"], out[f"{name}/imgs"], out[f"{name}/masks"] = bg.numpy(), imgs.numpy(), masks.numpy()
    D = MultiScaleDiscriminator(6, 6, ndf=32, n_layers=3, max_nf_mult=8, norm_type="instance", use_sigmoid=False).eval()
    shapes = {k: tuple(v.shape) for k, v in D.state_dict().items()}
    D.load_state_dict({k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=17).items()}, strict=True)
    gx = torch.tensor(synthetic.uniform_image((2, 6, S, S), 30, "global_x"))
    lx = torch.tensor(synthetic.uniform_image((2, 6, S, S), 31, "local_x"))
    with torch.no_grad():
        outs = D(gx, lx, None, None, get_avg=False)
    out["multi_scale/keys_sha"] = np.array(hashlib.sha256("\n".join(f"{k}:{shapes[k]}" for k in sorted(shapes)).encode()).hexdigest())
    for i, o in enumerate(outs):
        out[f"multi_scale/out{i}"] = o.numpy()
    dst = os.path.join(ROOT, "tests/golden/golden_concat_v1.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(out), "entries", {k: v.shape for k, v in out.items() if v.ndim > 1})


if __name__ == "__main__":
    main()
