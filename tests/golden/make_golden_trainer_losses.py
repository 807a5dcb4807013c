#!/usr/bin/env python3
"""Generate tests/golden/golden_trainer_losses_v1.npz with the REFERENCE's own LWGTrainer.optimize_G / optimize_D
(tools/trainers/lwg_trainer.py:732-832, called unbound on an attribute bag: the methods only read the tensors ``set_input`` left on the
trainer, the criterions and ``_train_opts``) and its own GlobalDiscriminator / LSGANLoss / TVLoss on seeded tensors and weights:
every loss term of the generator step, the discriminator loss and the d_real / d_fake averages.  cv2 / torchvision /
neural_renderer are stubbed (imported by the package, not used by these methods).

    python tests/golden/make_golden_trainer_losses.py
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("LWG_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ipercore_amd import synthetic  # noqa: E402
from tests.golden.make_golden_discriminators import seeded_state_dict  # noqa: E402

S, NS, NT = 64, 2, 1
DCFG = dict(cond_nc=6, bg_cond_nc=4, ndf=32, n_layers=3, max_nf_mult=8, norm_type="instance", use_sigmoid=False)
LAMBDAS = dict(lambda_rec=10.0, lambda_tsf=7.0, lambda_D_prob=1.5, lambda_mask=2.0, lambda_mask_smooth=0.3, lambda_face=5.0, use_face=False)


def tensors():
    """The tensors of one personalization step (bs = 1), seeded by name; masks strictly inside (0, 1) for the BCE."""
    u = lambda shape, seed, name: torch.tensor(synthetic.uniform_image(shape, seed, name))       # noqa: E731
    return {"fake_bg": u((1, 1, 3, S, S), 301, "fake_bg"), "fake_src_imgs": u((1, NS, 3, S, S), 302, "fake_src"),
            "fake_tsf_imgs": u((1, NT, 3, S, S), 303, "fake_tsf"),
            "fake_masks": u((1, NS + NT, 1, S, S), 304, "fake_masks") * 0.45 + 0.5,
            "real_src": u((1, NS, 3, S, S), 305, "real_src"), "real_tsf": u((1, NT, 3, S, S), 306, "real_tsf"),
            "real_bg": u((1, 3, S, S), 307, "real_bg"), "body_mask": (u((1, NS + NT, 1, S, S), 308, "body_mask") > 0).float(),
            "input_G_tsf": u((1, NT, 6, S, S), 309, "input_G_tsf")}


def main():
    sys.path.insert(0, REF)
    for m in ("cv2", "torchvision", "torchvision.models", "torchvision.transforms", "neural_renderer", "visdom", "tensorboardX"):
        if m not in sys.modules:
            try:
                __import__(m)
            except Exception:
                sys.modules[m] = types.ModuleType(m)
    import iPERCore.tools.trainers.lwg_trainer as lt
    from iPERCore.models.networks.criterions import LSGANLoss, TVLoss
    from iPERCore.models.networks.discriminators.multi_scale_dis import GlobalDiscriminator
    D = GlobalDiscriminator(synthetic.AttrDict(**DCFG), use_aug_bg=False)
    D.load_state_dict(seeded_state_dict(D, 23), strict=True)
    t = tensors()
    bag = types.SimpleNamespace(
        D=D, crt_gan=LSGANLoss(), crt_l1=torch.nn.L1Loss(), crt_tsf=torch.nn.L1Loss(), crt_mask=torch.nn.BCELoss(), crt_tv=TVLoss(),
        _train_opts=synthetic.AttrDict(**LAMBDAS), _use_gan=True, _loss_g_face=0.0, _loss_g_adv=0.0,
        _real_src=t["real_src"], _real_tsf=t["real_tsf"], _real_bg=t["real_bg"], _body_mask=t["body_mask"],
        _input_G_tsf=t["input_G_tsf"], _body_bbox=None, _head_bbox=None)
    with torch.no_grad():
        loss_g = lt.LWGTrainer.optimize_G(bag, t["fake_bg"], t["fake_src_imgs"], t["fake_tsf_imgs"], t["fake_masks"])
        loss_d = lt.LWGTrainer.optimize_D(bag, t["fake_bg"], t["fake_tsf_imgs"])
    out = {"loss_G": float(loss_g), "loss_D": float(loss_d), "g_rec": float(bag._loss_g_rec), "g_tsf": float(bag._loss_g_tsf),
           "g_adv": float(bag._loss_g_adv), "g_mask": float(bag._loss_g_mask), "g_mask_smooth": float(bag._loss_g_smooth),
           "d_real": float(bag._d_real), "d_fake": float(bag._d_fake)}
    dst = os.path.join(ROOT, "tests/golden/golden_trainer_losses_v1.npz")
    np.savez_compressed(dst, **{k: np.array(v) for k, v in out.items()})
    print("wrote", dst, out)


if __name__ == "__main__":
    main()
