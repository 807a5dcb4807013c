#!/usr/bin/env python3
"""Generate tests/golden/golden_smpl24_v1.npz with the REFERENCE's own 24-joint SMPL (bodynets/batch_smpl.py:283-436 +
BaseSMPL.get_details) on a synthetic smpl_model.pkl-shaped parameter file and seeded poses.

    python tests/golden/make_golden_smpl24.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

import torch  # noqa: E402

from ipercore_amd import synthetic  # noqa: E402


def main():
    mg.install_stubs()
    from iPERCore.tools.human_digitalizer.bodynets.batch_smpl import SMPL
    tmp = synthetic.tmp_asset_dir()
    net = SMPL(model_path=synthetic.write_smpl_pickle(os.path.join(tmp, "smpl24_synth.pkl"), seed=0)).eval()
    smpls = synthetic.smpl_sequence(3, seed=80, pose_dim=72)
    offsets = 0.002 * synthetic.uniform_image((6890, 3), 81, "offsets")
    with torch.no_grad():
        d = net.get_details(torch.tensor(smpls), torch.tensor(offsets))
    out = {"verts_sub": d["verts"].numpy()[:, ::10], "j3d": d["j3d"].numpy(), "j2d": d["j2d"].numpy(),
           "verts_mean": d["verts"].numpy().mean(axis=(1, 2))}
    dst = os.path.join(mg.ROOT, "tests/golden/golden_smpl24_v1.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
