#!/usr/bin/env python3
"""Generate tests/golden/golden_temporal_v1.npz: the reference's own AttentionLWBGenerator(temporal=True).forward_tsf with
temporal attention inputs (temp_enc_outs / temp_res_outs / Ttt, attlwb_spade_resunet.py:208-252,:480-535) on seeded inputs.

    python tests/golden/make_golden_temporal.py        (authoring container only: imports /root/reference)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

import torch  # noqa: E402

from ipercore_amd import synthetic  # noqa: E402

S, NS, NT = 64, 2, 2


def temporal_inputs(golden_v1):
    """Seeded inputs shared with the tests."""
    Tst = torch.tensor(golden_v1["render/Tst"]).view(1, NS, S, S, 2)
    Ttt = torch.roll(Tst, shifts=(3, -2), dims=(2, 3)).clone()            # a different, still smooth flow field
    src_inputs = torch.tensor(synthetic.uniform_image((1, NS, 6, S, S), 8, "src_inputs"))
    tmp_inputs = torch.tensor(synthetic.uniform_image((NT, 1, 6, S, S), 30, "tmp_inputs"))   # NT previous [pred, cond] frames
    tsf_inputs = torch.tensor(synthetic.uniform_image((1, 6, S, S), 9, "tsf_inputs"))
    return Tst, Ttt, src_inputs, tmp_inputs, tsf_inputs


def main():
    mg.install_stubs()
    from iPERCore.models.networks.generators.attlwb_spade_resunet import AttentionLWBGenerator
    g1 = np.load(os.path.join(mg.ROOT, "tests/golden/golden_v1.npz"))
    nf, nres, bgf = [64, 64, 128], 2, [64, 64, 128]
    G = AttentionLWBGenerator(mg.gen_cfg(nf, nres, bgf), temporal=True).eval()
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    sd = synthetic.fill_state_dict(shapes, seed=7)
    G.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    Tst, Ttt, src_inputs, tmp_inputs, tsf_inputs = temporal_inputs(g1)
    with torch.no_grad():
        enc, res = G.forward_src(src_inputs, only_enc=True)
        tencs, tress = zip(*[G.forward_src(tmp_inputs[k:k + 1], only_enc=True) for k in range(NT)])   # post_update, one frame each
        tenc = [torch.cat([tencs[k][l] for k in range(NT)], dim=0) for l in range(len(enc))]
        tres = [torch.cat([tress[k][l] for k in range(NT)], dim=0) for l in range(len(res))]
        img, mask = G.forward_tsf(tsf_inputs, enc, res, Tst, temp_enc_outs=tenc, temp_res_outs=tres, Ttt=Ttt)
        img0, mask0 = G.forward_tsf(tsf_inputs, enc, res, Tst)
    out = {"img": img.numpy(), "mask": mask.numpy(), "diff_vs_no_temporal": np.array(float((img - img0).abs().max()))}
    dst = os.path.join(mg.ROOT, "tests/golden/golden_temporal_v1.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes; temporal changes the image by", out["diff_vs_no_temporal"])


if __name__ == "__main__":
    main()
