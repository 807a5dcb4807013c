#!/usr/bin/env python3
"""Generate tests/golden/golden_swapper_v1.npz with the REFERENCE's own Swapper pieces (authoring container only):
``Swapper.get_selected_info_by_part_name`` (models/imitator.py:502-546), ``FlowCompositionForSwapper``
``add_rendered_selected_f2pts`` / ``merge_uv_img`` (models/flowcomposition.py:794-856) and ``make_trans_flow(use_selected_f2pts=True)``
(:514-582) for two people (2 + 1 source images) on seeded synthetic inputs.  Stubs as in make_golden.py / make_golden_source.py.

    python tests/golden/make_golden_swapper.py
"""
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_source as mgs  # noqa: E402

import torch  # noqa: E402

from ipercore_amd import synthetic  # noqa: E402

S = 128
PEOPLE = ((2, 20), (1, 40))                    # (number of source images, seed) per person
PART_SETS = {"head_body": (["head"], ["body"]), "leftover": (["upper"], ["left_leg", "right_foot"])}


def fids_sha(lists):
    return hashlib.sha256(";".join(",".join(str(int(f)) for f in sorted(l)) for l in lists).encode()).hexdigest()


def main():
    mg.install_stubs()
    mgs.install_cv2()
    from iPERCore.models.flowcomposition import FlowCompositionForSwapper
    from iPERCore.models.imitator import Swapper
    from iPERCore.tools.human_digitalizer.bodynets.batch_smplh import SMPLH

    tmp = synthetic.tmp_asset_dir()
    cfgdir = os.path.join(mg.REF, "assets/configs/pose3d")
    opt = mg.AttrDict(face_path=synthetic.write_smpl_faces_npy(os.path.join(tmp, "smpl_faces.npy")),
                      fim_enc_path=os.path.join(cfgdir, "mapper_fim_enc.txt"), uv_map_path=os.path.join(cfgdir, "mapper_uv.txt"),
                      part_path=os.path.join(cfgdir, "smpl_part_info.json"), map_name="uv_seg", image_size=S, only_vis=False,
                      num_source=3, time_step=1, **mgs.OPT_KS)
    cwd = os.getcwd()
    os.chdir(mg.REF)
    fc = FlowCompositionForSwapper(opt)
    os.chdir(cwd)
    smplh = SMPLH(model_path=synthetic.write_smplh_pickle(os.path.join(tmp, "smplh_synth.pkl"), seed=0))
    shim = types.SimpleNamespace(flow_comp=fc)
    out = {}
    with torch.no_grad():
        fc.make_uv_setup(1, 3, 1, torch.device("cpu"))
        for tag, parts in PART_SETS.items():
            _, fids = Swapper.get_selected_info_by_part_name(shim, list(parts), primary_ids=0)
            out[f"{tag}/fids_sha"] = np.array(fids_sha(fids))
            out[f"{tag}/fids_count"] = np.array([len(f) for f in fids])
            infos = []
            for i, (ns, seed) in enumerate(PEOPLE):
                smpls = synthetic.smpl_sequence(ns, seed=seed, pose_dim=72)
                info = smplh.get_details(torch.tensor(smpls), 0, links_ids=None)
                info["num_source"] = ns
                fc.add_rendered_f2verts_fim_wim(info, use_morph=False, get_uv_info=True)
                info["uv_img"] = torch.tensor(synthetic.uniform_image((1, 3, S, S), seed + 5, "uv_img"))
                fc.add_rendered_selected_f2pts(info, [fids[i]] * ns)
                infos.append(info)
            out[f"{tag}/uv_img"] = fc.merge_uv_img(infos).numpy()
            sel = torch.cat([i["selected_f2pts"] for i in infos], dim=0)
            out[f"{tag}/selected_count"] = np.array(int((sel[:, :, 0, 0] != -2).sum()))
            ref = smplh.get_details(torch.tensor(synthetic.smpl_sequence(1, seed=60, pose_dim=72)), 0, links_ids=None)
            fc.add_rendered_f2verts_fim_wim(ref, use_morph=False, get_uv_info=False)
            Tst, _ = fc.make_trans_flow(1, 3, 1, {"selected_f2pts": sel}, None, ref, temporal=False, use_selected_f2pts=True)
            out[f"{tag}/Tst"] = Tst.numpy().astype(np.float32)
    dst = os.path.join(mg.ROOT, "tests/golden/golden_swapper_v1.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
