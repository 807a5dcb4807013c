#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz by running the REFERENCE's own Python (imported from
/root/reference, authoring container only) on seeded synthetic inputs.

    python tests/golden/make_golden.py

Recipe: stub the three modules that are absent here (`neural_renderer`, `cv2`,
`torchvision`), shim `np.int`/`np.float`, then import the reference modules unchanged.  The
`neural_renderer` stub is wired to the oracle for the three calls the wrapper makes (look_at,
vertices_to_faces, rasterize_face_index_map_and_weight_map), so the fixtures pin everything the reference's
own code does AROUND the external rasterizer; the rasterizer itself stays "parity unpinned".

Inputs are regenerated in the tests from `ipercore_amd.synthetic` (same seeds); only outputs (and the
S=64 fim/wim maps used as inputs of the flow functions) are stored.
"""
import hashlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("LWG_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

np.int = int      # noqa: removed in NumPy >= 1.24; used at reference mesh.py:312,317
np.float = float  # noqa

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from ipercore_amd import synthetic  # noqa: E402
from oracle import lwg_oracle as orc  # noqa: E402

S = 64


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def install_stubs():
    nr = types.ModuleType("neural_renderer")
    nr.look_at = lambda vertices, eye: orc.look_at(vertices, eye)
    nr.vertices_to_faces = lambda vertices, faces: torch.stack(
        [vertices[b][faces[b].long()] for b in range(vertices.shape[0])], dim=0)

    def _rast(faces, image_size, anti_aliasing=False, near=0.1, far=100, eps=1e-3):
        return orc.rasterize_fim_wim(faces.detach().numpy(), image_size, near, far)
    nr.rasterize_face_index_map_and_weight_map = _rast
    sys.modules["neural_renderer"] = nr
    for m in ("cv2", "torchvision"):
        sys.modules[m] = types.ModuleType(m)
    sys.path.insert(0, REF)


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def gen_cfg(num_filters, n_res, bg_filters):
    return AttrDict(name="AttLWB-SPADE",
                    BGNet=AttrDict(norm_type="instance", cond_nc=4, n_res_block=n_res, num_filters=bg_filters),
                    SIDNet=AttrDict(norm_type="None", cond_nc=6, n_res_block=n_res, num_filters=num_filters),
                    TSFNet=AttrDict(norm_type="instance", cond_nc=6, n_res_block=n_res, num_filters=num_filters))


def main():
    install_stubs()
    from iPERCore.tools.human_digitalizer.renders.nmr import SMPLRenderer
    from iPERCore.tools.human_digitalizer.bodynets.batch_smplh import SMPLH
    from iPERCore.tools.utils.geometry.cam_pose_utils import WeakPerspectiveCamera
    from iPERCore.models.networks.generators.attlwb_spade_resunet import AttentionLWBGenerator

    out = {}
    tmp = synthetic.tmp_asset_dir()
    cfgdir = os.path.join(REF, "assets/configs/pose3d")
    face_npy = synthetic.write_smpl_faces_npy(os.path.join(tmp, "smpl_faces.npy"))
    pkl = synthetic.write_smplh_pickle(os.path.join(tmp, "smplh_synth.pkl"), seed=0)

    # ---------------- 1. renderer tables (bit-exact index/constant tables) ----------------
    render = SMPLRenderer(face_path=face_npy,
                          fim_enc_path=os.path.join(cfgdir, "mapper_fim_enc.txt"),
                          uv_map_path=os.path.join(cfgdir, "mapper_uv.txt"),
                          part_path=os.path.join(cfgdir, "smpl_part_info.json"),
                          front_path=os.path.join(cfgdir, "front_body.json"),
                          head_path=os.path.join(cfgdir, "head.json"),
                          facial_path=os.path.join(cfgdir, "front_facial.json"),
                          map_name="uv_seg", tex_size=3, image_size=S, fill_back=False,
                          anti_aliasing=True, background_color=(0, 0, 0), has_front=True, top_k=3)
    for name in ("smpl_faces", "obj_faces", "map_fn", "front_map_fn", "f_img2uvs", "face_k_nearest",
                 "f_uvs2img", "coords", "img2uv_sampler"):
        buf = getattr(render, name).numpy()
        out["table_sha/" + name] = np.array(sha(buf))
        out["table_shape/" + name] = np.array(buf.shape)
        out["table_dtype/" + name] = np.array(str(buf.dtype))
    out["table_head/map_fn"] = render.map_fn.numpy()[[0, 1, 2, 13775, 13776]]
    out["table_head/face_k_nearest"] = render.face_k_nearest.numpy()[:8]

    # ---------------- 2. SMPL-H body model ----------------
    smplh = SMPLH(model_path=pkl)
    smpls72 = synthetic.smpl_sequence(3, seed=1, pose_dim=72)
    smpls156 = synthetic.smpl_sequence(2, seed=2, pose_dim=156)
    offsets = (0.005 * synthetic._rs(3, "offsets").standard_normal((6890, 3))).astype(np.float32)
    r = synthetic._rs(4, "links")
    links = np.stack([r.randint(0, 6890, size=40), r.randint(0, 6890, size=40)], axis=1).astype(np.int64)
    with torch.no_grad():
        d72 = smplh.get_details(torch.tensor(smpls72), torch.tensor(offsets), links_ids=links)
        d156 = smplh.get_details(torch.tensor(smpls156), 0, links_ids=None)
    out["smplh72/verts"] = d72["verts"].numpy()
    out["smplh72/j3d"] = d72["j3d"].numpy()
    out["smplh72/j2d"] = d72["j2d"].numpy()
    out["smplh156/verts_sub"] = d156["verts"].numpy()[:, ::5]
    out["smplh156/j3d"] = d156["j3d"].numpy()

    # ---------------- 3. camera swap ----------------
    cams = synthetic._rs(5, "cams").uniform(0.5, 1.0, size=(3, 1, 3)).astype(np.float32)
    out["cam_swap/smooth"] = WeakPerspectiveCamera.cam_swap(
        torch.tensor(cams[0]), torch.tensor(cams[1]), torch.tensor(cams[2]), "smooth").numpy()

    # ---------------- 3b. stabilize (sequence-global pre-pass), with an injected jump ----------------
    seq = synthetic.smpl_sequence(14, seed=13, pose_dim=72)
    seq[4:9, 2] -= np.array([0.15, 0.45, 0.6, 0.4, 0.1], dtype=np.float32)
    with torch.no_grad():
        wcam = WeakPerspectiveCamera(smplh)
        out["stabilize/out"] = wcam.stabilize(torch.tensor(seq)).numpy()
        fy = wcam.infer_smpl_foot_y(torch.tensor(seq[:, 3:-10]), torch.tensor(seq[0:1, -10:]).repeat(14, 1))
        out["stabilize/jumps"] = np.array(wcam.get_jump_mask((fy + torch.tensor(seq[:, 2])).numpy())[0]).reshape(-1, 2)

    # ---------------- 4. renderer wrapper + flow functions (S=64) ----------------
    with torch.no_grad():
        cam = d72["cam"][0:1].clone()
        verts = d72["verts"][0:1].clone()
        f2pts, fim, wim = render.render_fim_wim(cam, verts, smpl_faces=True)
        cond, _ = render.encode_fim(fim=fim, transpose=True)
        uv_fim, uv_wim = render.render_uv_fim_wim(1)
        f_uvs2img = render.get_f_uvs2img(1)
        Tuv2t = render.cal_bc_transform(f_uvs2img.clone(), fim, wim)
        uv_img = torch.tensor(synthetic.uniform_image((1, 3, S, S), 6, "uv_img"))
        syn = F.grid_sample(uv_img, Tuv2t)
        # two "sources": frames 1 and 2 of the 72-dim sequence
        src_f2pts, _, _ = render.render_fim_wim(d72["cam"][1:3].clone(), d72["verts"][1:3].clone(), smpl_faces=True)
        Tst = render.cal_bc_transform(src_f2pts, fim.repeat(2, 1, 1), wim.repeat(2, 1, 1, 1))
        vis = render.get_vis_f2pts(f2pts, fim)
    out["render/f2pts_sha"] = np.array(sha(f2pts.numpy()))
    out["render/f2pts_head"] = f2pts.numpy()[0, :4]
    out["render/fim"] = fim.numpy()
    out["render/wim"] = wim.numpy()
    out["render/cond_sha"] = np.array(sha(cond.numpy()))
    out["render/uv_fim_sha"] = np.array(sha(uv_fim.numpy()))
    out["render/uv_cover"] = np.array(int((uv_fim != -1).sum()))
    out["render/Tuv2t"] = Tuv2t.numpy()
    out["render/syn"] = syn.numpy()
    out["render/Tst"] = Tst.numpy()
    out["render/vis_f2pts_sha"] = np.array(sha(vis.numpy()))

    # ---------------- 5. generator (tiny config stored fully; full config at S=64) ----------------
    for tag, nf, nres, bgf in (("tiny", [64, 64, 128], 2, [64, 64, 128]), ("full", [64, 128, 256], 6, [64, 128, 128, 256])):
        cfg = gen_cfg(nf, nres, bgf)
        G = AttentionLWBGenerator(cfg, temporal=False).eval()
        shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
        sd = synthetic.fill_state_dict(shapes, seed=7)
        G.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
        ns = 2
        src_inputs = torch.tensor(synthetic.uniform_image((1, ns, 6, S, S), 8, "src_inputs"))
        tsf_inputs = torch.tensor(synthetic.uniform_image((1, 6, S, S), 9, "tsf_inputs"))
        bg_inputs = torch.tensor(synthetic.uniform_image((1, 1, 4, S, S), 10, "bg_inputs"))
        with torch.no_grad():
            enc, res = G.forward_src(src_inputs, only_enc=True)
            img, mask = G.forward_tsf(tsf_inputs, enc, res, Tst.view(1, ns, S, S, 2))
            bg = G.forward_bg(bg_inputs)
        out[f"gen_{tag}/keys_sha"] = np.array(hashlib.sha256("\n".join(
            f"{k}:{shapes[k]}" for k in sorted(shapes)).encode()).hexdigest())
        out[f"gen_{tag}/nparams"] = np.array(sum(int(np.prod(s)) for s in shapes.values()))
        out[f"gen_{tag}/enc2_sub"] = enc[-1].numpy()[:, ::8]
        out[f"gen_{tag}/res_last_sub"] = res[-1].numpy()[:, ::8]
        out[f"gen_{tag}/img"] = img.numpy()
        out[f"gen_{tag}/mask"] = mask.numpy()
        out[f"gen_{tag}/bg"] = bg.numpy()
        if tag == "full":
            with open(os.path.join(ROOT, "tests/golden/attlwb_spade_state_dict_keys.txt"), "w") as fp:
                for k in G.state_dict().keys():
                    fp.write(f"{k} {tuple(shapes[k])}\n")

    dst = os.path.join(ROOT, "tests/golden/golden_v1.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(out), "entries")


if __name__ == "__main__":
    main()
