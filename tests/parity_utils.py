"""Shared builders for the parity tests, __graft_entry__.smoke() and bench.py: one synthetic clip,
run through the HIP product path (``run_hip``) and through the CPU oracle (``run_oracle``).  Test
infrastructure: it is the only place where product and oracle meet."""
import numpy as np
import torch

from ipercore_amd import synthetic
from ipercore_amd.geometry import mesh
from ipercore_amd.networks import generator_param_shapes


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def gen_cfg(num_filters, n_res, bg_filters):
    return AttrDict(name="AttLWB-SPADE",
                    BGNet=AttrDict(norm_type="instance", cond_nc=4, n_res_block=n_res, num_filters=list(bg_filters)),
                    SIDNet=AttrDict(norm_type="None", cond_nc=6, n_res_block=n_res, num_filters=list(num_filters)),
                    TSFNet=AttrDict(norm_type="instance", cond_nc=6, n_res_block=n_res, num_filters=list(num_filters)))


def build_case(image_size=512, num_filters=(64, 128, 256), n_res=6, bg_filters=(64, 128, 128, 256), n_frames=8, ns=2,
               seed=0):
    S = int(image_size)
    shapes = generator_param_shapes(num_filters, n_res, bg_filters)
    case = AttrDict(
        S=S, ns=ns, n_frames=n_frames, num_filters=list(num_filters), n_res=n_res, bg_filters=list(bg_filters),
        smplh=synthetic.smplh_model_dict(seed=seed),
        state=synthetic.fill_state_dict(shapes, seed=seed + 7),
        src_smpl=synthetic.smpl_sequence(ns, seed=seed + 11, pose_dim=72),
        tgt_smpls=synthetic.smpl_sequence(n_frames, seed=seed + 12, pose_dim=72),
        uv_img=synthetic.uniform_image((1, 3, S, S), seed + 6, "uv_img"),
        bg_img=synthetic.uniform_image((1, 3, S, S), seed + 5, "bg_img"),
        src_img=synthetic.uniform_image((1, ns, 3, S, S), seed + 4, "src_img"),
    )
    case.opt = AttrDict(image_size=S, gen_name="AttLWB-SPADE", temporal=False, only_vis=False, map_name="uv_seg",
                        smpl_model_hand=case.smplh, neural_render_cfg=AttrDict(Generator=gen_cfg(num_filters, n_res, bg_filters)))
    return case


def make_imitator(case, frame_batch=8, device="cuda:0"):
    from ipercore_amd.imitator import Imitator
    im = Imitator(case.opt, device=torch.device(device), frame_batch=frame_batch)
    im.generator.load_state_dict({k: torch.tensor(v) for k, v in case.state.items()}, strict=True)
    im.generator.to(im.device)
    im.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
    return im


def run_hip(case, frame_batch=8, cam_strategy="smooth", imitator=None):
    im = imitator or make_imitator(case, frame_batch)
    tgt = im.prepare_sequence(case.tgt_smpls, cam_strategy)
    return im.synthesize(tgt, cam_strategy)


def oracle_tables(topo=None):
    topo = topo or mesh.load_topology()
    uv, fim = mesh.obj_from_topology(topo, "uv"), mesh.obj_from_topology(topo, "fim")
    return {"smpl_faces": topo["faces_uv"].astype(np.int32),
            "map_fn": mesh.create_mapping("uv_seg", fim, contain_bg=True).astype(np.float32),
            "f_uvs2img": mesh.get_f2vts(uv, z=1)[:, :, 0:2].astype(np.float32)}


def oracle_source(case, tables=None):
    """The oracle's version of the cached source state (what source_setup leaves in src_info)."""
    from oracle import lwg_oracle as orc
    tables = tables or oracle_tables()
    model = orc.SMPLHModel(case.smplh)
    sd = {k: torch.tensor(v) for k, v in case.state.items()}
    src = orc.smplh_get_details(model, case.src_smpl, 0, None)
    f2pts, fim, wim = orc.render_fim_wim(src["cam"], src["verts"], tables["smpl_faces"], case.S)
    cond = orc.encode_fim(tables["map_fn"], fim)
    src_inputs = torch.cat([torch.tensor(case.src_img)[0], cond], dim=1).unsqueeze(0)
    with torch.no_grad():
        feats = orc.gen_forward_src(sd, src_inputs, n_down=len(case.num_filters), n_res=case.n_res)
    info = {"cam": src["cam"], "shape": src["shape"], "offsets": 0, "links_ids": None, "uv_img": torch.tensor(case.uv_img),
            "bg": torch.tensor(case.bg_img), "f2pts": f2pts, "feats": feats}
    return model, tables, sd, info


def run_oracle(case, cam_strategy="smooth", frames=None, return_all=False):
    """(n,3,S,S) CPU tensor: the reference algorithm for every frame (or the listed frame indices)."""
    from oracle import lwg_oracle as orc
    model, tables, sd, info = oracle_source(case)
    tgt = torch.tensor(case.tgt_smpls)
    if cam_strategy == "smooth":
        tgt = orc.stabilize(model, tgt)
    first_cam = tgt[0:1, 0:3].clone()
    outs, extra = [], []
    for t in (range(tgt.shape[0]) if frames is None else frames):
        with torch.no_grad():
            r = orc.imitate_frame(model, tables, sd, info, tgt[t], first_cam, case.S, cam_strategy)
        outs.append(r["pred"])
        extra.append(r)
    pred = torch.cat(outs, dim=0)
    return (pred, extra) if return_all else pred
