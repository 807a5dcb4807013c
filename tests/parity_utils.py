"""Shared builders for the parity tests, __graft_entry__.smoke() and bench.py: one synthetic clip,
run through the HIP product path (``run_hip``) and through the CPU oracle (``run_oracle``).  Test
infrastructure: it is the only place where product and oracle meet."""
import numpy as np
import torch

from ipercore_amd import synthetic
from ipercore_amd.geometry import mesh


from ipercore_amd.synthetic import AttrDict, build_case, gen_cfg, make_imitator  # noqa: E402,F401  (product-side builders)


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


def face_k_nearest_table(topo=None, k=3):
    """renders/nmr.py:186-190 with FlowComposition's top_k = 3 (flowcomposition.py:66): (nf, k) nearest same-part faces in UV space."""
    topo = topo or mesh.load_topology()
    f_img2uvs = mesh.get_f2vts(mesh.obj_from_topology(topo, "fim"), z=1).astype(np.float32)
    parts = {str(n): topo["part_" + str(n)] for n in topo["part_names"]}
    return mesh.find_part_k_nearest_faces(f_img2uvs, mesh.get_part_ids(f_img2uvs.shape[0], parts), k=k)


def oracle_source(case, tables=None, src_override=None):
    """The oracle's version of the cached source state (what source_setup leaves in src_info).

    src_override = (cam, verts): use these posed source vertices instead of the oracle's own skinning, so that
    the rasterizer sees bit-identical inputs on both sides (a 1e-7 vertex difference flips silhouette pixels,
    which is a discontinuity of the algorithm, not an error of either implementation)."""
    from oracle import lwg_oracle as orc
    tables = tables or oracle_tables()
    model = orc.SMPLHModel(case.smplh)
    sd = {k: torch.tensor(v) for k, v in case.state.items()}
    src = orc.smplh_get_details(model, case.src_smpl, 0, None)
    own_verts = src["verts"]
    if src_override is not None:
        src["cam"], src["verts"] = src_override
    f2pts, fim, wim = orc.render_fim_wim(src["cam"], src["verts"], tables["smpl_faces"], case.S)
    cond = orc.encode_fim(tables["map_fn"], fim)
    src_inputs = torch.cat([torch.tensor(case.src_img)[0], cond], dim=1).unsqueeze(0)
    with torch.no_grad():
        feats = orc.gen_forward_src(sd, src_inputs, n_down=len(case.num_filters), n_res=case.n_res)
    info = {"cam": src["cam"], "shape": src["shape"], "offsets": 0, "links_ids": None, "uv_img": torch.tensor(case.uv_img),
            "bg": torch.tensor(case.bg_img), "f2pts": f2pts, "feats": feats, "own_verts": own_verts, "fim": fim}
    if case.opt.get("only_vis", False):       # flowcomposition.py:559-562: Tst from the visible source faces (+ k nearest) only
        info["only_vis"], info["only_vis_f2pts"] = True, orc.get_vis_f2pts(f2pts, fim, face_k_nearest_table())
    return model, tables, sd, info


def run_oracle(case, cam_strategy="smooth", frames=None, return_all=False, src_override=None, ref_override=None):
    """(n,3,S,S) CPU tensor: the reference algorithm for every frame (or the listed frame indices).
    ref_override: {t: (cam (1,3), verts (1,nv,3))} posed target vertices to rasterize instead of the oracle's."""
    from oracle import lwg_oracle as orc
    model, tables, sd, info = oracle_source(case, src_override=src_override)
    tgt = torch.tensor(case.tgt_smpls)
    if cam_strategy == "smooth":
        tgt = orc.stabilize(model, tgt)
    first_cam = tgt[0:1, 0:3].clone()
    outs, extra = [], []
    for t in (range(tgt.shape[0]) if frames is None else frames):
        with torch.no_grad():
            r = orc.imitate_frame(model, tables, sd, info, tgt[t], first_cam, case.S, cam_strategy,
                                  ref_override=None if ref_override is None else ref_override[t])
        outs.append(r["pred"])
        extra.append(r)
    pred = torch.cat(outs, dim=0)
    return (pred, extra) if return_all else pred


def staged_parity(case, frame_batch=8, cam_strategy="smooth", frames=None, device="cuda:0", imitator=None, return_want=False,
                  e2e_fim=True):
    """HIP path vs oracle, stage by stage (the end-to-end map is discontinuous at silhouette pixels, so each stage
    is compared on identical inputs): (1) skinned vertices HIP vs oracle; (2) fim/wim exact given the HIP vertices;
    (3) generator input / flows; (4) final frames.  Returns a metrics dict; the caller asserts.
    ``frames``: the frame indices the oracle renders (default all).  ``e2e_fim``: additionally REPORT the un-overridden end-to-end
    agreement (expect >= 99.9 %): the oracle rasterizes ITS OWN skinned vertices (which differ from the HIP ones by
    <= 1e-5) and the face-index map is compared with the HIP one pixel by pixel -> ``fim_agree_e2e_min`` / ``_mean``."""
    im = imitator or make_imitator(case, frame_batch, device=device)
    tgt = im.prepare_sequence(case.tgt_smpls, cam_strategy)
    idx = list(range(tgt.shape[0])) if frames is None else list(frames)
    keep = set(idx)
    preds, refs = [], {}
    fb = im.frame_batch
    for s in range(0, tgt.shape[0], fb):
        tsf8, Tst, ref = im.make_inputs_for_tsf(im.src_info, tgt[s:s + fb], cam_strategy, t=s, want_aux=True)
        preds.append(im.forward(tsf8, Tst)[0])
        for i in range(ref["verts"].shape[0]):
            if s + i in keep:
                refs[s + i] = {"cam": ref["cam"][i:i + 1].cpu(), "verts": ref["verts"][i:i + 1].cpu(), "fim": ref["fim"][i].cpu(),
                               "wim": ref["wim"][i].cpu(), "Tst": Tst[i].cpu(), "tsf8": tsf8[i].cpu()}
    got = torch.cat(preds, dim=0).cpu()
    src_ov = (im.src_info["cam"].cpu(), im.src_info["verts"].cpu())
    want, extra = run_oracle(case, cam_strategy, frames=idx, return_all=True, src_override=src_ov,
                             ref_override={t: (refs[t]["cam"], refs[t]["verts"]) for t in idx})
    m = {"frames": idx, "frame_batch": fb}
    m["src_verts_max"] = (extra[0]["src_own_verts"] - src_ov[1]).abs().max().item()
    m["src_fim_equal"] = bool(torch.equal(im.src_info["fim"].cpu(), extra[0]["src_fim"]))
    m["verts_max"] = max((extra[k]["own_verts"] - refs[t]["verts"]).abs().max().item() for k, t in enumerate(idx))
    m["fim_equal"] = all(bool(torch.equal(refs[t]["fim"], extra[k]["fim"][0])) for k, t in enumerate(idx))
    m["wim_max"] = max((refs[t]["wim"] - extra[k]["wim"][0]).abs().max().item() for k, t in enumerate(idx))
    m["Tst_max"] = max((refs[t]["Tst"] - extra[k]["Tst"][0]).abs().max().item() for k, t in enumerate(idx))
    m["tsf_inputs_max"] = max((refs[t]["tsf8"][..., :6].permute(2, 0, 1) - extra[k]["tsf_inputs"][0]).abs().max().item()
                              for k, t in enumerate(idx))
    d = (got[idx] - want).abs()
    m["pred_max"], m["pred_mean"] = d.max().item(), d.mean().item()
    m["pred_finite"] = bool(torch.isfinite(got).all())
    if e2e_fim:
        from oracle import lwg_oracle as orc
        tables = oracle_tables()
        rates = []
        for k, t in enumerate(idx):
            _, fim_own, _ = orc.render_fim_wim(extra[k]["own_cam"], extra[k]["own_verts"], tables["smpl_faces"], case.S)
            rates.append((fim_own[0] == refs[t]["fim"]).float().mean().item())
        m["fim_agree_e2e_min"], m["fim_agree_e2e_mean"] = min(rates), float(np.mean(rates))
    if return_want:
        m["want"] = want
    return m, got, im


# ---- once-per-source stage: inputs shared by the golden script, the CPU pin test and the GPU check
def source_stage_inputs(S, ns=2, seed=20):
    """Same seeds as tests/golden/make_golden_source.py::source_inputs."""
    from ipercore_amd import synthetic as syn
    return syn.smpl_sequence(ns, seed=seed, pose_dim=72), syn.uniform_image((1, ns, 3, S, S), seed + 1, "src_img")


def fg_masks_from_sil(sil):
    """tests/golden/make_golden_source.py::fg_masks_from_sil (synthetic segmentation derived from the rendered silhouette)."""
    s = sil.clone()
    sh = torch.zeros_like(s)
    sh[:, :, 1:, 2:] = s[:, :, :-1, :-2]
    dil = (torch.nn.functional.max_pool2d(s, 3, 1, 1) > 0).float()
    m = ((sh + dil) > 0).float()
    m[:, :, :, : s.shape[-1] // 5] = 0
    return m


def oracle_source_stage(S, ks, smplh_dict=None, topo=None, verts_cam=None):
    """The oracle's process_source on the seeded source-stage inputs.  ks: dict(conf_erode_ks, out_dilate_ks, bg_ks).
    verts_cam = (cam, verts) overrides the oracle's own skinning (GPU check: identical rasterizer inputs)."""
    from ipercore_amd import synthetic as syn
    from oracle import lwg_oracle as orc
    topo = topo or mesh.load_topology()
    fim_obj = mesh.obj_from_topology(topo, "fim")
    tables = oracle_tables(topo)
    smpls, img = source_stage_inputs(S)
    if verts_cam is None:
        model = orc.SMPLHModel(smplh_dict or syn.smplh_model_dict(seed=0))
        d = orc.smplh_get_details(model, smpls, 0, None)
        cam, verts = d["cam"], d["verts"]
    else:
        cam, verts = verts_cam
    _, fim, _ = orc.render_fim_wim(cam, verts, tables["smpl_faces"], S)
    cond = orc.encode_fim(tables["map_fn"], fim)
    obj_f2pts, obj_fim, _ = orc.render_fim_wim(cam, verts, fim_obj["faces"].astype(np.int32), S)
    f_img2uvs = mesh.get_f2vts(fim_obj, z=1).astype(np.float32)
    uv_fim, uv_wim = orc.render_uv_fim_wim(f_img2uvs, 1, S)
    parts = {str(n): topo["part_" + str(n)] for n in topo["part_names"]}
    fkn = mesh.find_part_k_nearest_faces(f_img2uvs, mesh.get_part_ids(f_img2uvs.shape[0], parts), k=3)
    fg = fg_masks_from_sil((fim != -1).float().unsqueeze(1))
    out = orc.process_source(torch.tensor(img), cond, fim, obj_f2pts, obj_fim, uv_fim, uv_wim, fkn, masks=1.0 - fg, **ks)
    out.update(fg=fg, src_img=torch.tensor(img), smpls=smpls, obj_f2pts=obj_f2pts, uv_fim=uv_fim, uv_wim=uv_wim)
    return out
