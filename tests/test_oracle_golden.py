"""Pins the CPU oracle (oracle/) against fixtures produced by the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest
import torch

from ipercore_amd import synthetic
from ipercore_amd.geometry import mesh
from oracle import lwg_oracle as orc

S = 64


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _smplh():
    return orc.SMPLHModel(synthetic.smplh_model_dict(seed=0))


def _tables(topo):
    uv, fim = mesh.obj_from_topology(topo, "uv"), mesh.obj_from_topology(topo, "fim")
    return {
        "smpl_faces": topo["faces_uv"].astype(np.int32),
        "map_fn": mesh.create_mapping("uv_seg", fim, contain_bg=True).astype(np.float32),
        "f_uvs2img": mesh.get_f2vts(uv, z=1)[:, :, 0:2].astype(np.float32),
        "f_img2uvs": mesh.get_f2vts(fim, z=1).astype(np.float32),
    }


def _details72():
    smpls72 = synthetic.smpl_sequence(3, seed=1, pose_dim=72)
    offsets = (0.005 * synthetic._rs(3, "offsets").standard_normal((6890, 3))).astype(np.float32)
    r = synthetic._rs(4, "links")
    links = np.stack([r.randint(0, 6890, size=40), r.randint(0, 6890, size=40)], axis=1).astype(np.int64)
    return orc.smplh_get_details(_smplh(), smpls72, torch.tensor(offsets), links)


def test_smplh_matches_reference(golden):
    d = _details72()
    assert np.abs(d["verts"].numpy() - golden["smplh72/verts"]).max() <= 1e-5
    assert np.abs(d["j3d"].numpy() - golden["smplh72/j3d"]).max() <= 1e-5
    assert np.abs(d["j2d"].numpy() - golden["smplh72/j2d"]).max() <= 1e-5
    d2 = orc.smplh_get_details(_smplh(), synthetic.smpl_sequence(2, seed=2, pose_dim=156), 0, None)
    assert np.abs(d2["verts"].numpy()[:, ::5] - golden["smplh156/verts_sub"]).max() <= 1e-5
    assert np.abs(d2["j3d"].numpy() - golden["smplh156/j3d"]).max() <= 1e-5


def test_cam_swap(golden):
    cams = synthetic._rs(5, "cams").uniform(0.5, 1.0, size=(3, 1, 3)).astype(np.float32)
    got = orc.cam_swap(torch.tensor(cams[0]), torch.tensor(cams[1]), torch.tensor(cams[2]), "smooth")
    assert np.array_equal(got.numpy(), golden["cam_swap/smooth"])


def _jump_sequence():
    seq = synthetic.smpl_sequence(14, seed=13, pose_dim=72)
    seq[4:9, 2] -= np.array([0.15, 0.45, 0.6, 0.4, 0.1], dtype=np.float32)
    return seq


def test_stabilize_matches_reference(golden):
    got = orc.stabilize(_smplh(), _jump_sequence())
    assert len(golden["stabilize/jumps"]) >= 1          # the fixture really exercises the jump branch
    assert np.abs(got.numpy() - golden["stabilize/out"]).max() <= 1e-5


def test_render_wrapper_and_flows(golden, topo):
    t = _tables(topo)
    d = _details72()
    f2pts, fim, wim = orc.render_fim_wim(d["cam"][0:1], d["verts"][0:1], t["smpl_faces"], S)
    # the wrapper (projection, y flips, look_at) is bit-exact; fim/wim come from the same C oracle
    assert sha(f2pts.numpy()) == str(golden["render/f2pts_sha"])
    assert np.array_equal(fim.numpy(), golden["render/fim"])
    assert np.array_equal(wim.numpy(), golden["render/wim"])
    cond = orc.encode_fim(t["map_fn"], fim)
    assert sha(cond.numpy()) == str(golden["render/cond_sha"])
    uv_fim, _ = orc.render_uv_fim_wim(t["f_img2uvs"], 1, S)
    assert sha(uv_fim.numpy()) == str(golden["render/uv_fim_sha"])
    uv_img = torch.tensor(synthetic.uniform_image((1, 3, S, S), 6, "uv_img"))
    tsf_inputs, Tuv2t = orc.make_tsf_inputs(uv_img, t["f_uvs2img"], cond, fim, wim)
    assert np.abs(Tuv2t.numpy() - golden["render/Tuv2t"]).max() <= 1e-6
    assert np.abs(tsf_inputs[:, 0:3].numpy() - golden["render/syn"]).max() <= 1e-5
    src_f2pts, _, _ = orc.render_fim_wim(d["cam"][1:3], d["verts"][1:3], t["smpl_faces"], S)
    Tst = orc.make_trans_flow(src_f2pts, fim, wim)
    assert np.abs(Tst[0].numpy() - golden["render/Tst"]).max() <= 1e-6


def _gen_case(golden, tag, nf, nres, bgf):
    from ipercore_amd.networks import generator_param_shapes
    shapes = generator_param_shapes(nf, nres, bgf)
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(golden[f"gen_{tag}/nparams"])
    keys = hashlib.sha256("\n".join(f"{k}:{tuple(shapes[k])}" for k in sorted(shapes)).encode()).hexdigest()
    assert keys == str(golden[f"gen_{tag}/keys_sha"])
    sd = {k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=7).items()}
    ns = 2
    src_inputs = torch.tensor(synthetic.uniform_image((1, ns, 6, S, S), 8, "src_inputs"))
    tsf_inputs = torch.tensor(synthetic.uniform_image((1, 6, S, S), 9, "tsf_inputs"))
    bg_inputs = torch.tensor(synthetic.uniform_image((1, 1, 4, S, S), 10, "bg_inputs"))
    Tst = torch.tensor(golden["render/Tst"]).view(1, ns, S, S, 2)
    with torch.no_grad():
        enc, res = orc.gen_forward_src(sd, src_inputs, n_down=len(nf), n_res=nres)
        img, mask = orc.gen_forward_tsf(sd, tsf_inputs, enc, res, Tst, n_down=len(nf), n_res=nres)
        bg = orc.gen_forward_bg(sd, bg_inputs, n_down=len(bgf), n_res=nres)
    assert np.abs(enc[-1].numpy()[:, ::8] - golden[f"gen_{tag}/enc2_sub"]).max() <= 1e-5
    assert np.abs(res[-1].numpy()[:, ::8] - golden[f"gen_{tag}/res_last_sub"]).max() <= 1e-4
    assert np.abs(img.numpy() - golden[f"gen_{tag}/img"]).max() <= 1e-4
    assert np.abs(mask.numpy() - golden[f"gen_{tag}/mask"]).max() <= 1e-4
    assert np.abs(bg.numpy() - golden[f"gen_{tag}/bg"]).max() <= 1e-4


def test_generator_tiny(golden):
    _gen_case(golden, "tiny", [64, 64, 128], 2, [64, 64, 128])


def test_generator_full(golden):
    _gen_case(golden, "full", [64, 128, 256], 6, [64, 128, 128, 256])


def test_generator_full_at_256():
    """The oracle against the reference's own generator at S = 256, full width (tests/golden/make_golden_tsf256.py): the other generator goldens
    are S = 64 / 128 - this one pins the size-dependent parts (flow resizing down to 32 x 32, out-of-range and -2 flows, four times the area)."""
    from ipercore_amd.networks import generator_param_shapes
    from tests.golden.make_golden_tsf256 import synthetic_flow
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_tsf256_v1.npz"))
    nf, nres, bgf = [64, 128, 256], 6, [64, 128, 128, 256]
    sd = {k: torch.tensor(v) for k, v in synthetic.fill_state_dict(generator_param_shapes(nf, nres, bgf), seed=7).items()}
    src_inputs = torch.tensor(synthetic.uniform_image((1, 2, 6, 256, 256), 8, "src_inputs_256"))
    tsf_inputs = torch.tensor(synthetic.uniform_image((1, 6, 256, 256), 9, "tsf_inputs_256"))
    with torch.no_grad():
        enc, res = orc.gen_forward_src(sd, src_inputs, n_down=3, n_res=nres)
        img, mask = orc.gen_forward_tsf(sd, tsf_inputs, enc, res, torch.tensor(synthetic_flow()), n_down=3, n_res=nres)
    assert np.abs(enc[-1].numpy()[:, ::16, ::2, ::2] - g["enc2_sub"]).max() <= 1e-5
    assert np.abs(res[-1].numpy()[:, ::16, ::2, ::2] - g["res_last_sub"]).max() <= 1e-4
    assert np.abs(img.numpy()[:, :, ::2, ::2] - g["img_sub"]).max() <= 1e-4 and np.abs(mask.numpy()[:, :, ::2, ::2] - g["mask_sub"]).max() <= 1e-4
    assert abs(img.double().mean().item() - float(g["img_mean"])) <= 1e-6 and abs(mask.double().mean().item() - float(g["mask_mean"])) <= 1e-6


LWB_VARIANTS = (("AddLWB", "add", "plain"), ("AvgLWB", "avg", "plain"), ("SoftGateAddLWB", "sg_add", "softgate"),
                ("SoftGateAvgLWB", "sg_avg", "softgate"))


@pytest.mark.parametrize("name,kind,shapes_kind", LWB_VARIANTS)
def test_lwb_variant_generators_match_reference(golden, name, kind, shapes_kind):
    """AddLWB / AvgLWB / SoftGateAdd / SoftGateAvg transfer streams of the oracle against outputs of the reference's own
    generators (tests/golden/make_golden_lwb_variants.py), and the parameter inventory against their state_dict keys."""
    import hashlib
    from ipercore_amd.networks import generator_param_shapes
    gv = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_lwb_variants_v1.npz"))
    S, ns = 64, 2
    src_inputs = torch.tensor(synthetic.uniform_image((1, ns, 6, S, S), 8, "src_inputs"))
    tsf_inputs = torch.tensor(synthetic.uniform_image((1, 6, S, S), 9, "tsf_inputs"))
    Tst = torch.tensor(golden["render/Tst"]).view(1, ns, S, S, 2)
    for tag, nf, nres, bgf in (("tiny", [64, 64, 128], 2, [64, 64, 128]), ("full", [64, 128, 256], 6, [64, 128, 128, 256])):
        shapes = generator_param_shapes(nf, nres, bgf, lwb=shapes_kind)
        keys_sha = hashlib.sha256("\n".join(f"{k}:{tuple(shapes[k])}" for k in sorted(shapes)).encode()).hexdigest()
        assert keys_sha == str(gv[f"{name}/{tag}/keys_sha"])
        sd = {k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=11).items()}
        with torch.no_grad():
            enc, res = orc.gen_forward_src(sd, src_inputs, n_down=len(nf), n_res=nres)
            img, mask = orc.gen_forward_tsf(sd, tsf_inputs, enc, res, Tst, n_down=len(nf), n_res=nres, lwb=kind)
        assert np.abs(img.numpy() - gv[f"{name}/{tag}/img"]).max() <= 1e-4
        assert np.abs(mask.numpy() - gv[f"{name}/{tag}/mask"]).max() <= 1e-4


SWAP_PEOPLE = ((2, 20), (1, 40))
SWAP_PART_SETS = {"head_body": (["head"], ["body"]), "leftover": (["upper"], ["left_leg", "right_foot"])}


def _fids_sha(lists):
    return hashlib.sha256(";".join(",".join(str(int(f)) for f in sorted(l)) for l in lists).encode()).hexdigest()


def test_swapper_pieces_match_reference(topo):
    """Part-name face selection, selected_f2pts, merge_uv_img and the selected-face flows of the oracle against the reference's
    own Swapper / FlowCompositionForSwapper (tests/golden/make_golden_swapper.py)."""
    from tests import parity_utils as pu
    gs = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_swapper_v1.npz"))
    S = 128
    t = pu.oracle_tables(topo)
    fim_obj = mesh.obj_from_topology(topo, "fim")
    obj_faces = fim_obj["faces"].astype(np.int32)
    f_img2uvs = mesh.get_f2vts(fim_obj, z=1).astype(np.float32)
    uv_fim, uv_wim = orc.render_uv_fim_wim(f_img2uvs, 1, S)
    parts = {str(n): topo["part_" + str(n)] for n in topo["part_names"]}
    part_faces = list(mesh.get_part_ids(f_img2uvs.shape[0], parts).values())
    model = orc.SMPLHModel(synthetic.smplh_model_dict(seed=0))
    for tag, swap_parts in SWAP_PART_SETS.items():
        fids = orc.select_faces_by_part_name(part_faces, f_img2uvs.shape[0], swap_parts, primary_ids=0)
        assert [len(f) for f in fids] == list(gs[f"{tag}/fids_count"]) and _fids_sha(fids) == str(gs[f"{tag}/fids_sha"])
        uv_imgs, sel_obj, sel = [], [], []
        for i, (ns, seed) in enumerate(SWAP_PEOPLE):
            d = orc.smplh_get_details(model, synthetic.smpl_sequence(ns, seed=seed, pose_dim=72), 0, None)
            f2pts, _, _ = orc.render_fim_wim(d["cam"], d["verts"], t["smpl_faces"], S)
            obj_f2pts, _, _ = orc.render_fim_wim(d["cam"], d["verts"], obj_faces, S)
            sel.append(orc.get_selected_f2pts(f2pts, [fids[i]] * ns))
            sel_obj.append(orc.get_selected_f2pts(obj_f2pts, [fids[i]] * ns)[0:1])
            uv_imgs.append(torch.tensor(synthetic.uniform_image((1, 3, S, S), seed + 5, "uv_img")))
        sel = torch.cat(sel, dim=0)
        assert int((sel[:, :, 0, 0] != -2).sum()) == int(gs[f"{tag}/selected_count"])
        uv = orc.merge_uv_img(uv_imgs, sel_obj, uv_fim, uv_wim)
        assert np.abs(uv.numpy() - gs[f"{tag}/uv_img"]).max() <= 1e-5
        ref = orc.smplh_get_details(model, synthetic.smpl_sequence(1, seed=60, pose_dim=72), 0, None)
        _, rfim, rwim = orc.render_fim_wim(ref["cam"], ref["verts"], t["smpl_faces"], S)
        Tst = orc.cal_bc_transform(sel, rfim.repeat(3, 1, 1), rwim.repeat(3, 1, 1, 1))
        assert np.abs(Tst.numpy() - gs[f"{tag}/Tst"][0]).max() <= 1e-5


def test_smpl24_matches_reference():
    """The oracle's 24-joint SMPL (trainers' body model) against the reference's own SMPL.get_details
    (tests/golden/make_golden_smpl24.py): vertices, the 19 COCO+ joints and their projection."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_smpl24_v1.npz"))
    smpls = synthetic.smpl_sequence(3, seed=80, pose_dim=72)
    offsets = torch.tensor(0.002 * synthetic.uniform_image((6890, 3), 81, "offsets"))
    d = orc.smpl24_get_details(synthetic.smpl_model_dict(seed=0), smpls, offsets)
    assert np.abs(d["verts"].numpy()[:, ::10] - g["verts_sub"]).max() <= 2e-6
    assert np.abs(d["j3d"].numpy() - g["j3d"]).max() <= 2e-6 and np.abs(d["j2d"].numpy() - g["j2d"]).max() <= 2e-6


def test_textured_render_properties(topo):
    """The (parity-unpinned) textured renderer of the oracle: a constant texture reproduces its colour inside the silhouette and
    the background colour outside; unit ambient light is the identity, a directional light only dims; a texture that is linear in
    the barycentric texel index reproduces perspective-correct interpolation (sum of weights = 1 -> constant)."""
    from tests import parity_utils as pu
    t = pu.oracle_tables(topo)
    model = orc.SMPLHModel(synthetic.smplh_model_dict(seed=0))
    d = orc.smplh_get_details(model, synthetic.smpl_sequence(1, seed=20, pose_dim=72), 0, None)
    fv = orc.project_faces(d["cam"], d["verts"], t["smpl_faces"])
    nf, T, S = fv.shape[1], 3, 48
    col = torch.tensor([0.2, -0.4, 0.6])
    tex = torch.ones(1, nf, T, T, T, 3) * col
    fim, _ = orc.rasterize_fim_wim(fv.numpy(), S, 0.1, 25.0)
    img = orc.nr_rasterize(fv, tex, S, anti_aliasing=False, near=0.1, far=25.0, background_color=(-1, -1, -1))
    inside = fim[0] >= 0
    assert (img[0][:, inside] - col[:, None]).abs().max() <= 1e-6 and (img[0][:, ~inside] == -1).all()
    assert torch.equal(orc.nr_lighting(fv, tex, 1, 0), tex)
    lit = orc.nr_lighting(fv, tex.abs(), 0.7, 0.3, direction=(1, 0.5, 1))
    assert (lit <= tex.abs() * (0.7 + 0.3 * (1.5 ** 0.5 * 1.5)) + 1e-6).all() and (lit >= 0.7 * tex.abs() - 1e-6).all()
    # texel value = i0 + i1 + i2 (in texel units): perspective-correct weights sum to T - 1 -> constant image
    g = torch.arange(T, dtype=torch.float32)
    lin = (g[:, None, None] + g[None, :, None] + g[None, None, :])[None, None, ..., None].expand(1, nf, T, T, T, 3)
    img2 = orc.nr_rasterize(fv, lin.contiguous(), S, anti_aliasing=False, near=0.1, far=25.0)
    vals = img2[0][:, inside]
    assert (vals - (T - 1)).abs().max() <= 2e-3           # the clamp at T - 1 - eps moves saturated texels by eps
    aa = orc.nr_rasterize(fv, tex, S, anti_aliasing=True, near=0.1, far=25.0, background_color=(-1, -1, -1))
    assert aa.shape == (1, 3, S, S) and aa.min() >= -1 - 1e-6 and aa.max() <= 0.6 + 1e-6


def test_loss_network_parameter_inventories():
    """The frozen loss networks keep the parameter names of the checkpoints the reference loads: Sphere20a
    (criterions/faceloss.py:203-257, pinned by the golden generated from the reference class itself: the state_dict loaded there
    with strict=True) and torchvision's VGG19 ``features.{i}`` indices (criterions/vggloss.py:12-40)."""
    from ipercore_amd.trainers import Sphere20aFeatures, VGG19Features
    sp = {k: tuple(v.shape) for k, v in Sphere20aFeatures(None, allow_seeded=True).state_dict().items()}
    assert len(sp) == 62 and sp["conv1_1.weight"] == (64, 3, 3, 3) and sp["relu4_3.weight"] == (512,) and sp["fc5.weight"] == (512, 512 * 7 * 6)
    assert sum(1 for k in sp if k.startswith("conv") and k.endswith(".weight")) == 20
    vg = {k: tuple(v.shape) for k, v in VGG19Features(None, allow_seeded=True).state_dict().items()}
    assert sorted(int(k.split(".")[1]) for k in vg if k.endswith(".weight")) == [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28]
    assert vg["features.0.weight"] == (64, 3, 3, 3) and vg["features.28.weight"] == (512, 512, 3, 3)


def test_identity_warp_property(topo):
    """T = cal_bc_transform(f2pts, fim, wim) reproduces the grid_sample coordinate of each
    covered pixel - the property the reference relies on when it uses T as a sampling grid."""
    t = _tables(topo)
    d = _details72()
    for size in (64, 128):
        f2pts, fim, wim = orc.render_fim_wim(d["cam"][0:1], d["verts"][0:1], t["smpl_faces"], size)
        T = orc.cal_bc_transform(f2pts, fim, wim)[0].numpy()
        on = fim[0].numpy() >= 0
        assert 0.05 < on.mean() < 0.6
        rr, cc = np.nonzero(on)
        want = np.stack([(2 * cc + 1) / size - 1, (2 * rr + 1) / size - 1], axis=1)
        err = np.abs(T[on] - want)
        # interior pixels are exact to rounding; silhouette pixels have clamped weights (<= 1 pixel)
        assert np.median(err) < 1e-5
        assert err.max() < 2.0 / size


def test_uv_atlas_front_facing(topo):
    """All 13776 UV-atlas triangles of mapper_fim_enc.txt survive back-face culling."""
    t = _tables(topo)
    f = t["f_img2uvs"].copy()
    f[:, :, 1] *= -1
    keep = (f[:, 2, 1] - f[:, 0, 1]) * (f[:, 1, 0] - f[:, 0, 0]) >= (f[:, 1, 1] - f[:, 0, 1]) * (f[:, 2, 0] - f[:, 0, 0])
    assert keep.all()
    uv_fim, uv_wim = orc.render_uv_fim_wim(t["f_img2uvs"], 1, 128)
    assert (uv_fim >= 0).float().mean() > 0.3
    assert torch.allclose(uv_wim.sum(-1)[uv_fim >= 0], torch.ones(1), atol=1e-5)


# ---------------------------------------------------------------------------------------------- source stage (a15)
def _golden_source():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_source_v1.npz"))


def test_source_stage_matches_reference(topo):
    """oracle.process_source / morph / canny vs the reference's own FlowComposition.process_source, morph, CannyFilter
    (tests/golden/make_golden_source.py), S = 128, ns = 2, with a segmentation that differs from the rendered silhouette."""
    from tests import parity_utils as pu
    g = _golden_source()
    kg, kx, _, _ = orc.canny_kernels()
    assert np.array_equal(kg.numpy().reshape(3, 3), g["canny/gauss"]) and np.array_equal(kx.numpy().reshape(3, 3), g["canny/sobel_x"])
    o = pu.oracle_source_stage(128, dict(conf_erode_ks=3, out_dilate_ks=21, bg_ks=11), topo=topo)
    assert np.array_equal(o["fg"].numpy().astype(np.uint8), g["fg_masks"])
    assert np.array_equal(o["confidant_sil"].numpy().astype(np.uint8), g["confidant_sil"])
    assert np.array_equal(o["outpad_sil"].numpy().astype(np.uint8), g["outpad_sil"])
    assert np.array_equal(o["thin_edges"].numpy().astype(np.uint8), g["thin_edges"])
    assert int((o["only_vis_obj_f2pts"][:, :, 0, 0] != -2).sum()) == int(g["only_vis_obj_count"])
    # morphed image: exact up to fp32 summation order, except where the 3rd/4th nearest boundary distances tie (there the
    # reference's topk(sorted=False) choice is backend-defined; the oracle takes the lowest index)
    ties = o["tie_mask"][:, None].expand(-1, 3, -1, -1).numpy()
    d = np.abs(o["input_G_src"][0, :, 0:3].numpy() - g["input_G_src"][0, :, 0:3])
    assert d[~ties].max() <= 1e-5
    assert ties.mean() < 0.05
    assert np.abs(o["input_G_src"][0, :, 3:].numpy() - g["input_G_src"][0, :, 3:]).max() == 0
    assert np.abs(o["input_G_bg"].numpy() - g["input_G_bg"]).max() <= 1e-6
    # UV merge isolated from the tie pixels: feed the reference's morphed image to the oracle's make_uv_img
    uv = orc.make_uv_img(torch.tensor(g["input_G_src"][:, :, 0:3]), o["obj_f2pts"], o["only_vis_obj_f2pts"], o["uv_fim"], o["uv_wim"])
    assert np.abs(uv.numpy() - g["uv_img"]).max() <= 1e-5
    # standalone morph fixtures
    assert np.array_equal(orc.morph(o["fg"], 5, "erode").numpy().astype(np.uint8), g["morph/erode5"])
    frac = torch.tensor(synthetic.uniform_image((1, 1, 128, 128), 23, "frac")).abs() * 0.3
    assert np.array_equal(orc.morph(frac, 13, "dilate").numpy().astype(np.uint8), g["morph/dilate13_frac"])


# ---------------------------------------------------------------------------------------------- temporal attention
def _temporal_inputs(golden):
    """Same seeds as tests/golden/make_golden_temporal.py::temporal_inputs."""
    Tst = torch.tensor(golden["render/Tst"]).view(1, 2, S, S, 2)
    Ttt = torch.roll(Tst, shifts=(3, -2), dims=(2, 3)).clone()
    src_inputs = torch.tensor(synthetic.uniform_image((1, 2, 6, S, S), 8, "src_inputs"))
    tmp_inputs = torch.tensor(synthetic.uniform_image((2, 1, 6, S, S), 30, "tmp_inputs"))
    tsf_inputs = torch.tensor(synthetic.uniform_image((1, 6, S, S), 9, "tsf_inputs"))
    return Tst, Ttt, src_inputs, tmp_inputs, tsf_inputs


def test_temporal_forward_tsf_matches_reference(golden):
    """oracle.gen_forward_tsf with temporal inputs vs the reference's AttentionLWBGenerator(temporal=True).forward_tsf."""
    from ipercore_amd.networks import generator_param_shapes
    gt = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_temporal_v1.npz"))
    nf, nres, bgf = [64, 64, 128], 2, [64, 64, 128]
    sd = {k: torch.tensor(v) for k, v in synthetic.fill_state_dict(generator_param_shapes(nf, nres, bgf), seed=7).items()}
    Tst, Ttt, src_inputs, tmp_inputs, tsf_inputs = _temporal_inputs(golden)
    with torch.no_grad():
        enc, res = orc.gen_forward_src(sd, src_inputs, len(nf), nres)
        feats = [orc.gen_forward_src(sd, tmp_inputs[k:k + 1], len(nf), nres) for k in range(2)]
        tenc = [torch.cat([feats[k][0][l] for k in range(2)], dim=0) for l in range(len(nf))]
        tres = [torch.cat([feats[k][1][l] for k in range(2)], dim=0) for l in range(nres)]
        img, mask = orc.gen_forward_tsf(sd, tsf_inputs, enc, res, Tst, len(nf), nres, tenc, tres, Ttt)
    assert float(gt["diff_vs_no_temporal"]) > 1e-2                      # the fixture really exercises the temporal branch
    assert np.abs(img.numpy() - gt["img"]).max() <= 1e-5 and np.abs(mask.numpy() - gt["mask"]).max() <= 1e-5


def test_discriminators_against_reference_classes():
    """oracle.patch_discriminator / crop_img / discriminator_forward against the outputs of the reference's OWN
    GlobalDiscriminator / GlobalLocalDiscriminator / GlobalBodyHeadDiscriminator (golden_discriminators_v1.npz, generated by
    tests/golden/make_golden_discriminators.py on seeded weights; aug-bg branch on, one degenerate head box)."""
    from tests.golden import make_golden_discriminators as mk
    from ipercore_amd.trainers import create_discriminator
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_discriminators_v1.npz"))
    x, bg_x, body, head = mk.inputs()
    assert np.abs(orc.crop_img(x, body, 2).numpy()[:, ::2] - g["crop_body"]).max() <= 1e-6
    assert np.abs(orc.crop_img(x, head, 4).numpy() - g["crop_head"]).max() <= 1e-6 and g["crop_head"].shape[0] == 1
    for name in mk.NAMES:
        D = create_discriminator(name, synthetic.AttrDict(**mk.CFG), use_aug_bg=True)       # parameter inventory of the product class
        sd = mk.seeded_state_dict(D, 17)
        outs, avg = orc.discriminator_forward(name, sd, x, bg_x, body, head, mk.CFG["n_layers"], True)
        assert len(outs) == int(g[f"{name}/n"])
        for i, o in enumerate(outs):
            assert np.abs(o.numpy() - g[f"{name}/out{i}"]).max() <= 2e-5, (name, i)
        assert abs(float(avg) - float(g[f"{name}/avg"])) <= 1e-6


def _trainer_on_golden_tensors(D):
    """A LWGTrainer shell around the tensors of golden_trainer_losses_v1 (no generator, no optimizers: optimize_G / optimize_D only
    read ``inp``, ``opts`` and ``D``)."""
    from tests.golden import make_golden_trainer_losses as mk
    from ipercore_amd.trainers import LWGTrainer, TrainOpts
    t = mk.tensors()
    tr = object.__new__(LWGTrainer)
    tr.D, tr.crt_tsf, tr.crt_face, tr.losses = D, None, None, {}
    tr.opts = TrainOpts.l1_transfer()
    for k, v in mk.LAMBDAS.items():
        setattr(tr.opts, k, v)
    tr.inp = {"input_G_tsf": t["input_G_tsf"], "real_src": t["real_src"], "real_tsf": t["real_tsf"], "real_bg": t["real_bg"],
              "body_mask": t["body_mask"]}
    return tr, t


def test_trainer_loss_assembly_against_reference_methods():
    """LWGTrainer.optimize_G / optimize_D of the PRODUCT (their loss assembly is plain torch) against the values the reference's own
    LWGTrainer.optimize_G / optimize_D produced on the same seeded tensors (golden_trainer_losses_v1.npz, generated by
    tests/golden/make_golden_trainer_losses.py).  The discriminator inside is the oracle's restatement here (its HIP form is
    compared with the reference's classes by the GPU checks); the same comparison runs on the GPU with the HIP discriminator."""
    from tests.golden import make_golden_trainer_losses as mk
    from tests.golden.make_golden_discriminators import seeded_state_dict
    from ipercore_amd.trainers import create_discriminator
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_trainer_losses_v1.npz"))
    sd = seeded_state_dict(create_discriminator("patch_global", synthetic.AttrDict(**mk.DCFG)), 23)

    def D(inputs):
        return orc.discriminator_forward("patch_global", sd, inputs["x"], None, None, None, mk.DCFG["n_layers"], False)[0]
    tr, t = _trainer_on_golden_tensors(D)
    with torch.no_grad():
        loss_g = tr.optimize_G(t["fake_bg"], t["fake_src_imgs"], t["fake_tsf_imgs"], t["fake_masks"])
        loss_d = tr.optimize_D(t["fake_tsf_imgs"])
    got = dict(loss_G=loss_g, loss_D=loss_d, **{k: tr.losses[k] for k in ("g_rec", "g_tsf", "g_adv", "g_mask", "g_mask_smooth", "d_real", "d_fake")})
    for k, v in got.items():
        assert abs(float(v) - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), (k, float(v), float(g[k]))
