"""Host logic of the runner (Imitator / FlowComposition / SMPLRenderer / SMPLH wrappers, frame batching, camera
pre-pass) on CPU with the C-ABI ops emulated (tests/emu_ops.py), against the oracle's frame-by-frame result."""
import os

import numpy as np
import pytest
import torch

from tests import emu_ops
from tests import parity_utils as pu


def test_imitator_batched_equals_oracle_per_frame(monkeypatch):
    emu_ops.install(monkeypatch)
    case = pu.build_case(image_size=64, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=5, ns=2)
    im = pu.make_imitator(case, frame_batch=2, device="cpu")        # 5 frames in batches of 2,2,1
    got = pu.run_hip(case, imitator=im)
    want = pu.run_oracle(case)
    assert got.shape == (5, 3, 64, 64)
    assert (got - want).abs().max().item() <= 2e-4
    # reference-shaped API: inference() returns a list of (3,S,S) arrays in [-1,1]
    outs = im.inference(case.tgt_smpls, cam_strategy="smooth", output_dir="", verbose=False)
    assert len(outs) == 5 and outs[0].shape == (3, 64, 64)
    assert np.abs(np.stack(outs) - want.numpy()).max() <= 2e-4


def test_inference_writes_reference_file_names(monkeypatch, tmp_path):
    emu_ops.install(monkeypatch)
    case = pu.build_case(image_size=64, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=2, ns=1)
    im = pu.make_imitator(case, frame_batch=8, device="cpu")
    paths = im.inference(case.tgt_smpls, cam_strategy="copy", output_dir=str(tmp_path), prefix="pred_", verbose=False)
    assert [p.split("/")[-1] for p in paths] == ["pred_00000000.png", "pred_00000001.png"]
    from PIL import Image
    from oracle import lwg_oracle as orc
    want = pu.run_oracle(case, cam_strategy="copy")
    img = np.asarray(Image.open(paths[1]))
    assert np.array_equal(img[:, :, ::-1], orc.to_uint8_bgr(want[1].numpy())) or \
        np.abs(img[:, :, ::-1].astype(int) - orc.to_uint8_bgr(want[1].numpy()).astype(int)).max() <= 1


def test_reference_shaped_flow_methods(monkeypatch):
    emu_ops.install(monkeypatch)
    from oracle import lwg_oracle as orc
    case = pu.build_case(image_size=64, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=2, ns=2)
    im = pu.make_imitator(case, frame_batch=2, device="cpu")
    tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
    model, tables, sd, info = pu.oracle_source(case)
    ref = im.body_rec.get_details(im.swap_params(im.src_info["cam"][0:1], im.src_info["shape"][0:1], tgt[0:1]), 0, None)
    im.flow_comp.add_rendered_f2verts_fim_wim(ref, use_morph=False, get_uv_info=False)
    im.flow_comp.make_uv_setup(1, 2, 1, "cpu")
    tsf = im.flow_comp.make_tsf_inputs(im.src_info["uv_img"], ref)
    Tst, Ttt = im.flow_comp.make_trans_flow(1, 2, 1, im.src_info, None, ref, temporal=False)
    want, allr = pu.run_oracle(case, frames=[0], return_all=True)
    assert Ttt is None and Tst.shape == (1, 2, 64, 64, 2)
    assert (tsf[:, 0] - allr[0]["tsf_inputs"]).abs().max() <= 2e-4   # random-texture sampling amplifies 1e-7 flow rounding
    assert (Tst - allr[0]["Tst"]).abs().max() <= 1e-5


def test_staged_parity_harness_on_emulated_abi(monkeypatch):
    """The staged comparison used by the GPU parity tests and smoke(), exercised on CPU."""
    emu_ops.install(monkeypatch)
    case = pu.build_case(image_size=64, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=3, ns=2)
    m, got, im = pu.staged_parity(case, frame_batch=2, device="cpu")
    assert m["fim_equal"] and m["src_fim_equal"] and m["wim_max"] == 0.0
    assert m["verts_max"] <= 1e-5 and m["src_verts_max"] <= 1e-5
    assert m["pred_max"] <= 2e-4 and m["Tst_max"] <= 1e-5


def test_frame_writer_cpu(tmp_path):
    """The async output stage with CPU tensors: names (imitator.py:369), numerics (cv_utils.py:111-113), ordering."""
    import numpy as np
    import torch
    from PIL import Image
    from ipercore_amd.output import FrameWriter
    rs = np.random.RandomState(0)
    frames = torch.tensor(rs.uniform(-1, 1, size=(7, 3, 24, 24)).astype(np.float32))
    frames[0, 0, 0, 0], frames[0, 1, 0, 0], frames[0, 2, 0, 0] = 1.0, -1.0, 0.0
    w = FrameWriter(str(tmp_path), prefix="pred_", workers=3, ring=2)
    w.submit(frames[0:3], 0)
    w.submit(frames[3:6], 3)
    w.submit(frames[6:7], 6)           # a last, smaller batch
    paths = w.close()
    assert [os.path.basename(p) for p in paths] == ["pred_{:0>8}.png".format(t) for t in range(7)]
    for t, p in enumerate(paths):
        want = ((np.transpose(frames[t].numpy(), (1, 2, 0)) + 1) / 2.0 * 255).astype(np.uint8)
        assert np.array_equal(np.asarray(Image.open(p)), want)
    assert tuple(np.asarray(Image.open(paths[0]))[0, 0]) == (255, 0, 127)


def test_novel_view_smpls_and_viewer(monkeypatch):
    """create_T_pose_novel_view_smpl / add_hands_params_to_smpl (services/base_runner.py:11-55) and the Viewer runner on the
    emulated ABI: 156-dim hand-extended poses run through the same per-frame path as 72-dim ones with zero hands."""
    from ipercore_amd.imitator import ModelsFactory, add_hands_params_to_smpl, create_T_pose_novel_view_smpl
    smpls = create_T_pose_novel_view_smpl(5)
    assert smpls.shape == (5, 85) and np.allclose(np.linalg.norm(smpls[:, 3:6], axis=1), np.pi, atol=1e-5)
    assert np.allclose(smpls[0, 3:6], [np.pi, 0, 0], atol=1e-6) and np.allclose(np.abs(smpls[2, 3:6]), [0, 0, np.pi], atol=1e-5)
    emu_ops.install(monkeypatch)
    case = pu.build_case(image_size=64, num_filters=[64, 64, 128], n_res=1, bg_filters=[64, 64, 128], n_frames=2, ns=2)
    v = ModelsFactory.get_by_name("viewer", case.opt, device=torch.device("cpu"), frame_batch=2)
    v.generator.load_state_dict({k: torch.tensor(x) for k, x in case.state.items()}, strict=True)
    v.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
    smpls[:, 0:3] = case.src_smpl[0, 0:3]
    smpls[:, -10:] = case.src_smpl[0, -10:]
    h = add_hands_params_to_smpl(smpls, v.body_rec.np_hands_mean)
    assert h.shape == (5, 3 + 156 + 10)
    out156 = v.inference(h[:3], cam_strategy="smooth")
    out72 = v.inference(smpls[:3], cam_strategy="smooth")
    assert len(out156) == 3 and out156[0].shape == (3, 64, 64)
    assert all(np.abs(a - b).max() <= 1e-5 for a, b in zip(out156, out72))          # hands_mean = 0 in the synthetic model


def test_configs0_motion_imitate_256_cpu_plumbing(monkeypatch):
    """BASELINE configs[0] as stated (demo/motion_imitate.py:120-133 -> run_imitator): 256x256, ONE source image, an 8-frame reference
    clip, the whole path on CPU tensors - the runner's host logic over the emulated C ABI against the oracle's frame-by-frame result
    (full AttLWB-SPADE architecture; plumbing + parity, no GPU, no timing)."""
    emu_ops.install(monkeypatch)
    case = pu.build_case(image_size=256, n_frames=8, ns=1)
    im = pu.make_imitator(case, frame_batch=8, device="cpu")
    outs = im.inference(case.tgt_smpls, cam_strategy="smooth", output_dir="", verbose=False)
    assert len(outs) == 8 and outs[0].shape == (3, 256, 256)
    want = pu.run_oracle(case)
    d = np.abs(np.stack(outs) - want.numpy())
    assert np.isfinite(d).all() and d.max() <= 2e-3 and d.mean() <= 1e-4, (d.max(), d.mean())     # SURVEY 8c generator tolerance


def test_frame_batch_is_not_clamped(monkeypatch):
    """The 3 GiB per-tensor range of the conv kernels' buffer offsets no longer reaches the caller (launches are cut into batch slices
    inside the C entry points, csrc/lwg_conv_slices.h): ``frame_batch`` is what was asked for at any size / precision mode."""
    emu_ops.install(monkeypatch)
    case = pu.build_case(image_size=64, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=1, ns=1)
    im = pu.make_imitator(case, frame_batch=12, device="cpu")
    assert im.frame_batch == 12 and im.max_frame_batch() is None
    im.image_size = 1024
    im.frame_batch = 40
    assert im.frame_batch == 40
    im.generator.conv_precision = "bf16"
    assert im.frame_batch == 40


def test_get_vis_f2pts_matches_reference_golden(monkeypatch):
    """renders/nmr.py:639-681 ``get_vis_f2pts`` of the drop-in renderer (device-side index ops, no ``unique()``) and of the oracle
    against the sha of the reference's OWN output (tests/golden/make_golden.py section 4: ``render/vis_f2pts_sha``, S = 64,
    top_k = 3 as FlowComposition builds it)."""
    import hashlib
    emu_ops.install(monkeypatch)
    from ipercore_amd import synthetic
    from ipercore_amd.renders import SMPLRenderer
    from oracle import lwg_oracle as orc
    from tests.test_oracle_golden import _details72, S
    golden = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()          # noqa: E731
    render = SMPLRenderer(image_size=S, has_front=True, top_k=3)          # make_golden.py:84-92
    d = _details72()
    f2pts, fim, wim = render.render_fim_wim(d["cam"][0:1], d["verts"][0:1], smpl_faces=True)
    assert sha(f2pts.numpy()) == str(golden["render/f2pts_sha"]) and np.array_equal(fim.numpy(), golden["render/fim"])
    vis = render.get_vis_f2pts(f2pts, fim)
    assert sha(vis.numpy()) == str(golden["render/vis_f2pts_sha"])
    want = orc.get_vis_f2pts(f2pts, fim, render.face_k_nearest.numpy())
    assert torch.equal(vis, want)
    n_kept = int((vis[0, :, 0, 0] != -2).sum())
    assert 0 < n_kept < f2pts.shape[1]                 # the fixture really drops invisible faces
    # single (nf,3,2) form and a map with NO background pixel (the reference drops the smallest id whatever it is, nmr.py:660)
    assert torch.equal(render.get_vis_f2pts(f2pts[0], fim[0]), vis[0])
    full = fim.clone()
    full[full < 0] = 7
    assert torch.equal(render.get_vis_f2pts(f2pts, full), orc.get_vis_f2pts(f2pts, full, render.face_k_nearest.numpy()))
    assert synthetic is not None


def test_only_vis_frames_on_emulated_abi(monkeypatch):
    """opt.only_vis = True through Imitator's batched per-frame path (flowcomposition.py:559-562) against the oracle."""
    emu_ops.install(monkeypatch)
    case = pu.build_case(image_size=64, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=3, ns=2)
    plain = pu.run_hip(case, imitator=pu.make_imitator(case, frame_batch=2, device="cpu"))
    case.opt["only_vis"] = True
    im = pu.make_imitator(case, frame_batch=2, device="cpu")
    assert im.flow_comp.only_vis
    got = pu.run_hip(case, imitator=im)
    want = pu.run_oracle(case)
    assert (got - want).abs().max().item() <= 2e-4
    assert (got - plain).abs().max().item() > 1e-3       # the option changes the flows (hidden source faces no longer feed the warp)


def test_precision_modes_plumbing(monkeypatch):
    """generator.conv_precision = "winograd" (the default) / "fp32" / "split" route through the same engine wiring (the emulated ABI computes every
    mode exactly, so the frames must equal the default mode's): the quad-plane head input is used by the fp32 MFMA modes ("winograd": the
    last up-sampling layer is lwg_conv_transpose4_winograd_f32 with a quad-plane output; "fp32": the direct transposed convolution), "split" takes the NHWC head."""
    emu_ops.install(monkeypatch)
    from ipercore_amd import ops
    case = pu.build_case(image_size=64, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=3, ns=2)
    im = pu.make_imitator(case, frame_batch=2, device="cpu")
    calls = []
    real_head = ops.head_compose
    monkeypatch.setattr(ops, "head_compose", lambda *a, **k: (calls.append(bool(k.get("q4"))), real_head(*a, **k))[1])
    assert im.generator.conv_precision == "winograd"
    ref = pu.run_hip(case, imitator=im)
    assert calls and all(calls), "the default mode feeds the head channel-quad planes"
    for mode in ("fp32", "split"):
        del calls[:]
        im.generator.conv_precision = mode
        got = pu.run_hip(case, imitator=im)
        assert calls and (all(calls) if mode == "fp32" else not any(calls)), mode
        assert torch.equal(got, ref), mode
    with pytest.raises(AssertionError):
        ops.conv_precision("fp16")


def test_smplh_pickle_route_equals_dict_route(monkeypatch, tmp_path):
    """``SMPLH(model_path="...pkl")`` - the reference's constructor (batch_smplh.py:105-135: pickle with a scipy-sparse ``J_regressor``,
    ``hands_meanl/r``, ``hands_componentsl/r``) - builds the same model as the dict route every other test uses: buffers equal, the 72-dim
    pose is completed with the pickle's hand means, and ``get_details`` (emulated C ABI) gives identical vertices / joints."""
    import pickle
    import scipy.sparse as sp
    from ipercore_amd import synthetic
    from ipercore_amd.bodynets import SMPLH
    emu_ops.install(monkeypatch)
    d = synthetic.smplh_model_dict(seed=3)
    r = np.random.RandomState(5)
    d["hands_meanl"], d["hands_meanr"] = 0.1 * r.standard_normal(45), 0.1 * r.standard_normal(45)     # the licensed model's are non-zero
    d["hands_componentsl"], d["hands_componentsr"] = r.standard_normal((45, 45)), r.standard_normal((45, 45))
    on_disk = dict(d)
    jr = d["J_regressor"].copy()
    jr[np.abs(jr) < np.quantile(np.abs(jr), 0.9)] = 0.0                  # SMPL's regressor is sparse: stored as CSC in the licensed pickle
    d["J_regressor"] = jr
    on_disk["J_regressor"] = sp.csc_matrix(jr)
    path = tmp_path / "SMPLH_NEUTRAL.pkl"
    with open(path, "wb") as fp:
        pickle.dump(on_disk, fp, protocol=2)
    a, b = SMPLH(str(path)), SMPLH(d)
    sa, sb = dict(a.named_buffers()), dict(b.named_buffers())
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert np.array_equal(a.faces, b.faces) and np.array_equal(a.np_hands_mean, b.np_hands_mean)
    assert a.np_hands_mean.shape == (90,) and np.abs(a.np_hands_mean).max() > 0
    smpls = torch.tensor(synthetic.smpl_sequence(3, seed=2))
    full = a._full_pose(smpls[:, 3:75])
    assert full.shape == (3, 156) and torch.equal(full[:, 66:], a.hands_mean.repeat(3, 1))
    da, db = a.get_details(smpls), b.get_details(smpls)
    for k in ("verts", "j2d", "cam", "pose", "shape"):
        if k in da:
            assert torch.equal(da[k], db[k]), k
    # PCA hands (use_pca=True, batch_smplh.py:160-169): the last 12 pose entries are coefficients of the pickle's first six components
    c = SMPLH(str(path), use_pca=True, num_pca_comps=6)
    th = torch.cat([smpls[:, 3:69], 0.3 * torch.ones(3, 12)], dim=1)
    fp_ = c._full_pose(th)
    want_l = 0.3 * torch.tensor(d["hands_componentsl"][:6], dtype=torch.float32).sum(0)
    assert fp_.shape == (3, 156) and torch.allclose(fp_[0, 66:111], want_l, atol=1e-6)
