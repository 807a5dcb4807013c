"""INTEGRATION.md's levels EXECUTED against the reference's own code (authoring container only: skipped where /root/reference is
absent, i.e. on the GPU box).  The C ABI is emulated on CPU tensors (tests/emu_ops.py), so what is exercised here is the BINDING:
names, argument conventions and return types the reference's callers rely on.

Level 2 - ``sys.modules["neural_renderer"] = ipercore_amd.nr``, then the reference's unmodified ``SMPLRenderer``
(iPERCore/tools/human_digitalizer/renders/nmr.py:128-225, 298-358, 390-401, 639-681, 713-757) runs on it; its outputs must equal
those of ``ipercore_amd.renders.SMPLRenderer``.
Level 1 - the reference's unmodified runner function ``call_imitator_inference`` (iPERCore/services/run_imitator.py:19-84, with
``base_runner.add_bullet_time_effect / add_special_effect / add_hands_params_to_smpl`` :33-151) drives ``ipercore_amd.imitator.Imitator``:
file names, frame count and the frames themselves against the oracle.
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

from tests import emu_ops
from tests import parity_utils as pu

REF = os.environ.get("LWG_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "iPERCore")), reason="the reference checkout is not on this box")

S = 64


class _Stub(types.ModuleType):
    """An absent third-party module the reference imports at module scope but never calls on these paths."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        m = _Stub(self.__name__ + "." + k)
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return None


def _purge():
    for k in [k for k in sys.modules if k == "iPERCore" or k.startswith("iPERCore.") or k == "neural_renderer"]:
        del sys.modules[k]


@pytest.fixture
def reference(monkeypatch):
    """The reference importable with OUR renderer module standing in for the un-vendored CUDA package (INTEGRATION Level 2)."""
    import ipercore_amd.nr as our_nr
    emu_ops.install(monkeypatch)
    _purge()
    monkeypatch.setattr(np, "int", int, raising=False)          # removed in NumPy >= 1.24, used at reference mesh.py:312,317
    monkeypatch.setattr(np, "float", float, raising=False)
    for name in ("cv2", "toml", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "tensorboardX"):
        if name not in sys.modules:
            monkeypatch.setitem(sys.modules, name, _Stub(name))
    monkeypatch.setitem(sys.modules, "neural_renderer", our_nr)
    monkeypatch.syspath_prepend(REF)
    yield our_nr
    _purge()


def _ref_renderer(image_size):
    from ipercore_amd import synthetic
    from iPERCore.tools.human_digitalizer.renders.nmr import SMPLRenderer
    cfgdir = os.path.join(REF, "assets/configs/pose3d")
    tmp = synthetic.tmp_asset_dir()
    return SMPLRenderer(face_path=synthetic.write_smpl_faces_npy(os.path.join(tmp, "smpl_faces.npy")),
                        fim_enc_path=os.path.join(cfgdir, "mapper_fim_enc.txt"), uv_map_path=os.path.join(cfgdir, "mapper_uv.txt"),
                        part_path=os.path.join(cfgdir, "smpl_part_info.json"), front_path=os.path.join(cfgdir, "front_body.json"),
                        head_path=os.path.join(cfgdir, "head.json"), facial_path=os.path.join(cfgdir, "front_facial.json"),
                        map_name="uv_seg", tex_size=3, image_size=image_size, fill_back=False, anti_aliasing=True,
                        background_color=(0, 0, 0), has_front=True, top_k=3)


def test_level2_reference_smplrenderer_on_our_neural_renderer(reference):
    import iPERCore.tools.human_digitalizer.renders.nmr as ref_nmr
    from ipercore_amd.renders import SMPLRenderer
    from tests.test_oracle_golden import _details72
    assert ref_nmr.nr is reference, "the reference's `import neural_renderer as nr` did not bind ipercore_amd.nr"
    ref = _ref_renderer(S)
    ours = SMPLRenderer(image_size=S, has_front=True, top_k=3)
    d = _details72()
    cam, verts = d["cam"][0:2].clone(), d["verts"][0:2].clone()
    # nmr.py:319-342 (look_at + vertices_to_faces + rasterize_face_index_map_and_weight_map of OUR module inside THEIR wrapper)
    for smpl_faces in (True, False):
        rf, rfim, rwim = ref.render_fim_wim(cam, verts, smpl_faces=smpl_faces)
        of, ofim, owim = ours.render_fim_wim(cam, verts, smpl_faces=smpl_faces)
        assert rfim.dtype == torch.int32 and rwim.shape == (2, S, S, 3)
        assert torch.equal(rf, of) and torch.equal(rfim, ofim) and torch.equal(rwim, owim)
        assert int((rfim >= 0).sum()) > 100
    assert torch.equal(ref.render_fim(cam, verts, smpl_faces=True), ours.render_fim(cam, verts, smpl_faces=True))
    # nmr.py:344-358 UV atlas
    ruf, ruw = ref.render_uv_fim_wim(2)
    ouf, ouw = ours.render_uv_fim_wim(2)
    assert torch.equal(ruf, ouf) and torch.equal(ruw, ouw)
    # consumers of the maps: their torch code on their maps vs our kernels' contract on ours
    f2pts, fim, wim = ref.render_fim_wim(cam, verts, smpl_faces=True)
    enc_r, _ = ref.encode_fim(fim=fim, transpose=True)
    enc_o, _ = ours.encode_fim(fim=fim, transpose=True)
    assert torch.equal(enc_r, enc_o)
    T_r = ref.cal_bc_transform(f2pts.flip(0).contiguous(), fim, wim)
    T_o = ours.cal_bc_transform(f2pts.flip(0).contiguous(), fim, wim)
    assert (T_r - T_o).abs().max().item() <= 1e-6 and bool((T_r == -2).any())
    Tuv_r = ref.cal_bc_transform(ref.get_f_uvs2img(2), fim, wim)
    Tuv_o = ours.cal_bc_transform(ours.get_f_uvs2img(2), fim, wim)
    assert (Tuv_r - Tuv_o).abs().max().item() <= 1e-6
    assert torch.equal(ref.get_vis_f2pts(f2pts, fim), ours.get_vis_f2pts(f2pts, fim))
    # the textured renderer's calls (nr.lighting / nr.rasterize, nmr.py:243-296): signature-compatible and the same image through either
    # wrapper (the values themselves are parity-unpinned: the package is not vendored)
    uv_img = torch.rand(2, 3, S, S) * 2 - 1
    ref.set_ambient_light()
    ours.set_ambient_light()
    img_r, tex_r = ref.forward(cam, verts, uv_img, dynamic=True)
    img_o, tex_o = ours.forward(cam, verts, uv_img, dynamic=True)
    assert img_r.shape == (2, 3, S, S) and torch.isfinite(img_r).all()
    assert (tex_r - tex_o).abs().max().item() <= 1e-5 and (img_r - img_o).abs().max().item() <= 1e-4


class _Meta:
    def __init__(self, out_img_dir, view, bt):
        self.out_img_dir, self.effect_info, self.pose_fc, self.cam_fc = out_img_dir, {"View": view, "BT": bt}, 300, 100


def _frames_of(paths):
    from PIL import Image
    return np.stack([np.asarray(Image.open(p))[:, :, ::-1] for p in paths])          # PNGs hold BGR (cv2.imwrite of the reference)


def test_level1_reference_runner_drives_our_imitator(reference, tmp_path):
    from iPERCore.services import run_imitator as ref_runner                     # the reference's module, unmodified
    from iPERCore.services.base_runner import add_bullet_time_effect, add_hands_params_to_smpl, add_special_effect
    from oracle import lwg_oracle as orc
    case = pu.build_case(image_size=S, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=4, ns=2)
    im = pu.make_imitator(case, frame_batch=3, device="cpu")
    opt = pu.AttrDict(cam_strategy="smooth")
    ref_paths = [f"frame_{i:08d}.png" for i in range(4)]
    hands = im.body_rec.np_hands_mean

    # (a) no multi-view: bullet-time effect at frame 1 (2 novel-view frames inserted), hands appended -> (6,169) rows (cam 3 + pose 156 + shape 10), prefix "pred_"
    out_dir = str(tmp_path / "plain")
    res = ref_runner.call_imitator_inference(opt, im, _Meta(out_dir, [], [(1, 2)]), ref_paths, case.tgt_smpls.copy(), visualizer=None)
    smpls, img_paths = add_bullet_time_effect(case.tgt_smpls.copy(), ref_paths, bt_list=[(1, 2)])
    smpls = add_hands_params_to_smpl(smpls, hands)
    assert smpls.shape == (6, 3 + 156 + 10) and res["ref_imgs_paths"] == img_paths
    outs = [o[0] for o in res["outputs"]]
    assert [os.path.basename(p) for p in outs] == [f"pred_{t:0>8}.png" for t in range(6)]          # imitator.py:369
    case_a = pu.build_case(image_size=S, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=4, ns=2)
    case_a.tgt_smpls = smpls.astype(np.float32)
    want = pu.run_oracle(case_a)
    want_u8 = np.stack([orc.to_uint8_bgr(w.numpy()) for w in want])
    got_u8 = _frames_of(outs)
    assert np.abs(got_u8.astype(int) - want_u8.astype(int)).max() <= 1 and (got_u8 != want_u8).mean() < 1e-3

    # (b) multi-view outputs (run_imitator.py:62-75): one inference per view, prefix "pred_{i}_{view}_", outputs zipped per frame
    im.first_cam = None
    out_dir = str(tmp_path / "views")
    res = ref_runner.call_imitator_inference(opt, im, _Meta(out_dir, [0, 90], []), ref_paths, case.tgt_smpls.copy(), visualizer=None)
    assert len(res["outputs"]) == 4 and all(len(o) == 2 for o in res["outputs"])
    assert os.path.basename(res["outputs"][2][1]) == "pred_1_90_00000002.png"
    # the reference's add_view_effect rotates ref_smpls IN PLACE (base_runner.py:72-75): view 1 acts on the rows view 0 returned
    rows = case.tgt_smpls.copy()
    rows0, _ = add_special_effect(rows, ref_paths, view_dir=0, bt_list=[])
    rows1, _ = add_special_effect(rows, ref_paths, view_dir=90, bt_list=[])
    case_b = pu.build_case(image_size=S, num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128], n_frames=4, ns=2)
    case_b.tgt_smpls = add_hands_params_to_smpl(rows1, hands).astype(np.float32)
    want = pu.run_oracle(case_b)
    got_u8 = _frames_of([o[1] for o in res["outputs"]])
    want_u8 = np.stack([orc.to_uint8_bgr(w.numpy()) for w in want])
    assert np.abs(got_u8.astype(int) - want_u8.astype(int)).max() <= 1 and (got_u8 != want_u8).mean() < 1e-3
