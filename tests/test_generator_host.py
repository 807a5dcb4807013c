"""Host logic of the generator (weight packing, launch orchestration, layouts, API) on CPU: the C-ABI ops are
replaced by tests/emu_ops.py, the result is checked against the oracle and the reference-generated goldens."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ipercore_amd import synthetic
from ipercore_amd.networks import NetworksFactory, generator_param_shapes
from oracle import lwg_oracle as orc
from tests import emu_ops

S = 64


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def make_cfg(nf, nres, bgf):
    return AttrDict(name="AttLWB-SPADE",
                    BGNet=AttrDict(norm_type="instance", cond_nc=4, n_res_block=nres, num_filters=bgf),
                    SIDNet=AttrDict(norm_type="None", cond_nc=6, n_res_block=nres, num_filters=nf),
                    TSFNet=AttrDict(norm_type="instance", cond_nc=6, n_res_block=nres, num_filters=nf))


def build(nf, nres, bgf, seed=7):
    G = NetworksFactory.get_by_name("AttLWB-SPADE", cfg=make_cfg(nf, nres, bgf), temporal=False).eval()
    shapes = generator_param_shapes(nf, nres, bgf)
    sd = {k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=seed).items()}
    missing = G.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return G, sd


def test_state_dict_keys_match_reference(golden):
    G, _ = build([64, 128, 256], 6, [64, 128, 128, 256])
    keys = [ln.split(" ")[0] for ln in open("tests/golden/attlwb_spade_state_dict_keys.txt")]
    assert sorted(G.state_dict().keys()) == sorted(keys)
    assert sum(p.numel() for p in G.parameters()) == 36276992 == int(golden["gen_full/nparams"])


@pytest.mark.parametrize("tag,nf,nres,bgf", [("tiny", [64, 64, 128], 2, [64, 64, 128]),
                                              ("full", [64, 128, 256], 6, [64, 128, 128, 256])])
def test_generator_api_through_emulated_abi(monkeypatch, golden, tag, nf, nres, bgf):
    emu_ops.install(monkeypatch)
    G, sd = build(nf, nres, bgf)
    ns = 2
    src_inputs = torch.tensor(synthetic.uniform_image((1, ns, 6, S, S), 8, "src_inputs"))
    tsf_inputs = torch.tensor(synthetic.uniform_image((1, 6, S, S), 9, "tsf_inputs"))
    bg_inputs = torch.tensor(synthetic.uniform_image((1, 1, 4, S, S), 10, "bg_inputs"))
    Tst = torch.tensor(golden["render/Tst"]).view(1, ns, S, S, 2)
    enc, res = G.forward_src(src_inputs, only_enc=True)
    img, mask = G.forward_tsf(tsf_inputs, enc, res, Tst)
    bg = G.forward_bg(bg_inputs)
    # vs the reference's own outputs
    assert np.abs(enc[-1].numpy()[:, ::8] - golden[f"gen_{tag}/enc2_sub"]).max() <= 1e-4
    assert np.abs(res[-1].numpy()[:, ::8] - golden[f"gen_{tag}/res_last_sub"]).max() <= 1e-4
    assert np.abs(img.numpy() - golden[f"gen_{tag}/img"]).max() <= 2e-4
    assert np.abs(mask.numpy() - golden[f"gen_{tag}/mask"]).max() <= 2e-4
    assert np.abs(bg.numpy() - golden[f"gen_{tag}/bg"]).max() <= 2e-4
    # plain-list API (no engine cache attached) must give the same answer
    img2, mask2 = G.forward_tsf(tsf_inputs, list(enc), list(res), Tst)
    assert torch.allclose(img, img2, atol=1e-6) and torch.allclose(mask, mask2, atol=1e-6)
    # src decode branch + full forward
    enc_o, res_o = orc.gen_forward_src(sd, src_inputs, n_down=len(nf), n_res=nres)
    _, _, simg, smask = G.forward_src(src_inputs, only_enc=False)
    x = res_o[-1]
    for i in range(len(nf)):
        x = torch.relu(orc._convT(sd, f"src_net.decoders.layers.{i}.0", x))
    assert torch.allclose(simg[0], torch.tanh(orc._conv(sd, "src_net.img_reg.0", x, pad=2)), atol=2e-4)
    assert torch.allclose(smask[0], torch.sigmoid(orc._conv(sd, "src_net.att_reg.0", x, pad=2)), atol=2e-4)
    outs = G(bg_inputs, src_inputs, tsf_inputs.unsqueeze(1), Tst.unsqueeze(1), only_tsf=True)
    assert outs[0].shape == (1, 1, 3, S, S) and outs[1].shape == (1, 1, 3, S, S) and outs[2].shape == (1, 1, 1, S, S)
    assert torch.allclose(outs[1][:, 0], img, atol=1e-6)


@pytest.mark.parametrize("name", ["AddLWB", "AvgLWB", "SoftGateAddLWB", "SoftGateAvgLWB"])
def test_lwb_variant_generators_through_emulated_abi(monkeypatch, golden, name):
    """The other Liquid Warping Block generators of the reference's factory (networks/__init__.py:22-36) behind the same API:
    host logic (parameter tree, packing, block dispatch, scale factors) against the reference's own outputs."""
    from ipercore_amd.networks import generator_param_shapes
    emu_ops.install(monkeypatch)
    gv = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_lwb_variants_v1.npz"))
    nf, nres, bgf, ns = [64, 64, 128], 2, [64, 64, 128], 2
    G = NetworksFactory.get_by_name(name, cfg=synthetic.gen_cfg(nf, nres, bgf), temporal=False).eval()
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    assert shapes == {k: tuple(v) for k, v in generator_param_shapes(nf, nres, bgf, lwb="plain" if "Soft" not in name else "softgate").items()}
    G.load_state_dict({k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=11).items()}, strict=True)
    src_inputs = torch.tensor(synthetic.uniform_image((1, ns, 6, S, S), 8, "src_inputs"))
    tsf_inputs = torch.tensor(synthetic.uniform_image((1, 6, S, S), 9, "tsf_inputs"))
    Tst = torch.tensor(golden["render/Tst"]).view(1, ns, S, S, 2)
    enc, res = G.forward_src(src_inputs, only_enc=True)
    img, mask = G.forward_tsf(tsf_inputs, enc, res, Tst)
    assert np.abs(img.numpy() - gv[f"{name}/tiny/img"]).max() <= 2e-4
    assert np.abs(mask.numpy() - gv[f"{name}/tiny/mask"]).max() <= 2e-4
    img2, _ = G.forward_tsf(tsf_inputs, list(enc), list(res), Tst)
    assert torch.allclose(img, img2, atol=1e-6)


def test_non_square_inputs_are_refused_before_any_network_runs(monkeypatch):
    """image_size is one number in the reference; the attention blocks take flows resized to square feature maps: a non-square input raises at
    the API entry (forward_src / forward_tsf), not at the first attention site after the encoder has run."""
    emu_ops.install(monkeypatch)
    from ipercore_amd.networks import NetworksFactory
    G = NetworksFactory.get_by_name("AttLWB-SPADE", cfg=synthetic.gen_cfg([64, 64, 128], 2, [64, 64, 128]), temporal=False).eval()
    calls = []
    monkeypatch.setattr(emu_ops.real_ops, "conv2d", lambda *a, **k: calls.append(1))
    with pytest.raises(ValueError, match="square images only"):
        G.forward_src(torch.zeros(1, 2, 6, 64, 96))
    with pytest.raises(ValueError, match="square images only"):
        G.forward_tsf(torch.zeros(1, 6, 64, 96), None, None, torch.zeros(1, 2, 64, 96, 2))
    assert not calls


def test_cpu_tensors_fail_loudly():
    G, _ = build([64, 64, 128], 2, [64, 64, 128])
    with pytest.raises(RuntimeError):
        G.forward_bg(torch.zeros(1, 1, 4, S, S))


def test_unknown_network_name():
    with pytest.raises(ValueError):
        NetworksFactory.get_by_name("AttLWB-AdaIN", cfg=None)


def _concat_case(name):
    from tests.golden.make_golden_concat import concat_cfg, inputs
    gc = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_concat_v1.npz"))
    cfg = concat_cfg(name, 27, 4) if name == "InputConcat" else concat_cfg(name, 6)
    G = NetworksFactory.get_by_name(name, cfg=cfg, temporal=False).eval()
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    import hashlib
    assert hashlib.sha256("\n".join(f"{k}:{shapes[k]}" for k in sorted(shapes)).encode()).hexdigest() == str(gc[f"{name}/keys_sha"])
    G.load_state_dict({k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=13).items()}, strict=True)
    return G, gc, inputs()


@pytest.mark.parametrize("name", ["InputConcat", "TextureWarping"])
def test_concat_baseline_generators_through_emulated_abi(monkeypatch, name):
    """The input-concatenation baselines of the reference's factory (networks/__init__.py:38-44; input_concat_resunet.py:182-307,
    texture_warping_resunet.py:8-112): state_dict keys and the outputs of all four methods against the reference's OWN classes."""
    emu_ops.install(monkeypatch)
    G, gc, (bg_in, src_in, tsf_in) = _concat_case(name)
    enc, enc2 = G.forward_src(src_in, only_enc=True)
    assert tuple(enc.shape) == tuple(gc[f"{name}/src_enc_shape"]) and enc2 is enc
    assert G.forward_src(src_in, only_enc=False)[2:] == (None, None)
    img, mask = G.forward_tsf(tsf_in[:, 0], enc)
    assert np.abs(img.numpy() - gc[f"{name}/img"]).max() <= 2e-4 and np.abs(mask.numpy() - gc[f"{name}/mask"]).max() <= 2e-4
    bg, imgs, masks = G(bg_in, src_in, tsf_in)
    assert np.abs(bg.numpy() - gc[f"{name}/bg"]).max() <= 5e-4
    assert np.abs(imgs.numpy() - gc[f"{name}/imgs"]).max() <= 2e-4 and np.abs(masks.numpy() - gc[f"{name}/masks"]).max() <= 2e-4
    if name == "InputConcat":          # more sources than num_source are cut, fewer are repeated (:215-249)
        six = torch.cat([src_in, src_in, src_in], dim=1)
        assert torch.equal(G.forward_src(six)[0], six[:, :4].reshape(1, 24, S, S))
        assert torch.equal(G.forward_src(src_in[:, :1])[0], src_in[:, :1].repeat(1, 4, 1, 1, 1).reshape(1, 24, S, S))


def test_multi_scale_discriminator_host_logic(monkeypatch):
    """``multi_scale`` (discriminators/multi_scale_dis.py:287-332) through the factory: keys and logits against the reference's OWN class."""
    import hashlib
    import unittest.mock as um
    from ipercore_amd.networks import training as tr
    emu_ops.install(monkeypatch)
    monkeypatch.setattr(tr.ConvFn, "forward", staticmethod(_cpu_ok(tr.ConvFn.forward)))
    gc = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_concat_v1.npz"))
    D = NetworksFactory.get_by_name("multi_scale", 6, 6, ndf=32, n_layers=3, max_nf_mult=8, norm_type="instance", use_sigmoid=False)
    shapes = {k: tuple(v.shape) for k, v in D.state_dict().items()}
    assert hashlib.sha256("\n".join(f"{k}:{shapes[k]}" for k in sorted(shapes)).encode()).hexdigest() == str(gc["multi_scale/keys_sha"])
    D.load_state_dict({k: torch.tensor(v) for k, v in synthetic.fill_state_dict(shapes, seed=17).items()}, strict=True)
    gx = torch.tensor(synthetic.uniform_image((2, 6, S, S), 30, "global_x"))
    lx = torch.tensor(synthetic.uniform_image((2, 6, S, S), 31, "local_x"))
    with torch.no_grad(), um.patch.object(torch.Tensor, "is_cuda", new_callable=um.PropertyMock, return_value=True):
        outs, avg = D(gx, lx, None, None, get_avg=True)
    assert len(outs) == 3
    for i, o in enumerate(outs):
        assert np.abs(o.numpy() - gc[f"multi_scale/out{i}"]).max() <= 2e-4, i
    with pytest.raises(NotImplementedError, match="norm_type='instance'"):       # the reference's default argument: a targeted error, not a nested one
        NetworksFactory.get_by_name("multi_scale", 6, 6)
    assert abs(float(avg) - float(np.mean([gc[f"multi_scale/out{i}"].mean() for i in range(3)]))) <= 1e-4
    with pytest.raises(NotImplementedError):
        NetworksFactory.get_by_name("multi_scale", 6, 6)               # the reference default norm_type="batch" is not built


def test_training_conv_packing_cpu(monkeypatch):
    """Host logic of the backward path on CPU: the dgrad panels (stride 1, stride 2 parity launches, transposed conv),
    the wgrad K-order unpacking and ConvFn's plumbing (bias sums, channel padding, concat split), with the C ABI emulated
    (tests/emu_ops.py), against torch autograd of F.conv2d / F.conv_transpose2d."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from ipercore_amd.networks import training as tr
    emu_ops.install(monkeypatch)
    monkeypatch.setattr(tr.ConvFn, "forward", staticmethod(_cpu_ok(tr.ConvFn.forward)))
    rs = np.random.RandomState(3)
    rnd = lambda *s: torch.tensor(rs.standard_normal(s).astype(np.float32))        # noqa: E731
    cases = [("conv", 2, 8, 8, 64, 64, 3, 1, 1, 0, 1, None), ("conv", 1, 8, 8, 64, 128, 3, 2, 1, 0, 1, None),
             ("conv", 1, 6, 6, 96, 64, 3, 1, 1, 32, 0, None), ("convT", 1, 4, 4, 64, 64, 4, 2, 1, 0, 1, None),
             ("conv", 1, 8, 8, 64, 4, 5, 1, 2, 0, 0, 64), ("conv", 1, 8, 8, 64, 64, 4, 2, 1, 0, 0, None),
             ("conv", 1, 9, 9, 64, 64, 4, 1, 1, 0, 0, None), ("conv", 2, 8, 8, 64, 3, 7, 1, 3, 0, 0, 64)]
    thin_calls = []
    orig_thin = tr.ConvFn._backward_thin
    monkeypatch.setattr(tr.ConvFn, "_backward_thin", staticmethod(lambda *a: (thin_calls.append(1), orig_thin(*a))[1]))
    for kind, B, H, W, Cin, N, k, stride, pad, C1, act, n_pad in cases:
        w = rnd(*((N, Cin, k, k) if kind == "conv" else (Cin, N, 4, 4))) * 0.1
        b = rnd(N) * 0.1
        x = rnd(B, H, W, Cin)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        xn = xr.permute(0, 3, 1, 2)
        yr = F.conv2d(xn, wr, br, stride=stride, padding=pad) if kind == "conv" else F.conv_transpose2d(xn, wr, br, stride=2, padding=1)
        yr = F.relu(yr) if act else yr
        g = rnd(*yr.permute(0, 2, 3, 1).shape)
        (yr.permute(0, 2, 3, 1) * g).sum().backward()
        xd, wd, bd = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        x0, x1 = (xd, None) if not C1 else (xd[..., :Cin - C1], xd[..., Cin - C1:])
        y = tr.conv(x0, wd, bd, x1=x1, kind=kind, stride=stride, pad=pad, act=act, n_pad=n_pad)
        (y * g).sum().backward()
        for name, a_, r_ in (("y", y, yr.permute(0, 2, 3, 1)), ("dx", xd.grad, xr.grad), ("dw", wd.grad, wr.grad), ("db", bd.grad, br.grad)):
            err = (a_.detach() - r_.detach()).abs().max().item()
            assert err <= 1e-4 * max(1.0, r_.abs().max().item()), (kind, k, stride, name, err)
    assert len(thin_calls) == 2          # the two regressor shapes (N = 4 and 3) took the thin backward forms
    # a first layer that hands back dX (the discriminator under G's adversarial term): 6 real channels zero-extended to 8,
    # 32 output channels zero-extended to 64, dgrad columns 8 -> 64
    w, b, x = rnd(32, 6, 4, 4) * 0.1, rnd(32) * 0.1, rnd(1, 8, 8, 8)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr[..., :6].permute(0, 3, 1, 2), wr, b, stride=2, padding=1).permute(0, 2, 3, 1)
    g = rnd(*yr.shape)
    (yr * g).sum().backward()
    xd, wd = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = tr.conv(xd, wd, b, stride=2, pad=1, cin_pad=8, n_pad=64)
    (y * g).sum().backward()
    assert y.shape == yr.shape and (y - yr).abs().max().item() <= 1e-5
    assert (xd.grad - xr.grad).abs().max().item() <= 1e-5 and (wd.grad - wr.grad).abs().max().item() <= 1e-4
    # the fused regressor pair (HeadFn: csrc/head.hip forward, thin backward forms) against tanh / sigmoid of F.conv2d
    import unittest.mock as um
    x, wi, wa = rnd(2, 12, 12, 64), rnd(3, 64, 5, 5) * 0.05, rnd(1, 64, 5, 5) * 0.05
    xr, wir, war = (t.clone().requires_grad_(True) for t in (x, wi, wa))
    xn = xr.permute(0, 3, 1, 2)
    ir, mr = torch.tanh(F.conv2d(xn, wir, padding=2)), torch.sigmoid(F.conv2d(xn, war, padding=2))
    gi, gm = rnd(*ir.shape), rnd(*mr.shape)
    ((ir * gi).sum() + (mr * gm).sum()).backward()
    xd, wid, wad = (t.clone().requires_grad_(True) for t in (x, wi, wa))
    with um.patch.object(torch.Tensor, "is_cuda", new_callable=um.PropertyMock, return_value=True):
        img, mask = tr.HeadFn.apply(xd, wid, wad)
    ((img * gi).sum() + (mask * gm).sum()).backward()
    for name, a_, r_ in (("img", img, ir), ("mask", mask, mr), ("dx", xd.grad, xr.grad), ("dw_img", wid.grad, wir.grad), ("dw_att", wad.grad, war.grad)):
        err = (a_.detach() - r_.detach()).abs().max().item()
        assert err <= 1e-4 * max(1.0, r_.abs().max().item()), (name, err)


def _cpu_ok(fwd):
    """ConvFn refuses CPU tensors on the product path; the host-logic test runs it on the emulated ABI."""
    def wrapped(ctx, x0, x1, weight, bias, cfg):
        import unittest.mock as um
        with um.patch.object(torch.Tensor, "is_cuda", new_callable=um.PropertyMock, return_value=True):
            return fwd(ctx, x0, x1, weight, bias, cfg)
    return wrapped


def test_discriminator_variants_host_logic(monkeypatch):
    """multi_scale_dis.py:47-284 host side: sub-network inventory / parameter names, output order (bg, global, body, head; global
    first for patch_global), crop_img's resize and its dropping of degenerate boxes, reduce_tensor - with the sub-networks stubbed
    (their convolutions are GPU checks)."""
    from ipercore_amd import trainers as T
    from ipercore_amd.synthetic import AttrDict
    cfg = AttrDict(cond_nc=6, bg_cond_nc=4, ndf=32, n_layers=4, max_nf_mult=8, norm_type="instance", use_sigmoid=False)
    D = NetworksFactory.get_by_name("patch_global_body_head", cfg, use_aug_bg=True)
    keys = list(D.state_dict().keys())
    assert len(keys) == 48 and all(f"{m}.model.{i}.{p}" in keys for m in ("global_model", "body_model", "head_model", "bg_model")
                                   for i in (0, 2, 5, 8, 11, 14) for p in ("weight", "bias"))
    assert tuple(D.bg_model.model.__getattr__("0").weight.shape) == (32, 4, 4, 4) and tuple(D.head_model.model.__getattr__("14").weight.shape) == (1, 256, 4, 4)
    seen = []

    def stub(self, x):
        seen.append((self.input_nc, tuple(x.shape)))
        return x.mean(dim=1, keepdim=True)
    monkeypatch.setattr(T.PatchDiscriminator, "forward", stub)
    x, bg = torch.rand(3, 6, 64, 64), torch.rand(3, 4, 64, 64)
    body = torch.tensor([[4, 40, 2, 60], [8, 8, 0, 30], [0, 64, 0, 64]])
    head = torch.tensor([[10, 30, 2, 20], [12, 28, 4, 4], [5, 5, 0, 9]])
    outs, avg = D({"x": x, "bg_x": bg, "body_rects": body, "head_rects": head, "get_avg": True})
    assert seen == [(4, (3, 4, 64, 64)), (6, (3, 6, 64, 64)), (6, (2, 6, 32, 32)), (6, (1, 6, 16, 16))]
    assert abs(avg.item() - sum(o.mean().item() for o in outs) / 4) < 1e-6
    ref = F.interpolate(x[0:1, :, 2:60, 4:40], size=(32, 32), mode="bilinear", align_corners=True)
    assert torch.equal(T.crop_img(x, body, 2)[0:1], ref)
    assert len(T.crop_img(x, torch.tensor([[1, 1, 0, 5]] * 3), 2)) == 0
    G1 = NetworksFactory.get_by_name("patch_global", cfg, use_aug_bg=True)
    outs = G1({"x": x, "bg_x": bg, "get_avg": False})
    assert [o.shape[1] for o in outs] == [1, 1] and torch.equal(outs[0], x.mean(dim=1, keepdim=True))     # [global, bg]
    outs = G1(x)                                                                                           # the trainer's tensor form
    assert len(outs) == 1
    with pytest.raises(TypeError):                  # multi_scale has the reference's own positional signature (global_nc, input_nc, ...), not (cfg)
        NetworksFactory.get_by_name("multi_scale", cfg)


def _as_device(fn):
    """Run ``fn`` with torch.Tensor.is_cuda reporting True: the product's CUDA-only guards and its device-side packing /
    un-packing paths are taken, against the emulated C ABI (tests/emu_ops.py)."""
    import unittest.mock as um
    with um.patch.object(torch.Tensor, "is_cuda", new_callable=um.PropertyMock, return_value=True):
        return fn()


def test_training_graph_host_logic_cpu(monkeypatch):
    """The WHOLE training graph of the generator (bg + src with decoder + tsf: every ConvFn / NormAct / AttnFn / HeadFn, the thin
    regressor backward, concat splits, parity launches) and of the discriminator (incl. the gradient it hands back to its input)
    through the emulated C ABI on the CPU, against torch autograd through the oracle: outputs and EVERY parameter gradient.
    The GPU suite makes the same comparison on the kernels (check_generator_training_grads); this one keeps the host logic under
    the CPU suite."""
    from ipercore_amd.networks.training import TrainableGenerator
    from ipercore_amd.trainers import PatchDiscriminator
    emu_ops.install(monkeypatch)
    S_, ns, nf, nres, bgf = 32, 2, [64, 64, 128], 1, [64, 64, 128]
    G = NetworksFactory.get_by_name("AttLWB-SPADE", cfg=synthetic.gen_cfg(nf, nres, bgf), temporal=False)
    sdn = synthetic.fill_state_dict(generator_param_shapes(nf, nres, bgf), seed=7)
    G.load_state_dict({k: torch.tensor(v) for k, v in sdn.items()}, strict=True)
    G.train()
    u = lambda shape, seed, name: torch.tensor(synthetic.uniform_image(shape, seed, name))           # noqa: E731
    bg_in, src_in, tsf_in = u((1, 1, 4, S_, S_), 10, "bg_inputs"), u((1, ns, 6, S_, S_), 8, "src_inputs"), u((1, 1, 6, S_, S_), 9, "tsf_inputs")
    Tst = u((1, 1, ns, S_, S_, 2), 11, "Tst") * 1.1                                                   # some samples leave the image
    tgt = [u(s, 500 + i, "tgt") for i, s in enumerate(((1, 1, 3, S_, S_), (1, ns, 3, S_, S_), (1, ns, 1, S_, S_), (1, 1, 3, S_, S_), (1, 1, 1, S_, S_)))]
    loss_of = lambda outs: sum(((o - t) ** 2).mean() for o, t in zip(outs, tgt))                      # noqa: E731  (smooth: no sign flips)
    sd = {k: torch.tensor(v, requires_grad=True) for k, v in sdn.items()}
    outs_ref = orc.gen_forward_train(sd, bg_in, src_in, tsf_in, Tst, n_down=len(nf), n_res=nres, n_bg=len(bgf))
    loss_of(outs_ref).backward()

    def run():
        outs = TrainableGenerator(G).forward(bg_in, src_in, tsf_in, Tst)
        loss_of(outs).backward()
        return outs
    outs = _as_device(run)
    for name, a_, b_ in zip(("bg", "src_img", "src_mask", "tsf_img", "tsf_mask"), outs, outs_ref):
        assert (a_.detach() - b_.detach()).abs().max().item() <= 2e-4, name
    gmax = max(v.grad.abs().max().item() for v in sd.values())
    for k, p_ in G.named_parameters():
        assert p_.grad is not None, f"no gradient for {k}"
        rel = (p_.grad - sd[k].grad).abs().max().item() / max(sd[k].grad.abs().max().item(), 1e-3 * gmax)
        assert rel <= 2e-3, (k, rel)
    # the discriminator: logits, parameter gradients and the gradient handed back to the input (G's adversarial term)
    torch.manual_seed(1)
    D = PatchDiscriminator(6, 32, 3, 8)
    dsd = {"d." + k: v.detach().clone().requires_grad_(True) for k, v in D.state_dict().items()}
    x = u((2, 6, 64, 64), 12, "dis_x")
    xr = x.clone().requires_grad_(True)
    (orc.patch_discriminator(dsd, "d.", xr, 3) ** 2).mean().backward()
    xd = x.clone().requires_grad_(True)

    def run_d():
        out = D(xd)
        (out ** 2).mean().backward()
        return out
    out = _as_device(run_d)
    assert (out.detach() - orc.patch_discriminator(dsd, "d.", x, 3).detach()).abs().max().item() <= 1e-4
    assert xd.grad is not None and (xd.grad - xr.grad).abs().max().item() <= 2e-3 * xr.grad.abs().max().item()
    dmax = max(v.grad.abs().max().item() for v in dsd.values())
    for k, p_ in D.named_parameters():
        ref = dsd["d." + k].grad
        assert (p_.grad - ref).abs().max().item() <= 2e-3 * max(ref.abs().max().item(), 1e-3 * dmax), k


def test_trainer_step_host_logic_cpu(monkeypatch):
    """One LWGTrainer.optimize_parameters() (lwg_trainer.py:326-352: forward, G loss / backward / Adam, then the D loss on the
    detached fakes of the SAME forward / backward / Adam) through the emulated C ABI: both loss values and the gradients the two
    Adam updates consumed (the flat buffers of FlatAdam) against torch autograd through the oracle's generator and discriminator
    with the loss assembly restated here; the parameters move."""
    from ipercore_amd.trainers import LWGTrainer, PatchGlobalDiscriminator, TrainOpts
    emu_ops.install(monkeypatch)
    S_, ns, nf, nres, bgf = 32, 2, [64, 64, 128], 1, [64, 64, 128]
    G = NetworksFactory.get_by_name("AttLWB-SPADE", cfg=synthetic.gen_cfg(nf, nres, bgf), temporal=False)
    sdn = synthetic.fill_state_dict(generator_param_shapes(nf, nres, bgf), seed=7)
    G.load_state_dict({k: torch.tensor(v) for k, v in sdn.items()}, strict=True)
    G.train()
    torch.manual_seed(2)
    D = PatchGlobalDiscriminator(ndf=32, n_layers=3)
    u = lambda shape, seed, name: torch.tensor(synthetic.uniform_image(shape, seed, name))           # noqa: E731
    inp = {"input_G_bg": u((1, 1, 4, S_, S_), 10, "bg_inputs"), "input_G_src": u((1, ns, 6, S_, S_), 8, "src_inputs"),
           "input_G_tsf": u((1, 1, 6, S_, S_), 9, "tsf_inputs"), "Tst": u((1, 1, ns, S_, S_, 2), 11, "Tst"),
           "real_src": u((1, ns, 3, S_, S_), 700, "real_src"), "real_tsf": u((1, 1, 3, S_, S_), 701, "real_tsf"),
           "real_bg": u((1, 3, S_, S_), 702, "real_bg"), "body_mask": (u((1, ns + 1, 1, S_, S_), 703, "mask") > 0).float()}
    sdG = {k: torch.tensor(v, requires_grad=True) for k, v in sdn.items()}
    sdD = {k: v.detach().clone().requires_grad_(True) for k, v in D.state_dict().items()}
    o = TrainOpts.l1_transfer()
    # ---- the step restated on the oracle (plain torch, CPU)
    bg, s_col, s_mask, t_col, t_mask = orc.gen_forward_train(sdG, inp["input_G_bg"], inp["input_G_src"], inp["input_G_tsf"], inp["Tst"],
                                                             n_down=len(nf), n_res=nres, n_bg=len(bgf))
    fake_src, fake_tsf = s_mask * bg + (1 - s_mask) * s_col, t_mask * bg + (1 - t_mask) * t_col
    cond = inp["input_G_tsf"][:, :, -3:].reshape(1, 3, S_, S_)
    dis = lambda x: orc.patch_discriminator(sdD, "global_model.", x, 3)                                 # noqa: E731
    l1 = torch.nn.functional.l1_loss
    fm = torch.cat([s_mask, t_mask], dim=1).view(-1, 1, S_, S_)
    tv = (fm[:, :, :, :-1] - fm[:, :, :, 1:]).abs().mean() + (fm[:, :, :-1, :] - fm[:, :, 1:, :]).abs().mean()
    loss_g = ((l1(fake_src, inp["real_src"]) + l1(bg.view(-1, 3, S_, S_), inp["real_bg"])) / 2 * o.lambda_rec
              + l1(fake_tsf.view(1, 3, S_, S_), inp["real_tsf"].view(1, 3, S_, S_)) * o.lambda_tsf
              + (dis(torch.cat([fake_tsf.view(1, 3, S_, S_), cond], dim=1)) ** 2).mean() * o.lambda_D_prob
              + torch.nn.functional.binary_cross_entropy(fm, inp["body_mask"].view(-1, 1, S_, S_)) * o.lambda_mask + tv * o.lambda_mask_smooth)
    gG = torch.autograd.grad(loss_g, list(sdG.values()))
    d_real = dis(torch.cat([inp["real_tsf"].view(1, 3, S_, S_), cond], dim=1))
    d_fake = dis(torch.cat([fake_tsf.detach().view(1, 3, S_, S_), cond], dim=1))
    loss_d = ((d_real - 1) ** 2).mean() + ((d_fake + 1) ** 2).mean()
    gD = torch.autograd.grad(loss_d, list(sdD.values()))
    # ---- the product's step
    tr = LWGTrainer(G, D, opts=o)
    # FlatAdam lays the parameter pairs that run as ONE stacked convolution out back to back: their concatenation is a VIEW of the flat
    # buffer (no copy), for SPADE's gamma | beta weights and biases and the attention's fk | fv weights of every site
    from ipercore_amd.networks.training import _stacked
    n_pairs = 0
    for name, mod in G.named_modules():
        pairs = []
        if name.endswith(".spade"):
            pairs = [(mod.mlp_gamma.weight, mod.mlp_beta.weight), (mod.mlp_gamma.bias, mod.mlp_beta.bias)]
        elif hasattr(mod, "fk") and hasattr(mod, "fv"):
            pairs = [(mod.fk.weight, mod.fv.weight)]
        for a_, b_ in pairs:
            st = _stacked(a_, b_)
            assert st.data_ptr() == a_.data_ptr() and torch.equal(st, torch.cat([a_.detach(), b_.detach()], dim=0))
            n_pairs += 1
    assert n_pairs == 3 * (len(nf) + nres), n_pairs
    tr.set_input(inp)
    w0 = {k: v.detach().clone() for k, v in list(G.state_dict().items()) + list(D.state_dict().items())}
    lg, ld = _as_device(tr.optimize_parameters)
    assert abs(lg.item() - loss_g.item()) <= 1e-4 * abs(loss_g.item()) and abs(ld.item() - loss_d.item()) <= 1e-4 * abs(loss_d.item())
    for mod, ref_sd, ref_g in ((G, sdG, gG), (D, sdD, gD)):
        gmax = max(g_.abs().max().item() for g_ in ref_g)
        for (k, p_), g_ in zip(mod.named_parameters(), ref_g):
            assert list(ref_sd.keys()).index(k) >= 0 and p_.grad is not None, k
            g_ = ref_g[list(ref_sd.keys()).index(k)]
            assert (p_.grad - g_).abs().max().item() <= 2e-3 * max(g_.abs().max().item(), 1e-3 * gmax), k
    moved = {k: (v.detach() - w0[k]).abs().max().item() for k, v in list(G.state_dict().items()) + list(D.state_dict().items())}
    # Adam's first step: |dw| <= lr; every weight tensor is updated (a bias in front of an InstanceNorm has a zero gradient)
    assert max(moved.values()) <= 1.001e-4 and all(v > 0 for k, v in moved.items() if k.endswith("weight")), moved


def test_loss_networks_host_logic_cpu(monkeypatch):
    """The two frozen loss networks of the personalization step through the emulated C ABI on the CPU:
    Sphere20aFeatures / FaceLoss against outputs of the reference's OWN Sphere20a (golden_faceloss_v1.npz: five features + the loss),
    VGG19Features / VGGLoss against the same network written with F.conv2d / F.max_pool2d (value and image gradient; frozen
    weights take ConvFn's data-gradient-only path)."""
    from tests.golden.make_golden_faceloss import face_state_dict
    from ipercore_amd.trainers import FaceLoss, VGGLoss
    emu_ops.install(monkeypatch)
    gf = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_faceloss_v1.npz"))
    crt = FaceLoss(None, allow_seeded=True)
    crt.net.load_state_dict(face_state_dict(), strict=True)
    x, y = torch.tensor(synthetic.uniform_image((2, 3, 112, 96), 70, "face_x")), torch.tensor(synthetic.uniform_image((2, 3, 112, 96), 71, "face_y"))

    def run_face():
        with torch.no_grad():
            return crt.net(x), crt(x, y)
    fx, loss = _as_device(run_face)
    for i, f in enumerate(fx):
        got = f.permute(0, 3, 1, 2)[:, ::8] if f.dim() == 4 else f
        assert (got - torch.tensor(gf[f"fx{i}"])).abs().max().item() <= 2e-4, i
    assert abs(loss.item() - float(gf["loss"])) <= 2e-4 * abs(float(gf["loss"]))
    # head boxes stay device tensors (faceloss.py:384-406 reads them on the host and drops empty ones): crops for every sample + a validity flag,
    # the L1 means over the valid samples only = the reference's value on the kept samples; the gradient reaches the images through the crop
    imgs_a = torch.tensor(synthetic.uniform_image((3, 3, 128, 128), 72, "face_a")).requires_grad_(True)
    imgs_b = torch.tensor(synthetic.uniform_image((3, 3, 128, 128), 73, "face_b"))
    box = torch.tensor([[20, 84, 10, 90], [5, 5, 0, 10], [0, 128, 3, 128]])

    def run_boxes():
        l_ = crt(imgs_a, imgs_b, bbox1=box, bbox2=box)
        l_.backward()
        with torch.no_grad():
            keep = [0, 2]
            cut = lambda t: torch.cat([F.interpolate(t[i:i + 1, :, box[i, 2]:box[i, 3], box[i, 0]:box[i, 1]], size=(112, 96), mode="bilinear",   # noqa: E731
                                                     align_corners=True) for i in keep])
            return l_.detach(), crt(cut(imgs_a.detach()), cut(imgs_b))
    l_dev, l_ref = _as_device(run_boxes)
    assert abs(l_dev.item() - l_ref.item()) <= 1e-5 * abs(l_ref.item())
    assert imgs_a.grad is not None and float(imgs_a.grad[1].abs().max()) == 0.0 and float(imgs_a.grad[0].abs().max()) > 0      # dropped sample: no gradient
    assert float(imgs_a.grad[0, :, :10].abs().max()) == 0.0                                                                     # outside the box: none either
    # VGG19 perceptual loss at a small size
    vcrt = VGGLoss(ckpt_path=None, allow_seeded=True)
    sd = {k: v.detach() for k, v in vcrt.vgg.state_dict().items()}

    def ref_feats(t):
        outs = []
        for item in vcrt.vgg.CFG:
            if item == "M":
                t = F.max_pool2d(t, 2, 2)
                continue
            t = F.relu(F.conv2d(t, sd[f"features.{item[0]}.weight"], sd[f"features.{item[0]}.bias"], padding=1))
            if item[0] in vcrt.vgg.TAPS:
                outs.append(t)
        return outs
    a, b = torch.tensor(synthetic.uniform_image((1, 3, 32, 32), 990, "vgg_x")) * 0.5, torch.tensor(synthetic.uniform_image((1, 3, 32, 32), 991, "vgg_y")) * 0.5
    with torch.no_grad():
        fb = ref_feats(b)
    ar = a.clone().requires_grad_(True)
    sum(w * F.mse_loss(p, q) for w, p, q in zip(vcrt.WEIGHTS, ref_feats(ar), fb)).backward()
    ad = a.clone().requires_grad_(True)

    def run_vgg():
        with torch.no_grad():
            fbd = vcrt.vgg(b)
        feats = vcrt.vgg(ad)
        sum(w * F.mse_loss(p, q) for w, p, q in zip(vcrt.WEIGHTS, feats, fbd)).backward()
        return feats
    feats = _as_device(run_vgg)
    for p, q in zip(feats, ref_feats(a)):
        q = q.permute(0, 2, 3, 1) if p.shape != q.shape else q
        assert (p.detach() - q).abs().max().item() <= 1e-4 * max(1.0, q.abs().max().item())
    assert (ad.grad - ar.grad).abs().max().item() <= 2e-3 * ar.grad.abs().max().item()
    assert all(p.grad is None for p in vcrt.vgg.parameters())                  # frozen: no weight gradients were formed
