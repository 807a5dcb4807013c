"""AttLWB-SPADE generator behind the reference's generator API, executed by hand-written HIP kernels.

Drop-in for ``iPERCore.models.networks.generators.AttentionLWBGenerator``
(attlwb_spade_resunet.py:538-699) / ``AttentionLWBFrontGenerator`` (:702-834):

* same constructor ``(cfg, temporal=False)``, same ``state_dict`` keys (checkpoint-compatible), still an
  ``nn.Module`` (``.parameters()``, ``.to()``, ``.eval()``);
* same methods and tensor conventions: ``forward_bg`` (:615-631), ``forward_src`` (:450-478),
  ``forward_tsf`` (:480-535), ``forward`` (:633-699) - NCHW fp32 in / out.

MI355X design (not a translation of the reference's layer objects):
  - the module is a parameter tree; kernels read re-packed weight panels (``packing.py``) cached per weight
    version;
  - activations are NHWC fp32 end to end; every conv is one launch of the MFMA implicit-GEMM kernel with its
    bias / ReLU / residual / SPADE epilogue fused; a transposed conv is 4 parity launches, the skip concat is
    read in place (two input pointers), the output head + compositing is one kernel;
  - the 1x1 ``fk``/``fv`` convs are hoisted out of the frame loop: ``Wk x_src`` / ``Wv x_src`` depend only on
    the cached source features, and a 1x1 conv commutes with the zero-padded bilinear warp (bias added after
    the warp, as the reference does) - see ``csrc/lwb_attn.hip``;
  - frames are independent when ``temporal=False`` so the engine takes a batch of frames per call.

These methods are the inference engine: they run under ``torch.no_grad()`` on CUDA tensors and raise on CPU tensors (no
fallback).  The differentiable forward for the personalization step lives in ``networks/training.py``
(``TrainableGenerator`` over the same parameter tree; ``trainers.LWGTrainer``).
"""
import math

import torch
import torch.nn as nn

from .. import ops
from . import packing
from .params import generator_param_shapes


class _Node(nn.Module):
    """A bare container: parameters live at the leaves of a tree of these, named like the reference's modules."""

    def __init__(self):
        super().__init__()


class ParamTree(nn.Module):
    def __init__(self, shapes):
        super().__init__()
        self._shapes = dict(shapes)
        for name, shape in shapes.items():
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if not hasattr(node, p):
                    node.add_module(p, _Node())
                node = getattr(node, p)
            node.register_parameter(parts[-1], nn.Parameter(torch.empty(*shape)))
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        """PyTorch's default Conv2d/ConvTranspose2d init: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), ..)."""
        named = dict(self.named_parameters())
        for name, p in named.items():
            if name.endswith(".weight"):
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                bound = 1.0 / math.sqrt(fan_in)
                p.uniform_(-bound, bound)
                b = named.get(name[:-len("weight")] + "bias")
                if b is not None:
                    b.uniform_(-bound, bound)


class _Packed:
    """Packed weight panels of one generator, rebuilt when a parameter changes (version counters)."""

    def __init__(self, gen):
        sd = {k: v for k, v in gen.named_parameters()}
        dev = next(gen.parameters()).device
        self.device = dev
        self.version = _param_version(gen)
        nf, n_res, n_down = gen.num_filters, gen.n_res_block, len(gen.num_filters)
        P = packing

        def conv(name, stride=1, pad=None, cin_pad=None, n_pad=None):
            return P.spec_to(P.pack_conv(sd[name + ".weight"], sd.get(name + ".bias"), stride, pad, cin_pad, n_pad), dev)

        def convT(name, n_pad=None):
            return [P.spec_to(s, dev) for s in P.pack_conv_transpose(sd[name + ".weight"], sd.get(name + ".bias"), n_pad)]

        def site(p):
            # the query projection folded into the source side (csrc/lwb_attn_x.hip): Kq = (Wq^T Wk) f, kappa = (Wk^T bq) . f per source
            # texel - the per-frame fq convolution, the q tensor and bk (constant over the sources: it cancels in the softmax) disappear
            Wq = sd[p + ".fq.weight"].detach()[:, :, 0, 0].double()
            Wk = sd[p + ".fk.weight"].detach()[:, :, 0, 0].double()
            bq = sd[p + ".fq.bias"].detach().double()
            Wkq = (Wq.t() @ Wk).float()[:, :, None, None].contiguous()
            wkap = (Wk.t() @ bq).float()[None, :, None, None].contiguous()
            d = {
                "fkq": P.spec_to(P.pack_conv(Wkq, None, 1, 0), dev),
                "fkap": P.spec_to(P.pack_conv(wkap, None, 1, 0, None, 64), dev),          # one output column, zero-extended to the kernel's 64
                "fv": P.spec_to(P.pack_conv(sd[p + ".fv.weight"], None, 1, 0), dev),     # bias added after the warp
                "bv": sd[p + ".fv.bias"].detach().float().contiguous(),
                "shared": conv(p + ".spade.mlp_shared.0"),
                "gb": P.spec_to(P.pack_spade_gamma_beta(sd[p + ".spade.mlp_gamma.weight"], sd[p + ".spade.mlp_gamma.bias"],
                                                        sd[p + ".spade.mlp_beta.weight"], sd[p + ".spade.mlp_beta.bias"]), dev),
            }
            return d

        cin_pad = 8
        self.tsf_enc = [conv(f"tsf_net_enc.layers.{i}.0", stride=2, cin_pad=cin_pad if i == 0 else None) for i in range(n_down)]
        self.src_enc = [conv(f"src_net.encoders.layers.{i}.0", stride=2, cin_pad=cin_pad if i == 0 else None) for i in range(n_down)]
        self.src_res = [(conv(f"src_net.res_blocks.{i}.main.0"), conv(f"src_net.res_blocks.{i}.main.2")) for i in range(n_res)]
        self.src_dec = [convT(f"src_net.decoders.layers.{i}.0") for i in range(n_down)]
        self.src_head = P.pack_head(sd["src_net.img_reg.0.weight"], sd["src_net.att_reg.0.weight"]).to(dev)
        self.res = [(conv(f"res_blocks.{i}.main.0"), conv(f"res_blocks.{i}.main.2")) for i in range(n_res)]
        if gen.lwb_kind == "att":
            pass
        elif gen.lwb_kind in ("sg_add", "sg_avg"):
            site = lambda p: {"g0": conv(p + ".gate_conv.0"), "g2": conv(p + ".gate_conv.2")}      # noqa: E731
        else:
            site = lambda p: {}                                                                     # noqa: E731
        self.enc_sites = [site(f"enc_attlwbs.{i}") for i in range(n_down)]
        self.res_sites = [site(f"res_attlwbs.{i}") for i in range(n_res)]
        self.upconvs = [convT(f"tsf_net_dec.upconvs.{i}.0") for i in range(n_down)]
        self.skippers = [conv(f"tsf_net_dec.skippers.{i}.0") for i in range(n_down - 1)]
        self.head = P.pack_head(sd["tsf_img_reg.0.weight"], sd["tsf_att_reg.0.weight"]).to(dev)
        # bf16 mode: the same regressors as an MFMA operand panel (csrc/bf16_ops.hip); needs the 64-channel decoder output
        self.head16 = P.pack_head_bf16(sd["tsf_img_reg.0.weight"], sd["tsf_att_reg.0.weight"]).to(dev) if nf[0] == 64 else None
        self.bg = pack_bg_layers(sd, gen.bg_filters, n_res, dev) if gen.has_bg else None


def pack_bg_layers(sd, bgf, n_res, dev):
    """The background network (ResNetInpaintor, bg_inpaintor.py:24-60) as a list of (kind, packed spec) in execution order."""
    P = packing

    def conv(name, stride=1, pad=None, cin_pad=None, n_pad=None):
        return P.spec_to(P.pack_conv(sd[name + ".weight"], sd.get(name + ".bias"), stride, pad, cin_pad, n_pad), dev)

    def convT(name, n_pad=None):
        return [P.spec_to(s, dev) for s in P.pack_conv_transpose(sd[name + ".weight"], sd.get(name + ".bias"), n_pad)]
    i = 0
    layers = [("conv", conv(f"bg_net.main.{i}", stride=1, pad=3, cin_pad=4))]
    i += 3
    for d in range(1, len(bgf)):
        layers.append(("conv", conv(f"bg_net.main.{i}", stride=2)))
        i += 3
    for _ in range(n_res):
        layers.append(("res", (conv(f"bg_net.main.{i}.main.0"), conv(f"bg_net.main.{i}.main.3"))))
        i += 1
    for d in range(len(bgf) - 1, 0, -1):
        layers.append(("convT", convT(f"bg_net.main.{i}")))
        i += 3
    layers.append(("out", conv(f"bg_net.main.{i}", stride=1, pad=3, n_pad=64)))
    return layers


def run_bg_layers(layers, bg4):
    """bg4 (n,S,S,4) NHWC through the packed background network -> (n,3,S,S) NCHW (InstanceNorm after every conv, tanh at the end)."""
    scratch = _Scratch()
    x = bg4

    def norm(t, act, res=None):
        B, h, w, C = t.shape
        mean, rstd = t.new_empty(B, C), t.new_empty(B, C)
        nsplit = max(1, min(64, (h * w) // 64))
        ops.instnorm_stats(t, mean, rstd, scratch.get(B * C * nsplit * 3, t.device), eps=1e-5, nsplit=nsplit)
        return ops.instnorm_apply(t, mean, rstd, torch.empty_like(t), act=act, res=res)

    for kind, spec in layers:
        B, H, W, _ = x.shape
        if kind == "conv":
            y = ops.conv2d(x, spec, x.new_empty(B, H // spec.stride, W // spec.stride, spec.N))
            x = norm(y, ops.ACT_RELU)
        elif kind == "res":
            h = norm(ops.conv2d(x, spec[0], torch.empty_like(x)), ops.ACT_RELU)
            x = norm(ops.conv2d(h, spec[1], torch.empty_like(x)), ops.ACT_NONE, res=x)
        elif kind == "convT":
            y = x.new_empty(B, 2 * H, 2 * W, spec[0].N)
            x = norm(ops.conv_transpose2d(x, spec, y, act=ops.ACT_NONE), ops.ACT_RELU)
        else:
            y = ops.conv2d(x, spec, x.new_empty(B, H, W, spec.N), act=ops.ACT_TANH)
            return ops.nhwc_to_nchw(y, channels=3)


def _param_version(gen):
    return tuple((p.data_ptr(), p._version) for p in gen.parameters())


class SourceFeatures:
    """Per-source cache (NHWC): encoder / res-block features and their hoisted K/V projections per AttLWB site."""

    def __init__(self, enc, res, kv, ns, batched=False):
        self.enc, self.res, self.kv, self.ns, self.batched = enc, res, kv, ns, batched


class AttentionLWBGenerator(nn.Module):
    has_bg = True
    # the Liquid Warping Block of the transfer stream: "att" (SelfAttentionLWB + SPADE), "add" / "avg" (AddLWB / AvgLWB),
    # "sg_add" / "sg_avg" (SoftGateLWB) - see the subclasses at the end of this file
    lwb_kind = "att"

    def __init__(self, cfg, temporal=False):
        super().__init__()
        self._name = _get(cfg, "name", "AttLWB-SPADE")
        tsf = _get(cfg, "TSFNet")
        sid = _get(cfg, "SIDNet")
        self.num_filters = [int(c) for c in _get(tsf, "num_filters")]
        self.n_res_block = int(_get(tsf, "n_res_block"))
        if [int(c) for c in _get(sid, "num_filters")] != self.num_filters or int(_get(sid, "n_res_block")) != self.n_res_block:
            raise ValueError("SIDNet and TSFNet must share num_filters / n_res_block (the LWB sites pair them up)")
        for c in self.num_filters:
            if c not in (64, 128, 256):
                raise ValueError(f"num_filters entries must be in {{64,128,256}} for the HIP kernels, got {c}")
        self.cond_nc = int(_get(tsf, "cond_nc"))
        self.temporal = temporal
        self.bg_filters, bg_cond = None, 4
        if self.has_bg:
            bgc = _get(cfg, "BGNet")
            self.bg_filters = [int(c) for c in _get(bgc, "num_filters")]
            bg_cond = int(_get(bgc, "cond_nc"))
        shapes = generator_param_shapes(self.num_filters, self.n_res_block, self.bg_filters or (), cond_nc=self.cond_nc,
                                        bg_cond_nc=bg_cond, with_bg=self.has_bg,
                                        lwb={"att": "att", "add": "plain", "avg": "plain"}.get(self.lwb_kind, "softgate"))
        tree = ParamTree(shapes)
        for name, child in tree.named_children():     # graft the tree's top-level nodes onto this module
            self.add_module(name, child)
        self._packed = None
        # How the engine's convolutions run (ops.conv_precision):
        # "winograd" (default since round 5; BASELINE configs[1] / [2]: "AttLWB generator fp32"): fp32 tensors and fp32 MFMA arithmetic throughout;
        #   the 3x3 / stride 1 layers (SPADE, residual blocks, skip convolutions: 81 % of the flops) as fused Winograd convolutions - F(4x4,3x3)
        #   (csrc/conv_winograd4.hip, round 6: 36 products per 4 x 4 outputs instead of 144) for Cin >= 64, F(2x2,3x3) (csrc/conv_winograd.hip: 16 per
        #   2 x 2 instead of 36) below -, the transposed convolutions as F(2x2,2x2), everything else on the direct implicit-GEMM kernel;
        # "winograd2x2": the same with the F(4x4,3x3) kernel off (rounds 5 / 6's engine) - the latency engine of callers that render ONE frame per
        #   call (2.4 against 3.6 ms per frame at frame batch 1: a one-frame F(4x4,3x3) launch is 32 workgroups on a 64^2 layer), 20 % slower on batches;
        # "fp32": every layer on the direct kernel (the rounds 1-4 default; frames differ from "winograd" by ~3e-6);
        # "bf16": BASELINE configs[3] - every activation tensor of the engine is stored as bf16, the convs run on the bf16 MFMA kernel
        #   (fp32 accumulation), InstanceNorm statistics / attention / head read bf16; the first conv of a stream takes the fp32 input;
        # "split": fp32 tensors, every product formed from six bf16 MFMAs (exact three-way operand split).
        self.conv_precision = "winograd"

    # ------------------------------------------------------------------ plumbing
    def packed(self):
        if self._packed is None or self._packed.version != _param_version(self):
            self._packed = _Packed(self)
        return self._packed

    def _check(self, *tensors):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            raise NotImplementedError("these methods are the no_grad inference engine: call under torch.no_grad() / .eval(), or train "
                                      "through ipercore_amd.networks.training.TrainableGenerator (trainers.LWGTrainer)")
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise RuntimeError("ipercore_amd generator runs on the MI355X only: got a CPU tensor (no fallback)")

    @staticmethod
    def _check_square(t):
        """The attention blocks take flows resized to their (square) feature size; image_size is one number throughout the reference
        (deploy.toml image_size, options_base.py): a non-square input is refused HERE, before any network has run."""
        if t.shape[-1] != t.shape[-2]:
            raise ValueError(f"square images only (got {t.shape[-2]} x {t.shape[-1]}): the reference's image_size is one number and the "
                             "attention blocks' flow fields are resized to square feature maps")

    # ------------------------------------------------------------------ NHWC engine
    @torch.no_grad()
    def _encode_sources_impl(self, src8, batched=False, ns=None):
        """src8: (n, S, S, 8) NHWC (6 used) -> SourceFeatures with K/V panels for the 9 AttLWB sites."""
        pk = self.packed()
        x = src8
        adt = self._act_dtype()
        enc, res = [], []
        for i, spec in enumerate(pk.src_enc):
            n, H, W, _ = x.shape
            y = x.new_empty(n, H // 2, W // 2, spec.N, dtype=adt)
            x = ops.conv2d(x, spec, y, act=ops.ACT_RELU)
            enc.append(x)
        for c0, c1 in pk.src_res:
            h = ops.conv2d(x, c0, torch.empty_like(x), act=ops.ACT_RELU)
            x = ops.conv2d(h, c1, torch.empty_like(x), epi=ops.EPI_RESIDUAL, res=x)
            res.append(x)
        return SourceFeatures(enc, res, self._project_sources(pk, enc, res), ns if ns is not None else src8.shape[0], batched)

    def _project_sources(self, pk, enc, res):
        """Per LWB site what the per-frame kernel gathers from: the hoisted K / V projections (attention) or the source
        features themselves (Add / Avg / SoftGate blocks warp the raw features)."""
        kv = []
        for feats, sites in ((enc, pk.enc_sites), (res, pk.res_sites)):
            for f, st in zip(feats, sites):
                if self.lwb_kind == "att":
                    n, h, w, _ = f.shape
                    with ops.conv_precision("fp32" if self.conv_precision == "bf16" else self.conv_precision):     # the logit offset stays fp32
                        kap = ops.conv2d(f.float(), st["fkap"], f.new_empty(n, h, w, 64, dtype=torch.float32))[..., 0].contiguous()
                    kv.append((ops.conv2d(f, st["fkq"], torch.empty_like(f)), ops.conv2d(f, st["fv"], torch.empty_like(f)), kap))
                else:
                    kv.append((f, None, None))
        return kv

    def _attlwb(self, st, tsf_x, kv, Tst, batched, scratch):
        B, h, w, C = tsf_x.shape
        if self.lwb_kind != "att":
            ns = Tst.shape[1]
            Tst = scratch.flow(Tst, h, w)
            if self.lwb_kind == "add":
                return ops.lwb_fuse(tsf_x, kv[0], Tst, torch.empty_like(tsf_x), src_batched=batched)
            if self.lwb_kind == "avg":
                return ops.lwb_fuse(tsf_x, kv[0], Tst, torch.empty_like(tsf_x), scale_o=1.0 / (ns + 1), src_batched=batched)
            g = ops.conv2d(tsf_x, st["g0"], torch.empty_like(tsf_x), act=ops.ACT_RELU)
            g = ops.conv2d(g, st["g2"], torch.empty_like(tsf_x), act=ops.ACT_SIGMOID)
            return ops.lwb_fuse(tsf_x, kv[0], Tst, torch.empty_like(tsf_x), gate=g,
                                scale_w=1.0 if self.lwb_kind == "sg_add" else 1.0 / ns, src_batched=batched)
        # ONE pass over tsf_x: the attention (query projection folded into kv[0] = Kq / kv[2] = kappa) and the per-tile partial statistics
        # of SPADE's InstanceNorm of tsf_x
        mean = tsf_x.new_empty(B, C, dtype=torch.float32)
        rstd = tsf_x.new_empty(B, C, dtype=torch.float32)
        nrec = ops.attn_records(h, w, C, tsf_x.dtype)
        ws = scratch.get(ops.instnorm_finalize_ws(B, C, nrec), tsf_x.device)
        flow = scratch.flow(Tst, h, w)
        if tuple(flow.shape[2:4]) != (h, w):
            raise NotImplementedError("the attention block takes flows resized to its feature size: non-square feature maps are not built "
                                      "(image_size is one number throughout the reference)")
        att = ops.lwb_attention_x(tsf_x, kv[0], kv[2], kv[1], st["bv"], flow, torch.empty_like(tsf_x), stats=ws,
                                  src_batched=batched)
        ops.instnorm_finalize(ws, B, C, nrec, mean, rstd, eps=1e-5)
        actv = ops.conv2d(att, st["shared"], tsf_x.new_empty(B, h, w, st["shared"].N), act=ops.ACT_RELU)
        return ops.conv2d(actv, st["gb"], torch.empty_like(tsf_x), epi=ops.EPI_SPADE, xn=tsf_x, mean=mean, rstd=rstd)

    @staticmethod
    def _upconv(x, specs, act, q4=False):
        """q4: the output as channel-quad planes (B, N/4, 2H, 2W, 4) - the layer that feeds the fp32 output head (csrc/head.hip)."""
        B, H, W, _ = x.shape
        if q4:
            return ops.conv_transpose2d(x, specs, x.new_empty(B, specs[0].N // 4, 2 * H, 2 * W, 4), act=act, q4=True)
        y = x.new_empty(B, 2 * H, 2 * W, specs[0].N)
        return ops.conv_transpose2d(x, specs, y, act=act)

    @torch.no_grad()
    def _run_tsf_impl(self, tsf8, feats, Tst, bg=None, want_pred=True, want_mask=True, want_img=False):
        """tsf8 (B,S,S,8) NHWC; feats: SourceFeatures; Tst (B,ns,S,S,2) -> (pred, mask, img) NCHW (None if not asked)."""
        pk = self.packed()
        scratch = _Scratch()
        n_down = len(pk.tsf_enc)
        x = tsf8
        adt = self._act_dtype()
        if adt != torch.float32 and self.lwb_kind != "att":
            raise NotImplementedError("bf16 activation storage is built for the attention LWB (AttLWB-SPADE) generators")
        if feats.kv[0][0].dtype != adt:
            raise RuntimeError(f"source features are {feats.kv[0][0].dtype} but the generator runs in {self.conv_precision} mode: "
                               "rebuild them (Imitator.set_source / forward_src) after changing conv_precision")
        enc = []
        site = 0
        for i, spec in enumerate(pk.tsf_enc):
            B, H, W, _ = x.shape
            x = ops.conv2d(x, spec, x.new_empty(B, H // 2, W // 2, spec.N, dtype=adt), act=ops.ACT_RELU)
            x = self._attlwb(pk.enc_sites[i], x, feats.kv[site], Tst, feats.batched, scratch)
            site += 1
            enc.append(x)
        for i, (c0, c1) in enumerate(pk.res):
            h = ops.conv2d(x, c0, torch.empty_like(x), act=ops.ACT_RELU)
            x = ops.conv2d(h, c1, torch.empty_like(x), epi=ops.EPI_RESIDUAL, res=x)
            x = self._attlwb(pk.res_sites[i], x, feats.kv[site], Tst, feats.batched, scratch)
            site += 1
        # fp32 MFMA path ("fp32": direct transposed convolutions; "winograd": lwg_conv_transpose4_winograd_f32, incl. this last layer): the last
        # up-sampling layer writes channel-quad planes, the layout the fp32 head stages whole lines from
        q4 = adt == torch.float32 and self.conv_precision in ("fp32", "winograd", "winograd2x2")
        for i in range(n_down):
            if i == n_down - 1 and pk.head16 is not None and ops.up4_head_eligible(x, pk.upconvs[i], ops.ACT_RELU):
                # bf16 engine (BASELINE configs[3]): the last up-sampling layer, the 5x5 regressors and the compositing as ONE launch - the
                # (B, S, S, 64) tensor between them is never written (csrc/up4_head_bf16.hip)
                return ops.up4_head_compose_bf16(x, pk.upconvs[i], pk.head16, bg, want_pred=want_pred and bg is not None, want_mask=want_mask,
                                                 want_img=want_img)
            x = self._upconv(x, pk.upconvs[i], ops.ACT_RELU, q4=q4 and i == n_down - 1)
            if i != n_down - 1:
                skip = enc[n_down - 2 - i]
                sp = pk.skippers[i]
                x = ops.conv2d(skip, sp, x.new_empty(x.shape[0], x.shape[1], x.shape[2], sp.N), x1=x, act=ops.ACT_RELU)
        head = pk.head
        if x.dtype == torch.bfloat16:
            if pk.head16 is not None:
                head = pk.head16
            else:
                x = x.float()
        return ops.head_compose(x, head, bg, want_pred=want_pred and bg is not None, want_mask=want_mask, want_img=want_img, q4=q4)

    def _act_dtype(self):
        """Storage type of the engine's activation tensors: bf16 in the "bf16" precision mode (BASELINE configs[3]), else fp32."""
        return torch.bfloat16 if self.conv_precision == "bf16" else torch.float32

    @torch.no_grad()
    def _run_bg_impl(self, bg4):
        """bg4 (n,S,S,4) NHWC -> (n,3,S,S) NCHW."""
        return run_bg_layers(self.packed().bg, bg4)

    @torch.no_grad()
    def _run_src_decode_impl(self, x):
        """SIDNet decoder + regressors on the last res-block feature (forward_src(only_enc=False))."""
        pk = self.packed()
        for specs in pk.src_dec:
            x = self._upconv(x, specs, ops.ACT_RELU)
        _, mask, img = ops.head_compose(x.float(), pk.src_head, None, want_pred=False, want_mask=True, want_img=True)
        return img, mask

    def encode_sources(self, *a, **k):
        with ops.conv_precision(self.conv_precision):
            return self._encode_sources_impl(*a, **k)

    def run_tsf(self, *a, **k):
        with ops.conv_precision(self.conv_precision):
            return self._run_tsf_impl(*a, **k)

    def run_bg(self, *a, **k):
        # once per source, InstanceNorm after every conv: fp32 tensors in every mode ("split" still speeds up its products)
        with ops.conv_precision("fp32" if self.conv_precision == "bf16" else self.conv_precision):
            return self._run_bg_impl(*a, **k)

    def run_src_decode(self, *a, **k):
        with ops.conv_precision(self.conv_precision):
            return self._run_src_decode_impl(*a, **k)

    # ------------------------------------------------------------------ reference API (NCHW)
    @torch.no_grad()
    def forward_bg(self, bg_inputs):
        """(bs, ns, 4, h, w) -> (bs, ns, 3, h, w)  (attlwb_spade_resunet.py:615-631)."""
        if not self.has_bg:
            raise AttributeError("this generator has no background network")
        self._check(bg_inputs)
        bs, ns, c, h, w = bg_inputs.shape
        x = ops.nchw_to_nhwc(bg_inputs.reshape(bs * ns, c, h, w).contiguous().float(), c_pad=4)
        return self.run_bg(x).view(bs, ns, 3, h, w)

    @torch.no_grad()
    def forward_src(self, src_inputs, only_enc=True):
        """(bs, ns, 6, h, w) -> (enc_outs, res_outs[, img, mask]) as NCHW lists (:450-478).  The returned lists carry
        the NHWC/KV cache so a following ``forward_tsf`` does not recompute it."""
        self._check(src_inputs)
        self._check_square(src_inputs)
        bs, ns, c, h, w = src_inputs.shape
        src8 = ops.nchw_to_nhwc(src_inputs.reshape(bs * ns, c, h, w).contiguous().float(), c_pad=8)
        feats = self.encode_sources(src8, batched=bs > 1, ns=ns)
        enc = _FeatList(ops.nhwc_to_nchw(t.float()) for t in feats.enc)
        res = _FeatList(ops.nhwc_to_nchw(t.float()) for t in feats.res)
        enc.lwg_cache = res.lwg_cache = feats
        if only_enc:
            return enc, res
        img, mask = self.run_src_decode(feats.res[-1])
        return enc, res, img.view(bs, ns, 3, h, w), mask.view(bs, ns, 1, h, w)

    def _features_from_api(self, src_enc_outs, src_res_outs, bs):
        cache = getattr(src_enc_outs, "lwg_cache", None)
        if cache is not None and getattr(src_res_outs, "lwg_cache", None) is cache:
            return cache
        pk = self.packed()
        adt = self._act_dtype()
        enc = [ops.nchw_to_nhwc(t.contiguous().float()).to(adt) for t in src_enc_outs]
        res = [ops.nchw_to_nhwc(t.contiguous().float()).to(adt) for t in src_res_outs]
        n = enc[0].shape[0]
        return SourceFeatures(enc, res, self._project_sources(pk, enc, res), n // bs, batched=bs > 1)

    @torch.no_grad()
    def forward_tsf(self, tsf_inputs, src_enc_outs, src_res_outs, Tst, temp_enc_outs=None, temp_res_outs=None, Ttt=None):
        """(bs,6,h,w), src feats, Tst (bs,ns,h,w,2) -> (tsf_img (bs,3,h,w), tsf_mask (bs,1,h,w))  (:480-535).
        temp_enc_outs / temp_res_outs ((bs*nt,c,h,w) lists) + Ttt (bs,nt,h,w,2): the temporal attention inputs (:232-243) -
        their K / V join the sources' along the attention axis."""
        self._check(tsf_inputs, Tst)
        self._check_square(tsf_inputs)
        bs = tsf_inputs.shape[0]
        feats = self._features_from_api(src_enc_outs, src_res_outs, bs)
        T = Tst.contiguous().float()
        if temp_enc_outs is not None and Ttt is not None and self.lwb_kind == "att":      # the other blocks ignore temp_x / Ttt
            if bs != 1:
                raise NotImplementedError("temporal attention runs one clip per process (bs = 1), as Imitator.inference does")
            tfe = self._features_from_api(temp_enc_outs, temp_res_outs, bs)
            feats = SourceFeatures(feats.enc, feats.res, [tuple(torch.cat([a, ta], dim=0) for a, ta in zip(site, tsite))
                                                          for site, tsite in zip(feats.kv, tfe.kv)],
                                   feats.ns + Ttt.shape[1], batched=False)
            T = torch.cat([T, Ttt.contiguous().float()], dim=1).contiguous()
        tsf8 = ops.nchw_to_nhwc(tsf_inputs.contiguous().float(), c_pad=8)
        _, mask, img = self.run_tsf(tsf8, feats, T, bg=None, want_pred=False, want_mask=True, want_img=True)
        return img, mask

    @torch.no_grad()
    def _forward_streams(self, src_inputs, tsf_inputs, Tst, only_tsf):
        bs, nt = Tst.shape[0], Tst.shape[1]
        if only_tsf:
            enc, res = self.forward_src(src_inputs, only_enc=True)
            src_imgs = src_masks = None
        else:
            enc, res, src_imgs, src_masks = self.forward_src(src_inputs, only_enc=False)
        imgs, masks = [], []
        for t in range(nt):
            if t != 0 and self.temporal:
                raise NotImplementedError("multi-step temporal training forward (Ttt inside forward()) is not built; "
                                          "temporal INFERENCE goes through Imitator(temporal=True) / forward_tsf(temp_*)")
            img, mask = self.forward_tsf(tsf_inputs[:, t], enc, res, Tst[:, t].contiguous())
            imgs.append(img)
            masks.append(mask)
        return src_imgs, src_masks, torch.stack(imgs, dim=1), torch.stack(masks, dim=1)

    @torch.no_grad()
    def forward(self, bg_inputs, src_inputs, tsf_inputs, Tst, Ttt=None, only_tsf=True):
        """:633-699 with temporal=False: Tst (bs, nt, ns, h, w, 2), tsf_inputs (bs, nt, 6, h, w)."""
        bg_img = self.forward_bg(bg_inputs)
        src_imgs, src_masks, imgs, masks = self._forward_streams(src_inputs, tsf_inputs, Tst, only_tsf)
        if only_tsf:
            return bg_img, imgs, masks
        return bg_img, src_imgs, src_masks, imgs, masks


class AttentionLWBFrontGenerator(AttentionLWBGenerator):
    """attlwb_spade_resunet.py:702-834: same network without the background branch."""
    has_bg = False

    @torch.no_grad()
    def forward(self, src_inputs, tsf_inputs, Tst, Ttt=None, only_tsf=True):
        """:778-834: the same two streams, no background branch."""
        src_imgs, src_masks, imgs, masks = self._forward_streams(src_inputs, tsf_inputs, Tst, only_tsf)
        if only_tsf:
            return imgs, masks
        return src_imgs, src_masks, imgs, masks


class AddLWBGenerator(AttentionLWBGenerator):
    """generators/lwb_resunet.py:509-518 (BaseLWBGenerator :315-506 with AddLWB :77-111): same three streams, the transfer
    stream fuses by  tsf_x + sum_s warp_s(src_x)  - no block parameters."""
    lwb_kind = "add"


class AvgLWBGenerator(AttentionLWBGenerator):
    """generators/lwb_resunet.py:521-531 with AvgLWB :114-152: mean over [tsf_x, warped sources]."""
    lwb_kind = "avg"


class SoftGateAddLWBGenerator(AttentionLWBGenerator):
    """generators/lwb_softgate_resunet.py:522-525 (SoftGateLWBGenerator :317-519, SoftGateLWB :77-123):
    tsf_x + sigmoid(conv3(relu(conv3(tsf_x)))) * sum_s warp_s(src_x)."""
    lwb_kind = "sg_add"


class SoftGateAvgLWBGenerator(AttentionLWBGenerator):
    """generators/lwb_softgate_resunet.py:528-531: the gate times the mean of the warped sources."""
    lwb_kind = "sg_avg"


class _FeatList(list):
    """A list that can carry the engine's cache as an attribute (plain lists cannot)."""


class _Scratch:
    def __init__(self):
        self.buf = None
        self.flows = {}

    def flow(self, Tst, h, w):
        """The (B,ns,S,S,2) flows resized to (h,w) - LWB.resize_trans - once per frame batch and resolution (seven of the nine attention
        sites share one): the block kernels then read one coalesced value per pixel and source instead of resizing per pixel."""
        if Tst.shape[2] == h and Tst.shape[3] == w:
            return Tst
        if h != w:                                        # the Add / Avg / SoftGate kernels take a square field: their in-kernel resize
            return Tst
        key = (h, w, Tst.data_ptr())
        if key not in self.flows:
            self.flows[key] = ops.flow_resize(Tst, h, w)
        return self.flows[key]

    def get(self, n, device):
        if self.buf is None or self.buf.numel() < n or self.buf.device != device:
            self.buf = torch.empty(max(n, 1 << 16), device=device, dtype=torch.float32)
        return self.buf


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        if key in cfg:
            return cfg[key]
    elif hasattr(cfg, key):
        return getattr(cfg, key)
    if default is not None:
        return default
    raise KeyError(key)


class _ConcatPacked:
    """Packed panels of ``bg_net`` + ``tsf_net`` (ResAutoEncoder) for the input-concatenation baselines."""

    def __init__(self, gen):
        sd = {k: v for k, v in gen.named_parameters()}
        dev = next(gen.parameters()).device
        self.version = _param_version(gen)
        P = packing
        nf, n_down = gen.num_filters, len(gen.num_filters)

        def conv(name, stride=1, pad=None, cin_pad=None):
            return P.spec_to(P.pack_conv(sd[name + ".weight"], sd.get(name + ".bias"), stride, pad, cin_pad, None), dev)
        self.enc = [conv(f"tsf_net.encoders.layers.{i}.0", stride=2, cin_pad=gen.cin_pad if i == 0 else None) for i in range(n_down)]
        self.res = [(conv(f"tsf_net.res_blocks.{i}.main.0"), conv(f"tsf_net.res_blocks.{i}.main.2")) for i in range(gen.n_res_block)]
        self.dec = [[P.spec_to(sp, dev) for sp in P.pack_conv_transpose(sd[f"tsf_net.decoders.layers.{i}.0.weight"],
                                                                       sd[f"tsf_net.decoders.layers.{i}.0.bias"], None)] for i in range(n_down)]
        self.head = P.pack_head(sd["tsf_net.img_reg.0.weight"], sd["tsf_net.att_reg.0.weight"]).to(dev)
        self.bg = pack_bg_layers(sd, gen.bg_filters, gen.bg_n_res_block, dev)


class InputConcatGenerator(nn.Module):
    """generators/input_concat_resunet.py:182-307 (factory name ``InputConcat``, networks/__init__.py:38-40): NO warp - the sources
    (padded by repetition / truncated to ``cfg.TSFNet.num_source``, :215-249) are concatenated with the target's condition channels and go
    through ONE ``ResAutoEncoder`` (:126-179; ``cond_nc`` = 6 num_source + 3 = 27), plus the background network.  A baseline of the
    paper behind the same four methods as every generator; the convolutions run on the MFMA kernels (the 27-channel input is
    zero-extended to 32), the regressors + tanh / sigmoid on the head kernel."""
    has_bg = True

    def __init__(self, cfg, temporal=False):
        super().__init__()
        self._name = _get(cfg, "name", "InputConcat")
        tsf, bgc = _get(cfg, "TSFNet"), _get(cfg, "BGNet")
        self.num_filters = [int(c) for c in _get(tsf, "num_filters")]
        self.n_res_block = int(_get(tsf, "n_res_block"))
        self.cond_nc = int(_get(tsf, "cond_nc"))
        self.num_source = int(_get(tsf, "num_source", 4)) if self._uses_sources() else None
        self.bg_filters = [int(c) for c in _get(bgc, "num_filters")]
        self.bg_n_res_block = int(_get(bgc, "n_res_block"))
        self.temporal = temporal
        if self.num_filters[0] != 64 or any(c not in (64, 128, 256) for c in self.num_filters):
            raise ValueError(f"num_filters must start with 64 and stay in {{64,128,256}} for the HIP kernels, got {self.num_filters}")
        # the conv kernels take Cin in {4, 8, 16} or a multiple of 32: the network input is zero-extended
        self.cin_pad = 8 if self.cond_nc <= 8 else (16 if self.cond_nc <= 16 else (self.cond_nc + 31) // 32 * 32)
        from .params import concat_generator_param_shapes
        tree = ParamTree(concat_generator_param_shapes(self.num_filters, self.n_res_block, self.bg_filters, self.cond_nc,
                                                        int(_get(bgc, "cond_nc", 4)), self.bg_n_res_block))
        for name, child in tree.named_children():
            self.add_module(name, child)
        self._packed = None

    def _uses_sources(self):
        return True

    def packed(self):
        if self._packed is None or self._packed.version != _param_version(self):
            self._packed = _ConcatPacked(self)
        return self._packed

    @staticmethod
    def _check(*tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise RuntimeError("ipercore_amd generator runs on the MI355X only: got a CPU tensor (no fallback)")

    @torch.no_grad()
    def forward_bg(self, bg_inputs):
        """(bs, ns, 4, h, w) -> (bs, ns, 3, h, w)."""
        self._check(bg_inputs)
        bs, ns, c, h, w = bg_inputs.shape
        x = ops.nchw_to_nhwc(bg_inputs.reshape(bs * ns, c, h, w).contiguous().float(), c_pad=4)
        return run_bg_layers(self.packed().bg, x).view(bs, ns, 3, h, w)

    def forward_src(self, src_inputs, only_enc=True):
        """:215-249: no network - the sources, repeated / cut to ``num_source``, as one (bs, ns * 6, h, w) tensor (returned twice)."""
        bs, ns, _, h, w = src_inputs.shape
        need = self.num_source
        if ns > need:
            src_inputs = src_inputs[:, 0:need]
        elif ns < need:
            src_inputs = torch.cat([src_inputs, torch.stack([src_inputs[:, s % ns] for s in range(need - ns)], dim=1)], dim=1)
        enc = src_inputs.reshape(bs, -1, h, w)
        return (enc, enc) if only_enc else (enc, enc, None, None)

    @torch.no_grad()
    def _autoencode(self, inputs):
        """(n, cond_nc, h, w) NCHW -> (img (n,3,h,w), mask (n,1,h,w)): ResAutoEncoder.forward (:160-169)."""
        self._check(inputs)
        pk = self.packed()
        x = ops.nchw_to_nhwc(inputs.contiguous().float(), c_pad=self.cin_pad)
        for spec in pk.enc:
            n, H, W, _ = x.shape
            x = ops.conv2d(x, spec, x.new_empty(n, H // 2, W // 2, spec.N), act=ops.ACT_RELU)
        for c0, c1 in pk.res:
            t = ops.conv2d(x, c0, torch.empty_like(x), act=ops.ACT_RELU)
            x = ops.conv2d(t, c1, torch.empty_like(x), epi=ops.EPI_RESIDUAL, res=x)
        for specs in pk.dec:
            n, H, W, _ = x.shape
            x = ops.conv_transpose2d(x, specs, x.new_empty(n, 2 * H, 2 * W, specs[0].N), act=ops.ACT_RELU)
        _, mask, img = ops.head_compose(x, pk.head, None, want_pred=False, want_mask=True, want_img=True)
        return img, mask

    @torch.no_grad()
    def forward_tsf(self, tsf_inputs, src_enc_outs, src_res_outs=None, Tst=None, temp_enc_outs=None, temp_res_outs=None, Ttt=None):
        """:251-277: cat[sources, the target's last three (condition) channels] -> ResAutoEncoder."""
        return self._autoencode(torch.cat([src_enc_outs, tsf_inputs[:, -3:]], dim=1))

    @torch.no_grad()
    def forward(self, bg_inputs, src_inputs, tsf_inputs, Tst=None, Ttt=None, only_tsf=True):
        """:279-307 -> (bg_img (bs,ns,3,h,w), tsf_imgs (bs,nt,3,h,w), tsf_masks (bs,nt,1,h,w))."""
        bg_img = self.forward_bg(bg_inputs)
        enc, _ = self.forward_src(src_inputs, only_enc=True)
        outs = [self.forward_tsf(tsf_inputs[:, t], enc) for t in range(tsf_inputs.shape[1])]
        return bg_img, torch.stack([o[0] for o in outs], dim=1), torch.stack([o[1] for o in outs], dim=1)


class TextureWarpingGenerator(InputConcatGenerator):
    """generators/texture_warping_resunet.py:8-112 (factory name ``TextureWarping``): the transfer network sees ONLY ``tsf_inputs`` -
    the UV-warped synthetic image already inside them (flowcomposition.py:206-248) plus the condition; ``cond_nc`` = 6."""

    def __init__(self, cfg, temporal=False):
        super().__init__(cfg, temporal)
        self._name = _get(cfg, "name", "TextureWarping")

    def _uses_sources(self):
        return False

    def forward_src(self, src_inputs, only_enc=True):
        """:40-60: the sources as one (bs, ns * 6, h, w) view; nothing consumes it."""
        bs, ns, _, h, w = src_inputs.shape
        enc = src_inputs.reshape(bs, -1, h, w)
        return (enc, enc) if only_enc else (enc, enc, None, None)

    @torch.no_grad()
    def forward_tsf(self, tsf_inputs, src_enc_outs=None, src_res_outs=None, Tst=None, temp_enc_outs=None, temp_res_outs=None, Ttt=None):
        """:62-84."""
        return self._autoencode(tsf_inputs)

    @torch.no_grad()
    def forward(self, bg_inputs, src_inputs, tsf_inputs, Tst=None, Ttt=None, only_tsf=True):
        """:86-112: every target frame in one batch."""
        bg_img = self.forward_bg(bg_inputs)
        bs, nt, c, h, w = tsf_inputs.shape
        img, mask = self._autoencode(tsf_inputs.reshape(bs * nt, c, h, w))
        return bg_img, img.view(bs, nt, 3, h, w), mask.view(bs, nt, 1, h, w)
