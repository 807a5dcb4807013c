"""Training (autograd) path of the AttLWB-SPADE generator for the personalization step.

Reference: ``LWGTrainer.forward / optimize_G`` (tools/trainers/lwg_trainer.py:699-789) call
``AttentionLWBGenerator.forward(bg, src, tsf, Tst, only_tsf=False)`` (attlwb_spade_resunet.py:633-699) and
``loss.backward()`` (lwg_trainer.py:345).

What runs where, this round:
* every convolution / transposed convolution - forward, data gradient and weight gradient (> 99 % of the 1857 GFLOP of
  a G step) - runs on the hand-written MFMA kernels: ``ConvFn`` is a ``torch.autograd.Function`` whose forward is
  ``lwg_conv2d_nhwc_f32``, whose data gradient is the same kernel on dY with a transposed panel
  (``packing.pack_dgrad_*``) and whose weight gradient is ``lwg_conv2d_wgrad_nhwc_f32``;
* InstanceNorm (+ ReLU / LeakyReLU), SPADE's modulation, the ReLU masks and Adam are HIP kernels too (``NormAct`` / ``SpadeNormFn``,
  csrc/train_ops.hip), and so is the attention block: ``AttnFn`` / ``AttnKVFn`` run the hoisted-K/V gather kernel forward
  (csrc/lwb_attn.hip ``lwg_lwb_attention[_kv]_f32``) and ``lwg_lwb_attention[_kv]_bwd_f32`` backward (gathers recomputed, fp32 atomics for
  the bilinear scatter into dK / dV); the 5x5 / 7x7 regressors run the thin VALU kernels (``HeadFn`` / ``ThinConvFn``);
* what remains PyTorch-ROCm autograd: the tanh / sigmoid derivatives of the regressors' outputs, the mask compositing, the scalar
  losses (L1, LSGAN, BCE, TV) and the fan-in additions of tensors with two consumers - HBM-bound elementwise work (~90 small aten
  launches per step inside the captured graph).
(The per-frame INFERENCE engine uses a different attention form - the query projection folded into the cached K, csrc/lwb_attn_x.hip;
here fq is trainable, so q = fq(x) stays an explicit convolution.)
There is no CPU fallback: ``ConvFn`` raises on CPU tensors.

The module reuses the parameter tree of ``generator.AttentionLWBGenerator`` (same ``state_dict`` keys), so a
personalized checkpoint saved from here loads into the inference engine unchanged.
"""
import contextlib
import math

import torch
import torch.nn.functional as F

from .. import ops
from . import packing

_RELU, _NONE = 1, 0


class ConvCfg(object):
    """Static description of one layer: kind ('conv' | 'convT'), stride, pad, fused act (0 none, 1 relu),
    zero-extension of input channels (cin_pad) / output channels (n_pad) to what the kernels need."""

    def __init__(self, kind="conv", stride=1, pad=None, act=_NONE, cin_pad=None, n_pad=None, need_dx=True, mask_dx=False, premasked=False):
        self.kind, self.stride, self.pad, self.act, self.cin_pad, self.n_pad, self.need_dx = kind, stride, pad, act, cin_pad, n_pad, need_dx
        # A ReLU layer whose output feeds exactly ONE convolution: that convolution (mask_dx=True) applies the ReLU backward where it
        # writes its data gradient (dX = x0 > 0 ? ... : 0 in the launch's epilogue), and the ReLU layer (premasked=True) skips its own
        # act_bwd pass.  The two flags are set together by the wiring (TrainableGenerator); check_generator_training_grads pins them.
        self.mask_dx, self.premasked = mask_dx and FUSED_RELU_MASK, premasked and FUSED_RELU_MASK


FUSED_CONVT_WGRAD = True    # lab switch: False = a transposed convolution's weight gradient as four parity launches
FUSED_S2_DGRAD = True       # lab switch: False = the data gradient of a 4x4 stride-2 convolution as four parity launches
FUSED_RELU_MASK = True      # lab switch: False = every ReLU convolution runs its own act_bwd pass
FUSED_KV_PAIR = True        # lab switch: False = the fk / fv projections of an attention site as two 1x1 convolutions
FUSED_SPADE_PAIR = True     # lab switch: False = SPADE's mlp_gamma / mlp_beta as two convolutions (two launches per pass + gradient add)
FUSED_BIAS_GRAD = True      # lab switch: False = bias gradients by the separate column-sum kernel
FUSED_CONVT_FWD = True      # lab switch: False = four parity launches (split-K where the library plans it)


def _stacked(a, b):
    """cat([a, b], dim 0) of two parameters WITHOUT a copy when b starts where a ends in the same storage (``trainers.FlatAdam``
    lays paired parameters out that way) - else a concatenation (modules outside a trainer, the host-logic tests)."""
    a, b = a.detach(), b.detach()
    if (a.is_contiguous() and b.is_contiguous() and a.shape[1:] == b.shape[1:] and a.dtype == b.dtype
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel()):
        shape = (a.shape[0] + b.shape[0],) + tuple(a.shape[1:])
        return torch.as_strided(a, shape, a.stride())
    return torch.cat([a, b], dim=0)


class ConvFn(torch.autograd.Function):
    """y = act(conv(cat[x0, x1], weight) + bias) on NHWC tensors, all three passes on the MFMA kernels.
    weight2 / bias2: a second nn.Conv2d of the same geometry on the same input - the two run as ONE launch of N + N2 output columns
    (y = [conv(x, weight) | conv(x, weight2)] along the channels): SPADE's mlp_gamma | mlp_beta (attlwb_spade_resunet.py:66-67)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, cfg, weight2=None, bias2=None):
        if not x0.is_cuda:
            raise RuntimeError("ipercore_amd training convs run on the MI355X only (no CPU fallback)")
        ctx.n_first = None
        if weight2 is not None:
            assert cfg.kind == "conv" and (bias is None) == (bias2 is None)
            ctx.n_first = weight.shape[0]
            weight = _stacked(weight, weight2)
            bias = None if bias is None else _stacked(bias, bias2)
        x0 = x0.contiguous()
        x1 = None if x1 is None else x1.contiguous()
        B, H, W, _ = x0.shape
        dev = x0.device
        if cfg.kind == "conv":
            N = weight.shape[0]
            spec = packing.spec_to(packing.pack_conv(weight, bias, cfg.stride, cfg.pad, cfg.cin_pad, cfg.n_pad), dev)
            kh = weight.shape[2]
            pad = kh // 2 if cfg.pad is None else cfg.pad
            OH, OW = (H + 2 * pad - kh) // cfg.stride + 1, (W + 2 * pad - kh) // cfg.stride + 1
            y = torch.empty(B, OH, OW, spec.N, device=dev, dtype=torch.float32)
            ops.conv2d(x0, spec, y, x1=x1, act=cfg.act, splitk=True)
            specs = [spec]
        else:
            N = weight.shape[1]
            specs = [packing.spec_to(s, dev) for s in packing.pack_conv_transpose(weight, bias, cfg.n_pad)]
            y = torch.empty(B, 2 * H, 2 * W, specs[0].N, device=dev, dtype=torch.float32)
            if FUSED_CONVT_FWD:
                ops.conv_transpose2d(x0, specs, y, act=cfg.act, splitk=True)     # small launches: the four parities as one grid (lwg_conv_transpose4_nhwc_f32)
            else:
                for s in specs:
                    ops.conv2d(x0, s, y, act=cfg.act, splitk=True)
        ctx.cfg, ctx.specs, ctx.N, ctx.has_bias = cfg, specs, N, bias is not None
        ctx.has_x1 = x1 is not None
        # a regressor (3 / 4 / 1 output channels zero-extended to 64): its backward runs on the thin forms below
        ctx.thin = (cfg.kind == "conv" and cfg.stride == 1 and N <= 16 and x1 is None and cfg.act == _NONE and specs[0].Cin % 64 == 0
                    and specs[0].Cin == x0.shape[3])
        ctx.save_for_backward(x0, x1, weight, y if cfg.act == _RELU else None)
        return y if y.shape[3] == N else y[..., :N]

    @staticmethod
    def backward(ctx, dy):
        x0, x1, weight, y = ctx.saved_tensors
        cfg, specs, N = ctx.cfg, ctx.specs, ctx.N
        Np = specs[0].N
        dy = dy.contiguous()
        if ctx.thin:
            return ConvFn._backward_thin(ctx, x0, weight, dy)
        if Np != N:                                                   # zero-extended output channels carry no gradient
            full = dy.new_zeros(dy.shape[0], dy.shape[1], dy.shape[2], Np)
            full[..., :N] = dy
            dy = full
        if cfg.act == _RELU and not cfg.premasked:
            dy = ops.act_bwd(dy, y, ops.ACT_RELU)
        dev = dy.device
        C0 = x0.shape[3]
        Cin_packed = specs[0].Cin
        want_dw = ctx.needs_input_grad[2]                            # frozen weights (the VGG19 of the perceptual loss): data gradient only
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        # (the weight gradient on a second stream next to the data gradient was tried twice - eager launches: 53.6 vs 48.6 ms in round 1,
        # captured step: 32.4 vs 32.2 ms in round 2 - and removed: both launches already fill the chip)
        # the bias gradient of a convolution rides along with its weight gradient (the column sums of dy in the launch that stages dy
        # anyway: FUSED_BIAS_GRAD); a transposed convolution's four parity launches each see a quarter of dy -> the column-sum kernel
        fused_db = FUSED_BIAS_GRAD and want_db and want_dw and cfg.kind == "conv"
        db = None
        if want_db:
            db = torch.empty(N, device=dev, dtype=torch.float32) if fused_db else ops.colsum(dy)[:N]
        dw = None
        if want_dw:
            if cfg.kind == "conv":
                dw = packing.wgrad_conv(x0, specs[0], dy, x1, weight.shape[2], weight.shape[3], weight.shape[1], N, db=db if fused_db else None)
            else:
                adj = packing.spec_to(packing.pack_dgrad_conv_transpose(weight, n_pad=Np)[0], dev) if FUSED_CONVT_WGRAD else None
                dw = packing.wgrad_conv_transpose(x0, specs, dy, weight.shape[0], N, adj_spec=adj)
        if cfg.kind == "conv":
            Nw, Cin, kh, kw = weight.shape
            dx = None
            if cfg.need_dx and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
                pad = kh // 2 if cfg.pad is None else cfg.pad            # the dgrad panel sees the padded channel counts
                Cd = (Cin_packed + 63) // 64 * 64                        # GEMM columns of the dgrad launch: the kernel's granularity is 64
                dspecs = [packing.spec_to(s, dev) for s in packing.pack_dgrad_conv(weight, cfg.stride, pad, n_pad=Np, cin_pad=Cd)]
                B, H, W, _ = x0.shape
                dx = torch.empty(B, H, W, Cd, device=dev, dtype=torch.float32)
                if cfg.stride == 1 and cfg.mask_dx:
                    assert Cd == C0 and x1 is None, "mask_dx: the data gradient must have exactly the forward input's channels"
                    ops.conv2d(dy, dspecs[0], dx, epi=ops.EPI_RESIDUAL, act=ops.ACT_RELU_MASK, res=x0, splitk=True)
                elif cfg.stride == 1:
                    ops.conv2d(dy, dspecs[0], dx, splitk=True)
                else:
                    assert not cfg.mask_dx
                    if FUSED_S2_DGRAD and kh == 4 and kw == 4 and pad == 1 and H % 2 == 0 and W % 2 == 0 and dy.shape[1] * 2 == H and dy.shape[2] * 2 == W:
                        # the data gradient of Conv2d(4, 2, 1) IS ConvTranspose2d(4, 2, 1) with the same weight: its four input-parity
                        # launches have four taps each, parity p's shifted by p - the one-grid form of the decoder's up-sampling layers
                        # (lwg_conv_transpose4_nhwc_f32) runs them as ONE launch when a parity is small (the discriminator's layers)
                        ops.conv_transpose2d(dy, dspecs, dx, splitk=True, out_hw=lambda s_: ((H - s_.ooy + 1) // 2, (W - s_.oox + 1) // 2))
                    else:
                        for s in dspecs:         # one launch per input parity (py, px): rows py, py + 2, .. < H - ceil for odd sizes
                            ops.conv2d(dy, s, dx, out_hw=((H - s.ooy + 1) // 2, (W - s.oox + 1) // 2), splitk=True)
        else:
            Cin, Nw = weight.shape[0], weight.shape[1]
            dx = None
            if cfg.need_dx and ctx.needs_input_grad[0]:
                dspec = packing.spec_to(packing.pack_dgrad_conv_transpose(weight, n_pad=Np)[0], dev)
                B, H, W, _ = x0.shape
                dx = torch.empty(B, H, W, Cin, device=dev, dtype=torch.float32)
                if cfg.mask_dx:
                    ops.conv2d(dy, dspec, dx, epi=ops.EPI_RESIDUAL, act=ops.ACT_RELU_MASK, res=x0, splitk=True)
                else:
                    ops.conv2d(dy, dspec, dx, splitk=True)
        dx0 = dx1 = None
        if dx is not None:
            dx0 = dx[..., :C0] if (ctx.has_x1 or dx.shape[3] != C0) else dx
            if ctx.has_x1:
                dx1 = dx[..., C0:C0 + x1.shape[3]]
        if ctx.n_first is not None:          # the pair's gradients: views of the stacked ones
            k = ctx.n_first
            return (dx0, dx1, None if dw is None else dw[:k], None if db is None else db[:k], None,
                    None if dw is None else dw[k:], None if db is None else db[k:])
        return dx0, dx1, dw, db, None, None, None

    @staticmethod
    def _backward_thin(ctx, x0, weight, dy):
        """A regressor's backward on the thin forms (``thin_backward``): dY zero-extended to 4 / 8 / 16 channels, not 64."""
        cfg, N = ctx.cfg, ctx.N
        Ns = 4 if N <= 4 else (8 if N <= 8 else 16)
        if Ns != N:
            dy = F.pad(dy, (0, Ns - N))
        db = ops.colsum(dy)[:N] if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        kh = weight.shape[2]
        dx, dw = thin_backward(x0, weight, dy, kh // 2 if cfg.pad is None else cfg.pad, cfg.need_dx and ctx.needs_input_grad[0],
                               ctx.needs_input_grad[2])
        return dx, None, dw, db, None, None, None


def thin_backward(x0, weight, dy, pad, want_dx, want_dw):
    """Backward of a stride-1 conv with N <= 16 outputs WITHOUT the zero-extension to 64 columns a forward MFMA launch needs
    (the 5x5 / 7x7 image and mask regressors at full resolution: 16x fewer MFMA flops than the padded forms).
    dy (B,H,W,Ns), Ns = 4 / 8 / 16 >= N, becomes the INPUT of a small-Cin convolution with negated taps:
      dX = that conv of dY (the forward kernel's K-slot path, K = taps * Ns),
      dW = that conv's weight gradient with x in the role of the output gradient:
           dW'[(tap, n)][c] = sum_q dY[q - tap][n] x[q][c] = sum_p x[p + tap][c] dY[p][n]  - the forward conv's dW[n][c][tap]."""
    N, Cin, kh, kw = weight.shape
    dx = dw = None
    if want_dx or want_dw:
        dspec = packing.spec_to(packing.pack_dgrad_conv(weight, 1, pad, n_pad=dy.shape[3], cin_pad=x0.shape[3])[0], dy.device)
        if want_dx:
            dx = torch.empty(x0.shape, device=dy.device, dtype=torch.float32)
            ops.conv2d(dy, dspec, dx, splitk=True)
        if want_dw:
            dw = packing.wgrad_thin(dy, dspec, x0, kh, kw, dy.shape[3], N, Cin)
    return dx, dw


def conv(x0, weight, bias=None, x1=None, **kw):
    return ConvFn.apply(x0, x1, weight, bias, ConvCfg(**kw))


def conv_pair(x0, weight, bias, weight2, bias2, **kw):
    """[conv(x0, weight, bias) | conv(x0, weight2, bias2)] along the channels as one launch per pass."""
    return ConvFn.apply(x0, None, weight, bias, ConvCfg(**kw), weight2, bias2)


class ThinConvFn(torch.autograd.Function):
    """y = conv_ks(x, weight) for a bias-free stride-1 regressor with N <= 4 outputs and ks in {5, 7} (the 7x7 image head of the
    background network, bg_inpaintor.py:53) on the vector-ALU kernel (csrc/head.hip lwg_thin_conv_f32) instead of an MFMA launch
    zero-extended to 64 columns (21x the useful flops); backward = ``thin_backward``.  x (B,S,S,C) NHWC -> (B,S,S,N)."""

    @staticmethod
    def forward(ctx, x, weight):
        if not x.is_cuda:
            raise RuntimeError("ipercore_amd training ops run on the MI355X only (no CPU fallback)")
        x = x.contiguous()
        N, _, ks, _ = weight.shape
        y = ops.thin_conv(x, packing.pack_thin(weight).to(x.device), ks)
        ctx.save_for_backward(x, weight)
        return y[..., :N]

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, _, ks, _ = weight.shape
        dy = F.pad(dy, (0, 4 - N)).contiguous() if N != 4 else dy.contiguous()
        dx, dw = thin_backward(x, weight.detach(), dy, ks // 2, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dw


class HeadFn(torch.autograd.Function):
    """img = tanh(conv5x5(x, w_img)), mask = sigmoid(conv5x5(x, w_att)) (attlwb_spade_resunet.py:375-384, 604-613; no bias) on
    the inference path's fused regressor kernel (csrc/head.hip: 4 output channels per pixel on the vector ALUs instead of an
    MFMA launch zero-extended to 64 columns); backward = the activations' derivatives + ``thin_backward``.
    x (B,S,S,C) NHWC -> img (B,3,S,S), mask (B,1,S,S) NCHW."""

    @staticmethod
    def forward(ctx, x, w_img, w_att):
        if not x.is_cuda:
            raise RuntimeError("ipercore_amd training ops run on the MI355X only (no CPU fallback)")
        x = x.contiguous()
        _, mask, img = ops.head_compose(x, packing.pack_head(w_img, w_att).to(x.device), None, want_pred=False, want_mask=True, want_img=True)
        ctx.save_for_backward(x, w_img, w_att, img, mask)
        return img, mask

    @staticmethod
    def backward(ctx, dimg, dmask):
        x, w_img, w_att, img, mask = ctx.saved_tensors
        dpre = torch.cat([dimg * (1.0 - img * img), dmask * (mask * (1.0 - mask))], dim=1)            # (B,4,S,S)
        dy = ops.nchw_to_nhwc(dpre, c_pad=4)
        dx, dw = thin_backward(x, torch.cat([w_img, w_att], dim=0).detach(), dy, 2, ctx.needs_input_grad[0],
                               ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, None if dw is None else dw[:3], None if dw is None else dw[3:4]


# ---------------------------------------------------------------------------------------------- NHWC glue (autograd)
class NormAct(torch.autograd.Function):
    """y = act(InstanceNorm2d(x) * (1 + gamma) + beta) on (B,H,W,C), forward and backward on csrc/train_ops.hip
    (statistics by the inference path's lwg_instnorm_stats).  gamma = beta = None: plain InstanceNorm + activation."""

    @staticmethod
    def forward(ctx, x, gamma, beta, act):
        y, mean, rstd = ops.norm_fwd(x, gamma, beta, act)
        ctx.act = act
        ctx.save_for_backward(x.contiguous(), mean, rstd, None if gamma is None else gamma.contiguous(), y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, y = ctx.saved_tensors
        dx, dg, db = ops.norm_bwd(dy, y, x, mean, rstd, gamma, ctx.act)
        return dx, dg, db, None


class SpadeNormFn(torch.autograd.Function):
    """y = act(InstanceNorm2d(x) * (1 + gamma) + beta) with gb = gamma | beta (B,H,W,2C) as ONE tensor: the kernels read the halves in
    place and the backward writes d(gamma | beta) in place - no slicing / concatenation kernels around the fused SPADE convolution."""

    @staticmethod
    def forward(ctx, x, gb, act):
        gb = gb.contiguous()
        y, mean, rstd = ops.norm_fwd(x, act=act, gb=gb)
        ctx.act = act
        ctx.save_for_backward(x.contiguous(), mean, rstd, gb, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gb, y = ctx.saved_tensors
        dx, dgb, _ = ops.norm_bwd(dy, y, x, mean, rstd, act=ctx.act, gb=gb)
        return dx, dgb, None


def instance_norm(x, act=_NONE):
    """nn.InstanceNorm2d(affine=False) (+ fused activation) on (B,H,W,C)."""
    return NormAct.apply(x, None, None, act)


class AttnFn(torch.autograd.Function):
    """Attention-form LWB (flow resize + warp + softmax over the sources) with its HIP backward.
    q (B,h,w,C) incl. bias; Ks/Vs (B*ns,h,w,C) = Wk x / Wv x without bias; T (B,ns,S,S,2) constant."""

    @staticmethod
    def forward(ctx, q, Ks, Vs, bk, bv, T):
        q, Ks, Vs, T = q.contiguous(), Ks.contiguous(), Vs.contiguous(), T.contiguous()
        out = ops.lwb_attention(q, Ks, Vs, bk, bv, T, torch.empty_like(q), src_batched=True)
        ctx.save_for_backward(q, Ks, Vs, bk, bv, T)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, Ks, Vs, bk, bv, T = ctx.saved_tensors
        dq, dKs, dVs = ops.lwb_attention_bwd(q, Ks, Vs, bk, bv, T, dout, src_batched=True)
        dbv = ops.colsum(dout) if ctx.needs_input_grad[4] else None
        dbk = torch.zeros_like(bk) if ctx.needs_input_grad[3] else None
        return dq, dKs, dVs, dbk, dbv, None


class AttnKVFn(torch.autograd.Function):
    """``AttnFn`` with K | V as ONE tensor kv (B*ns,h,w,2C) - the output of the stacked fk | fv projection (``conv_pair``) - read and
    differentiated in place (lwg_lwb_attention_kv_f32 / _kv_bwd_f32)."""

    @staticmethod
    def forward(ctx, q, kv, bk, bv, T):
        q, kv, T = q.contiguous(), kv.contiguous(), T.contiguous()
        out = ops.lwb_attention_kv(q, kv, bk, bv, T, torch.empty_like(q), src_batched=True)
        ctx.save_for_backward(q, kv, bk, bv, T)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, bk, bv, T = ctx.saved_tensors
        dq, dkv = ops.lwb_attention_kv_bwd(q, kv, bk, bv, T, dout, src_batched=True)
        dbv = ops.colsum(dout) if ctx.needs_input_grad[3] else None
        dbk = torch.zeros_like(bk) if ctx.needs_input_grad[2] else None
        return dq, dkv, dbk, dbv, None


class MaxPool2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) on NHWC (csrc/train_ops.hip)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2_bwd(x, dy)


def lwb_transform(x, T):
    """attlwb_spade_resunet.py:175-191 on NHWC features: flow resize (align_corners=True) + grid_sample (zeros)."""
    h, w = x.shape[1:3]
    if T.shape[1] != h or T.shape[2] != w:
        T = F.interpolate(T.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    out = F.grid_sample(x.permute(0, 3, 1, 2), T, mode="bilinear", padding_mode="zeros", align_corners=False)
    return out.permute(0, 2, 3, 1)


class TrainableGenerator(object):
    """Functional training forward over the parameters of an ``AttentionLWBGenerator`` (reference :633-699)."""

    def __init__(self, gen):
        self.gen = gen
        self.n_down = len(gen.num_filters)
        self.n_res = gen.n_res_block
        self.n_bg = len(gen.bg_filters) if gen.has_bg else 0
        self.fused_attention = True      # False: the eager warp + softmax chain (kept as the in-framework cross-check)

    def p(self, name):
        obj = self.gen
        for part in name.split("."):
            obj = getattr(obj, part)
        return obj

    def cv(self, name, x, x1=None, **kw):
        node = self.p(name)
        return conv(x, node.weight, getattr(node, "bias", None), x1=x1, **kw)

    # -- SelfAttentionLWB (:194-252) + SPADE (:80-93)
    def attlwb(self, pfx, tsf_x, src_x, Tst):
        bs, ns, S, _, _ = Tst.shape
        h, w, C = tsf_x.shape[1:]
        fk, fv = self.p(pfx + ".fk"), self.p(pfx + ".fv")
        q = self.cv(pfx + ".fq", tsf_x, pad=0)
        if self.fused_attention and FUSED_KV_PAIR:
            # ... and the two projections are ONE stacked 1x1 convolution (K | V along the channels), gathered in place by the kernel
            x = AttnKVFn.apply(q, conv_pair(src_x, fk.weight, None, fv.weight, None, pad=0), fk.bias, fv.bias, Tst)
        elif self.fused_attention:
            # a 1x1 conv commutes with the zero-padded warp: project the source features, warp inside the kernel
            Ks = conv(src_x, fk.weight, None, pad=0)
            Vs = conv(src_x, fv.weight, None, pad=0)
            x = AttnFn.apply(q, Ks, Vs, fk.bias, fv.bias, Tst)
        else:
            warp = lwb_transform(src_x, Tst.reshape(bs * ns, S, S, 2))
            K = self.cv(pfx + ".fk", warp, pad=0).view(bs, ns, h, w, C)
            V = self.cv(pfx + ".fv", warp, pad=0).view(bs, ns, h, w, C)
            logits = (K * q.unsqueeze(1)).sum(dim=4, keepdim=True) / math.sqrt(C)
            x = (torch.softmax(logits, dim=1) * V).sum(dim=1)
        if FUSED_SPADE_PAIR:         # mlp_shared's ReLU output feeds only the stacked gamma | beta convolution: its mask rides in that dgrad
            actv = self.cv(pfx + ".spade.mlp_shared.0", x, act=_RELU, premasked=True)
            g, b = self.p(pfx + ".spade.mlp_gamma"), self.p(pfx + ".spade.mlp_beta")
            return SpadeNormFn.apply(tsf_x, conv_pair(actv, g.weight, g.bias, b.weight, b.bias, mask_dx=True), _NONE)
        actv = self.cv(pfx + ".spade.mlp_shared.0", x, act=_RELU)
        gamma = self.cv(pfx + ".spade.mlp_gamma", actv)
        beta = self.cv(pfx + ".spade.mlp_beta", actv)
        return NormAct.apply(tsf_x, gamma, beta, _NONE)

    def res_block(self, pfx, x):
        return x + self.cv(pfx + ".main.2", self.cv(pfx + ".main.0", x, act=_RELU, premasked=True), mask_dx=True)

    def head(self, img_name, att_name, x):
        """The two 5x5 regressors with their tanh / sigmoid: one fused launch (``HeadFn``); NHWC views of its NCHW outputs."""
        img, mask = HeadFn.apply(x, self.p(img_name).weight, self.p(att_name).weight)
        return img.permute(0, 2, 3, 1), mask.permute(0, 2, 3, 1)

    # -- the three branches (NHWC in / out)
    def forward_bg(self, bg4):
        """(n,S,S,4) -> (n,S,S,3)   (bg_inpaintor.py:24-60)."""
        x, i = bg4, 0
        x = instance_norm(self.cv(f"bg_net.main.{i}", x, pad=3, cin_pad=4, need_dx=False), _RELU)
        i += 3
        for _ in range(self.n_bg - 1):
            x = instance_norm(self.cv(f"bg_net.main.{i}", x, stride=2), _RELU)
            i += 3
        for _ in range(self.n_res):
            y = instance_norm(self.cv(f"bg_net.main.{i}.main.0", x), _RELU)
            x = x + instance_norm(self.cv(f"bg_net.main.{i}.main.3", y))
            i += 1
        for _ in range(self.n_bg - 1):
            x = instance_norm(self.cv(f"bg_net.main.{i}", x, kind="convT"), _RELU)
            i += 3
        w = self.p(f"bg_net.main.{i}").weight            # Conv2d(nf, 3, 7, 1, 3, bias=False) + Tanh (bg_inpaintor.py:53-54)
        # ThinConvFn's backward is the thin MFMA form (dX = a conv with N = Cin columns: the kernel's granularity is 64 - ConvFn's ctx.thin rule)
        if w.shape[0] <= 4 and w.shape[2] in (5, 7) and x.shape[3] % 64 == 0 and x.shape[1] == x.shape[2]:
            return torch.tanh(ThinConvFn.apply(x, w))
        return torch.tanh(self.cv(f"bg_net.main.{i}", x, pad=3, n_pad=64))

    def forward_src(self, src8, side=None):
        """(n,S,S,8) -> (enc list, res list, img (n,S,S,3), mask (n,S,S,1))   (:450-478, only_enc=False).  side: a stream for the
        decoder + regressors (the caller joins it)."""
        x, enc, res = src8, [], []
        for i in range(self.n_down):
            x = self.cv(f"src_net.encoders.layers.{i}.0", x, stride=2, act=_RELU, cin_pad=8 if i == 0 else None, need_dx=i != 0)
            enc.append(x)
        for i in range(self.n_res):
            x = self.res_block(f"src_net.res_blocks.{i}", x)
            res.append(x)
        # the source decoder + regressors feed only the reconstruction losses: on the branch stream (behind the background
        # network) they overlap the transfer stream, which needs just enc / res
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())   # x is final on the main stream
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if side is not None:
                x.record_stream(side)
            for i in range(self.n_down):
                # a decoder layer's ReLU output feeds only the next layer: that layer's data gradient carries the mask
                x = self.cv(f"src_net.decoders.layers.{i}.0", x, kind="convT", act=_RELU, mask_dx=i > 0, premasked=i < self.n_down - 1)
            img, mask = self.head("src_net.img_reg.0", "src_net.att_reg.0", x)
        return enc, res, img, mask

    def forward_tsf(self, tsf8, enc_src, res_src, Tst):
        """(B,S,S,8), source features, Tst (B,ns,S,S,2) -> (img (B,S,S,3), mask (B,S,S,1))   (:480-535)."""
        x, enc = tsf8, []
        for i in range(self.n_down):
            x = self.cv(f"tsf_net_enc.layers.{i}.0", x, stride=2, act=_RELU, cin_pad=8 if i == 0 else None, need_dx=i != 0)
            x = self.attlwb(f"enc_attlwbs.{i}", x, enc_src[i], Tst)
            enc.append(x)
        for i in range(self.n_res):
            x = self.res_block(f"res_blocks.{i}", x)
            x = self.attlwb(f"res_attlwbs.{i}", x, res_src[i], Tst)
        for i in range(self.n_down):
            x = self.cv(f"tsf_net_dec.upconvs.{i}.0", x, kind="convT", act=_RELU, mask_dx=i > 0)      # input: the skipper's ReLU output
            if i != self.n_down - 1:
                x = self.cv(f"tsf_net_dec.skippers.{i}.0", enc[self.n_down - 2 - i], x1=x, act=_RELU, premasked=True)
        return self.head("tsf_img_reg.0", "tsf_att_reg.0", x)

    def forward(self, bg_inputs, src_inputs, tsf_inputs, Tst):
        """Reference signature (:633-699, temporal=False, only_tsf=False), NCHW in / out:
        bg_inputs (bs,nb,4,h,w), src_inputs (bs,ns,6,h,w), tsf_inputs (bs,nt,6,h,w), Tst (bs,nt,ns,h,w,2)
        -> bg (bs,nb,3,h,w), src_img (bs,ns,3,h,w), src_mask (bs,ns,1,h,w), tsf_img (bs,nt,3,h,w), tsf_mask (bs,nt,1,h,w)."""
        bs, nb = bg_inputs.shape[:2]
        ns, nt = src_inputs.shape[1], tsf_inputs.shape[1]
        h, w = tsf_inputs.shape[-2:]
        nhwc = lambda t, cp: F.pad(t.reshape(-1, t.shape[2], h, w).permute(0, 2, 3, 1), (0, cp - t.shape[2])).contiguous()   # noqa: E731
        nchw = lambda t, n: t.permute(0, 3, 1, 2).reshape(bs, n, t.shape[3], h, w)                                             # noqa: E731
        # the background network shares nothing with the source / transfer streams until the losses: on ops.BRANCH_STREAM (installed by
        # the trainer) it runs next to them - one training sample leaves most layers with fewer workgroups than the chip holds.
        # Autograd replays each node on the stream of its forward, so the two backward chains overlap the same way.
        side = ops.BRANCH_STREAM if bg_inputs.is_cuda else None
        if side is not None:
            cur = torch.cuda.current_stream()
            bg4 = nhwc(bg_inputs, 4)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                bg4.record_stream(side)
                bg = self.forward_bg(bg4)
        else:
            bg = self.forward_bg(nhwc(bg_inputs, 4))
        enc, res, s_img, s_mask = self.forward_src(nhwc(src_inputs, 8), side=side)
        imgs, masks = [], []
        for t in range(nt):
            if bs == 1:
                e, r = enc, res
            else:
                raise NotImplementedError("bs > 1 per rank: run one sample per process (config 5 is 1 sample / GPU)")
            img, mask = self.forward_tsf(nhwc(tsf_inputs[:, t:t + 1], 8), e, r, Tst[:, t].contiguous())
            imgs.append(img)
            masks.append(mask)
        if side is not None:
            cur.wait_stream(side)
            for v in (bg, s_img, s_mask):
                v.record_stream(cur)
        return (nchw(bg, nb), nchw(s_img, ns), nchw(s_mask, ns),
                nchw(torch.cat(imgs, dim=0), nt), nchw(torch.cat(masks, dim=0), nt))
