"""Personalization training step on the MI355X.

Reference: ``LWGTrainer`` (iPERCore/tools/trainers/lwg_trainer.py:609-832) driven by
``services/personalization.py:95-151``: per iteration ``forward`` (G with only_tsf=False), ``optimize_G`` (LSGAN +
L1 reconstruction + transfer loss + BCE mask + TV), Adam step, ``optimize_D`` (LSGAN real/fake), Adam step
(``optimize_parameters`` :326-352).  Discriminator: ``GlobalDiscriminator`` over ``PatchDiscriminator``
(models/networks/discriminators/multi_scale_dis.py:47-107, patch_dis.py:8-70), factory name ``patch_global``.

Every convolution of G and D (forward, data gradient, weight gradient) runs on the hand-written MFMA kernels through
``networks.training.ConvFn``; part of the elementwise glue is PyTorch-ROCm autograd (see that module).
The composed discriminators ``patch_global_local`` / ``patch_global_body_head`` (multi_scale_dis.py:110-284) are built on the same
``PatchDiscriminator``.  The VGG19 perceptual loss (``vggloss.py``) and the SphereFace loss (``faceloss.py``) run on the same conv
kernels with frozen weights.  ``TrainOpts`` defaults to the reference's loss set (deploy.toml: use_vgg = "VGG19", use_face = true);
their checkpoints are not distributable, and a missing file raises - as the reference does - unless
``TrainOpts.allow_seeded_loss_nets`` (benchmarks / tests: seeded weights) is set or the two losses are switched off
(``use_vgg = "None"`` / ``use_face = False`` is the reference's L1 transfer loss, lwg_trainer.py:154-158).

Data parallelism (BASELINE config 5: one sample per GPU): parameters and gradients of a network live in ONE flat fp32
buffer each (``FlatAdam``); the gradient buffer is averaged in place with a single RCCL all-reduce - large, few
collectives for point-to-point xGMI instead of DDP's 25 MB buckets - and updated by one fused Adam kernel.  The reference's personalization itself is single-GPU (no collective).
"""
import math
import os

import time

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .bodynets import SMPL, SMPLH
from .flowcomposition import FlowComposition
from .morphology import morph
from .networks.training import MaxPool2Fn, TrainableGenerator, conv, instance_norm

_RELU = 1


class PatchDiscriminator(nn.Module):
    """discriminators/patch_dis.py:8-70 with norm_type = "instance", use_sigmoid = False: 4x4 convolutions (n_layers of stride 2,
    then two of stride 1), InstanceNorm + LeakyReLU(0.2) between them, on the MFMA kernels (``ConvFn``) + ``NormAct``.
    Parameter names follow the reference Sequential: ``model.{0,2,5,8,11,14}``."""

    def __init__(self, input_nc=6, ndf=64, n_layers=4, max_nf_mult=8, norm_type="instance", use_sigmoid=False):
        super().__init__()
        if norm_type != "instance" or use_sigmoid:
            raise NotImplementedError("only norm_type='instance', use_sigmoid=False (deploy.toml / AttLWB-SPADE.toml) are built")
        if ndf % 32:
            raise ValueError("ndf must be a multiple of 32 (channel granularity of the MFMA conv kernels)")
        chans = [input_nc, ndf]
        for n in range(1, n_layers):
            chans.append(ndf * min(2 ** n, max_nf_mult))
        chans.append(ndf * min(2 ** n_layers, max_nf_mult))
        self.n_layers, self.input_nc = n_layers, input_nc
        idx = [0] + [2 + 3 * (n - 1) for n in range(1, n_layers + 1)]          # Sequential indices of the convs
        self.layer_names = [str(i) for i in idx] + [str(idx[-1] + 3)]
        self.model = nn.Module()
        for i, name in enumerate(self.layer_names[:-1]):
            m = nn.Module()
            m.weight = nn.Parameter(torch.empty(chans[i + 1], chans[i], 4, 4))
            m.bias = nn.Parameter(torch.empty(chans[i + 1]))
            self.model.add_module(name, m)
        m = nn.Module()
        m.weight = nn.Parameter(torch.empty(1, chans[-1], 4, 4))
        m.bias = nn.Parameter(torch.empty(1))
        self.model.add_module(self.layer_names[-1], m)
        for name in self.layer_names:                                          # PyTorch Conv2d default init
            layer = getattr(self.model, name)
            bound = 1.0 / (layer.weight.shape[1] * 16) ** 0.5
            layer.weight.data.uniform_(-bound, bound)
            layer.bias.data.uniform_(-bound, bound)

    def forward(self, x_nchw):
        """(N, input_nc, H, W) -> patch logits (N, 1, h, w)."""
        cp = 8 if x_nchw.shape[1] <= 8 else (x_nchw.shape[1] + 3) // 4 * 4
        x = F.pad(x_nchw.permute(0, 2, 3, 1), (0, cp - x_nchw.shape[1])).contiguous()
        n = len(self.layer_names)
        for i, name in enumerate(self.layer_names):
            layer = getattr(self.model, name)
            stride = 2 if i < self.n_layers else 1
            npad = None if layer.weight.shape[0] % 64 == 0 else (layer.weight.shape[0] + 63) // 64 * 64
            if i == 0:
                x = F.leaky_relu(conv(x, layer.weight, layer.bias, stride=stride, pad=1, cin_pad=cp, n_pad=npad), 0.2)   # dX only when the input asks for it: G's adversarial term
            elif i < n - 1:
                x = instance_norm(conv(x, layer.weight, layer.bias, stride=stride, pad=1, n_pad=npad), ops.ACT_LRELU)
            else:
                x = conv(x, layer.weight, layer.bias, stride=stride, pad=1, n_pad=64)
        return x.permute(0, 3, 1, 2)


def _cfg_get(cfg, key, default):
    if cfg is None:
        return default
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


def crop_img(imgs, rects, fact=2):
    """multi_scale_dis.py:21-44: crop (min_x, max_x, min_y, max_y) boxes and resize them to (H / fact, W / fact); degenerate boxes
    are dropped (a host read of the N x 4 integers, as in the reference)."""
    _, _, H, W = imgs.shape
    crops = []
    for i, (x0, x1, y0, y1) in enumerate(torch.as_tensor(rects).tolist()):
        if x0 != x1 and y0 != y1:
            crops.append(F.interpolate(imgs[i:i + 1, :, y0:y1, x0:x1], size=(H // fact, W // fact), mode="bilinear", align_corners=True))
    return torch.cat(crops, dim=0) if crops else crops


def _reduce_outs(outs):
    with torch.no_grad():
        return sum(o.mean() for o in outs) / len(outs)


class GlobalDiscriminator(nn.Module):
    """multi_scale_dis.py:47-107 (``patch_global``).  ``forward`` takes the reference's dict ({"x", "bg_x"[, "get_avg"]}) or, as the
    trainer here uses it, the image tensor itself (-> list of logits)."""
    CROPS = ()                                             # (model attribute, rect key, size factor) of the sub-classes

    def __init__(self, cfg=None, use_aug_bg=False, **kw):
        super().__init__()
        g = lambda k, d: kw.get(k, _cfg_get(cfg, k, d))                                           # noqa: E731
        mk = lambda nc: PatchDiscriminator(nc, g("ndf", 64), g("n_layers", 4), g("max_nf_mult", 8), g("norm_type", "instance"),     # noqa: E731
                                           g("use_sigmoid", False))
        self.global_model = mk(g("cond_nc", 6))
        for attr, _, _ in self.CROPS:
            setattr(self, attr, mk(g("cond_nc", 6)))
        self.bg_model = mk(g("bg_cond_nc", 4)) if use_aug_bg else None
        self.use_aug_bg = use_aug_bg

    # the attributes the single-network form exposed
    n_layers = property(lambda self: self.global_model.n_layers)
    layer_names = property(lambda self: self.global_model.layer_names)

    def forward(self, inputs):
        if torch.is_tensor(inputs):
            inputs = {"x": inputs, "bg_x": None}
        x, bg_x = inputs["x"], inputs.get("bg_x")
        outs = []
        if bg_x is not None and self.use_aug_bg:
            outs.append(self.bg_model(bg_x))
        outs.append(self.global_model(x))
        if self.use_aug_bg and not self.CROPS:                 # GlobalDiscriminator lists [global, bg] (:96-100)
            outs = outs[::-1]
        for attr, key, fact in self.CROPS:
            if inputs.get(key) is None:
                raise ValueError(f"{type(self).__name__} needs inputs['{key}'] (N, 4 = (min_x, max_x, min_y, max_y))")
            crops = crop_img(x, inputs[key], fact=fact)
            if len(crops) != 0:
                outs.append(getattr(self, attr)(crops))
        if inputs.get("get_avg", False):
            return outs, _reduce_outs(outs)
        return outs


class GlobalLocalDiscriminator(GlobalDiscriminator):
    """multi_scale_dis.py:110-191 (``patch_global_local``): + a second PatchDiscriminator on the body crop at half size."""
    CROPS = (("local_model", "body_rects", 2),)


class GlobalBodyHeadDiscriminator(GlobalDiscriminator):
    """multi_scale_dis.py:194-284 (``patch_global_body_head``): + body crop at half size and head crop at quarter size."""
    CROPS = (("body_model", "body_rects", 2), ("head_model", "head_rects", 4))


class PatchGlobalDiscriminator(GlobalDiscriminator):
    """``patch_global`` as the personalization step uses it: GlobalDiscriminator without the augmented-background branch
    (use_aug_bg=False, the deploy.toml default), keyword arguments instead of a cfg."""

    def __init__(self, cond_nc=6, ndf=64, n_layers=4, max_nf_mult=8):
        super().__init__(None, False, cond_nc=cond_nc, ndf=ndf, n_layers=n_layers, max_nf_mult=max_nf_mult)


class MultiScaleDiscriminator(nn.Module):
    """multi_scale_dis.py:287-332 (factory name ``multi_scale``): an optional global PatchDiscriminator on ``global_x`` plus two
    PatchDiscriminators on ``local_x`` and on its half-size bilinear (align_corners=True) resize; ``forward`` returns the list
    [global?, scale 0, scale 1].  The reference's positional signature; ``norm_type`` must be "instance" (its default, "batch", is not
    built: no runner or config of the reference constructs this class).  The reference's ``get_avg=True`` path calls an undefined
    ``self.reduce_tensor`` and raises; here it returns the mean logit as its other discriminators do (``reduce_tensor`` :9-18)."""

    def __init__(self, global_nc, input_nc, ndf=32, n_layers=3, max_nf_mult=8, norm_type="batch", use_sigmoid=False):
        super().__init__()
        if norm_type != "instance":
            raise NotImplementedError(f"MultiScaleDiscriminator(norm_type={norm_type!r}): only norm_type='instance' is built (the reference's default, "
                                      "'batch', is reached by none of its runners or configs): pass norm_type='instance'")
        self.n_scales = 2
        self.scale_models = nn.ModuleList([PatchDiscriminator(input_nc, ndf, n_layers, max_nf_mult, norm_type, use_sigmoid)
                                           for _ in range(self.n_scales)])
        self.global_model = PatchDiscriminator(global_nc, ndf, n_layers, max_nf_mult, norm_type, use_sigmoid) if global_nc is not None else None

    def forward(self, global_x, local_x, body_rects=None, head_rects=None, get_avg=True):
        outs = []
        if self.global_model is not None:
            outs.append(self.global_model(global_x))
        _, _, H, W = local_x.shape
        x = local_x
        for i in range(self.n_scales):
            outs.append(self.scale_models[i](x))
            if i < self.n_scales - 1:
                fact = 2 ** (i + 1)
                x = F.interpolate(local_x, size=(H // fact, W // fact), mode="bilinear", align_corners=True)
        return (outs, _reduce_outs(outs)) if get_avg else outs


def create_discriminator(name, cfg=None, use_aug_bg=False):
    """The discriminator entries of the reference's NetworksFactory (networks/__init__.py:50-60)."""
    table = {"patch_global": GlobalDiscriminator, "patch_global_local": GlobalLocalDiscriminator,
             "patch_global_body_head": GlobalBodyHeadDiscriminator}
    if name not in table:
        raise ValueError(f"Network {name} not recognized (built: {sorted(table)}; multi_scale has its own constructor signature: "
                         "NetworksFactory.get_by_name('multi_scale', global_nc, input_nc, ...))")
    return table[name](cfg, use_aug_bg=use_aug_bg)


def lsgan_loss(outs, target):
    """criterions/ganloss.py:7-21."""
    return sum(torch.mean((o - target) ** 2) for o in outs) / len(outs)


def tv_loss(mat):
    """criterions/generals.py:7-13."""
    return torch.mean(torch.abs(mat[:, :, :, :-1] - mat[:, :, :, 1:])) + torch.mean(torch.abs(mat[:, :, :-1, :] - mat[:, :, 1:, :]))


class FlatAdam(object):
    """torch.optim.Adam (lwg_trainer.py:140-146: lr, betas, eps 1e-8, no weight decay) over ONE flat fp32 buffer.

    The parameters of the module are re-pointed at slices of a single buffer and their .grad at slices of a single gradient
    buffer, so (a) the update is one HIP kernel (lwg_adam_step_f32) instead of a multi-tensor loop and (b) the data-parallel
    exchange all-reduces the gradient buffer in place - no flatten / unflatten copies."""

    # parameters that run as ONE stacked convolution in the training step (training.conv_pair): the second one is placed right behind
    # the first in the flat buffer, so cat([first, second]) is a VIEW of it (no per-step concatenation / panel re-registration)
    PAIRS = ((".mlp_gamma.weight", ".mlp_beta.weight"), (".mlp_gamma.bias", ".mlp_beta.bias"), (".fk.weight", ".fv.weight"))

    @classmethod
    def _ordered(cls, module):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        by_name = dict(named)
        follower = {}                                    # first name -> second name
        for a, b in cls.PAIRS:
            for n, _ in named:
                if n.endswith(a) and n[:-len(a)] + b in by_name:
                    follower[n] = n[:-len(a)] + b
        placed, out = set(follower.values()), []
        for n, p in named:
            if n in placed:
                continue
            out.append(p)
            if n in follower:
                out.append(by_name[follower[n]])
        return out

    def __init__(self, module, lr, betas=(0.9, 0.999), eps=1e-8):
        self.module, self.lr, self.betas, self.eps, self.t = module, lr, betas, eps, 0
        params = self._ordered(module)
        n = sum(p.numel() for p in params)
        n4 = (n + 3) // 4 * 4
        dev = params[0].device
        self.flat = torch.zeros(n4, device=dev)
        self.grad = torch.zeros(n4, device=dev)
        self.m, self.v = torch.zeros(n4, device=dev), torch.zeros(n4, device=dev)
        self.t_dev = torch.zeros(1, device=dev, dtype=torch.int32)      # the step count lives on the device: graph replays advance it
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            off += k
        self.params = params
        self._slots, off = [], 0
        for p in params:                           # a parameter's range of the flat gradient buffer
            self._slots.append(self.grad[off:off + p.numel()].view_as(p))
            p.grad = self._slots[-1]
            off += p.numel()

    def zero_grad(self):
        """Gradients start each step as None: autograd then KEEPS the tensor a backward function hands it (no accumulation
        kernel per parameter - ~330 small adds, 1.6 ms of a 34 ms step) and ``_gather`` moves them into the flat buffer with one
        multi-tensor copy before they are exchanged / applied.  Ranges of parameters that receive no gradient stay zero."""
        self.grad.zero_()
        for p in self.params:
            p.grad = None

    def _gather(self, indices=None):
        """p.grad (whatever tensor autograd left there) -> its range of the flat buffer; p.grad becomes that view."""
        idx = range(len(self.params)) if indices is None else indices
        src, dst = [], []
        for i in idx:
            p, slot = self.params[i], self._slots[i]
            if p.grad is None or p.grad.data_ptr() == slot.data_ptr():
                continue
            src.append(p.grad.detach())
            dst.append(slot)
            p.grad = slot
        if dst:
            torch._foreach_copy_(dst, src)

    # ---- data-parallel exchange, overlapped with backward ---------------------------------------------------------------
    def arm(self, group=None, n_buckets=4, force=False):
        """Call before the backward whose gradients this optimizer will apply.  The flat gradient buffer is cut into
        ``n_buckets`` contiguous ranges (a few large collectives: xGMI rings are per-link bound, so 145 MB of G gradients go
        out as ~36 MB pieces, not DDP's 25 MB x many); a post-accumulate hook per parameter counts the range down and hands it to
        an ``async_op`` all-reduce the moment its last gradient has been accumulated - the exchange of the late layers runs
        on RCCL's stream while backward is still computing the early ones.  ``allreduce()`` afterwards only finishes the job."""
        self._works, self._issued = [], set()
        self._group = group
        self._armed = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)   # force: one-rank RCCL check
        if not self._armed:
            return
        if not hasattr(self, "_bucket_of"):
            total = self.grad.numel()
            bounds = [(total * b // n_buckets) // 4 * 4 for b in range(n_buckets)] + [total]
            self._ranges = [(bounds[b], bounds[b + 1]) for b in range(n_buckets) if bounds[b + 1] > bounds[b]]
            self._bucket_of, self._members = [], [0] * len(self._ranges)
            self._params_of = [[] for _ in self._ranges]
            off = 0
            for i, p in enumerate(self.params):
                touched = [b for b, (lo, hi) in enumerate(self._ranges) if lo < off + p.numel() and off < hi]
                self._bucket_of.append(touched)            # a parameter may straddle a boundary: it counts for both ranges
                for b in touched:
                    self._members[b] += 1
                    self._params_of[b].append(i)
                off += p.numel()
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i))
        self._pending = list(self._members)

    def _on_grad(self, i):
        if not getattr(self, "_armed", False):
            return
        for b in self._bucket_of[i]:
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._issue(b)

    def _issue(self, b):
        lo, hi = self._ranges[b]
        self._gather(self._params_of[b])
        self._issued.add(b)
        self._works.append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self._group, async_op=True))

    def allreduce(self, group=None, force=False):
        """Average the flat gradient buffer over the ranks.  After ``arm()``: wait for the ranges already in flight and send the
        ones whose parameters received no gradient in this backward; without ``arm()``: one all-reduce of the whole buffer."""
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
            return
        self._gather()
        if getattr(self, "_armed", False):
            for b in range(len(self._ranges)):
                if b not in self._issued:
                    self._issue(b)
            for w in self._works:
                w.wait()
            self.overlapped_ranges = len(self._issued)
            self._armed = False
        else:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
        self.grad.div_(dist.get_world_size(group))

    def allreduce_async(self, group=None, n_ranges=1, force=False):
        """Hand the (already complete) flat gradient buffer to RCCL as ``n_ranges`` async all-reduces and return the work handles
        (None when there is nothing to exchange): the collectives run on RCCL's stream from this point of the current stream on,
        next to whatever the caller launches afterwards on OTHER streams.  ``allreduce_finish`` waits and applies the 1/N."""
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
            return None
        self._gather()
        total = self.grad.numel()
        bounds = [(total * b // n_ranges) // 4 * 4 for b in range(n_ranges)] + [total]
        return [dist.all_reduce(self.grad[bounds[b]:bounds[b + 1]], op=dist.ReduceOp.SUM, group=group, async_op=True)
                for b in range(n_ranges) if bounds[b + 1] > bounds[b]]

    def allreduce_finish(self, works, group=None):
        if works is None:
            return
        for w in works:
            w.wait()                                   # the current stream waits for RCCL's stream
        self.grad.div_(dist.get_world_size(group))

    def snapshot(self):
        """Parameters, both moments and the step count (device + host mirror): what a throw-away step must not leave changed."""
        return (self.flat.clone(), self.m.clone(), self.v.clone(), self.t_dev.clone(), self.t)

    def restore(self, snap):
        flat, m, v, t_dev, t = snap
        self.flat.copy_(flat)
        self.m.copy_(m)
        self.v.copy_(v)
        self.t_dev.copy_(t_dev)
        self.t = t

    def note_replayed(self, n=1):
        """A captured ``step()`` was replayed ``n`` times: the device did the update (raw pointers - no tensor version counter moved),
        so advance the host mirror of the step count and drop the module's packed inference panels here."""
        self.t += n
        if hasattr(self.module, "_packed"):
            self.module._packed = None

    def step(self):
        self.t += 1                                    # host mirror; graph replays advance it through note_replayed()
        self._gather()
        ops.adam_step_dev(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t_dev)
        if hasattr(self.module, "_packed"):
            self.module._packed = None             # the inference engine's packed panels are stale now


def allreduce_grads(params, group=None):
    """Average the gradients of ``params`` over the ranks with ONE all-reduce of a flat fp32 buffer."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def _load_frozen(net, ckpt_path, allow_seeded, what):
    """Load the pretrained weights of a frozen loss network, or fail the way the reference does (its torch.load raises on a missing
    file, vggloss.py:30-33 / faceloss.py:300-303).  Seeded weights are an explicit opt-in for benchmarks and tests
    (``allow_seeded``): a step then has the right cost and gradient structure, NOT the trained metric."""
    if ckpt_path and os.path.exists(ckpt_path):
        sd = torch.load(ckpt_path, map_location="cpu")
        if isinstance(sd, dict) and "state_dict" in sd and all(not torch.is_tensor(v) for k, v in sd.items() if k != "state_dict"):
            sd = sd["state_dict"]
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        res = net.load_state_dict(sd, strict=False)       # unexpected keys are fine (torchvision's classifier, Sphere20a's fc6)
        if res.missing_keys:
            raise RuntimeError(f"{what}: checkpoint {ckpt_path} lacks {len(res.missing_keys)} of the network's tensors "
                               f"(first: {res.missing_keys[:3]}) - refusing to train against partly random features")
        return
    if not allow_seeded:
        raise FileNotFoundError(f"{what}: pretrained weights not found at {ckpt_path!r}.  The reference cannot personalize without them "
                                "either; set TrainOpts.allow_seeded_loss_nets = True ONLY for benchmarks / tests (seeded weights).")


class VGG19Features(nn.Module):
    """criterions/vggloss.py:10-96 (VGG19, before_relu=False): torchvision's ``vgg19().features`` cut after relu1_1, relu2_1,
    relu3_1, relu4_1, relu5_1 - 13 frozen 3x3 convolutions (+ReLU) and four 2x2 max-pools on the MFMA / HIP kernels.
    ``state_dict`` keys follow torchvision (``features.{i}.weight``), so ``vgg19-dcbb9e9d.pth`` loads unchanged; without a
    checkpoint the weights are seeded He-normal (the loss then has the right cost and gradient structure, not the trained metric)."""
    CFG = ((0, 3, 64), (2, 64, 64), "M", (5, 64, 128), (7, 128, 128), "M", (10, 128, 256), (12, 256, 256), (14, 256, 256), (16, 256, 256), "M",
           (19, 256, 512), (21, 512, 512), (23, 512, 512), (25, 512, 512), "M", (28, 512, 512))
    TAPS = (0, 5, 10, 19, 28)                # conv indices whose ReLU output is a loss feature (slice_ids [2, 7, 12, 21, 30])

    def __init__(self, ckpt_path=None, seed=0, allow_seeded=False):
        super().__init__()
        self.features = nn.Module()
        g = torch.Generator().manual_seed(seed)
        for item in self.CFG:
            if item == "M":
                continue
            idx, cin, cout = item
            layer = nn.Module()
            layer.weight = nn.Parameter(torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin)), requires_grad=False)
            layer.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
            self.features.add_module(str(idx), layer)
        _load_frozen(self, ckpt_path, allow_seeded, "VGG19 perceptual loss (use_vgg)")

    def forward(self, x_nchw):
        """(N,3,H,W) in the generator's [-1,1] range (the reference feeds it un-normalised too) -> five NHWC feature maps."""
        x = F.pad(x_nchw.permute(0, 2, 3, 1), (0, 64 - x_nchw.shape[1])).contiguous()      # 3 -> 64 channels: the data gradient of
        outs = []                                                                           # the first conv needs N' % 64 == 0
        for item in self.CFG:
            if item == "M":
                x = MaxPool2Fn.apply(x)
                continue
            idx = item[0]
            layer = getattr(self.features, str(idx))
            x = conv(x, layer.weight, layer.bias, act=1, cin_pad=64 if idx == 0 else None)
            if idx in self.TAPS:
                outs.append(x)
        return outs


# The frozen loss networks (VGG19, Sphere20a) see [fake | target] as ONE batch (True: half the launches, but the data gradients then run over the
# target rows too - zeros) or as two passes with the target under no_grad (False).  Measured in bench_personalize.py --two-pass-loss.
LOSS_NETS_ONE_PASS = True


class VGGLoss(nn.Module):
    """criterions/vggloss.py:261-292: sum_i w_i * L1(vgg_i(x), vgg_i(y).detach()), inputs resized to 224x224 (bilinear,
    align_corners=True) as the trainers do (lwg_trainer.py:153-155, resize=True)."""
    WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)

    def __init__(self, ckpt_path=None, resize=True, allow_seeded=False):
        super().__init__()
        self.vgg, self.resize = VGG19Features(ckpt_path, allow_seeded=allow_seeded), resize

    def forward(self, x, y):
        if self.resize:
            x = F.interpolate(x, size=(224, 224), mode="bilinear", align_corners=True)
            y = F.interpolate(y, size=(224, 224), mode="bilinear", align_corners=True)
        # one pass over [x | y] (the network is frozen and has no batch statistics: the same features as two passes, half the launches - a
        # one-sample launch leaves most of the chip idle); the target half is cut out of the graph
        n = x.shape[0]
        if not LOSS_NETS_ONE_PASS:          # two passes: the target half under no_grad (no activations saved, no data gradients over its rows)
            fx = self.vgg(x)
            with torch.no_grad():
                fy = self.vgg(y)
            return sum(w * F.l1_loss(a, b) for w, a, b in zip(self.WEIGHTS, fx, fy))
        f = self.vgg(torch.cat([x, y.detach()], dim=0))
        return sum(w * F.l1_loss(a[:n], a[n:].detach()) for w, a in zip(self.WEIGHTS, f))


class _PReLUFn(torch.autograd.Function):
    """res + PReLU(x) with frozen per-channel slopes: one launch forward, one backward (ops.prelu / lwg_prelu_f32) - the ten launches of the
    torch.where form per activation were 1.6 ms of the step with the reference's default loss set."""

    @staticmethod
    def forward(ctx, x, slope, res):
        if slope.requires_grad:
            raise RuntimeError("_PReLUFn: the slopes are frozen (the Sphere20a of the face loss); a trainable nn.PReLU is not part of this path")
        ctx.save_for_backward(x, slope)
        ctx.has_res = res is not None
        return ops.prelu(x, slope, res)

    @staticmethod
    def backward(ctx, dy):
        x, slope = ctx.saved_tensors
        dy = dy.contiguous()
        return (ops.prelu_bwd(x, slope, dy) if ctx.needs_input_grad[0] else None), None, (dy if ctx.has_res and ctx.needs_input_grad[2] else None)


class Sphere20aFeatures(nn.Module):
    """criterions/faceloss.py:203-285 (Sphere20a): 20 frozen 3x3 convolutions (four of them stride 2) with per-channel PReLU and
    residual adds, then fc5; parameter names as in ``sphere20a_20171020.pth`` (conv{b}_{i}, relu{b}_{i}, fc5) so the checkpoint
    loads when present (strict=False: its fc6 classifier head is not part of the loss).  The convolutions run on the MFMA kernels
    (data gradient only: the network is frozen), PReLU + residual add on one HIP launch each way; fc5 is one library GEMM."""
    BLOCKS = ((1, 3, 64, 3), (2, 64, 128, 5), (3, 128, 256, 9), (4, 256, 512, 3))          # (block, cin, cout, number of convs)

    def __init__(self, ckpt_path=None, seed=0, allow_seeded=False):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        for b, cin, cout, n in self.BLOCKS:
            for i in range(1, n + 1):
                ci = cin if i == 1 else cout
                conv_ = nn.Module()
                conv_.weight = nn.Parameter(torch.randn(cout, ci, 3, 3, generator=g) * math.sqrt(2.0 / (9 * ci)), requires_grad=False)
                conv_.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
                self.add_module(f"conv{b}_{i}", conv_)
                act = nn.Module()
                act.weight = nn.Parameter(torch.full((cout,), 0.25), requires_grad=False)
                self.add_module(f"relu{b}_{i}", act)
        self.fc5 = nn.Module()
        self.fc5.weight = nn.Parameter(torch.randn(512, 512 * 7 * 6, generator=g) * math.sqrt(1.0 / (512 * 7 * 6)), requires_grad=False)
        self.fc5.bias = nn.Parameter(torch.zeros(512), requires_grad=False)
        _load_frozen(self, ckpt_path, allow_seeded, "SphereFace loss (use_face)")

    def _cp(self, b, i, x, stride=1, first=False, res=None):
        c, a = getattr(self, f"conv{b}_{i}"), getattr(self, f"relu{b}_{i}")
        y = conv(x, c.weight, c.bias, stride=stride, cin_pad=64 if first else None)
        return _PReLUFn.apply(y, a.weight, res)                                          # nn.PReLU(C) on NHWC (+ the block's residual add)

    def forward(self, x_nchw):
        """(N,3,112,96) -> [block1 (N,56,48,64), block2 (N,28,24,128), block3 (N,14,12,256), block4 (N,7,6,512), fc5 (N,512)]."""
        x = F.pad(x_nchw.permute(0, 2, 3, 1), (0, 64 - x_nchw.shape[1])).contiguous()
        outs = []
        for b, cin, cout, n in self.BLOCKS:
            x = self._cp(b, 1, x, stride=2, first=b == 1)
            for i in range(2, n + 1, 2):
                x = self._cp(b, i + 1, self._cp(b, i, x), res=x)
            outs.append(x)
        outs.append(F.linear(x.permute(0, 3, 1, 2).reshape(x.shape[0], -1), self.fc5.weight, self.fc5.bias))
        return outs


class _CropResizeFn(torch.autograd.Function):
    """Head crops with the boxes read on the device (ops.crop_resize / lwg_crop_resize_bilinear_f32): (crops, valid)."""

    @staticmethod
    def forward(ctx, imgs, box, oh, ow):
        y, valid = ops.crop_resize(imgs, box, (oh, ow))
        ctx.save_for_backward(box)
        ctx.hw = (imgs.shape[2], imgs.shape[3])
        ctx.mark_non_differentiable(valid)
        return y, valid

    @staticmethod
    def backward(ctx, dy, _dvalid):
        (box,) = ctx.saved_tensors
        return ops.crop_resize_bwd(dy.contiguous(), box, ctx.hw), None, None, None


class FaceLoss(nn.Module):
    """criterions/faceloss.py:288-406 (Sphere20a branch): heads cropped by bounding box, resized to 112x96 (bilinear,
    align_corners=True), weighted L1 between the five Sphere20a features; the second argument is the target (detached).
    The reference reads the boxes on the host (``bboxs[i]`` indexing, :384-400) and drops samples whose box is empty; here the boxes stay on
    the device (``crop_head_bbox`` -> crops + a validity flag per sample, zeros for an empty box) and the L1 means run over the valid samples
    only - the same value, with static shapes: the step with the reference's default loss set is captured as a hipGraph like the others."""
    WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)
    HEIGHT, WIDTH = 112, 96

    def __init__(self, pretrained_path=None, allow_seeded=False):
        super().__init__()
        self.net = Sphere20aFeatures(pretrained_path, allow_seeded=allow_seeded)

    def crop_head_bbox(self, imgs, bboxs):
        """:384-406; bboxs (N,4) = [min_x, max_x, min_y, max_y] int64 on the device -> (crops (N,3,112,96), valid (N))."""
        if bboxs.device != imgs.device or bboxs.dtype != torch.int64:      # (the trainer binds device int64 boxes: nothing to do in a captured step)
            bboxs = bboxs.to(device=imgs.device, dtype=torch.int64)
        return _CropResizeFn.apply(imgs.contiguous().float(), bboxs, self.HEIGHT, self.WIDTH)

    def forward(self, imgs1, imgs2, bbox1=None, bbox2=None):
        valid = None
        if bbox1 is not None:
            h1, valid = self.crop_head_bbox(imgs1, bbox1)
        else:
            h1 = F.interpolate(imgs1, size=(self.HEIGHT, self.WIDTH), mode="bilinear", align_corners=True)
        if bbox2 is not None:
            h2, v2 = self.crop_head_bbox(imgs2, bbox2)
            valid = v2 if valid is None else valid * v2
        else:
            h2 = F.interpolate(imgs2, size=(self.HEIGHT, self.WIDTH), mode="bilinear", align_corners=True)
        n = h1.shape[0]
        if not LOSS_NETS_ONE_PASS:
            f1 = self.net(h1)
            with torch.no_grad():
                f2 = self.net(h2)
        else:
            f = self.net(torch.cat([h1, h2.detach()], dim=0))      # one pass over [fake | target] heads (see VGGLoss.forward)
            f1, f2 = [a[:n] for a in f], [a[n:].detach() for a in f]
        if valid is None:
            return sum(w * F.l1_loss(a, b) for w, a, b in zip(self.WEIGHTS, f1, f2))
        nv = valid.sum().clamp_min(1.0)                      # no valid head at all: every term is zero (the reference returns 0)
        return sum(w * (((a - b).abs().flatten(1).mean(dim=1) * valid).sum() / nv) for w, a, b in zip(self.WEIGHTS, f1, f2))


class TrainOpts(object):
    """deploy.toml:76-102 defaults, incl. the reference's loss set (use_vgg = "VGG19", use_face = true).  Their checkpoints are not
    distributable: without the files the trainer raises unless ``allow_seeded_loss_nets`` is set (benchmarks / tests), or the two
    losses are switched off explicitly (use_vgg = "None", use_face = False: the reference's L1 transfer loss, lwg_trainer.py:154-158)."""
    lambda_rec, lambda_tsf, lambda_mask, lambda_mask_smooth, lambda_D_prob = 10.0, 10.0, 5.0, 1.0, 1.0
    lr_G, lr_D = 1e-4, 1e-4
    G_adam_b1, G_adam_b2, D_adam_b1, D_adam_b2 = 0.9, 0.999, 0.9, 0.999
    # "winograd" (default since round 5): the 3 x 3 / stride 1 forward and data-gradient convolutions run as F(2x2,3x3) Winograd
    # convolutions on the fp32 matrix pipe (csrc/conv_winograd.hip; launches too small to fill the chip keep the direct split-K form:
    # ops.WINO_MIN_GRID), everything else - and every weight gradient - on the direct fp32 MFMA kernels; "fp32": all direct;
    # "split": forward and data-gradient convs with Cin % 32 == 0 on the bf16x6 kernel (fp32-level accuracy)
    conv_precision = "winograd"
    # "VGG19": the transfer loss is the VGG19 perceptual loss (deploy.toml:83, the reference's default) instead of L1;
    # vgg_loss_path: torchvision vgg19 state_dict (used when the file exists, seeded weights otherwise)
    use_vgg = "VGG19"
    vgg_loss_path = "./assets/checkpoints/losses/vgg19-dcbb9e9d.pth"
    # the SphereFace (Sphere20a) loss on the head crop of the transferred image (deploy.toml:77-79, lambda_face :87)
    use_face = True
    face_loss_path = "./assets/checkpoints/losses/sphere20a_20171020.pth"
    lambda_face = 5.0
    # replay the static-shape step as hipGraphs (three segments: G forward / loss / backward, Adam(G) + D loss / backward, Adam(D); the
    # data-parallel all-reduces run between them): one step is ~2700 kernel launches and the Python / ctypes launch path needs
    # 26 ms to enqueue what the GPU executes in 33 ms - any kernel-side gain would otherwise be hidden behind the host
    use_graph = True
    # one launch re-packs every weight panel of the step (ops.PanelCache / lwg_pack_panels_f32) instead of one launch per panel
    use_panel_cache = True
    # the background network (forward and, through autograd's stream replay, backward) on a second stream next to the source / transfer
    # networks: independent until the losses (32.2 -> 30.3 ms per step); the source decoder + regressors ride the same stream
    branch_streams = True
    # captured step only: D's own forward / backward next to G's backward (it needs the fake images and D's weights, not G's update)
    overlap_d_step = True
    # data-parallel runs (N > 1): "segmented" = the step as segments [G fwd/bwd | D fwd/bwd | Adam(G) | Adam(D)] with G's gradient
    # all-reduce on RCCL's stream WHILE D's segment computes and D's all-reduce while Adam(G) runs (the captured step always has this
    # form at N > 1); "hooks" = the eager step with hook-driven range all-reduces during backward (FlatAdam.arm)
    dp_schedule = "hooks"
    allow_seeded_loss_nets = False                  # True: seeded VGG19 / Sphere20a weights when a checkpoint is absent (NOT a trained metric)

    @classmethod
    def l1_transfer(cls):
        """The reference's configuration without the two pretrained loss networks (use_vgg = "None", use_face = false)."""
        o = cls()
        o.use_vgg, o.use_face = "None", False
        return o


class FlowCompositionForTrainer(FlowComposition):
    """tools/trainers/base.py:90-141: the per-sample input stage of the trainers - body model, renders, the once-per-source
    image stage, target conditions and the (bs, nt, ns) flows - on the same HIP kernels the runner uses.
    ``body_model``: any model with ``get_details``; default: the trainers' 24-joint ``SMPL`` (bodynets/batch_smpl.py:283-436) from
    ``opt.smpl_model`` (raises when the pickle is missing, as the reference does).  An explicitly passed SMPL-H has no COCO+
    keypoints: its head / body boxes are then None and the box-dependent losses / discriminator crops refuse to run."""

    def __init__(self, opt, body_model=None):
        super().__init__(opt)
        g = lambda k, d=None: getattr(opt, k, d) if not isinstance(opt, dict) else opt.get(k, d)    # noqa: E731
        if body_model is None:
            p24 = g("smpl_model")
            if p24 is None or not (isinstance(p24, dict) or os.path.exists(p24)):
                # the head / body boxes below index the 19 COCO+ keypoints of the trainers' SMPL (base.py:205-285); with SMPL-H's 52
                # joints they would silently crop shoulders and hands - the reference raises here too (base.py:95)
                raise FileNotFoundError(f"opt.smpl_model = {p24!r}: the trainers' 24-joint SMPL pickle is required "
                                        "(pass body_model=SMPLH(...) explicitly to train without keypoint-based crops)")
            body_model = SMPL(model_path=p24)
        self.smpl = body_model
        self.has_cocoplus = isinstance(body_model, SMPL)
        self.ft_ks = int(g("ft_ks", 1))                      # deploy.toml:12
        self.share_bg = bool(g("share_bg", True))

    @torch.no_grad()
    def forward(self, src_img, ref_img, src_smpl, ref_smpl, src_mask=None, ref_mask=None, links_ids=None, offsets=0, temporal=False):
        """-> input_G_bg (bs,nb,4,h,w), input_G_src (bs,ns,6,h,w), input_G_tsf (bs,nt,6,h,w), Tst (bs,nt,ns,h,w,2), Ttt,
        src_mask (bs,ns,1,h,w), tsf_mask (bs,nt,1,h,w), head_bbox (bs*nt,4), body_bbox (bs*nt,4), uv_img (bs,3,h,w)."""
        bs, ns, _, h, w = src_img.shape
        nt = ref_img.shape[1]
        self.make_uv_setup(bs, self.num_source, self.time_step, src_img.device)
        sl = rl = None
        if links_ids is not None:
            nv, c = links_ids.shape[-2:]
            sl = links_ids.expand(bs, ns, nv, c).reshape(bs * ns, nv, c)
            rl = links_ids.expand(bs, nt, nv, c).reshape(bs * nt, nv, c)
        src_info = self.smpl.get_details(src_smpl.reshape(bs * ns, -1).contiguous(), offsets, links_ids=sl)
        ref_info = self.smpl.get_details(ref_smpl.reshape(bs * nt, -1).contiguous(), offsets, links_ids=rl)
        if src_mask is not None:
            src_info["masks"] = src_mask.reshape(bs * ns, 1, h, w).contiguous()
        if ref_mask is not None:
            ref_info["masks"] = ref_mask.reshape(bs * nt, 1, h, w).contiguous()
        self.add_rendered_f2verts_fim_wim(src_info, use_morph=True, get_uv_info=True)
        self.add_rendered_f2verts_fim_wim(ref_info, use_morph=False, get_uv_info=False)
        primary = None if self.share_bg else list(range(ns))
        uv_img, input_G_bg, input_G_src = self.process_source(src_img, src_info, primary_ids=primary)
        src_info.pop("_edge_counts", None)
        input_G_tsf = self.make_tsf_inputs(uv_img, ref_info)
        Tst, Ttt = self.make_batch_trans_flow(bs, ns, nt, src_info, ref_info, temporal=temporal)
        sm = src_info["masks"] if src_mask is not None else src_info["cond"][:, -1:]
        tm = ref_info["masks"] if ref_mask is not None else ref_info["cond"][:, -1:]
        sm = morph(sm.contiguous(), ks=self.ft_ks, mode="erode").view(bs, ns, 1, h, w)
        tm = morph(tm.contiguous(), ks=self.ft_ks, mode="erode").view(bs, nt, 1, h, w)
        head_bbox = self.cal_head_bbox_by_kps(ref_info["j2d"]) if self.has_cocoplus else None
        body_bbox = self.cal_body_bbox_by_kps(ref_info["j2d"]) if self.has_cocoplus else None
        return input_G_bg, input_G_src, input_G_tsf, Tst, Ttt, sm, tm, head_bbox, body_bbox, uv_img

    def cal_head_bbox_by_kps(self, kps, neck_ids=12):
        """base.py:205-246 -> (N, 4) long (min_x, max_x, min_y, max_y) in pixels (tiny host-visible bookkeeping: torch ops)."""
        S = self.image_size
        k = (kps + 1) / 2.0
        min_x = (k[:, neck_ids:, 0] - 0.05).min(dim=1)[0].clamp_min(0.0)
        max_x = (k[:, neck_ids:, 0] + 0.05).max(dim=1)[0].clamp_max(1.0)
        min_y = (k[:, neck_ids:, 1] - 0.05).min(dim=1)[0].clamp_min(0.0)
        max_y = k[:, neck_ids:, 1].max(dim=1)[0].clamp_max(1.0)
        return torch.stack([(min_x * S).long(), (max_x * S).long(), (min_y * S).long(), (max_y * S).long()], dim=1)

    def cal_body_bbox_by_kps(self, kps, factor=1.2):
        """base.py:248-285."""
        S = self.image_size
        k = (kps + 1) / 2.0
        out = []
        for d in (0, 1):
            lo, hi = k[:, :, d].min(dim=1)[0], k[:, :, d].max(dim=1)[0]
            mid, ext = (lo + hi) / 2, (hi - lo) * factor
            out += [((mid - ext / 2).clamp_min(0.0) * S).long(), ((mid + ext / 2).clamp_max(1.0) * S).long()]
        return torch.stack(out, dim=1)


class LWGTrainer(object):
    """lwg_trainer.py:609-832 for bs = 1 sample per process, share_bg = True, temporal = False, use_gan = True."""

    def __init__(self, G, D=None, opts=None, group=None, flow_comp=None):
        self.G, self.D, self.group, self.flow_comp = G, D, group, flow_comp
        self.opts = opts or TrainOpts()
        self.tg = TrainableGenerator(G)
        o = self.opts
        self.optimizer_G = FlatAdam(G, lr=o.lr_G, betas=(o.G_adam_b1, o.G_adam_b2))
        self.optimizer_D = None if D is None else FlatAdam(D, lr=o.lr_D, betas=(o.D_adam_b1, o.D_adam_b2))
        self.losses = {}
        self.crt_tsf = None
        if o.use_vgg == "VGG19":
            self.crt_tsf = VGGLoss(ckpt_path=o.vgg_loss_path, allow_seeded=o.allow_seeded_loss_nets).to(next(G.parameters()).device)
        elif o.use_vgg not in ("None", None, False):
            raise NotImplementedError(f"use_vgg = {o.use_vgg}: only VGG19 (the reference's default) is built")
        self.crt_face = FaceLoss(o.face_loss_path, allow_seeded=o.allow_seeded_loss_nets).to(next(G.parameters()).device) if o.use_face else None

    def set_input(self, inputs, device=None, flow_comp=None, ns=None):
        """lwg_trainer.py:624-697.  ``inputs`` is either the dataset sample of the reference (``PersonalizedDataset.__getitem__``,
        data/personalized_dataset.py:163-191: images (1,ns+nt,3,h,w), smpls (1,ns+nt,85), masks (1,ns+nt,1,h,w), bg (1,3,h,w),
        offsets, links_ids) - then ``flow_comp`` (a FlowCompositionForTrainer) builds the network inputs on the device - or the
        tensors that stage leaves on the trainer: input_G_bg (1,nb,4,h,w), input_G_src (1,ns,6,h,w), input_G_tsf (1,nt,6,h,w),
        Tst (1,nt,ns,h,w,2), real_src (1,ns,3,h,w), real_tsf (1,nt,3,h,w), real_bg (nb,3,h,w), body_mask (1,ns+nt,1,h,w)."""
        if "images" not in inputs:
            self._bind_inputs(inputs)
            return
        fc = flow_comp if flow_comp is not None else self.flow_comp
        assert fc is not None, "set_input(sample) needs a FlowCompositionForTrainer (LWGTrainer(..., flow_comp=...))"
        dev = device if device is not None else next(self.G.parameters()).device
        to = lambda k: torch.as_tensor(inputs[k], dtype=torch.float32).to(dev)     # noqa: E731
        images, smpls, masks, bg = to("images"), to("smpls"), to("masks"), to("bg")
        offsets = to("offsets") if "offsets" in inputs else 0
        links = torch.as_tensor(inputs["links_ids"]).to(dev) if inputs.get("links_ids") is not None else None
        ns = fc.num_source if ns is None else ns
        S = images.shape[-1]
        g_bg, g_src, g_tsf, Tst, _, _, tsf_mask, head_bbox, body_bbox, uv_img = fc(
            images[:, :ns].contiguous(), images[:, ns:].contiguous(), smpls[:, :ns].contiguous(), smpls[:, ns:].contiguous(),
            src_mask=masks[:, :ns].contiguous(), ref_mask=masks[:, ns:].contiguous(), links_ids=links, offsets=offsets)
        if head_bbox is None and (self.crt_face is not None or (self.D is not None and getattr(self.D, "CROPS", ()))):
            raise RuntimeError("the body model of this FlowCompositionForTrainer has no COCO+ keypoints (SMPL-H passed explicitly): the "
                               "head / body boxes of FaceLoss and of the patch_global_local / patch_global_body_head discriminators "
                               "cannot be formed - use the trainers' 24-joint SMPL (opt.smpl_model)")
        if not fc.share_bg:
            tsf_img = images[:, ns:]
            g_bg = torch.cat([g_bg, torch.cat([tsf_img * tsf_mask, tsf_mask], dim=2)], dim=1)
        self._bind_inputs({"input_G_bg": g_bg.contiguous(), "input_G_src": g_src.contiguous(), "input_G_tsf": g_tsf.contiguous(),
                           "Tst": Tst.contiguous(), "real_src": images[:, :ns].contiguous(), "real_tsf": images[:, ns:].contiguous(),
                           "real_bg": bg.view(-1, 3, S, S), "body_mask": masks, "uv_img": uv_img, "head_bbox": head_bbox,
                           "body_bbox": body_bbox})

    def _bind_inputs(self, inp):
        """The captured step reads its inputs from fixed buffers: once graphs exist, a new sample is COPIED into them (same shapes:
        the personalization loop cycles over samples of one video); a sample of another shape drops the graphs (re-captured)."""
        # the boxes are read by device kernels (FaceLoss crops): int64 tensors next to the images, moved there ONCE per sample - a pageable
        # host-to-device copy inside a stream capture is not permitted
        dev_ = next((v.device for v in inp.values() if torch.is_tensor(v) and v.is_cuda), None)
        if dev_ is not None:
            inp = dict(inp)
            for k in ("head_bbox", "body_bbox"):
                if torch.is_tensor(inp.get(k)) and (inp[k].device != dev_ or inp[k].dtype != torch.int64):
                    inp[k] = inp[k].to(device=dev_, dtype=torch.int64)
        st = getattr(self, "_static_inp", None)
        if getattr(self, "_graphs", None) is None or st is None:
            self.inp = inp
            return
        same = set(inp) == set(st) and all((not torch.is_tensor(st[k])) or (torch.is_tensor(inp[k]) and inp[k].shape == st[k].shape
                                                                             and inp[k].dtype == st[k].dtype) for k in st)
        if not same:
            self._graphs = self._static_inp = None
            self.inp = inp
            return
        for k, v in st.items():
            if torch.is_tensor(v):
                if inp[k].data_ptr() != v.data_ptr():
                    v.copy_(inp[k])
            else:
                st[k] = inp[k]
        self.inp = st

    def forward(self):
        """:699-730."""
        i = self.inp
        fake_bg, src_color, src_mask, tsf_color, tsf_mask = self.tg.forward(i["input_G_bg"], i["input_G_src"], i["input_G_tsf"], i["Tst"])
        fake_src_imgs = src_mask * fake_bg + (1 - src_mask) * src_color             # share_bg: (1,1,3,h,w) broadcasts
        fake_tsf_imgs = tsf_mask * fake_bg + (1 - tsf_mask) * tsf_color
        return fake_bg, fake_src_imgs, fake_tsf_imgs, torch.cat([src_mask, tsf_mask], dim=1)

    def optimize_G(self, fake_bg, fake_src_imgs, fake_tsf_imgs, fake_masks):
        """:732-789 (use_vgg None -> crt_tsf = L1; use_face off)."""
        i, o = self.inp, self.opts
        bs, nt, c, h, w = fake_tsf_imgs.shape
        fake_tsf = fake_tsf_imgs.view(bs * nt, c, h, w)
        real_tsf = i["real_tsf"].view(bs * nt, c, h, w)
        loss_adv = 0.0
        if self.D is not None:
            tsf_cond = i["input_G_tsf"][:, :, -3:].reshape(bs * nt, 3, h, w)
            loss_adv = lsgan_loss(self.D(self._d_inputs(torch.cat([fake_tsf, tsf_cond], dim=1))), 0) * o.lambda_D_prob
        loss_rec = (F.l1_loss(fake_src_imgs, i["real_src"]) + F.l1_loss(fake_bg.view(-1, 3, h, w), i["real_bg"])) / 2 * o.lambda_rec
        loss_tsf = (F.l1_loss(fake_tsf, real_tsf) if self.crt_tsf is None else self.crt_tsf(fake_tsf, real_tsf)) * o.lambda_tsf
        loss_face = 0.0
        if self.crt_face is not None:                                     # :775-779
            loss_face = self.crt_face(fake_tsf, real_tsf, bbox1=i["head_bbox"], bbox2=i["head_bbox"]) * o.lambda_face
        fm = fake_masks.view(-1, 1, h, w)
        loss_mask = F.binary_cross_entropy(fm, i["body_mask"].view(-1, 1, h, w)) * o.lambda_mask
        loss_smooth = tv_loss(fm) * o.lambda_mask_smooth
        # detached: a stored loss that still carries its graph keeps every AccumulateGrad node of the step alive; the next step then
        # reuses those nodes ON THE STREAM THEY WERE CREATED ON - inside a hipGraph capture that is work on a non-capturing stream
        # (the capture crashed in hipStreamEndCapture when an eager step had run first)
        det = lambda v: v.detach() if torch.is_tensor(v) else v            # noqa: E731
        self.losses.update(g_rec=det(loss_rec), g_tsf=det(loss_tsf), g_face=det(loss_face), g_adv=det(loss_adv), g_mask=det(loss_mask),
                           g_mask_smooth=det(loss_smooth))
        return loss_rec + loss_tsf + loss_face + loss_adv + loss_mask + loss_smooth

    def _d_inputs(self, x):
        """:755-765 / :808-826: the discriminator's dict (no augmented background in personalization; the body / head boxes feed
        the patch_global_local / patch_global_body_head variants)."""
        return {"x": x, "bg_x": None, "body_rects": self.inp.get("body_bbox"), "head_rects": self.inp.get("head_bbox"), "get_avg": False}

    def optimize_D(self, fake_tsf_imgs):
        """:791-832."""
        i = self.inp
        bs, nt, c, h, w = fake_tsf_imgs.shape
        tsf_cond = i["input_G_tsf"][:, :, -3:].reshape(bs * nt, 3, h, w)
        fake_in = torch.cat([fake_tsf_imgs.detach().view(bs * nt, c, h, w), tsf_cond], dim=1)
        real_in = torch.cat([i["real_tsf"].reshape(bs * nt, c, h, w), tsf_cond], dim=1)
        # real and fake ride through D as ONE batch (its InstanceNorm is per sample, so the logits are those of two separate
        # calls; the deep layers of D have M = 31^2 rows per sample - two samples fill twice the workgroups per launch)
        n = real_in.shape[0]
        both = self._d_inputs(torch.cat([real_in, fake_in], dim=0))
        for k in ("body_rects", "head_rects"):
            if both[k] is not None:
                both[k] = torch.cat([torch.as_tensor(both[k])] * 2, dim=0)
        outs = self.D(both)
        d_real, d_fake = [o[:o.shape[0] // 2] for o in outs], [o[o.shape[0] // 2:] for o in outs]
        assert outs[0].shape[0] == 2 * n
        self.losses.update(d_real=_reduce_outs(d_real).detach(), d_fake=_reduce_outs(d_fake).detach())   # reduce_tensor, multi_scale_dis.py:9-18
        return lsgan_loss(d_real, 1) + lsgan_loss(d_fake, -1)

    def optimize_parameters(self):
        """:326-352, plus the gradient all-reduce when the step is data parallel."""
        on_gpu = torch.cuda.is_available() and next(self.G.parameters()).is_cuda
        if on_gpu and getattr(self.opts, "use_panel_cache", False) and getattr(self, "_panel_cache", None) is None:
            nets = [self.G, self.D, self.crt_tsf, self.crt_face]               # the loss criteria are nn.Modules holding their frozen networks
            self._panel_cache = ops.PanelCache([p for n in nets if n is not None for p in list(n.parameters()) + list(n.buffers())])
        if on_gpu and getattr(self.opts, "branch_streams", False) and getattr(self, "_branch_stream", None) is None:
            self._branch_stream = torch.cuda.Stream()
        prev = ops.PANEL_CACHE, ops.BRANCH_STREAM
        ops.PANEL_CACHE, ops.BRANCH_STREAM = getattr(self, "_panel_cache", None), getattr(self, "_branch_stream", None)
        try:
            with ops.conv_precision(self.opts.conv_precision):
                if self._graphable():
                    return self._graph_step()
                if getattr(self.opts, "dp_schedule", "hooks") == "segmented" and self._multi():
                    self.step_mode = "eager launches, segmented data-parallel schedule"
                    seg, st = self._eager_segments()
                    self._run_dp_schedule(seg)
                    return st["lg"], st.get("ld")
                self.step_mode = "eager launches"
                return self._optimize_parameters()
        finally:
            ops.PANEL_CACHE, ops.BRANCH_STREAM = prev

    def _multi(self):
        """Is this step data parallel?  ``force_dp`` (tests): run the data-parallel schedule in a one-rank group too, so the RCCL calls
        and the stream hand-offs of the N > 1 form execute on a single GPU."""
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or bool(getattr(self, "force_dp", False))

    # ---- the data-parallel schedule of the segmented step (graph replays on the GPU, eager closures in the CPU / gloo test) -------
    def _eager_segments(self):
        st = {}

        def seg_A():
            lg, fake = self._seg_G()
            lg.backward()
            self.optimizer_G._gather()
            st.update(lg=lg.detach(), fake=fake)

        def seg_D():
            ld = self._seg_D(st["fake"])
            ld.backward()
            self.optimizer_D._gather()
            st["ld"] = ld.detach()
        return {"A": seg_A, "D": seg_D, "adamG": self.optimizer_G.step,
                "adamD": None if self.D is None else self.optimizer_D.step}, st

    def _run_dp_schedule(self, seg):
        """N > 1.  A: G forward / losses / backward (gradients complete).  Then, concurrently: G's flat gradient buffer (145 MB) goes to
        RCCL (async: its stream starts where the compute stream stands) and D's own forward / backward runs on a second stream - the
        reference's DDP overlaps its bucketed all-reduce with backward (services/train.py:89-95); here the exchange hides behind the
        discriminator's step, which needs the fake images and D's weights, not G's update.  Adam(G) follows the exchange; D's buffer
        (28 MB) is exchanged while Adam(G) runs; Adam(D) last.  Same values as the reference order (lwg_trainer.py:326-352).
        Exposed exchange time (compute stream idle, waiting for RCCL) is kept as events -> exposed_allreduce_ms()."""
        on_gpu = torch.cuda.is_available() and next(self.G.parameters()).is_cuda
        oG, oD = self.optimizer_G, self.optimizer_D
        ev = {}

        def rec(name, stream=None):
            if on_gpu:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream if stream is not None else torch.cuda.current_stream())
                ev[name] = e
            else:                                           # CPU tensors (gloo tests): the host clock - segments run synchronously, the exchange does not
                ev[name] = time.perf_counter()
        seg["A"]()
        rec("A_done")
        force = bool(getattr(self, "force_dp", False))
        wG = oG.allreduce_async(self.group, n_ranges=4, force=force)
        ds = None
        if self.D is not None:
            if on_gpu:
                if getattr(self, "_d_stream", None) is None:
                    self._d_stream = torch.cuda.Stream()
                ds = self._d_stream
                ds.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(ds):
                    seg["D"]()
                rec("D_done", ds)
            else:
                seg["D"]()
        oG.allreduce_finish(wG, self.group)
        rec("G_reduced")
        wD = None
        if self.D is not None:
            if ds is not None:
                torch.cuda.current_stream().wait_stream(ds)
            wD = oD.allreduce_async(self.group, force=force)
        seg["adamG"]()
        rec("adamG_done")
        if self.D is not None:
            oD.allreduce_finish(wD, self.group)
            rec("D_reduced")
            seg["adamD"]()
        self._dp_events = ev
        self.allreduce_overlap = ("G's gradient all-reduce (4 ranges) behind D's forward / backward, D's behind Adam(G)"
                                  if self.D is not None else "G's gradient all-reduce exposed (no discriminator step to hide it behind)")

    # ---- the step in three segments (the data-parallel exchanges sit between them) ------------------------------------------------
    def _seg_G(self):
        if ops.PANEL_CACHE is not None:
            ops.PANEL_CACHE.refresh()                       # every weight panel of the step (G, D, the frozen loss networks): one launch
        fake_bg, fake_src_imgs, fake_tsf_imgs, fake_masks = self.forward()
        d_params = [] if self.D is None else list(self.D.parameters())
        for p in d_params:                                  # G's adversarial term needs D's data gradients only (the reference
            p.requires_grad_(False)                         # computes, then discards, D's weight gradients here)
        loss_G = self.optimize_G(fake_bg, fake_src_imgs, fake_tsf_imgs, fake_masks)
        for p in d_params:
            p.requires_grad_(True)
        self.optimizer_G.zero_grad()
        return loss_G, fake_tsf_imgs

    def _seg_D(self, fake_tsf_imgs):
        self.optimizer_D.zero_grad()
        return self.optimize_D(fake_tsf_imgs)

    def _optimize_parameters(self):
        loss_G, fake_tsf_imgs = self._seg_G()
        self.optimizer_G.arm(self.group)
        loss_G.backward()
        self.optimizer_G.allreduce(self.group)
        self.optimizer_G.step()
        loss_D = None
        if self.D is not None:
            loss_D = self._seg_D(fake_tsf_imgs)
            self.optimizer_D.arm(self.group, n_buckets=1)
            loss_D.backward()
            self.optimizer_D.allreduce(self.group)
            self.optimizer_D.step()
        return loss_G.detach(), None if loss_D is None else loss_D.detach()

    # ---- hipGraph replay of the step -----------------------------------------------------------------------------------------
    def _graphable(self):
        """The step is captured when its shapes are static and nothing in it reads the device from the host: no crop discriminators (their
        boxes are host-read integers, as in the reference; FaceLoss crops on the device since round 5), CUDA tensors, use_graph on."""
        if not getattr(self.opts, "use_graph", False) or getattr(self, "_graph_failed", False):
            return False
        if not torch.cuda.is_available() or not next(self.G.parameters()).is_cuda:
            return False
        if self.D is not None and getattr(self.D, "CROPS", ()):
            return False
        return True

    def _graph_step(self):
        if getattr(self, "_graphs", None) is None:
            try:
                self._capture()
            except Exception as e:                          # fall back to eager launches, loudly
                import traceback
                import warnings
                where = "".join(traceback.format_tb(e.__traceback__)[-4:])
                warnings.warn(f"LWGTrainer: capturing the step as a hipGraph failed ({type(e).__name__}: {e}); running eager launches.  "
                              f"Raised at:\n{where}")
                self._graph_failed, self._graphs = True, None
                torch.cuda.synchronize()
                self.step_mode = "eager launches (graph capture failed)"
                return self._optimize_parameters()
        if self._captured_lr != (self.optimizer_G.lr, None if self.optimizer_D is None else self.optimizer_D.lr):
            # the learning rate is a kernel argument frozen into the graph: a changed lr needs a new capture
            self._graphs = None
            return self._graph_step()
        multi = self._multi()
        if self._graphs["dp"] != multi:                      # captured for the other form (a process group appeared / went away)
            self._graphs = None
            return self._graph_step()
        gr = self._graphs
        if multi:
            # [A: G fwd / bwd] -> {G's all-reduce on RCCL's stream || [D: D fwd / bwd on the second stream]} -> {Adam(G) || D's all-reduce}
            # -> Adam(D): _run_dp_schedule
            self._run_dp_schedule({"A": gr["A"].replay, "D": None if gr["D"] is None else gr["D"].replay, "adamG": gr["B"].replay,
                                   "adamD": None if gr["C"] is None else gr["C"].replay})
        else:
            gr["A"].replay()
            gr["B"].replay()
            if self.D is not None:
                gr["C"].replay()
        # the replay updated the weights through raw pointers: host-side mirrors follow here (stale inference panels, step counts)
        self.optimizer_G.note_replayed()
        if self.D is not None:
            self.optimizer_D.note_replayed()
        # the static loss tensors are overwritten by the next replay: hand out copies (a caller may keep a history, as eager mode allows)
        lg, ld = self._static_losses
        return lg.clone(), None if ld is None else ld.clone()

    def exposed_allreduce_ms(self):
        """Time the compute stream of the LAST data-parallel step spent waiting for RCCL (synchronizes): G's exchange beyond the end of
        D's forward / backward segment, D's beyond Adam(G).  None when the last step exchanged nothing (one GPU)."""
        dp = getattr(self, "_dp_events", None)
        if not dp:
            return None
        if isinstance(dp["A_done"], float):
            t = lambda a, b: (dp[b] - dp[a]) * 1e3                                       # noqa: E731  (host clock: CPU / gloo runs)
        else:
            torch.cuda.synchronize()
            t = lambda a, b: dp[a].elapsed_time(dp[b])                                   # noqa: E731  (ms from a to b)
        g_exposed = t("A_done", "G_reduced") if "D_done" not in dp else min(t("A_done", "G_reduced"), t("D_done", "G_reduced"))
        d_exposed = t("adamG_done", "D_reduced") if "D_reduced" in dp else 0.0
        return max(0.0, g_exposed) + max(0.0, d_exposed)

    def _capture(self):
        """Warm up on a side stream (every kernel variant launched once: dynamic-LDS attributes are set outside the capture), then
        capture the three segments into graphs that share one memory pool."""
        import gc
        self.losses = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in self.losses.items()}
        gc.collect()                                        # no autograd graph of an earlier (eager) step may outlive this point
        # the warm-up steps are throw-away: the reference does exactly n_iters updates (services/personalization.py:95-151), so the
        # parameters, Adam moments and step counts of G and D are put back afterwards - capture + first replay = ONE update
        opts_ = [o for o in (self.optimizer_G, self.optimizer_D) if o is not None]
        snaps = [o.snapshot() for o in opts_]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._optimize_parameters()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for o, sn in zip(opts_, snaps):
            o.restore(sn)
        torch.cuda.synchronize()
        self.optimizer_G._armed = False                     # no hook-driven collectives inside a capture
        if self.optimizer_D is not None:
            self.optimizer_D._armed = False
        gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        gC = torch.cuda.CUDAGraph() if self.D is not None else None      # no discriminator: no third segment (an empty capture is not a graph)
        gD = None
        multi = self._multi()
        # The discriminator's own step needs this iteration's fake images (detached) and D's weights - not G's update.  Same values as
        # the reference order (lwg_trainer.py:326-352: optimize_G, G step, optimize_D, D step) in both forms:
        #  * one GPU: D's forward / backward runs NEXT TO G's backward inside graph A (which reads D's weights but never writes them or
        #    their gradients: they are frozen while G's adversarial term is built);
        #  * data parallel: D's forward / backward is its OWN graph, replayed on the second stream while RCCL all-reduces G's gradient
        #    buffer (_run_dp_schedule) - the exchange hides behind it instead of sitting exposed between the graphs.
        overlap_d = self.D is not None and bool(getattr(self.opts, "overlap_d_step", False)) and not multi
        loss_D = None
        if self.D is not None and getattr(self, "_d_stream", None) is None:
            self._d_stream = torch.cuda.Stream()
        with torch.cuda.graph(gA):
            loss_G, fake_tsf_imgs = self._seg_G()
            if overlap_d:
                cur = torch.cuda.current_stream()
                self._d_stream.wait_stream(cur)
                with torch.cuda.stream(self._d_stream):
                    loss_D = self._seg_D(fake_tsf_imgs)
                    loss_D.backward()
                    self.optimizer_D._gather()
            loss_G.backward()
            self.optimizer_G._gather()
            if overlap_d:
                cur.wait_stream(self._d_stream)
        pool = gA.pool()
        if multi and self.D is not None:
            gD = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gD, pool=pool, stream=self._d_stream):
                loss_D = self._seg_D(fake_tsf_imgs)
                loss_D.backward()
                self.optimizer_D._gather()
        with torch.cuda.graph(gB, pool=pool):
            self.optimizer_G.step()
            if self.D is not None and not overlap_d and not multi:
                loss_D = self._seg_D(fake_tsf_imgs)
                loss_D.backward()
                self.optimizer_D._gather()
        if gC is not None:
            with torch.cuda.graph(gC, pool=pool):
                self.optimizer_D.step()
        for o, sn in zip(opts_, snaps):                     # capturing step() advanced the host mirrors only; nothing ran on the device
            o.t = sn[4]
        self._graphs = {"A": gA, "D": gD, "B": gB, "C": gC, "dp": multi}
        self._captured_lr = (self.optimizer_G.lr, None if self.optimizer_D is None else self.optimizer_D.lr)
        self._static_losses = (loss_G.detach(), None if loss_D is None else loss_D.detach())
        self._static_inp = self.inp
        if multi:
            self.step_mode = ("4 hipGraph segments per step (G fwd/bwd | D fwd/bwd on a second stream next to G's gradient all-reduce | Adam(G) "
                              "next to D's all-reduce | Adam(D))")
        else:
            self.step_mode = ("3 hipGraph segments per step (G fwd/bwd with D's own fwd/bwd on a second stream | Adam(G) | Adam(D))"
                              if overlap_d else "3 hipGraph segments per step (G fwd/bwd | Adam(G) + D fwd/bwd | Adam(D))")
        torch.cuda.synchronize()


def personalize(trainer, samples, n_iters, ckpt_path=None, log_every=0):
    """services/personalization.py:95-151 without the process / DataLoader / TensorBoard shell: cycle over ``samples`` (dataset
    samples or prepared input dicts) for ``n_iters`` optimisation steps and save ``G.state_dict()`` where ``Imitator`` looks for
    ``personalized_ckpt_path`` (imitator.py:160-168).  Returns the (loss_G, loss_D) history as floats."""
    hist, it = [], 0
    while it < n_iters:
        for sample in samples:
            trainer.set_input(sample)
            lg, ld = trainer.optimize_parameters()
            it += 1
            if log_every and it % log_every == 0:
                hist.append((float(lg), None if ld is None else float(ld)))
            if it >= n_iters:
                break
    if ckpt_path:
        torch.save({k: v.detach().cpu() for k, v in trainer.G.state_dict().items()}, ckpt_path)
    return hist
