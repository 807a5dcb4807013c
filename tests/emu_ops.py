"""CPU emulation of the C-ABI ops' *contracts* (include/lwg_hip.h) - TEST INFRASTRUCTURE ONLY.

Lets the `-m "not gpu"` suite exercise the host logic (weight packing, launch orchestration, layouts, the
generator/renderer/runner classes) without a GPU by monkeypatching ``ipercore_amd.ops``.  It follows the
kernels' documented semantics (packed panels, tap tables, epilogues), NOT the oracle, so a packing or
orchestration bug shows up as a mismatch against the oracle.  Never imported by the product.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from ipercore_amd import ops as real_ops

ACT = {0: lambda v: v, 1: torch.relu, 2: torch.tanh, 3: torch.sigmoid, 4: lambda v: F.leaky_relu(v, 0.2)}


def _unpanel(w, ntaps, cin):
    """Packed panel [K/4][N][4] (include/lwg_hip.h: chunk-major / tap-minor K when Cin % 32 == 0) -> (K, N) with the
    logical k = tap * Cin + c."""
    K4, N, _ = w.shape
    wk = w.permute(0, 2, 1).reshape(K4 * 4, N)
    if cin % 32 == 0:
        wk = wk[:ntaps * cin].view(cin // 32, ntaps, 32, N).permute(1, 0, 2, 3).reshape(ntaps * cin, N)
    return wk


def conv2d(x0, spec, y, x1=None, epi=0, act=0, res=None, xn=None, mean=None, rstd=None, out_hw=None, ycoff=0, splitk=False, q4=False):
    if q4:      # LWG_DT_F32_Q4: y (B, YC/4, YH, YW, 4) channel-quad planes: run the launch on the NHWC image of y and write it back
        assert epi == 0 and not splitk and y.dim() == 5 and y.shape[4] == 4
        B_, Cq, YH_, YW_, _ = y.shape
        t = y.permute(0, 2, 3, 1, 4).reshape(B_, YH_, YW_, 4 * Cq).clone()
        conv2d(x0, spec, t, x1, epi, act, res, xn, mean, rstd, out_hw, ycoff, False)
        y.copy_(t.view(B_, YH_, YW_, Cq, 4).permute(0, 3, 1, 2, 4))
        return y
    x = x0 if x1 is None else torch.cat([x0, x1], dim=3)
    B, H, W, Cin = x.shape
    assert Cin == spec.Cin
    YB, YH, YW, YC = y.shape
    # the argument contract lwg_conv2d_nhwc_f32 enforces on the host (csrc/conv_igemm.hip, hipErrorInvalidValue otherwise)
    assert spec.N % 64 == 0 and Cin % 4 == 0 and YC % 4 == 0 and ycoff % 4 == 0, (spec.N, Cin, YC, ycoff)
    if Cin % 32:
        assert x1 is None and Cin <= 16 and Cin & (Cin - 1) == 0 and epi == 0, Cin
    elif x1 is not None:
        assert x0.shape[3] % 32 == 0
    if out_hw is None:
        OH, OW = (YH, YW) if spec.omul == 1 else (YH // spec.omul, YW // spec.omul)
    else:
        OH, OW = out_hw
    Wk = _unpanel(spec.w, spec.ntaps, Cin)
    assert Wk.shape[0] >= spec.ntaps * Cin and Wk.shape[1] == spec.N
    oy = torch.arange(OH) * spec.stride
    ox = torch.arange(OW) * spec.stride
    acc = torch.zeros(B, OH, OW, spec.N)
    for t in range(spec.ntaps):
        iy, ix = oy + spec.dy[t], ox + spec.dx[t]
        vy, vx = (iy >= 0) & (iy < H), (ix >= 0) & (ix < W)
        g = x[:, iy.clamp(0, H - 1)][:, :, ix.clamp(0, W - 1)]
        g = g * (vy[:, None] & vx[None, :]).float()[None, :, :, None]
        acc += g.reshape(-1, Cin) .matmul(Wk[t * Cin:(t + 1) * Cin]).view(B, OH, OW, spec.N)
    ys = slice(spec.ooy, None, spec.omul) if spec.omul > 1 else slice(None)
    xs = slice(spec.oox, None, spec.omul) if spec.omul > 1 else slice(None)
    if epi == real_ops.EPI_SPADE:
        C = spec.N // 2
        assert YC == C and spec.omul == 1
        v = (acc + spec.bias).view(B, OH, OW, C // 32, 2, 32)
        gamma, beta = v[..., 0, :].reshape(B, OH, OW, C), v[..., 1, :].reshape(B, OH, OW, C)
        out = (xn - mean[:, None, None, :]) * rstd[:, None, None, :] * (1 + gamma) + beta
        y.copy_(ACT[act](out))
        return y
    if spec.bias is not None:
        acc = acc + spec.bias
    if epi == real_ops.EPI_RESIDUAL and act == real_ops.ACT_RELU_MASK:       # the ReLU backward of the producer of the forward input
        y[:, ys, xs, ycoff:ycoff + spec.N] = acc * (res[:, ys, xs, ycoff:ycoff + spec.N] > 0).float()
        return y
    if epi == real_ops.EPI_RESIDUAL:
        acc = acc + res[:, ys, xs, ycoff:ycoff + spec.N]
    y[:, ys, xs, ycoff:ycoff + spec.N] = ACT[act](acc)
    return y


def conv_transpose2d(x, specs, y, act=0, splitk=False, out_hw=None, q4=False):
    """ops.conv_transpose2d's contract: ConvTranspose2d(4, 2, 1) given its four parity specs = the four parity launches."""
    for s in specs:
        conv2d(x, s, y, act=act, splitk=splitk, out_hw=None if out_hw is None else out_hw(s), q4=q4)
    return y


def conv2d_wgrad(x0, spec, dy, x1=None, out_hw=None, ycoff=0):
    """CPU emulation of lwg_conv2d_wgrad_nhwc_f32: dW (ntaps*Cin, N) in the panel's K order (include/lwg_hip.h)."""
    x = x0 if x1 is None else torch.cat([x0, x1], dim=3)
    B, H, W, Cin = x.shape
    YB, YH, YW, YC = dy.shape
    assert spec.N % 4 == 0 and Cin % 4 == 0 and YC % 4 == 0 and ycoff % 4 == 0       # lwg_conv2d_wgrad_nhwc_f32's host contract
    if Cin % 32:
        assert x1 is None and Cin <= 16 and Cin & (Cin - 1) == 0, Cin
    elif x1 is not None:
        assert x0.shape[3] % 32 == 0
    OH, OW = out_hw if out_hw is not None else ((YH, YW) if spec.omul == 1 else (YH // spec.omul, YW // spec.omul))
    ys = slice(spec.ooy, None, spec.omul) if spec.omul > 1 else slice(None)
    xs = slice(spec.oox, None, spec.omul) if spec.omul > 1 else slice(None)
    g = dy[:, ys, xs, ycoff:ycoff + spec.N][:, :OH, :OW].reshape(-1, spec.N)
    oy, ox = torch.arange(OH) * spec.stride, torch.arange(OW) * spec.stride
    rows = []
    for t in range(spec.ntaps):
        iy, ix = oy + spec.dy[t], ox + spec.dx[t]
        vy, vx = (iy >= 0) & (iy < H), (ix >= 0) & (ix < W)
        a = x[:, iy.clamp(0, H - 1)][:, :, ix.clamp(0, W - 1)] * (vy[:, None] & vx[None, :]).float()[None, :, :, None]
        rows.append(a.reshape(-1, Cin).t().matmul(g))                      # (Cin, N) for tap t
    dw = torch.stack(rows, dim=0)                                          # (ntaps, Cin, N), tap-major
    if Cin % 32 == 0:
        dw = dw.view(spec.ntaps, Cin // 32, 32, spec.N).permute(1, 0, 2, 3)
    return dw.reshape(spec.ntaps * Cin, spec.N).contiguous()


def act_bwd(dy, y, act):
    if act == real_ops.ACT_RELU:
        return dy * (y > 0)
    if act == real_ops.ACT_LRELU:
        return dy * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
    if act == real_ops.ACT_TANH:
        return dy * (1 - y * y)
    if act == real_ops.ACT_SIGMOID:
        return dy * y * (1 - y)
    return dy


def colsum(x):
    return x.reshape(-1, x.shape[-1]).sum(dim=0)


def instnorm_stats(x, mean, rstd, ws, eps=1e-5, nsplit=None):
    B, H, W, C = x.shape
    v = x.reshape(B, H * W, C)
    mean.copy_(v.mean(dim=1))
    rstd.copy_(1.0 / torch.sqrt(v.var(dim=1, unbiased=False) + eps))


def instnorm_apply(x, mean, rstd, y, act=0, res=None):
    out = ACT[act]((x - mean[:, None, None, :]) * rstd[:, None, None, :])
    if res is not None:
        out = out + res
    y.copy_(out)
    return y


def flow_resize(T, h, w):
    """lwg_flow_resize_f32 = LWB.resize_trans: F.interpolate(bilinear, align_corners=True) of every (S,S,2) flow field."""
    B, ns, S = T.shape[0], T.shape[1], T.shape[2]
    Tf = F.interpolate(T.reshape(B * ns, S, S, 2).permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=True)
    return Tf.permute(0, 2, 3, 1).reshape(B, ns, h, w, 2).contiguous()


def lwb_attention(q, Ks, Vs, bk, bv, T, out, src_batched=False, _differentiable=False):
    B, h, w, C = q.shape
    ns = T.shape[1]
    Tf = T.reshape(B * ns, T.shape[2], T.shape[3], 2)
    if (T.shape[2], T.shape[3]) != (h, w):
        Tf = F.interpolate(Tf.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    if src_batched:
        Kn, Vn = Ks, Vs
    else:
        Kn, Vn = Ks.repeat(B, 1, 1, 1), Vs.repeat(B, 1, 1, 1)
    Kw = F.grid_sample(Kn.permute(0, 3, 1, 2), Tf, mode="bilinear", padding_mode="zeros", align_corners=False)
    Vw = F.grid_sample(Vn.permute(0, 3, 1, 2), Tf, mode="bilinear", padding_mode="zeros", align_corners=False)
    Kw = Kw.view(B, ns, C, h, w) + bk.view(1, 1, C, 1, 1)
    Vw = Vw.view(B, ns, C, h, w) + bv.view(1, 1, C, 1, 1)
    logits = (Kw * q.permute(0, 3, 1, 2).unsqueeze(1)).sum(dim=2, keepdim=True) / math.sqrt(C)
    a = torch.softmax(logits, dim=1)
    res = (a * Vw).sum(dim=1).permute(0, 2, 3, 1)
    if _differentiable:
        return res
    out.copy_(res)
    return out


def lwb_attention_x(x, Kq, kappa, Vs, bv, T, out, stats=None, src_batched=False):
    """lwg_lwb_attention_x_*'s contract (include/lwg_hip.h): logit_s = (warp_s(Kq) . x + warp_s(kappa)) / sqrt(C),
    out = sum_s softmax_s(logit) warp_s(Vs) + bv; stats: per 8 x 8 tile the (count, mean, M2) record of x."""
    B, h, w, C = x.shape
    ns = T.shape[1]
    assert tuple(T.shape) == (B, ns, h, w, 2)
    Tf = T.reshape(B * ns, h, w, 2).to(x.dtype if x.dtype == torch.float64 else torch.float32)
    rep = (lambda t: t) if src_batched else (lambda t: t.repeat(B, *([1] * (t.dim() - 1))))
    cdt = Tf.dtype
    Kw = F.grid_sample(rep(Kq).to(cdt).permute(0, 3, 1, 2), Tf, mode="bilinear", padding_mode="zeros", align_corners=False)
    Vw = F.grid_sample(rep(Vs).to(cdt).permute(0, 3, 1, 2), Tf, mode="bilinear", padding_mode="zeros", align_corners=False)
    aw = F.grid_sample(rep(kappa).to(cdt).unsqueeze(1), Tf, mode="bilinear", padding_mode="zeros", align_corners=False)
    Kw, Vw, aw = Kw.view(B, ns, C, h, w), Vw.view(B, ns, C, h, w), aw.view(B, ns, 1, h, w)
    logits = ((Kw * x.to(cdt).permute(0, 3, 1, 2).unsqueeze(1)).sum(dim=2, keepdim=True) + aw) / math.sqrt(C)
    a = torch.softmax(logits, dim=1)
    res = (a * Vw).sum(dim=1).permute(0, 2, 3, 1) + bv.to(cdt).view(1, 1, 1, C)
    out.copy_(res.to(out.dtype))
    if stats is not None:
        ty, tx = (h + 7) // 8, (w + 7) // 8
        npart = attn_records(h, w, C, x.dtype) // (ty * tx)
        rec = stats[:B * ty * tx * npart * C * 3].view(B, ty * tx, npart, C, 3)
        xf = x.float()
        per = 64 // npart                                   # consecutive pixels of the tile (row-major 8 x 8) per workgroup
        for i in range(ty):
            for j in range(tx):
                for q in range(npart):
                    idx = [k for k in range(q * per, (q + 1) * per) if i * 8 + k // 8 < h and j * 8 + k % 8 < w]
                    if not idx:
                        rec[:, i * tx + j, q, :, 0] = 0
                        rec[:, i * tx + j, q, :, 1:] = 0
                        continue
                    blk = torch.stack([xf[:, i * 8 + k // 8, j * 8 + k % 8, :] for k in idx], dim=1)
                    mu = blk.mean(dim=1)
                    rec[:, i * tx + j, q, :, 0] = len(idx)
                    rec[:, i * tx + j, q, :, 1] = mu
                    rec[:, i * tx + j, q, :, 2] = ((blk - mu[:, None, :]) ** 2).sum(dim=1)
    return out


def instnorm_finalize(ws, B, C, nrec, mean, rstd, eps=1e-5):
    """lwg_instnorm_finalize_f32's contract: merge the (count, mean, M2) records of an image (exactly, in fp64)."""
    rec = ws[:B * nrec * C * 3].view(B, nrec, C, 3).double()
    n, mu, m2 = rec[..., 0], rec[..., 1], rec[..., 2]
    tot = n.sum(dim=1)
    gm = (n * mu).sum(dim=1) / tot
    var = (m2 + n * (mu - gm[:, None, :]) ** 2).sum(dim=1) / tot          # zero-count records (parts beyond the image edge) drop out
    mean.copy_(gm.float())
    rstd.copy_((1.0 / torch.sqrt(var + eps)).float())


def attn_records(h, w, C, dtype=torch.float32):
    """lwg_lwb_attention_x_records: one record per 8 x 8 tile (the product build runs one workgroup per tile and frame)."""
    return ((h + 7) // 8) * ((w + 7) // 8)


def instnorm_finalize_ws(B, C, nrec):
    return B * nrec * C * 3 + (B * ((nrec + 255) // 256) * C * 3 if nrec > 512 else 0)


def lwb_fuse(tsf_x, src_x, T, out, gate=None, scale_w=1.0, scale_o=1.0, src_batched=False):
    B, h, w, C = tsf_x.shape
    ns, S = T.shape[1], T.shape[2]
    Tf = T.reshape(B * ns, S, S, 2)
    if S != h:
        Tf = F.interpolate(Tf.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    Sn = src_x if src_batched else src_x.repeat(B, 1, 1, 1)
    warp = F.grid_sample(Sn.permute(0, 3, 1, 2), Tf, mode="bilinear", padding_mode="zeros", align_corners=False)
    fused = warp.view(B, ns, C, h, w).sum(dim=1).permute(0, 2, 3, 1) * scale_w
    out.copy_((tsf_x + (fused if gate is None else gate * fused)) * scale_o)
    return out


def head_compose(x, wpk, bg, want_pred=True, want_mask=True, want_img=False, q4=False):
    if q4:      # x (B, C/4, S, S, 4) channel-quad planes
        x = x.permute(0, 2, 3, 1, 4).reshape(x.shape[0], x.shape[2], x.shape[3], -1)
    B, S, _, C = x.shape
    w = wpk.view(5, 5, C, 4).permute(3, 2, 0, 1)
    o = F.conv2d(x.permute(0, 3, 1, 2), w, None, padding=2)
    img, mask = torch.tanh(o[:, 0:3]), torch.sigmoid(o[:, 3:4])
    pred = mask * bg + (1 - mask) * img if (want_pred and bg is not None) else None
    return pred, (mask if want_mask else None), (img if want_img else None)


def thin_conv(x, wpk, ks):
    """lwg_thin_conv_f32's contract: x (B,S,S,C), wpk (ks*ks, C, 4) -> (B,S,S,4), stride 1, pad ks // 2, no bias."""
    B, S, _, C = x.shape
    assert ks in (5, 7) and C % 8 == 0 and tuple(wpk.shape) == (ks * ks, C, 4)
    w = wpk.view(ks, ks, C, 4).permute(3, 2, 0, 1)
    return F.conv2d(x.permute(0, 3, 1, 2), w, None, padding=ks // 2).permute(0, 2, 3, 1).contiguous()


def nchw_to_nhwc(x, c_pad=None):
    B, C, H, W = x.shape
    Cp = C if c_pad is None else c_pad
    y = torch.zeros(B, H, W, Cp)
    y[..., :C] = x.permute(0, 2, 3, 1)
    return y


def nhwc_to_nchw(x, channels=None):
    C = x.shape[3] if channels is None else channels
    return x[..., :C].permute(0, 3, 1, 2).contiguous()


# ---- renderer / flow / body ops follow the ABI text, implemented with the same formulas as the kernels ----
def project_faces(verts, cam, faces, want_faces_v=True, want_f2pts=True):
    s, t = cam[:, 0].view(-1, 1, 1), cam[:, 1:3].view(-1, 1, 2)
    xy = s * (verts[:, :, :2] + t)
    v = verts[:, faces.long()]
    pxy = xy[:, faces.long()]
    fv = torch.stack([pxy[..., 0], -pxy[..., 1], v[..., 2] + np.float32(real_ops.EYE_DIST)], dim=-1) if want_faces_v else None
    return fv, (pxy.clone() if want_f2pts else None)


def rasterize_fim_wim(faces_v, image_size, near=0.1, far=100.0):
    from oracle import lwg_oracle as orc          # the emulator borrows the C rasterizer: same spec by construction
    return orc.rasterize_fim_wim(faces_v.numpy(), image_size, near, far)


def bc_transform(f2pts, fim, wim):
    B, S, _ = fim.shape
    T = torch.full((B, S, S, 2), -2.0)
    for b in range(B):
        on = fim[b] >= 0
        idx = fim[b][on].long()
        T[b][on] = (f2pts[b][idx] * wim[b][on][:, :, None]).sum(dim=1)
    return T


def encode_fim(fim, map_fn):
    return map_fn[fim.long()].permute(0, 3, 1, 2).contiguous()


def flow_compose(fim, wim, map_fn, f_uvs2img, uv_img4, src_f2pts, want_cond=False, want_tuv=False):
    B, S, _ = fim.shape
    ns = src_f2pts.shape[0]
    cond = encode_fim(fim, map_fn)
    tuv = bc_transform(f_uvs2img.unsqueeze(0).expand(B, -1, -1, -1), fim, wim)
    uv = uv_img4[..., :3].permute(2, 0, 1).unsqueeze(0).expand(B, -1, -1, -1)
    syn = F.grid_sample(uv, tuv, mode="bilinear", padding_mode="zeros", align_corners=False)
    tsf = torch.zeros(B, S, S, 8)
    tsf[..., 0:3] = syn.permute(0, 2, 3, 1)
    tsf[..., 3:6] = cond.permute(0, 2, 3, 1)
    Tst = torch.zeros(B, 0, S, S, 2) if ns == 0 else torch.stack(
        [bc_transform(src_f2pts[s:s + 1].expand(B, -1, -1, -1), fim, wim) for s in range(ns)], dim=1)
    return tsf, Tst, (cond if want_cond else None), (tuv if want_tuv else None)


def smpl_lbs(model, pose, beta, cam, offsets=None, links=None):
    from oracle import lwg_oracle as orc
    vt = model["v_template"] if offsets is None else model["v_template"] + offsets
    verts, j3d = orc.lbs(beta, pose, vt, model["shapedirs"], model["posedirs"], model["J_regressor"],
                         model["parents"].long(), model["lbs_weights"])
    if links is not None:
        out = verts.clone()
        out[:, links[:, 0].long()] = verts[:, links[:, 1].long()]
        verts = out
    j2d = cam[:, None, 0:1] * (j3d[:, :, :2] + cam[:, None, 1:3]) if cam is not None else None
    return verts, j3d, j2d


def _korder(ntaps, cin_pad):
    """k index of (tap, c) in the kernel's K order (include/lwg_hip.h, lwg_pack_panel_f32)."""
    tap = torch.arange(ntaps).view(-1, 1)
    c = torch.arange(cin_pad).view(1, -1)
    if cin_pad % 32 == 0:
        return ((c // 32) * ntaps + tap) * 32 + c % 32
    return tap * cin_pad + c


def pack_panel(w, transposed, kidx, cin, cin_pad, nout, n_pad):
    w = w.detach().float()
    D0, D1, KH, KW = w.shape
    kidx = list(kidx)
    ntaps = len(kidx)
    Kp = (ntaps * cin_pad + 31) // 32 * 32
    flat = torch.zeros(Kp, n_pad)
    wk = w.reshape(D0, D1, KH * KW)[:, :, kidx]                                   # (D0, D1, ntaps)
    m = wk.permute(2, 0, 1) if transposed else wk.permute(2, 1, 0)                # (ntaps, c, n)
    k = _korder(ntaps, cin_pad)[:, :cin]
    flat[k.reshape(-1), :nout] = m[:, :cin, :nout].reshape(ntaps * cin, nout)
    return flat.view(Kp // 4, 4, n_pad).permute(0, 2, 1).contiguous()


def unpack_wgrad(dwk, dw, transposed, kidx, cin, cin_pad, nout):
    D0, D1, KH, KW = dw.shape
    kidx = list(kidx)
    ntaps = len(kidx)
    k = _korder(ntaps, cin_pad)[:, :cin]
    m = dwk[k.reshape(-1), :nout].view(ntaps, cin, nout)                          # (tap, c, n)
    flat = dw.view(D0, D1, KH * KW)
    flat[:cin if transposed else nout, :nout if transposed else cin][:, :, kidx] = m.permute(1, 2, 0) if transposed else m.permute(2, 1, 0)
    return dw


def _norm_ref(x, mean, rstd, gamma, beta, act):
    xn = (x - mean[:, None, None, :]) * rstd[:, None, None, :]
    if gamma is not None:
        xn = xn * (1 + gamma) + beta
    return ACT[act](xn)


def norm_fwd(x, gamma=None, beta=None, act=0, eps=1e-5, gb=None):
    """lwg_instnorm_stats + lwg_norm_fwd_nhwc_f32: y = act(IN(x) * (1 + gamma) + beta), biased variance.  gb = gamma | beta fused."""
    if gb is not None:
        C = x.shape[3]
        gamma, beta = gb[..., :C], gb[..., C:]
    v = x.reshape(x.shape[0], -1, x.shape[3])
    mean, rstd = v.mean(1), 1.0 / torch.sqrt(v.var(1, unbiased=False) + eps)
    return _norm_ref(x, mean, rstd, gamma, beta, act), mean, rstd


def norm_bwd(dy, y, x, mean, rstd, gamma=None, act=0, gb=None):
    """lwg_norm_bwd_nhwc_f32 by torch autograd through the same formula (statistics are functions of x)."""
    if gb is not None:
        C = x.shape[3]
        dx, dg, db = norm_bwd(dy, y, x, mean, rstd, gb[..., :C].contiguous(), act)
        return dx, torch.cat([dg, db], dim=3), None
    with torch.enable_grad():
        xr = x.detach().clone().requires_grad_(True)
        gr = None if gamma is None else gamma.detach().clone().requires_grad_(True)
        br = None if gamma is None else torch.zeros_like(gamma).requires_grad_(True)
        v = xr.reshape(xr.shape[0], -1, xr.shape[3])
        eps = 1e-5
        out = _norm_ref(xr, v.mean(1), 1.0 / torch.sqrt(v.var(1, unbiased=False) + eps), gr, br, act)
        gs = torch.autograd.grad(out, [xr] if gamma is None else [xr, gr, br], dy)
    return (gs[0], None, None) if gamma is None else tuple(gs)


def lwb_attention_bwd(q, Ks, Vs, bk, bv, T, dout, src_batched=False):
    """lwg_lwb_attention_bwd_f32 by torch autograd through the emulated forward."""
    with torch.enable_grad():
        qr, kr, vr = (t.detach().clone().requires_grad_(True) for t in (q, Ks, Vs))
        out = lwb_attention(qr, kr, vr, bk, bv, T, torch.empty_like(q), src_batched=src_batched, _differentiable=True)
        return torch.autograd.grad(out, [qr, kr, vr], dout)


def lwb_attention_kv(q, kv, bk, bv, T, out, src_batched=False):
    C = q.shape[3]
    return lwb_attention(q, kv[..., :C].contiguous(), kv[..., C:].contiguous(), bk, bv, T, out, src_batched=src_batched)


def lwb_attention_kv_bwd(q, kv, bk, bv, T, dout, src_batched=False):
    C = q.shape[3]
    dq, dk, dv = lwb_attention_bwd(q, kv[..., :C].contiguous(), kv[..., C:].contiguous(), bk, bv, T, dout, src_batched=src_batched)
    return dq, torch.cat([dk, dv], dim=3)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, t):
    """lwg_adam_step_f32 = torch.optim.Adam's update (no weight decay, no amsgrad)."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    step = lr / (1 - beta1 ** t)
    p.addcdiv_(m, (v / (1 - beta2 ** t)).sqrt() + eps, value=-step)


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, t_dev):
    """lwg_adam_step_dev_f32: the step count lives in a (1,) int32 tensor that the call increments first."""
    t_dev += 1
    adam_step(p, g, m, v, lr, beta1, beta2, eps, int(t_dev.item()))


def maxpool2_fwd(x):
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous()


def maxpool2_bwd(x, dy):
    with torch.enable_grad():
        xr = x.detach().clone().requires_grad_(True)
        return torch.autograd.grad(F.max_pool2d(xr.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1), xr, dy)[0]


def conv2d_wgrad_unpacked(x0, spec, dy, dw, transposed, kidx, cin, nout, x1=None, out_hw=None, ycoff=0, db=None):
    if db is not None:      # the launch's own rows only (out_hw / ycoff select them); dense launches: every row of dy
        assert out_hw is None and ycoff == 0
        db.copy_(colsum(dy)[:nout])
    return unpack_wgrad(conv2d_wgrad(x0, spec, dy, x1=x1, out_hw=out_hw, ycoff=ycoff), dw, transposed, kidx, cin, spec.Cin, nout)


def frames_to_u8(pred, bgr=False):
    """lwg_frames_to_u8's contract: save_cv2_img(normalize=True) arithmetic (cv_utils.py:111-113) -> (B,S,S,3) uint8."""
    import numpy as np
    a = np.transpose(pred.detach().cpu().numpy().astype(np.float32), (0, 2, 3, 1))
    u8 = ((a + np.float32(1)) / np.float32(2.0) * np.float32(255)).astype(np.uint8)
    return torch.from_numpy(np.ascontiguousarray(u8[..., ::-1] if bgr else u8))


def texture_sample(fim, wim, faces_v, textures, eps=1e-3, background_color=(0.0, 0.0, 0.0)):
    """lwg_texture_sample_f32's contract (include/lwg_hip.h): textures (B | 1, nf, T, T, T, 3) -> rgb (B,S,S,3)."""
    from oracle import lwg_oracle as orc
    B = fim.shape[0]
    tex = textures if textures.shape[0] == B else textures.expand(B, *textures.shape[1:])
    return orc.texture_sample(fim, wim, faces_v.float(), tex.float(), eps, tuple(float(c) for c in background_color))


def grid_sample(x, grid):
    """lwg_grid_sample_nchw_f32's contract: F.grid_sample(bilinear, zeros, align_corners=False)."""
    return torch.nn.functional.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def winograd_panel(spec):
    """lwg_winograd_panel_f32's contract in torch (fp64, rounded once): Upk[16][Cin/8][2][N][4] from the fp32 GEMM panel of a 3x3 ConvSpec."""
    K4, N, _ = spec.w.shape
    cin, nt = spec.Cin, spec.ntaps
    assert nt == 9 and cin % 32 == 0 and K4 * 4 == nt * cin
    wk = spec.w.permute(0, 2, 1).reshape(K4 * 4, N)                                    # k' = ((c // 32) * ntaps + tap) * 32 + c % 32
    w = wk.view(cin // 32, nt, 32, N).permute(1, 0, 2, 3).reshape(nt, cin, N).double()     # [tap][c][n]
    g = w.new_zeros(3, 3, cin, N)
    for t in range(nt):
        g[spec.dy[t] + 1, spec.dx[t] + 1] = w[t]
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    U = torch.einsum("ij,jkcn,lk->ilcn", G, g, G).reshape(16, cin, N)
    return U.view(16, cin // 8, 4, 2, N).permute(0, 1, 3, 4, 2).contiguous().float()


W4_G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
W4_BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
W4_AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]


def winograd4_product(q, j):
    """(xi, nu) of product j (0..8) of wave set q (0..3) of csrc/conv_winograd4.hip: the row xi = q, then three products of row 4 + q // 2."""
    return (q, j) if j < 6 else (4 + q // 2, 3 * (q % 2) + j - 6)


def winograd4_panel(spec):
    """lwg_winograd4_panel_f32's contract in torch (fp64, rounded once): Upk[4][Cin/8][4][2][9 N] from the fp32 GEMM panel of a 3x3 ConvSpec - per block
    (q, s, kk, kh): [N][4] products 0-3, [N][4] products 4-7, [N] product 8."""
    K4, N, _ = spec.w.shape
    cin, nt = spec.Cin, spec.ntaps
    assert nt == 9 and cin % 32 == 0 and K4 * 4 == nt * cin
    wk = spec.w.permute(0, 2, 1).reshape(K4 * 4, N)
    w = wk.view(cin // 32, nt, 32, N).permute(1, 0, 2, 3).reshape(nt, cin, N).double()     # [tap][c][n]
    g = w.new_zeros(3, 3, cin, N)
    for t in range(nt):
        g[spec.dy[t] + 1, spec.dx[t] + 1] = w[t]
    G = torch.tensor(W4_G, dtype=torch.float64)
    U = torch.einsum("ij,jkcn,lk->ilcn", G, g, G)                                        # [xi][nu][c][n]
    out = torch.zeros(4, cin // 8, 4, 2, 9 * N, dtype=torch.float64)
    for q in range(4):
        for j in range(9):
            xi, nu = winograd4_product(q, j)
            u = U[xi, nu].view(cin // 8, 4, 2, N)                                          # c = 8 s + 2 kk + kh
            if j < 8:
                out[q, :, :, :, (j // 4) * 4 * N + (j % 4):(j // 4 + 1) * 4 * N:4] = u
            else:
                out[q, :, :, :, 8 * N:] = u
    return out.float()


def winograd4_panel_products(panel):
    """[q][j][c][n] view of a panel (c = 8 s + 2 kk + kh)."""
    _, ns, _, _, n9 = panel.shape
    N = n9 // 9
    ab = panel[..., :8 * N].reshape(4, ns, 4, 2, 2, N, 4).permute(0, 4, 6, 1, 2, 3, 5).reshape(4, 8, ns * 8, N)
    c8 = panel[..., 8 * N:].reshape(4, 1, ns * 8, N)
    return torch.cat([ab, c8], 1)


def winograd4_conv(x, panel, bias=None):
    """The F(4x4, 3x3) algorithm of csrc/conv_winograd4.hip restated around the panel (NHWC fp32 in, NHWC out; fp64 arithmetic): V = B^T d B per 6 x 6 patch
    (stride 4, halo origin -1), 36 channel contractions with the panel's products, the nu fold per wave set (whole rows 0..3, half rows 4 / 5 as three
    partial sums each), the reader's reconstruction of rows 4 / 5 and the xi fold - the bias enters as the start value of product (1, 1)."""
    B, H, W, C = x.shape
    U = winograd4_panel_products(panel.double())                                           # [q][j][c][n]
    N = U.shape[3]
    BT, AT = torch.tensor(W4_BT, dtype=torch.float64), torch.tensor(W4_AT, dtype=torch.float64)
    ph, pw = -(-H // 4), -(-W // 4)
    xp = F.pad(x.double().permute(0, 3, 1, 2), [1, 4 * pw - W + 1, 1, 4 * ph - H + 1])
    d = xp.unfold(2, 6, 4).unfold(3, 6, 4)                                                 # (B, C, ph, pw, 6, 6)
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", BT, d, BT)
    M = x.new_zeros(6, 6, B, ph, pw, N, dtype=torch.float64)
    for q in range(4):
        for j in range(9):
            xi, nu = winograd4_product(q, j)
            M[xi, nu] = torch.einsum("bcyx,cn->byxn", V[..., xi, nu], U[q, j])
    if bias is not None:
        M[1, 1] += bias.double()
    planes = {}
    for q in range(4):
        m = M[q]
        s12, d12, s34, d34 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
        for b_, f in enumerate(((m[0] + s12) + s34, 2 * d34 + d12, 4 * s34 + s12, 8 * d34 + d12 + m[5])):
            planes[4 * q + b_] = f
        h = M[4 + q // 2][3 * (q % 2):3 * (q % 2) + 3]
        P = (h[0] + h[1], h[0] - h[1], h[2]) if q % 2 else ((h[0] + h[1]) + h[2], h[1] - h[2], h[1] + h[2])
        for i in range(3):
            planes[16 + 3 * q + i] = P[i]
    Y = x.new_zeros(4, 4, B, ph, pw, N, dtype=torch.float64)
    for rb in range(4):
        Fs = [planes[4 * xi + rb] for xi in range(4)]
        ia, ib, cb = (0 if rb == 0 else 2 if rb == 2 else 1), (4 if rb & 1 else 3), float(1 << rb)
        for r in range(2):
            f = planes[16 + 6 * r + ia] + cb * planes[16 + 6 * r + ib]
            Fs.append(f + planes[16 + 6 * r + 5] if rb == 3 else f)
        s12, d12, s34, d34 = Fs[1] + Fs[2], Fs[1] - Fs[2], Fs[3] + Fs[4], Fs[3] - Fs[4]
        for a_, yv in enumerate(((Fs[0] + s12) + s34, 2 * d34 + d12, 4 * s34 + s12, 8 * d34 + d12 + Fs[5])):
            Y[a_, rb] = yv
    return Y.permute(2, 3, 0, 4, 1, 5).reshape(B, 4 * ph, 4 * pw, N)[:, :H, :W]


def _crop_ref(x, box, out_hw):
    """lwg_crop_resize_bilinear_f32's contract in the reference's own formulation (faceloss.py:384-406): per-sample slice + F.interpolate."""
    N = x.shape[0]
    ys, valid = [], []
    for i, (x0, x1, y0, y1) in enumerate(box.tolist()):
        # Python slice semantics of the reference's imgs[i, :, y0:y1, x0:x1]: a stop beyond the image is clamped (sample kept); an empty slice
        # (the reference's F.interpolate would raise) and negative, i.e. wrapping, starts are dropped
        ok = x0 != x1 and y0 != y1 and 0 <= x0 < min(x1, x.shape[3]) and 0 <= y0 < min(y1, x.shape[2])
        valid.append(1.0 if ok else 0.0)
        ys.append(F.interpolate(x[i:i + 1, :, y0:y1, x0:x1], size=tuple(out_hw), mode="bilinear", align_corners=True) if ok
                  else x.new_zeros(1, x.shape[1], *out_hw))
    return torch.cat(ys, dim=0), x.new_tensor(valid)


def crop_resize(x, box, out_hw, want_valid=True):
    with torch.no_grad():
        y, valid = _crop_ref(x.detach(), box, out_hw)
    return y, (valid if want_valid else None)


def crop_resize_bwd(dy, box, in_hw):
    x = torch.zeros(dy.shape[0], dy.shape[1], *in_hw, requires_grad=True)
    with torch.enable_grad():
        y, _ = _crop_ref(x, box, dy.shape[2:])
        (y * dy).sum().backward()
    return x.grad.detach()


def prelu(x, slope, res=None):
    y = torch.where(x >= 0, x, x * slope)
    return y if res is None else res + y


def prelu_bwd(x, slope, dy):
    return torch.where(x >= 0, dy, dy * slope)


def install(monkeypatch):
    """Route ipercore_amd.ops.* to the emulation and relax the CUDA-only guards (tests only)."""
    for name in ("conv2d", "instnorm_stats", "instnorm_apply", "lwb_attention", "head_compose", "nchw_to_nhwc",
                 "nhwc_to_nchw", "project_faces", "rasterize_fim_wim", "bc_transform", "encode_fim", "flow_compose",
                 "smpl_lbs", "conv2d_wgrad", "colsum", "act_bwd", "lwb_fuse", "pack_panel", "unpack_wgrad", "norm_fwd", "norm_bwd",
                 "lwb_attention_bwd", "lwb_attention_kv", "lwb_attention_kv_bwd", "adam_step", "adam_step_dev", "conv2d_wgrad_unpacked", "maxpool2_fwd", "maxpool2_bwd", "flow_resize", "frames_to_u8", "thin_conv", "conv_transpose2d", "texture_sample", "grid_sample", "lwb_attention_x", "instnorm_finalize", "attn_records", "instnorm_finalize_ws", "crop_resize", "crop_resize_bwd", "prelu", "prelu_bwd"):
        monkeypatch.setattr(real_ops, name, globals()[name])
    from ipercore_amd.networks import generator
    monkeypatch.setattr(generator.AttentionLWBGenerator, "_check", lambda self, *a: None)
    monkeypatch.setattr(generator.InputConcatGenerator, "_check", staticmethod(lambda *a: None))
