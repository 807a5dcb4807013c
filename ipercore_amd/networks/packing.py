"""Weight re-layout for the HIP kernels (done once per weight load, on whatever device the weights live).

The implicit-GEMM kernel (csrc/conv_igemm.hip) wants the GEMM B panel as ``[K/4][N][4]`` fp32, K ordered
channel-chunk major / tap minor (``k = ((c // 32) * ntaps + tap) * 32 + c % 32``; ``tap * Cin + c`` for the
small-Cin first layers) and padded to a multiple of 32, so one lane's 16-byte load is four consecutive k of
one output channel.  ``state_dict`` tensors keep PyTorch's OIHW / IOHW layouts (checkpoint compatibility);
only these packed copies are what the kernels read.
"""
import torch

from .. import ops
from ..ops import ConvSpec


def _panel(wk, ntaps=None, cin=None):
    """(K, N) with k = tap * Cin + c -> contiguous (ceil32(K)/4, N, 4) in the kernel's K order: when Cin % 32 == 0 the
    32-channel chunks are the outer index and the taps the inner one (k' = ((c // 32) * ntaps + tap) * 32 + c % 32)."""
    K, N = wk.shape
    if ntaps is not None and cin % 32 == 0:
        assert K == ntaps * cin
        wk = wk.view(ntaps, cin // 32, 32, N).permute(1, 0, 2, 3).reshape(K, N)
    Kp = (K + 31) // 32 * 32
    if Kp != K:
        wk = torch.cat([wk, wk.new_zeros(Kp - K, N)], dim=0)
    return wk.view(Kp // 4, 4, N).permute(0, 2, 1).contiguous()


def _pad_vec(b, n):
    if b is None:
        return None
    b = b.detach().float()
    if b.numel() == n:
        return b.contiguous()
    return torch.cat([b, b.new_zeros(n - b.numel())]).contiguous()


def pack_conv(weight, bias=None, stride=1, pad=None, cin_pad=None, n_pad=None):
    """nn.Conv2d weight (N, Cin, kh, kw) -> ConvSpec.  ``cin_pad``: zero-extend input channels (6 -> 8);
    ``n_pad``: zero-extend output channels to the kernel's 64-column granularity."""
    weight = weight.detach().float()
    N, Cin, kh, kw = weight.shape
    pad = kh // 2 if pad is None else pad
    Cp = Cin if cin_pad is None else cin_pad
    Np = N if n_pad is None else n_pad
    taps = [(ky - pad, kx - pad) for ky in range(kh) for kx in range(kw)]
    if weight.is_cuda:                      # one HIP launch instead of a chain of view / permute / copy kernels
        return ConvSpec(ops.pack_panel(weight.contiguous(), False, range(kh * kw), Cin, Cp, N, Np), _pad_vec(bias, Np), Np, Cp, taps,
                        stride=stride, algo_kn=kh * kw * Cin * N)
    w = weight.new_zeros(kh * kw, Cp, Np)
    w[:, :Cin, :N] = weight.permute(2, 3, 1, 0).reshape(kh * kw, Cin, N)
    return ConvSpec(_panel(w.reshape(kh * kw * Cp, Np), kh * kw, Cp), _pad_vec(bias, Np), Np, Cp, taps, stride=stride,
                    algo_kn=kh * kw * Cin * N)


# output parity -> [(kernel index, input offset)] for ConvTranspose2d(kernel 4, stride 2, padding 1):
# out[2a]   = x[a] * w[1] + x[a-1] * w[3];   out[2a+1] = x[a+1] * w[0] + x[a] * w[2]
_CT_TAPS = {0: [(1, 0), (3, -1)], 1: [(0, 1), (2, 0)]}


def pack_conv_transpose(weight, bias=None, n_pad=None):
    """nn.ConvTranspose2d(k=4, s=2, p=1) weight (Cin, Cout, 4, 4) -> four ConvSpecs, one per output parity."""
    weight = weight.detach().float()
    Cin, N, kh, kw = weight.shape
    assert kh == 4 and kw == 4
    Np = N if n_pad is None else n_pad
    specs = []
    for py in (0, 1):
        for px in (0, 1):
            taps, mats, kidx = [], [], []
            for ky, dy in _CT_TAPS[py]:
                for kx, dx in _CT_TAPS[px]:
                    taps.append((dy, dx))
                    kidx.append(ky * 4 + kx)
                    if not weight.is_cuda:
                        m = weight.new_zeros(Cin, Np)
                        m[:, :N] = weight[:, :, ky, kx]
                        mats.append(m)
            if weight.is_cuda:
                specs.append(ConvSpec(ops.pack_panel(weight.contiguous(), True, kidx, Cin, Cin, N, Np), _pad_vec(bias, Np), Np, Cin, taps,
                                      stride=1, omul=2, ooy=py, oox=px, algo_kn=len(taps) * Cin * N))
                continue
            wk = torch.stack(mats, dim=0).reshape(len(taps) * Cin, Np)
            specs.append(ConvSpec(_panel(wk, len(taps), Cin), _pad_vec(bias, Np), Np, Cin, taps, stride=1, omul=2, ooy=py, oox=px,
                                  algo_kn=len(taps) * Cin * N))
    return specs


def pack_spade_gamma_beta(w_gamma, b_gamma, w_beta, b_beta):
    """mlp_gamma / mlp_beta (C, 128, 3, 3) each -> one ConvSpec with N = 2C whose columns alternate in blocks of
    32: [gamma c0..c31 | beta c0..c31 | gamma c32..c63 | ...] so one wave holds gamma and beta of the same
    channels (SPADE epilogue of the conv kernel)."""
    C = w_gamma.shape[0]
    assert C % 32 == 0
    wg = w_gamma.detach().float().view(C // 32, 32, *w_gamma.shape[1:])
    wb = w_beta.detach().float().view(C // 32, 32, *w_beta.shape[1:])
    w = torch.stack([wg, wb], dim=1).reshape(2 * C, *w_gamma.shape[1:])
    b = torch.stack([b_gamma.detach().float().view(C // 32, 32), b_beta.detach().float().view(C // 32, 32)], dim=1).reshape(2 * C)
    return pack_conv(w, b, stride=1)


def pack_head(w_img, w_att):
    """tsf_img_reg (3,C,5,5) + tsf_att_reg (1,C,5,5) -> (25, C, 4) fp32 for csrc/head.hip."""
    w = torch.cat([w_img.detach().float(), w_att.detach().float()], dim=0)       # (4, C, 5, 5)
    return w.permute(2, 3, 1, 0).reshape(25, w.shape[1], 4).contiguous()


def pack_thin(w):
    """A thin regressor (N <= 4, Cin, ks, ks) -> (ks*ks, Cin, 4) fp32 for csrc/head.hip lwg_thin_conv_f32 (unused columns zero)."""
    N, C, kh, kw = w.shape
    assert N <= 4 and kh == kw
    out = w.new_zeros(kh * kw, C, 4, dtype=torch.float32)
    out[:, :, :N] = w.detach().float().permute(2, 3, 1, 0).reshape(kh * kw, C, N)
    return out.contiguous()


def pack_head_bf16(w_img, w_att):
    """tsf_img_reg (3,64,5,5) + tsf_att_reg (1,64,5,5) -> the bf16 MFMA operand panel of csrc/bf16_ops.hip lwg_head_bf16_kernel:
    [ky 5][pass 2][channel half 2][lane 64][8]; lane l holds row (l % 16) = 4 * tap + output and channels half*32 + 8*(l // 16) + e;
    pass 0: taps kx = 0..3; pass 1: tap kx = 4 in rows 0..3, rows 4..15 zero."""
    w = torch.cat([w_img.detach().float(), w_att.detach().float()], dim=0).cpu()  # (4 outputs, 64, 5, 5); a 12.8 KB table: built on the host
    assert w.shape == (4, 64, 5, 5), w.shape
    out = w.new_zeros(5, 2, 2, 64, 8)
    lane = torch.arange(64)
    row, koct = lane % 16, lane // 16
    tap, o = row // 4, row % 4
    for ky in range(5):
        for ch in range(2):
            for e in range(8):
                c = ch * 32 + koct * 8 + e
                out[ky, 0, ch, :, e] = w[o, c, ky, tap]
                out[ky, 1, ch, :, e] = torch.where(tap == 0, w[o, c, ky, 4], torch.zeros(()))
    return out.contiguous().to(torch.bfloat16)


def spec_to(spec, device):
    spec.w = spec.w.to(device)
    if spec.bias is not None:
        spec.bias = spec.bias.to(device)
    return spec


# ---------------------------------------------------------------------------------------------- backward (training)
def _taps_weights(weight, stride, pad):
    """nn.Conv2d weight (N, Cin, kh, kw) -> (taps [(dy,dx)], W (ntaps, Cin, N))."""
    N, Cin, kh, kw = weight.shape
    taps = [(ky - pad, kx - pad) for ky in range(kh) for kx in range(kw)]
    return taps, weight.permute(2, 3, 1, 0).reshape(kh * kw, Cin, N)


def _dgrad_spec(taps_w, n_out, **kw):
    """[( (dy,dx), Wt (Cin', N') )] -> ConvSpec whose GEMM columns are the ORIGINAL input channels."""
    taps = [t for t, _ in taps_w]
    wk = torch.cat([w for _, w in taps_w], dim=0)                 # (ntaps * Cin', N')
    cin = taps_w[0][1].shape[0]
    return ConvSpec(_panel(wk, len(taps), cin), None, n_out, cin, taps, **kw)


def pack_dgrad_conv(weight, stride=1, pad=None, n_pad=None, cin_pad=None):
    """Data gradient of nn.Conv2d(weight (N, Cin, k, k), stride, pad) as forward-kernel launches on dY (B,OH,OW,N):
    stride 1 -> one spec (taps negated, panels transposed); stride 2 -> four specs, one per input parity, that scatter
    into dX with omul = 2 (dX[2a + p] = sum_{taps d == p mod 2} dY[a + (p - d) / 2] W_d^T).
    n_pad / cin_pad: the zero-extended channel counts the forward launch used (dY has n_pad channels, dX gets cin_pad)."""
    weight = weight.detach().float()
    N, Cin, kh, kw = weight.shape
    Np = N if n_pad is None else n_pad
    Cp = Cin if cin_pad is None else cin_pad
    pad = kh // 2 if pad is None else pad
    taps = [(ky - pad, kx - pad) for ky in range(kh) for kx in range(kw)]
    if stride == 1:
        groups = [(dict(stride=1), [((-d[0], -d[1]), i) for i, d in enumerate(taps)])]
    else:
        assert stride == 2
        groups = [(dict(stride=1, omul=2, ooy=py, oox=px),
                   [(((py - d[0]) // 2, (px - d[1]) // 2), i) for i, d in enumerate(taps) if (py - d[0]) % 2 == 0 and (px - d[1]) % 2 == 0])
                  for py in (0, 1) for px in (0, 1)]
    if weight.is_cuda:                      # GEMM input channels = N (dim 0), GEMM columns = Cin (dim 1): the "transposed" read
        w = weight.contiguous()
        return [ConvSpec(ops.pack_panel(w, True, [i for _, i in tl], N, Np, Cin, Cp), None, Cp, Np, [t for t, _ in tl], **kw)
                for kw, tl in groups]
    if Np != N or Cp != Cin:
        wpad = weight.new_zeros(Np, Cp, kh, kw)
        wpad[:N, :Cin] = weight
        weight = wpad
    _, W = _taps_weights(weight, stride, pad)
    Wt = W.permute(0, 2, 1).contiguous()                          # (ntaps, N, Cin)
    return [_dgrad_spec([(t, Wt[i]) for t, i in tl], Cp, **kw) for kw, tl in groups]


def pack_dgrad_conv_transpose(weight, n_pad=None):
    """Data gradient of nn.ConvTranspose2d(k=4, s=2, p=1) weight (Cin, N, 4, 4): a stride-2, 16-tap convolution over dY
    (B,2H,2W,n_pad): dX[a] = sum_{parity p, tap e} dY[2 a + (p - 2 e)] W_{p,e}^T."""
    weight = weight.detach().float()
    Cin, N, kh, kw = weight.shape
    Np = N if n_pad is None else n_pad
    tl = [((py - 2 * ey, px - 2 * ex), ky * 4 + kx) for py in (0, 1) for px in (0, 1) for ky, ey in _CT_TAPS[py] for kx, ex in _CT_TAPS[px]]
    if weight.is_cuda:                      # GEMM input channels = N (dim 1), columns = Cin (dim 0)
        return [ConvSpec(ops.pack_panel(weight.contiguous(), False, [i for _, i in tl], N, Np, Cin, Cin), None, Cin, Np,
                         [t for t, _ in tl], stride=2)]
    if Np != N:
        wpad = weight.new_zeros(Cin, Np, 4, 4)
        wpad[:, :N] = weight
        weight = wpad
    tw = [(t, weight[:, :, i // 4, i % 4].t().contiguous()) for t, i in tl]                              # (N, Cin)
    return [_dgrad_spec(tw, Cin, stride=2)]


def unpack_wgrad(dwk, ntaps, cin):
    """(ntaps*Cin, N) in the kernel's K order -> (ntaps, Cin, N)."""
    N = dwk.shape[1]
    if cin % 32 == 0:
        return dwk.view(cin // 32, ntaps, 32, N).permute(1, 0, 2, 3).reshape(ntaps, cin, N)
    return dwk.view(ntaps, cin, N)


def wgrad_to_conv(dwk, ntaps, cin_packed, cin, n, kh, kw):
    """-> nn.Conv2d weight gradient (N, Cin, kh, kw) (drops the zero-padded input / output channels)."""
    if dwk.is_cuda:
        return ops.unpack_wgrad(dwk, torch.empty(n, cin, kh, kw, device=dwk.device, dtype=torch.float32), False, range(ntaps), cin,
                                cin_packed, n)
    return unpack_wgrad(dwk, ntaps, cin_packed)[:, :cin, :n].reshape(kh, kw, cin, n).permute(3, 2, 0, 1).contiguous()


def wgrad_conv(x0, spec, dy, x1, kh, kw, cin, n, db=None):
    """nn.Conv2d weight gradient (N, Cin, kh, kw) of the launch ``spec``: on the device ONE reduction launch writes it in place
    (ops.conv2d_wgrad_unpacked) - and the bias gradient into ``db`` (n,) when given; the host-logic tests take the two-step form."""
    if dy.is_cuda:
        return ops.conv2d_wgrad_unpacked(x0, spec, dy, torch.empty(n, cin, kh, kw, device=dy.device, dtype=torch.float32), False,
                                         range(kh * kw), cin, n, x1=x1, db=db)
    if db is not None:
        db.copy_(ops.colsum(dy)[:n])
    return wgrad_to_conv(ops.conv2d_wgrad(x0, spec, dy, x1=x1), kh * kw, spec.Cin, cin, n, kh, kw)


def wgrad_conv_transpose(x0, specs, dy, cin, n, adj_spec=None):
    """nn.ConvTranspose2d(4, 2, 1) weight gradient (Cin, N, 4, 4).  With ``adj_spec`` (pack_dgrad_conv_transpose: the adjoint convolution
    dY -> dX, stride 2, 16 taps) ONE launch: that convolution's weight gradient with the roles swapped - input dY (B,2H,2W,N), "output
    gradient" x (B,H,W,Cin): dW'[(tap, n)][c] = sum_a dY[2a + off(tap)][n] x[a][c] = dW[c][n][tap] - instead of four parity launches
    (and four slab reductions)."""
    if dy.is_cuda:
        g = torch.empty(cin, n, 4, 4, device=dy.device, dtype=torch.float32)           # all 16 positions are covered either way
        if adj_spec is not None:
            kidx = [ky * 4 + kx for py in (0, 1) for px in (0, 1) for ky, _ in _CT_TAPS[py] for kx, _ in _CT_TAPS[px]]   # pack_dgrad_conv_transpose's tap order
            return ops.conv2d_wgrad_unpacked(dy, adj_spec, x0, g, False, kidx, n, cin)
        i = 0
        for py in (0, 1):
            for px in (0, 1):
                ops.conv2d_wgrad_unpacked(x0, specs[i], dy, g, True, [ky * 4 + kx for ky, _ in _CT_TAPS[py] for kx, _ in _CT_TAPS[px]], cin, n)
                i += 1
        return g
    return wgrad_to_conv_transpose([ops.conv2d_wgrad(x0, s, dy) for s in specs], cin, n)


def wgrad_thin(dy, dspec, x0, kh, kw, ns, n, cin):
    """Weight gradient of the thin backward form (training.thin_backward): the role-swapped launch's rows are (tap, n), its
    columns the input channels -> (N, Cin, kh, kw)."""
    if dy.is_cuda:
        return ops.conv2d_wgrad_unpacked(dy, dspec, x0, torch.empty(n, cin, kh, kw, device=dy.device, dtype=torch.float32), True,
                                         range(kh * kw), n, cin)
    return wgrad_thin_to_conv(ops.conv2d_wgrad(dy, dspec, x0), kh, kw, ns, n, cin)


def wgrad_thin_to_conv(dwk, kh, kw, ns, n, cin):
    """Weight gradient of the thin backward form (training.ConvFn._backward_thin): (kh*kw*ns, Cd) rows k = tap * ns + n
    (the small-Cin K order), columns = input channels -> nn.Conv2d gradient (N, Cin, kh, kw)."""
    if dwk.is_cuda:
        return ops.unpack_wgrad(dwk, torch.empty(n, cin, kh, kw, device=dwk.device, dtype=torch.float32), True, range(kh * kw), n, ns, cin)
    return dwk.view(kh, kw, ns, dwk.shape[1])[:, :, :n, :cin].permute(2, 3, 0, 1).contiguous()


def wgrad_to_conv_transpose(dwks, cin, n):
    """Four parity gradients (each (4*Cin, Np) in kernel order, taps as in pack_conv_transpose) -> (Cin, N, 4, 4)."""
    if dwks[0].is_cuda:
        g = torch.empty(cin, n, 4, 4, device=dwks[0].device, dtype=torch.float32)      # the four parities cover all 16 positions
        i = 0
        for py in (0, 1):
            for px in (0, 1):
                ops.unpack_wgrad(dwks[i], g, True, [ky * 4 + kx for ky, _ in _CT_TAPS[py] for kx, _ in _CT_TAPS[px]], cin, cin, n)
                i += 1
        return g
    g = dwks[0].new_zeros(cin, n, 4, 4)
    i = 0
    for py in (0, 1):
        for px in (0, 1):
            w = unpack_wgrad(dwks[i], 4, cin)                     # (4, Cin, Np)
            t = 0
            for ky, _ in _CT_TAPS[py]:
                for kx, _ in _CT_TAPS[px]:
                    g[:, :, ky, kx] = w[t][:, :n]
                    t += 1
            i += 1
    return g
