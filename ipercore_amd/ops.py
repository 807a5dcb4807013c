"""Thin tensor-level wrappers over the C ABI (include/lwg_hip.h).

PyTorch-ROCm is plumbing here: it owns device memory and the stream; every compute call goes through
``liblwg_hip.so``.  All wrappers require contiguous CUDA tensors and raise otherwise - there is no eager /
CPU fallback on the product path.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_RELU_MASK, ACT_SIGMOID, ACT_TANH, EPI_NONE, EPI_RESIDUAL, EPI_SPADE  # noqa: F401

EYE_DIST = 2.7320508075688776   # 1/tan(30 deg) + 1 (reference renders/nmr.py:225)


def _ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("ipercore_amd ops need CUDA (HIP) tensors: the MI355X path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


# "fp32": native fp32 MFMA.  "bf16": the engine keeps every activation tensor in bf16 and the convs with Cin % 64 == 0 run on the bf16
# MFMA kernel (dispatch is by tensor dtype: ops.conv2d on bf16 tensors; this flag tells the generator what to allocate).  "split": the same convs run on the bf16x6 kernel - both operands split EXACTLY into three bf16 parts,
# six bf16 MFMAs per fp32 product, fp32 accumulation: fp32-level accuracy at 0.375 of the native matrix-pipe time.  "winograd": the 3x3 / stride 1
# convolutions with Cin % 32 == 0 - one input or a skip concatenation (both channel counts % 8 == 0), plain / residual / SPADE (gamma | beta stacked)
# epilogue, fp32 NHWC output - run as fused F(2x2, 3x3) Winograd convolutions on the fp32 matrix pipe (csrc/conv_winograd.hip: 16 multiplies per
# 2x2 outputs instead of 36; fp32-grade, not bitwise the direct result); everything else exactly as "fp32" (same kernels, same head).
CONV_PRECISION = "fp32"


class conv_precision(object):
    """Context manager: ``with ops.conv_precision("bf16"): ...``.  "winograd2x2" = "winograd" with the F(4x4, 3x3) kernel off (every eligible layer on
    the F(2x2, 3x3) kernel - rounds 5 / 6's engine): the latency engine of single-frame callers (a one-frame launch of the F(4x4, 3x3) kernel is 32
    workgroups on a 64^2 layer: 3.6 against 2.4 ms per frame at frame batch 1), 20 % slower on frame batches.  Each engine is batch-invariant in itself."""

    def __init__(self, mode):
        assert mode in ("fp32", "bf16", "split", "winograd", "winograd2x2")
        self.mode = mode

    def __enter__(self):
        global CONV_PRECISION, WINO4
        self.prev = CONV_PRECISION, WINO4
        if self.mode == "winograd2x2":
            CONV_PRECISION, WINO4 = "winograd", False
        else:
            CONV_PRECISION = self.mode

    def __exit__(self, *exc):
        global CONV_PRECISION, WINO4
        CONV_PRECISION, WINO4 = self.prev


# bench.py installs a callable(begin, M, spec, epi, info) to bracket conv entry-point calls with HIP events; info (the closing call only, else
# None) = {"kernels": launches behind the call (lwg_conv_slice_count: batch slices), "kind": which kernel family ran ("direct", "winograd",
# "split", "bf16", "up4")} - launch / executed-flop accounting only
CONV_HOOK = None


def _stream():
    return torch.cuda.current_stream().cuda_stream


class ConvSpec:
    """Host description of one packed convolution (weights already in the kernel's layout)."""
    __slots__ = ("w", "bias", "N", "Cin", "ntaps", "dy", "dx", "stride", "cshift", "omul", "ooy", "oox", "algo_kn", "_w16v2", "_w16hr", "_w16x3", "_w16c8", "_w16up", "_w32up", "_wwino", "_wwino_t", "_wwino4")

    def __init__(self, w, bias, N, Cin, taps, stride=1, omul=1, ooy=0, oox=0, algo_kn=None):
        self.w, self.bias, self.N, self.Cin = w, bias, int(N), int(Cin)
        # un-padded K*N of the convolution this panel implements: 2*M*algo_kn = its algorithmic flops
        self.algo_kn = int(algo_kn) if algo_kn is not None else len(taps) * int(Cin) * int(N)
        self.ntaps = len(taps)
        self.dy = [int(t[0]) for t in taps]
        self.dx = [int(t[1]) for t in taps]
        self.stride, self.omul, self.ooy, self.oox = stride, omul, ooy, oox
        self._w16v2 = None
        self._w16hr = None
        self._w16x3 = None
        self._w16c8 = None
        self._w16up = None
        self._w32up = None
        self._wwino = None
        self._wwino_t = None
        self._wwino4 = None
        self.cshift = 0
        if self.Cin % 32 != 0:
            q = self.Cin // 4
            assert q in (1, 2, 4), "small-Cin path handles Cin in {4, 8, 16}"
            self.cshift = q.bit_length() - 1


def _w16v2(spec):
    """The bf16 panel of lwg_conv2d_nhwc_bf16, built once per spec from the fp32 panel [K/4][N][4] (K order: 32-channel chunk major,
    tap minor): [ntaps*Cin/64][N][64] with k = ((c/64)*ntaps + tap)*64 + c%64, the eight 16-byte k-octets of row n stored at slot
    octet ^ ((n >> 1) & 7) (include/lwg_hip.h)."""
    if spec._w16v2 is None or spec._w16v2.device != spec.w.device:
        K4, N, _ = spec.w.shape
        cin, nt = spec.Cin, spec.ntaps
        assert cin % 64 == 0 and K4 * 4 == nt * cin, (cin, nt, K4)
        wk = spec.w.permute(0, 2, 1).reshape(cin // 64, 2, nt, 32, N)             # [c64][half][tap][c%32][n]
        wk = wk.permute(0, 2, 1, 3, 4).reshape(cin // 64 * nt, 64, N)              # [kstep][k%64][n]
        rows = wk.permute(0, 2, 1).reshape(cin // 64 * nt, N, 8, 8)                # [kstep][n][octet][8]
        n = torch.arange(N, device=spec.w.device)
        slot = torch.arange(8, device=spec.w.device)
        src = slot[None, :] ^ ((n[:, None] >> 1) & 7)                              # octet stored at slot s of row n
        rows = torch.gather(rows, 2, src[None, :, :, None].expand(rows.shape[0], N, 8, 8))
        spec._w16v2 = rows.reshape(cin // 64 * nt, N, 64).contiguous().to(torch.bfloat16)
    return spec._w16v2


def _hr_eligible(spec, x0, y, out_hw):
    """lwg_conv2d_nhwc_bf16_hr applies to the 3x3 / 2x2-tap stride-1 launches on a shared input / output grid."""
    if not BF16_HR or spec.stride != 1 or spec.ntaps not in (9, 4) or spec.N % 64 != 0 or spec.Cin % 64 != 0:
        return False
    if any(abs(d) > 1 for d in spec.dy) or any(abs(d) > 1 for d in spec.dx):
        return False
    OH, OW = ((y.shape[1], y.shape[2]) if spec.omul == 1 else (y.shape[1] // spec.omul, y.shape[2] // spec.omul)) if out_hw is None else out_hw
    return (OH, OW) == (x0.shape[1], x0.shape[2])


def _hr_tap_order(spec):
    """Tap permutation of the register-streamed panel: ascending in (dy, dx) - the row-renaming kernel wants an NDY x NDX grid in
    row-major order (a transposed convolution's parity launches list their taps descending)."""
    return sorted(range(spec.ntaps), key=lambda t: (spec.dy[t], spec.dx[t]))


def _w16hr(spec, spade):
    """(panel, bias) of lwg_conv2d_nhwc_bf16_hr: [ntaps*Cin/64][4][N][16] bf16, k = ((c/64)*ntaps + tap)*64 + c%64 split as
    ks*16 + e; for the SPADE epilogue the gamma | beta columns (and the bias) are re-interleaved from blocks of 32 to blocks of 16
    (include/lwg_hip.h).  Built once per (spec, spade)."""
    key = bool(spade)
    if spec._w16hr is None or spec._w16hr[0] != key or spec._w16hr[1].device != spec.w.device:
        K4, N, _ = spec.w.shape
        cin, nt = spec.Cin, spec.ntaps
        wk = spec.w.permute(0, 2, 1).reshape(cin // 64, 2, nt, 32, N).permute(0, 2, 1, 3, 4).reshape(cin // 64, nt, 64, N)    # [chunk][tap][k%64][n]
        wk = wk[:, _hr_tap_order(spec)].reshape(cin // 64 * nt, 64, N)                                                          # taps ascending in (dy, dx)
        bias = spec.bias
        if spade:
            j = torch.arange(N, device=spec.w.device)
            q, r = j // 32, j % 32
            ch = 16 * q + (r % 16)
            old = 64 * (ch // 32) + (ch % 32) + 32 * (r // 16)
            wk = wk[:, :, old]
            bias = None if bias is None else bias[old].contiguous()
        panel = wk.reshape(cin // 64 * nt, 4, 16, N).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)
        spec._w16hr = (key, panel, bias)
    return spec._w16hr[1], spec._w16hr[2]


BF16_HR = True          # lab switch: False routes every bf16 convolution to lwg_conv2d_nhwc_bf16 (LDS-DMA kernels)
BF16_PW = True          # lab switch: the pointwise (1x1, C -> C) launches on the register-resident-weights kernel
BF16_C8 = True          # lab switch: the fp32-input first layer on the bf16 MFMA kernel (False: fp32 MFMA kernel, bf16 output)


def _pw_eligible(spec, x0, y, x1, epi, out_hw):
    """The 1x1 / stride-1 / C -> C launches (the attention blocks' query projections) go to lwg_conv2d_nhwc_bf16_hr with ntaps = 1."""
    if not BF16_PW or spec.ntaps != 1 or spec.stride != 1 or spec.omul != 1 or x1 is not None or epi != EPI_NONE:
        return False
    if spec.dy[0] != 0 or spec.dx[0] != 0 or spec.N != spec.Cin or spec.N not in (64, 128, 256):
        return False
    OH, OW = (y.shape[1], y.shape[2]) if out_hw is None else out_hw
    return (OH, OW) == (x0.shape[1], x0.shape[2])


def _w16c8(spec):
    """bf16 panel of lwg_conv2d_nhwc_c8_bf16: [ceil(ntaps/2)][N][16], k = tap*8 + c (the small-Cin K order of the fp32 panel)."""
    if spec._w16c8 is None or spec._w16c8.device != spec.w.device:
        K4, N, _ = spec.w.shape
        nks = (spec.ntaps + 1) // 2
        wk = spec.w.permute(0, 2, 1).reshape(K4 * 4, N)
        if wk.shape[0] < nks * 16:
            wk = torch.cat([wk, wk.new_zeros(nks * 16 - wk.shape[0], N)], dim=0)
        spec._w16c8 = wk[:nks * 16].reshape(nks, 16, N).permute(0, 2, 1).contiguous().to(torch.bfloat16)
    return spec._w16c8


def _w16x3(spec):
    """The three-plane bf16 panel [3][K/8][N][8] of a spec: w = hi + mid + lo exactly (round-to-nearest residual splits)."""
    if spec._w16x3 is None or spec._w16x3.device != spec.w.device:
        K4, N, _ = spec.w.shape
        wk = spec.w.permute(0, 2, 1).reshape(K4 * 4, N)
        hi = wk.to(torch.bfloat16)
        r1 = wk - hi.float()
        mid = r1.to(torch.bfloat16)
        lo = (r1 - mid.float()).to(torch.bfloat16)
        spec._w16x3 = torch.stack([hi, mid, lo]).view(3, K4 // 2, 8, N).permute(0, 1, 3, 2).contiguous()
    return spec._w16x3


_WINO_TAPS = sorted((dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1))


def _wino_eligible(spec, x0, y, x1, epi, act, out_hw, q4, ycoff=0):
    """Launches lwg_conv2d_winograd_f32 takes: 3x3 / stride 1 / pad 1, fp32, Cin % 32 == 0 (one input or a skip concatenation with both channel
    counts % 8 == 0), plain / residual / SPADE (gamma | beta stacked, N = 2 C) epilogue, dense output."""
    if q4 or x0.dtype != torch.float32 or y.dtype != torch.float32 or epi not in (EPI_NONE, EPI_RESIDUAL, EPI_SPADE) or \
            (act == ACT_RELU_MASK and epi != EPI_RESIDUAL):
        return False
    if x1 is not None and (x1.dtype != torch.float32 or x0.shape[3] % 8 != 0 or x1.shape[3] % 8 != 0):
        return False
    if spec.ntaps != 9 or spec.stride != 1 or spec.omul != 1 or spec.Cin % 32 != 0 or spec.N % 64 != 0:
        return False
    if epi == EPI_SPADE and (y.shape[3] * 2 != spec.N or ycoff != 0):
        return False
    if sorted(zip(spec.dy, spec.dx)) != _WINO_TAPS:
        return False
    hw = (y.shape[1], y.shape[2]) if out_hw is None else tuple(out_hw)
    return hw == (x0.shape[1], x0.shape[2]) == (y.shape[1], y.shape[2])


WINO_MIN_GRID = 128     # workgroups (64 patches x 32 channels, times the K slices) below which a splitk=True caller gets the direct split-K form instead
WINO_SPLITK = True      # lab switch: False = training launches never split the Winograd kernel's K loop


def _wino_plan(a, spec, y, splitk):
    """How an eligible launch runs in the "winograd" mode: 0 = the Winograd kernel whole, n > 0 = split over K through a workspace of n floats
    (lwg_conv2d_winograd_f32_ws), None = not on the Winograd kernel.  Only a training launch (splitk=True) asks: one whose Winograd grid would leave
    most of the 256 CUs idle - a 28 x 28 x 512 VGG19 layer is 64 workgroups of 64 K stages each, 79 us whole against the direct kernel's split-K
    51 us - runs its K loop in slices when that fills the chip, and stays on the direct split-K kernel when even the slices do not
    (profiles/r05_g_winograd_small_launches.txt).  The synthesis path never asks (splitk=False: batch invariance)."""
    if not splitk:
        return 0
    # the kernel's own plan (lwg_conv2d_winograd_plan) - not a host copy of its tile geometry
    blocks, slices, nbv = ctypes.c_longlong(0), ctypes.c_int(0), ctypes.c_int(0)
    if _lib.lib().lwg_conv2d_winograd_plan(a, 1 if WINO_SPLITK else 0, ctypes.byref(blocks), ctypes.byref(slices), ctypes.byref(nbv), None) != 0:
        return None
    units32 = blocks.value * (nbv.value // 32)              # 64-patch x 32-channel units of work, times the K slices
    return slices.value * a.M * spec.N if units32 >= WINO_MIN_GRID else None


def _wwino(spec):
    """The transformed-weight fragment panel of lwg_conv2d_winograd_f32, built once per spec from the fp32 GEMM panel by ONE launch
    (lwg_winograd_panel_f32): U = G w G^T per (input, output) channel pair in fp64, rounded once, stored [16][Cin/8][2][N][4] - element
    (p, s, kh, n, kk) = U[p // 4][p % 4] of input channel 8 s + 2 kk + kh."""
    if spec._wwino is None or spec._wwino.device != spec.w.device:
        K4, N, _ = spec.w.shape
        cin, nt = spec.Cin, spec.ntaps
        assert nt == 9 and cin % 32 == 0 and K4 * 4 == nt * cin, (cin, nt, K4)
        if not spec.w.is_cuda:
            raise RuntimeError("ipercore_amd ops need CUDA (HIP) tensors: the MI355X path has no CPU fallback")
        tap9 = (ctypes.c_int * 9)()
        for t in range(nt):
            tap9[3 * (spec.dy[t] + 1) + spec.dx[t] + 1] = t
        if PANEL_CACHE is not None:                          # a training step: panels of registered weights are refreshed once per step
            U = PANEL_CACHE.winograd(spec, list(tap9))
            if U is not None:
                spec._wwino = U
                return U
        U = torch.empty(16, cin // 8, 2, N, 4, device=spec.w.device, dtype=torch.float32)
        _lib.check(_lib.lib().lwg_winograd_panel_f32(_ptr(spec.w), _ptr(U), cin, N, tap9, _stream()), "lwg_winograd_panel_f32")
        spec._wwino = U
    return spec._wwino


WINO4 = True            # lab switch: in the "winograd" mode the synthesis path's eligible launches with Cin >= WINO4_MIN_CIN run the F(4x4, 3x3) kernel
WINO4_MIN_CIN = 64      # below it the F(2x2, 3x3) kernel stays (a block's K loop must pay for the 28-plane exchange of the F(4x4, 3x3) epilogue)


def _wino4_use(spec, splitk):
    """Which launches of the "winograd" mode run lwg_conv2d_winograd4_f32 (F(4x4, 3x3): 2.25 multiplies per output instead of 4): a rule on the LAYER
    (its Cin), never on the batch - a frame's result must not depend on how many frames share its launch.  Training launches (splitk=True) keep the
    F(2x2, 3x3) kernel and its split plan."""
    return WINO4 and not splitk and spec.Cin >= WINO4_MIN_CIN


def _wwino4(spec):
    """The fragment panel of lwg_conv2d_winograd4_f32, built once per spec from the fp32 GEMM panel by ONE launch (lwg_winograd4_panel_f32):
    U = G w G^T (6 x 6) per (input, output) channel pair in fp64, rounded once, stored [4][Cin/8][4][2][9 N] (include/lwg_hip.h)."""
    if spec._wwino4 is None or spec._wwino4.device != spec.w.device:
        K4, N, _ = spec.w.shape
        cin, nt = spec.Cin, spec.ntaps
        assert nt == 9 and cin % 32 == 0 and K4 * 4 == nt * cin, (cin, nt, K4)
        if not spec.w.is_cuda:
            raise RuntimeError("ipercore_amd ops need CUDA (HIP) tensors: the MI355X path has no CPU fallback")
        tap9 = (ctypes.c_int * 9)()
        for t in range(nt):
            tap9[3 * (spec.dy[t] + 1) + spec.dx[t] + 1] = t
        U = torch.empty(4, cin // 8, 4, 2, 9 * N, device=spec.w.device, dtype=torch.float32)
        _lib.check(_lib.lib().lwg_winograd4_panel_f32(_ptr(spec.w), _ptr(U), cin, N, tap9, _stream()), "lwg_winograd4_panel_f32")
        spec._wwino4 = U
    return spec._wwino4


def conv_args(x0, spec, y, x1=None, epi=EPI_NONE, act=ACT_NONE, res=None, xn=None, mean=None, rstd=None,
              out_hw=None, ycoff=0, q4=False):
    """Fill the C-ABI argument block of one conv launch (see ``conv2d``).  q4: y is (B, YC/4, YH, YW, 4) - channel-quad planes
    (LWG_DT_F32_Q4), fp32 launches without a fused residual / SPADE epilogue only."""
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    assert C0 + C1 == spec.Cin, (C0, C1, spec.Cin)
    if q4:
        YB, YCq, YH, YW, four = y.shape
        assert four == 4 and y.dtype == torch.float32 and x0.dtype == torch.float32 and epi == EPI_NONE and y.is_contiguous()
        YC = 4 * YCq
    else:
        YB, YH, YW, YC = y.shape
    if out_hw is None:
        OH, OW = (YH, YW) if spec.omul == 1 else (YH // spec.omul, YW // spec.omul)
    else:
        OH, OW = out_hw
    a = _lib.LwgConvArgs()
    xdt, ydt = x0.dtype, y.dtype
    a.xdt = _lib.DT_BF16 if xdt == torch.bfloat16 else _lib.DT_F32
    a.ydt = _lib.DT_F32_Q4 if q4 else (_lib.DT_BF16 if ydt == torch.bfloat16 else _lib.DT_F32)
    a.x0, a.x1 = _ptr(x0, xdt), _ptr(x1, xdt)
    a.C0, a.C1 = C0, C1
    a.B, a.H, a.W = B, H, W
    a.OH, a.OW, a.M = OH, OW, B * OH * OW
    a.stride, a.ntaps, a.cshift = spec.stride, spec.ntaps, spec.cshift
    a.w, a.N, a.bias = _ptr(spec.w), spec.N, _ptr(spec.bias)
    a.y, a.YH, a.YW, a.YC, a.ycoff = _ptr(y, ydt), YH, YW, YC, ycoff
    a.omul, a.ooy, a.oox = spec.omul, spec.ooy, spec.oox
    a.epi, a.act = epi, act
    a.res, a.xn, a.mean, a.rstd = _ptr(res, ydt), _ptr(xn, ydt), _ptr(mean), _ptr(rstd)
    for i in range(spec.ntaps):
        a.dy[i] = spec.dy[i]
        a.dx[i] = spec.dx[i]
    return a


def _hook_end(a, spec, epi, kind, slices=True, extra=0):
    """Closing hook call: how many kernel launches the entry point made (the C rule: lwg_conv_slice_count; extra: a split launch's finishing
    kernel) and which kernel family ran."""
    n = (max(1, int(_lib.lib().lwg_conv_slice_count(a))) if slices else 1) + extra
    CONV_HOOK(False, a.M, spec, epi, {"kernels": n, "kind": kind})


def conv2d(x0, spec, y, x1=None, epi=EPI_NONE, act=ACT_NONE, res=None, xn=None, mean=None, rstd=None,
           out_hw=None, ycoff=0, splitk=False, q4=False):
    """y <- conv(cat[x0, x1]) per ``spec``; x*: (B,H,W,C) NHWC; y: (B,YH,YW,YC) NHWC (written in place).
    splitk: let the library split small-M / large-K launches over K (the training step's one-sample launches).  Off on the
    synthesis path: a frame's result must not depend on how many frames share its launch (batch invariance is a parity check).
    q4: y is (B, YC/4, YH, YW, 4), channel-quad planes (the fp32 MFMA path only: what ``head_compose(..., q4=True)`` reads)."""
    a = conv_args(x0, spec, y, x1, epi, act, res, xn, mean, rstd, out_hw, ycoff, q4)
    if q4 and (splitk or CONV_PRECISION not in ("fp32", "winograd") or x0.dtype != torch.float32):
        raise ValueError("channel-quad-plane outputs: fp32 activations on the fp32 MFMA path, no split-K")
    if CONV_HOOK is not None:
        CONV_HOOK(True, a.M, spec, epi, None)
    kind, sliced, extra = "direct", True, 0
    if x0.dtype == torch.bfloat16:
        kind = "bf16"
        # bf16 activation storage (BASELINE configs[3]): bf16 in, bf16 out, bf16 MFMA operands, fp32 accumulation
        if y.dtype != torch.bfloat16 or spec.Cin % 64 != 0:
            raise ValueError("bf16 convolutions need bf16 outputs and Cin % 64 == 0")
        if _hr_eligible(spec, x0, y, out_hw) or _pw_eligible(spec, x0, y, x1, epi, out_hw):
            panel, bias = _w16hr(spec, epi == EPI_SPADE)
            a.w, a.bias = _ptr(panel, torch.bfloat16), _ptr(bias)
            for i, t in enumerate(_hr_tap_order(spec)):              # the panel's tap order
                a.dy[i], a.dx[i] = spec.dy[t], spec.dx[t]
            _lib.check(_lib.lib().lwg_conv2d_nhwc_bf16_hr(a, _stream()), "lwg_conv2d_nhwc_bf16_hr")
        else:
            a.w = _ptr(_w16v2(spec), torch.bfloat16)
            _lib.check(_lib.lib().lwg_conv2d_nhwc_bf16(a, _stream()), "lwg_conv2d_nhwc_bf16")
    elif y.dtype == torch.bfloat16:
        # the first layer of a network in bf16 mode: fp32 image-like input (Cin < 32), bf16 output
        if BF16_C8 and spec.Cin == 8 and spec.N == 64 and spec.ntaps <= 10 and x1 is None and epi == EPI_NONE and spec.omul == 1:
            a.w = _ptr(_w16c8(spec), torch.bfloat16)
            _lib.check(_lib.lib().lwg_conv2d_nhwc_c8_bf16(a, _stream()), "lwg_conv2d_nhwc_c8_bf16")
        else:
            _lib.check(_lib.lib().lwg_conv2d_nhwc_f32(a, _stream()), "lwg_conv2d_nhwc_f32")
    elif CONV_PRECISION == "winograd" and _wino_eligible(spec, x0, y, x1, epi, act, out_hw, q4, ycoff) and \
            (nws := _wino_plan(a, spec, y, splitk)) is not None:
        kind, sliced = "winograd", False            # per-image buffer descriptors: one launch at any batch size
        if _wino4_use(spec, splitk):
            a.w = _ptr(_wwino4(spec))
            kind = "winograd4"
            _lib.check(_lib.lib().lwg_conv2d_winograd4_f32(a, _stream()), "lwg_conv2d_winograd4_f32")
        elif nws:
            a.w = _ptr(_wwino(spec))
            ws, extra = torch.empty(nws, device=x0.device, dtype=torch.float32), 1
            _lib.check(_lib.lib().lwg_conv2d_winograd_f32_ws(a, _ptr(ws), _stream()), "lwg_conv2d_winograd_f32_ws")
        else:
            a.w = _ptr(_wwino(spec))
            _lib.check(_lib.lib().lwg_conv2d_winograd_f32(a, _stream()), "lwg_conv2d_winograd_f32")
    elif CONV_PRECISION == "split" and spec.Cin % 32 == 0:
        a.w = _ptr(_w16x3(spec), torch.bfloat16)
        kind = "split"
        _lib.check(_lib.lib().lwg_conv2d_nhwc_f32_split(a, _stream()), "lwg_conv2d_nhwc_f32_split")
    else:
        nws = _lib.lib().lwg_conv2d_ws_floats(a) if splitk else 0    # > 0: a small-M / large-K launch the library runs split-K
        if nws:
            ws = torch.empty(nws, device=x0.device, dtype=torch.float32)
            _lib.check(_lib.lib().lwg_conv2d_nhwc_f32_ws(a, _ptr(ws), _stream()), "lwg_conv2d_nhwc_f32_ws")
        else:
            _lib.check(_lib.lib().lwg_conv2d_nhwc_f32(a, _stream()), "lwg_conv2d_nhwc_f32")
    if CONV_HOOK is not None:
        _hook_end(a, spec, epi, kind, sliced, extra)
    return y


WINO_UP4 = True         # lab switch: in the "winograd" mode the fp32 transposed convolutions of the synthesis path run lwg_conv_transpose4_winograd_f32
F32_UP4 = True          # lab switch: fp32 transposed convolutions through lwg_conv_transpose4_nhwc_f32 (False: four conv2d calls)
BF16_UP4 = True         # lab switch: the four parity launches of a bf16 transposed convolution fused into one (Cin <= 128)


def _parity_specs_ok(specs):
    """The four parity specs of ONE ConvTranspose2d(4, 2, 1) in order (packing.pack_conv_transpose): parity p's taps are parity 0's shifted by p."""
    s0 = specs[0]
    return (len(specs) == 4 and all(s.ntaps == 4 and s.omul == 2 and s.stride == 1 and (s.ooy, s.oox) == (i >> 1, i & 1) and s.N == s0.N and s.Cin == s0.Cin
                                    and [(dy - (i >> 1), dx - (i & 1)) for dy, dx in zip(s.dy, s.dx)] == list(zip(s0.dy, s0.dx)) for i, s in enumerate(specs)))


def _wwino_t(specs):
    """The transformed-weight panel of lwg_conv_transpose4_winograd_f32 from the four parity GEMM panels, built once per layer (a weight transform at
    load time, like the packing itself): Upk[4][Cin/8][4][2][9 N] - per (parity 2 py + px, s, kk, kh) [N][4] products 0-3, [N][4] products 4-7, [N] product 8 -, product 3 xi + nu of column n = sgn (G g G^T)[xi][nu] for input
    channel 8 s + 2 kk + kh - g the parity's 2 x 2 sub-kernel in input-offset order, G = [[1,0],[1,1],[0,1]], formed in fp64 and rounded once; the
    sign (-1 where a parity-1 row / column takes the form the parity-0 one already holds negated) is documented in include/lwg_hip.h."""
    s0 = specs[0]
    U = s0._wwino_t
    if U is not None and U.device == s0.w.device:
        return U
    Cin, N = s0.Cin, s0.N
    G = torch.tensor([[1.0, 0.0], [1.0, 1.0], [0.0, 1.0]], dtype=torch.float64, device=s0.w.device)
    parts = []
    for par, sp in enumerate(specs):
        py, px = par >> 1, par & 1
        W = sp.w.double().permute(0, 2, 1).reshape(4 * Cin, N)                  # rows k = ((c / 32) 4 + tap) 32 + c % 32
        W = W.view(Cin // 32, 4, 32, N).permute(1, 0, 2, 3).reshape(4, Cin, N)  # [tap][c][n]
        g = torch.zeros(2, 2, Cin, N, dtype=torch.float64, device=W.device)
        for t, (dy, dx) in enumerate(zip(sp.dy, sp.dx)):
            g[dy - (py - 1), dx - (px - 1)] = W[t]                               # g[r][q] multiplies x[i + py - 1 + r][j + px - 1 + q]
        u = torch.einsum("ar,rqcn,bq->abcn", G, g, G)                          # (3, 3, Cin, N)
        if py:
            u[0] = -u[0]
        if px:
            u[:, 0] = -u[:, 0]
        parts.append(u.reshape(9, Cin, N))
    U9 = torch.stack(parts).float()                                              # (4, 9, Cin, N)
    # per (parity, input channel): 9 N floats - [N][4] products 0-3, [N][4] products 4-7, [N] product 8 (every load of the kernel reads contiguous memory)
    U = torch.cat([U9[:, 0:4].permute(0, 2, 3, 1).reshape(4, Cin, 4 * N), U9[:, 4:8].permute(0, 2, 3, 1).reshape(4, Cin, 4 * N), U9[:, 8]], dim=2)
    U = U.view(4, Cin // 8, 4, 2, 9 * N).contiguous()                            # c = 8 s + 2 kk + kh
    s0._wwino_t = U
    return U


class _FusedTransposeSpec(object):
    """What a launch-accounting hook (bench.ConvTimer) sees for the fused transposed convolution: per INPUT pixel 16 taps and 4 N
    outputs; flops = 2 M algo_kn, bytes = M Cin in + 4 M N out + the four panels."""

    def __init__(self, s0, panel):
        self.N, self.Cin, self.ntaps, self.stride, self.omul = 4 * s0.N, s0.Cin, 16, 1, 2
        self.algo_kn, self.w = 4 * s0.algo_kn, panel


def conv_transpose2d(x, specs, y, act=ACT_NONE, splitk=False, out_hw=None, q4=False):
    """y (B,2H,2W,N) <- ConvTranspose2d(4, 2, 1) of x (B,H,W,Cin) given its four parity specs (packing.pack_conv_transpose).
    bf16 activations with Cin <= 128: ONE launch (lwg_conv_transpose4_nhwc_bf16: the input block is staged once for the four
    parities); fp32 in the "winograd" mode on the synthesis path (splitk=False): ONE fused F(2x2, 2x2) Winograd launch
    (lwg_conv_transpose4_winograd_f32: 36 products per 4 x 4 input patch instead of 64; fp32-grade, not the direct forms' bits);
    otherwise the one-grid form for small fp32 launches or the four parity launches of ``conv2d``.  ``splitk`` / ``out_hw`` are handed to those four launches (the
    training callers' plan: the one-grid form and the four-launch fall-back then differ only in launch count); ``out_hw`` is a
    callable spec -> (OH, OW) or None.  q4 (fp32 only): y is (B, N/4, 2H, 2W, 4), channel-quad planes (see ``conv2d``)."""
    s0 = specs[0]
    if (BF16_UP4 and BF16_HR and x.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and len(specs) == 4 and s0.Cin in (64, 128)
            and s0.N % 64 == 0 and all(s.ntaps == 4 and s.omul == 2 and (s.ooy, s.oox) == (i >> 1, i & 1) for i, s in enumerate(specs))):
        a = conv_args(x, s0, y, act=act)
        panel = getattr(s0, "_w16up", None)
        if panel is None or panel.device != s0.w.device:
            panel = torch.stack([_w16hr(s, False)[0] for s in specs]).contiguous()
            s0._w16up = panel
        a.w = _ptr(panel, torch.bfloat16)
        if CONV_HOOK is not None:                # one launch = the whole transposed convolution: 16 taps, 4 N output values per input pixel
            whole = _FusedTransposeSpec(s0, panel)
            CONV_HOOK(True, a.M, whole, EPI_NONE, None)
        _lib.check(_lib.lib().lwg_conv_transpose4_nhwc_bf16(a, _stream()), "lwg_conv_transpose4_nhwc_bf16")
        if CONV_HOOK is not None:
            _hook_end(a, whole, EPI_NONE, "up4")
        return y
    if (WINO_UP4 and CONV_PRECISION == "winograd" and not splitk and out_hw is None and x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32
            and s0.Cin % 32 == 0 and s0.N % 32 == 0 and _parity_specs_ok(specs)):
        # the synthesis path in the "winograd" mode: the layer as ONE fused F(2x2, 2x2) Winograd launch (36 products per 4 x 4 input patch instead
        # of 64; per-image work in a fixed order: a frame does not depend on its batch).  Training callers (splitk=True) keep the direct forms.
        a = conv_args(x, s0, y, act=act, q4=q4)
        panel = _wwino_t(specs)
        a.w = _ptr(panel)
        if CONV_HOOK is not None:
            whole = _FusedTransposeSpec(s0, panel)
            CONV_HOOK(True, a.M, whole, EPI_NONE, None)
        _lib.check(_lib.lib().lwg_conv_transpose4_winograd_f32(a, _stream()), "lwg_conv_transpose4_winograd_f32")
        if CONV_HOOK is not None:
            _hook_end(a, whole, EPI_NONE, "winograd_up4", False)
        return y
    if (F32_UP4 and x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and CONV_PRECISION in ("fp32", "winograd") and s0.Cin % 32 == 0
            and _parity_specs_ok(specs)):
        # fp32, small launch (one frame: a parity is a workgroup per CU or less): ONE grid of four times the workgroups
        # (lwg_conv_transpose4_nhwc_f32); large launches stay four conv2d calls (the library would issue the same four launches). The
        # values are those of the four conv2d calls bit for bit (same tiles, same K order).
        a = conv_args(x, s0, y, act=act, q4=q4)
        if _lib.lib().lwg_conv_transpose4_is_one_grid(a):
            panel = getattr(s0, "_w32up", None)
            if panel is None or panel.device != s0.w.device:
                panel = torch.stack([s.w.reshape(-1) for s in specs]).contiguous()
                s0._w32up = panel
            a.w = _ptr(panel)
            if CONV_HOOK is not None:
                whole = _FusedTransposeSpec(s0, panel)
                CONV_HOOK(True, a.M, whole, EPI_NONE, None)
            _lib.check(_lib.lib().lwg_conv_transpose4_nhwc_f32(a, _stream()), "lwg_conv_transpose4_nhwc_f32")
            if CONV_HOOK is not None:
                _hook_end(a, whole, EPI_NONE, "up4")
            return y
    for s in specs:
        conv2d(x, s, y, act=act, splitk=splitk, out_hw=None if out_hw is None else out_hw(s), q4=q4)
    return y


def instnorm_stats(x, mean, rstd, ws, eps=1e-5, nsplit=None):
    B, H, W, C = x.shape
    HW = H * W
    if nsplit is None:
        nsplit = max(1, min(64, HW // 64))
    assert ws.numel() >= B * C * nsplit * 3
    if x.dtype == torch.bfloat16:
        _lib.check(_lib.lib().lwg_instnorm_stats_nhwc_bf16(_ptr(x, torch.bfloat16), B, HW, C, eps, _ptr(mean), _ptr(rstd), _ptr(ws), nsplit,
                                                            _stream()), "lwg_instnorm_stats_nhwc_bf16")
        return
    _lib.check(_lib.lib().lwg_instnorm_stats_nhwc_f32(_ptr(x), B, HW, C, eps, _ptr(mean), _ptr(rstd), _ptr(ws), nsplit,
                                                       _stream()), "lwg_instnorm_stats_nhwc_f32")


def instnorm_apply(x, mean, rstd, y, act=ACT_NONE, res=None):
    B, H, W, C = x.shape
    _lib.check(_lib.lib().lwg_instnorm_apply_nhwc_f32(_ptr(x), _ptr(mean), _ptr(rstd), _ptr(res), _ptr(y), B, H * W, C, act,
                                                       _stream()), "lwg_instnorm_apply_nhwc_f32")
    return y


def flow_resize(T, h, w):
    """(B,ns,S,S,2) flows -> (B,ns,h,w,2): F.interpolate(bilinear, align_corners=True) of LWB.resize_trans as its own pass."""
    B, ns, S = T.shape[0], T.shape[1], T.shape[2]
    T = T.contiguous()
    out = torch.empty(B, ns, h, w, 2, device=T.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_flow_resize_f32(_ptr(T), B * ns, S, h, w, _ptr(out), _stream()), "lwg_flow_resize_f32")
    return out


def lwb_attention(q, Ks, Vs, bk, bv, T, out, src_batched=False):
    B, h, w, C = q.shape
    ns = T.shape[1]
    S = T.shape[2]
    assert T.shape[0] == B and Ks.shape[0] == (B * ns if src_batched else ns) and tuple(Ks.shape[1:]) == (h, w, C)
    if q.dtype == torch.bfloat16:
        bf = torch.bfloat16
        _lib.check(_lib.lib().lwg_lwb_attention_bf16(_ptr(q, bf), _ptr(Ks, bf), _ptr(Vs, bf), _ptr(bk), _ptr(bv), _ptr(T), _ptr(out, bf), B, ns, h,
                                                      w, C, S, 1 if src_batched else 0, _stream()), "lwg_lwb_attention_bf16")
        return out
    _lib.check(_lib.lib().lwg_lwb_attention_f32(_ptr(q), _ptr(Ks), _ptr(Vs), _ptr(bk), _ptr(bv), _ptr(T), _ptr(out), B, ns, h,
                                                 w, C, S, 1 if src_batched else 0, _stream()), "lwg_lwb_attention_f32")
    return out


def attn_records(h, w, C, dtype=torch.float32):
    """Records per image the statistics form of ``lwb_attention_x`` leaves: one per workgroup = per 8 x 8-pixel tile, or per part of a
    tile on small feature maps (a function of (h, w, C, storage type) only - never of the batch)."""
    n = _lib.lib().lwg_lwb_attention_x_records(int(h), int(w), int(C), 2 if dtype == torch.bfloat16 else 4)
    if n <= 0:
        raise ValueError(f"lwb_attention_x: unsupported (h, w, C, dtype) = {(h, w, C, dtype)}")
    return n


def lwb_attention_x(x, Kq, kappa, Vs, bv, T, out, stats=None, src_batched=False):
    """The attention-form Liquid Warping Block with the query projection folded into the source side (csrc/lwb_attn_x.hip):
    logit_s = (warp_s(Kq) . x + warp_s(kappa)) / sqrt(C), out = sum_s softmax_s(logit) warp_s(Vs) + bv.
    x (B,h,w,C) fp32 | bf16; Kq, Vs (nsrc,h,w,C) in x's dtype; kappa (nsrc,h,w) fp32; T (B,ns,h,w,2) flows ALREADY at (h,w).
    stats: None or a fp32 buffer of >= B * attn_records(h, w, C, dtype) * C * 3 floats receiving the partial InstanceNorm records of x."""
    B, h, w, C = x.shape
    ns = T.shape[1]
    nsrc = B * ns if src_batched else ns
    assert tuple(T.shape) == (B, ns, h, w, 2), (tuple(T.shape), (B, ns, h, w, 2))
    assert tuple(Kq.shape) == (nsrc, h, w, C) and tuple(Vs.shape) == (nsrc, h, w, C) and tuple(kappa.shape) == (nsrc, h, w)
    assert Kq.dtype == x.dtype and Vs.dtype == x.dtype and out.dtype == x.dtype and kappa.dtype == torch.float32
    assert stats is None or stats.numel() >= B * attn_records(h, w, C, x.dtype) * C * 3
    if x.dtype == torch.bfloat16:
        bf = torch.bfloat16
        _lib.check(_lib.lib().lwg_lwb_attention_x_bf16(_ptr(x, bf), _ptr(Kq, bf), _ptr(kappa), _ptr(Vs, bf), _ptr(bv), _ptr(T), _ptr(out, bf),
                                                        _ptr(stats), B, ns, h, w, C, 1 if src_batched else 0, _stream()), "lwg_lwb_attention_x_bf16")
        return out
    _lib.check(_lib.lib().lwg_lwb_attention_x_f32(_ptr(x), _ptr(Kq), _ptr(kappa), _ptr(Vs), _ptr(bv), _ptr(T), _ptr(out), _ptr(stats), B, ns, h, w,
                                                   C, 1 if src_batched else 0, _stream()), "lwg_lwb_attention_x_f32")
    return out


def instnorm_finalize_ws(B, C, nrec):
    """Floats of the record buffer ``instnorm_finalize`` takes: the (B, nrec, C, 3) records + scratch for the segment partials of long lists."""
    return int(_lib.lib().lwg_instnorm_finalize_ws_floats(int(B), int(C), int(nrec)))


def instnorm_finalize(ws, B, C, nrec, mean, rstd, eps=1e-5):
    """(B, nrec, C, 3) records (count, mean, M2) -> mean, rstd (B, C) of nn.InstanceNorm2d (biased variance).  ws: >= instnorm_finalize_ws floats."""
    assert ws.numel() >= instnorm_finalize_ws(B, C, nrec)
    _lib.check(_lib.lib().lwg_instnorm_finalize_f32(_ptr(ws), B, C, nrec, float(eps), _ptr(mean), _ptr(rstd), _stream()), "lwg_instnorm_finalize_f32")


def lwb_fuse(tsf_x, src_x, T, out, gate=None, scale_w=1.0, scale_o=1.0, src_batched=False):
    """out = (tsf_x + gate * scale_w * sum_s warp_s(src_x)) * scale_o - AddLWB / AvgLWB / SoftGateLWB fusion."""
    B, h, w, C = tsf_x.shape
    ns, S = T.shape[1], T.shape[2]
    assert T.shape[0] == B and src_x.shape[0] == (B * ns if src_batched else ns) and tuple(src_x.shape[1:]) == (h, w, C)
    _lib.check(_lib.lib().lwg_lwb_fuse_f32(_ptr(tsf_x), _ptr(src_x), _ptr(gate), _ptr(T), _ptr(out), B, ns, h, w, C, S,
                                           1 if src_batched else 0, float(scale_w), float(scale_o), _stream()), "lwg_lwb_fuse_f32")
    return out


def lwb_attention_bwd(q, Ks, Vs, bk, bv, T, dout, src_batched=False):
    """Gradients of ``lwb_attention`` w.r.t. q, Ks, Vs (the flows are constants).  dbv = colsum(dout), dbk = 0."""
    B, h, w, C = q.shape
    ns, S = T.shape[1], T.shape[2]
    dout = dout.contiguous()
    dq = torch.empty_like(q)
    dKs = torch.zeros_like(Ks)
    dVs = torch.zeros_like(Vs)
    _lib.check(_lib.lib().lwg_lwb_attention_bwd_f32(_ptr(q), _ptr(Ks), _ptr(Vs), _ptr(bk), _ptr(bv), _ptr(T), _ptr(dout), _ptr(dq),
                                                     _ptr(dKs), _ptr(dVs), B, ns, h, w, C, S, 1 if src_batched else 0, _stream()),
               "lwg_lwb_attention_bwd_f32")
    return dq, dKs, dVs


def lwb_attention_kv(q, kv, bk, bv, T, out, src_batched=False):
    """``lwb_attention`` with K | V as one tensor kv (nsrc,h,w,2C) (the stacked fk | fv projection of the training step)."""
    B, h, w, C = q.shape
    ns, S = T.shape[1], T.shape[2]
    assert T.shape[0] == B and kv.shape[0] == (B * ns if src_batched else ns) and tuple(kv.shape[1:]) == (h, w, 2 * C)
    _lib.check(_lib.lib().lwg_lwb_attention_kv_f32(_ptr(q), _ptr(kv), _ptr(bk), _ptr(bv), _ptr(T), _ptr(out), B, ns, h, w, C, S,
                                                    1 if src_batched else 0, _stream()), "lwg_lwb_attention_kv_f32")
    return out


def lwb_attention_kv_bwd(q, kv, bk, bv, T, dout, src_batched=False):
    """Gradients of ``lwb_attention_kv`` w.r.t. q and kv -> (dq, dkv)."""
    B, h, w, C = q.shape
    ns, S = T.shape[1], T.shape[2]
    dout = dout.contiguous()
    dq = torch.empty_like(q)
    dkv = torch.zeros_like(kv)
    _lib.check(_lib.lib().lwg_lwb_attention_kv_bwd_f32(_ptr(q), _ptr(kv), _ptr(bk), _ptr(bv), _ptr(T), _ptr(dout), _ptr(dq), _ptr(dkv),
                                                        B, ns, h, w, C, S, 1 if src_batched else 0, _stream()), "lwg_lwb_attention_kv_bwd_f32")
    return dq, dkv


def project_faces(verts, cam, faces, want_faces_v=True, want_f2pts=True):
    B, nv, _ = verts.shape
    nf = faces.shape[0]
    fv = torch.empty(B, nf, 3, 3, device=verts.device, dtype=torch.float32) if want_faces_v else None
    f2 = torch.empty(B, nf, 3, 2, device=verts.device, dtype=torch.float32) if want_f2pts else None
    cam = cam.contiguous()
    _lib.check(_lib.lib().lwg_project_faces_f32(_ptr(verts), _ptr(cam), _ptr(faces, torch.int32), B, nv, nf, EYE_DIST, _ptr(fv),
                                                 _ptr(f2), _stream()), "lwg_project_faces_f32")
    return fv, f2


def rasterize_fim_wim(faces_v, image_size, near=0.1, far=100.0):
    B, nf = faces_v.shape[0], faces_v.shape[1]
    S = int(image_size)
    dev = faces_v.device
    fim = torch.empty(B, S, S, device=dev, dtype=torch.int32)
    wim = torch.empty(B, S, S, 3, device=dev, dtype=torch.float32)
    ws = torch.empty(_lib.lib().lwg_rasterize_ws_bytes(B, nf, S), device=dev, dtype=torch.uint8)
    _lib.check(_lib.lib().lwg_rasterize_fim_wim_f32(_ptr(faces_v), B, nf, S, near, far, _ptr(fim, torch.int32), _ptr(wim),
                                                     ws.data_ptr(), _stream()), "lwg_rasterize_fim_wim_f32")
    return fim, wim


def texture_sample(fim, wim, faces_v, textures, eps=1e-3, background_color=(0.0, 0.0, 0.0)):
    """rgb (B,S,S,3) from the index / weight maps, faces_v (B,nf,3,3) and per-face textures (B|1,nf,T,T,T,3)."""
    B, S = fim.shape[0], fim.shape[1]
    nf, T = faces_v.shape[1], textures.shape[2]
    assert textures.shape[0] in (1, B) and textures.shape[1] == nf
    rgb = torch.empty(B, S, S, 3, device=fim.device, dtype=torch.float32)
    bg = (ctypes.c_float * 3)(*[float(c) for c in background_color])
    _lib.check(_lib.lib().lwg_texture_sample_f32(_ptr(fim, torch.int32), _ptr(wim), _ptr(faces_v.contiguous()), _ptr(textures.contiguous()), B, nf,
                                                 S, T, 1 if textures.shape[0] == B and B > 1 else (1 if textures.shape[0] == B else 0),
                                                 float(eps), bg, _ptr(rgb), _stream()), "lwg_texture_sample_f32")
    return rgb


def flow_compose(fim, wim, map_fn, f_uvs2img, uv_img4, src_f2pts, want_cond=False, want_tuv=False):
    B, S, _ = fim.shape
    nf = f_uvs2img.shape[0]
    ns = src_f2pts.shape[0]
    dev = fim.device
    tsf = torch.empty(B, S, S, 8, device=dev, dtype=torch.float32)
    Tst = torch.empty(B, ns, S, S, 2, device=dev, dtype=torch.float32)
    cond = torch.empty(B, 3, S, S, device=dev, dtype=torch.float32) if want_cond else None
    tuv = torch.empty(B, S, S, 2, device=dev, dtype=torch.float32) if want_tuv else None
    Hu, Wu = uv_img4.shape[0], uv_img4.shape[1]
    _lib.check(_lib.lib().lwg_flow_compose_f32(_ptr(fim, torch.int32), _ptr(wim), B, S, _ptr(map_fn), nf, _ptr(f_uvs2img),
                                                _ptr(uv_img4), Hu, Wu, _ptr(src_f2pts), ns, _ptr(tsf), _ptr(Tst), _ptr(cond),
                                                _ptr(tuv), _stream()), "lwg_flow_compose_f32")
    return tsf, Tst, cond, tuv


def bc_transform(f2pts, fim, wim):
    B, S, _ = fim.shape
    nf = f2pts.shape[1]
    T = torch.empty(B, S, S, 2, device=fim.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_bc_transform_f32(_ptr(f2pts), _ptr(fim, torch.int32), _ptr(wim), B, S, nf, _ptr(T), _stream()),
               "lwg_bc_transform_f32")
    return T


def encode_fim(fim, map_fn):
    B, S, _ = fim.shape
    D = map_fn.shape[1]
    out = torch.empty(B, D, S, S, device=fim.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_encode_fim_f32(_ptr(fim, torch.int32), _ptr(map_fn), B, S, map_fn.shape[0] - 1, D, _ptr(out),
                                              _stream()), "lwg_encode_fim_f32")
    return out


def head_compose(x, wpk, bg, want_pred=True, want_mask=True, want_img=False, q4=False):
    """q4: x is (B, C/4, S, S, 4) fp32, channel-quad planes (a convolution launched with q4=True wrote it)."""
    if q4:
        B, Cq, S, _, four = x.shape
        C = 4 * Cq
        assert four == 4 and x.dtype == torch.float32 and x.is_contiguous()
    else:
        B, S, _, C = x.shape
    dev = x.device
    pred = torch.empty(B, 3, S, S, device=dev, dtype=torch.float32) if want_pred else None
    mask = torch.empty(B, 1, S, S, device=dev, dtype=torch.float32) if want_mask else None
    img = torch.empty(B, 3, S, S, device=dev, dtype=torch.float32) if want_img else None
    bstride = 0
    if bg is not None and bg.shape[0] != 1:
        assert bg.shape[0] == B
        bstride = 3 * S * S
    if x.dtype == torch.bfloat16:                    # wpk: packing.pack_head_bf16
        _lib.check(_lib.lib().lwg_head_compose_bf16(_ptr(x, torch.bfloat16), _ptr(wpk, torch.bfloat16), _ptr(bg), bstride, B, S, C, _ptr(pred),
                                                     _ptr(mask), _ptr(img), _stream()), "lwg_head_compose_bf16")
        return pred, mask, img
    if q4:
        _lib.check(_lib.lib().lwg_head_compose_q4_f32(_ptr(x), _ptr(wpk), _ptr(bg), bstride, B, S, C, _ptr(pred), _ptr(mask), _ptr(img),
                                                       _stream()), "lwg_head_compose_q4_f32")
        return pred, mask, img
    _lib.check(_lib.lib().lwg_head_compose_f32(_ptr(x), _ptr(wpk), _ptr(bg), bstride, B, S, C, _ptr(pred), _ptr(mask), _ptr(img),
                                                _stream()), "lwg_head_compose_f32")
    return pred, mask, img


BF16_UP4_HEAD = True    # lab switch: False = the bf16 engine's last up-sampling layer and its output head as two launches (rounds 2-5)


def up4_head_eligible(x, specs, act):
    """The fused last stage of the bf16 engine (lwg_up4_head_compose_bf16): ConvTranspose2d(128 -> 64, 4, 2, 1) + ReLU on a bf16 input."""
    s0 = specs[0]
    return (BF16_UP4_HEAD and BF16_UP4 and BF16_HR and x.is_cuda and x.dtype == torch.bfloat16 and len(specs) == 4 and s0.Cin == 128 and s0.N == 64
            and act == ACT_RELU and s0.bias is not None
            and all(s.ntaps == 4 and s.omul == 2 and (s.ooy, s.oox) == (i >> 1, i & 1) for i, s in enumerate(specs)))


def up4_head_compose_bf16(x, specs, head16, bg, want_pred=True, want_mask=True, want_img=False):
    """BASELINE configs[3]'s last stage as ONE launch (csrc/up4_head_bf16.hip): x (B,H,W,128) bf16 -> ReLU(ConvTranspose2d(4, 2, 1)) (B,2H,2W,64), never
    written -> the 5x5 regressors + tanh / sigmoid + compositing -> (pred, mask, img) fp32 NCHW at (2H, 2W), as ``head_compose`` returns them.
    specs: the layer's four parity specs (packing.pack_conv_transpose); head16: packing.pack_head_bf16's panel."""
    B, H, W, _ = x.shape
    s0 = specs[0]
    dev = x.device
    S2h, S2w = 2 * H, 2 * W
    pred = torch.empty(B, 3, S2h, S2w, device=dev, dtype=torch.float32) if want_pred else None
    mask = torch.empty(B, 1, S2h, S2w, device=dev, dtype=torch.float32) if want_mask else None
    img = torch.empty(B, 3, S2h, S2w, device=dev, dtype=torch.float32) if want_img else None
    bstride = 0
    if bg is not None and bg.shape[0] != 1:
        assert bg.shape[0] == B
        bstride = 3 * S2h * S2w
    # the layer's output does not exist: a zero-batch tensor gives the launch description its geometry
    a = conv_args(x, s0, torch.empty(0, S2h, S2w, s0.N, device=dev, dtype=torch.bfloat16), act=ACT_RELU, out_hw=(H, W))
    panel = getattr(s0, "_w16up", None)
    if panel is None or panel.device != s0.w.device:
        panel = torch.stack([_w16hr(s, False)[0] for s in specs]).contiguous()
        s0._w16up = panel
    a.w = _ptr(panel, torch.bfloat16)
    if CONV_HOOK is not None:          # accounted as the transposed convolution it contains (16 taps, 4 N outputs per input pixel) + nothing for the head (as before)
        whole = _FusedTransposeSpec(s0, panel)
        CONV_HOOK(True, a.M, whole, EPI_NONE, None)
    _lib.check(_lib.lib().lwg_up4_head_compose_bf16(a, _ptr(head16, torch.bfloat16), _ptr(bg), bstride, _ptr(pred), _ptr(mask), _ptr(img), _stream()),
               "lwg_up4_head_compose_bf16")
    if CONV_HOOK is not None:
        per = H * W * 256
        CONV_HOOK(False, a.M, whole, EPI_NONE, {"kernels": max(1, -(-B // max(1, (0xC0000000 - 1) // per))), "kind": "up4_head"})
    return pred, mask, img


def thin_conv(x, wpk, ks):
    """Stride-1 ks x ks convolution (pad ks // 2, no bias) with <= 4 outputs: x (B,S,S,C) NHWC, wpk (ks*ks, C, 4) -> (B,S,S,4)
    pre-activation (csrc/head.hip lwg_thin_conv_f32)."""
    B, S, S2, C = x.shape
    assert S == S2 and tuple(wpk.shape) == (ks * ks, C, 4), (x.shape, wpk.shape, ks)
    y = torch.empty(B, S, S, 4, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_thin_conv_f32(_ptr(x), _ptr(wpk), B, S, C, ks, _ptr(y), _stream()), "lwg_thin_conv_f32")
    return y


def nchw_to_nhwc(x, c_pad=None):
    B, C = x.shape[0], x.shape[1]
    H, W = x.shape[2], x.shape[3]
    Cp = C if c_pad is None else c_pad
    y = torch.empty(B, H, W, Cp, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_nchw_to_nhwc_f32(_ptr(x), _ptr(y), B, C, Cp, H * W, _stream()), "lwg_nchw_to_nhwc_f32")
    return y


def nhwc_to_nchw(x, channels=None):
    B, H, W, Cs = x.shape
    C = Cs if channels is None else channels
    y = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_nhwc_to_nchw_f32(_ptr(x), _ptr(y), B, C, Cs, H * W, _stream()), "lwg_nhwc_to_nchw_f32")
    return y


def smpl_lbs(model, pose, beta, cam, offsets=None, links=None):
    """model: dict of device buffers (v_template, shapedirs, posedirs, J_regressor, parents int32, lbs_weights)."""
    B = pose.shape[0]
    nv = model["v_template"].shape[0]
    nj = model["J_regressor"].shape[0]
    dev = pose.device
    assert pose.shape[1] == 3 * nj, (pose.shape, nj)
    verts = torch.empty(B, nv, 3, device=dev, dtype=torch.float32)
    j3d = torch.empty(B, nj, 3, device=dev, dtype=torch.float32)
    j2d = torch.empty(B, nj, 2, device=dev, dtype=torch.float32) if cam is not None else None
    ws = torch.empty(_lib.lib().lwg_smpl_lbs_ws_floats(B, nv, nj), device=dev, dtype=torch.float32)
    off_batched = 0
    if offsets is not None:
        off_batched = 1 if offsets.dim() == 3 else 0
        if off_batched:
            assert offsets.shape[0] == B
    nlinks = 0 if links is None else links.shape[0]
    nbeta = beta.shape[1]
    _lib.check(_lib.lib().lwg_smpl_lbs_f32(
        _ptr(pose), pose.shape[1], _ptr(beta), nbeta, nbeta, _ptr(cam), 0 if cam is None else cam.shape[1],
        _ptr(model["v_template"]), _ptr(offsets), off_batched, _ptr(model["shapedirs"]), _ptr(model["posedirs"]),
        _ptr(model["J_regressor"]), _ptr(model["parents"], torch.int32), _ptr(model["lbs_weights"]),
        _ptr(links, torch.int32), nlinks, B, nv, nj, _ptr(verts), _ptr(j3d), _ptr(j2d), _ptr(ws), _stream()),
        "lwg_smpl_lbs_f32")
    return verts, j3d, j2d


# ---------------------------------------------------------------------------------------------- source_setup stage
MORPH_MODES = {"erode": 0, "dilate": 1, "soft_dilate": 2}


def morph(x, ks, mode="erode"):
    """(n,1,H,W) mask -> morphed mask (tools/utils/morphology/morph_ops.py:7-63)."""
    n, c, H, W = x.shape
    assert c == 1
    x = x.contiguous().float()
    out = torch.empty_like(x)
    ws = torch.empty_like(x)
    _lib.check(_lib.lib().lwg_morph_f32(_ptr(x), _ptr(out), n, H, W, int(ks), MORPH_MODES[mode], _ptr(ws), _stream()),
               "lwg_morph_f32")
    return out


def canny_edges(sil, gauss9, sobelx9, low, high):
    """(n,1,H,W) -> thin edges in {0,1} (canny_ops.py:137-212 with hysteresis).  gauss9/sobelx9: 9 python floats."""
    import ctypes
    n, c, H, W = sil.shape
    assert c == 1
    sil = sil.contiguous().float()
    edges = torch.empty_like(sil)
    ws = torch.empty(3 * sil.numel(), device=sil.device, dtype=torch.float32)
    g = (ctypes.c_float * 9)(*[float(v) for v in gauss9])
    s = (ctypes.c_float * 9)(*[float(v) for v in sobelx9])
    _lib.check(_lib.lib().lwg_canny_f32(_ptr(sil), n, H, W, g, s, float(low), float(high), _ptr(edges), _ptr(ws), _stream()),
               "lwg_canny_f32")
    return edges


def boundary_fill(src, confidant, outpad, edges, want_top3=False):
    """flowcomposition.py:268-386 body -> (morph_img (n,3,H,W), edge counts (n,) int32 device, top3 or None)."""
    n, _, H, W = src.shape
    src, confidant, outpad, edges = (t.contiguous().float() for t in (src, confidant, outpad, edges))
    out = torch.empty_like(src)
    top3 = torch.empty(n, 3, H, W, device=src.device, dtype=torch.int32) if want_top3 else None
    ws = torch.empty(n * (H * W + 1), device=src.device, dtype=torch.int32)
    _lib.check(_lib.lib().lwg_boundary_fill_f32(_ptr(src), _ptr(confidant), _ptr(outpad), _ptr(edges), n, H, W, _ptr(out),
                                                 _ptr(top3, torch.int32), _ptr(ws, torch.int32), _stream()),
               "lwg_boundary_fill_f32")
    return out, ws[n * H * W:], top3


def grid_sample(img, grid):
    """F.grid_sample(img (n|1,C,H,W), grid (n,Ho,Wo,2)) bilinear / zeros / align_corners=False."""
    n, Ho, Wo, _ = grid.shape
    nb, C, H, W = img.shape
    assert nb in (1, n)
    img, grid = img.contiguous().float(), grid.contiguous().float()
    out = torch.empty(n, C, Ho, Wo, device=img.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_grid_sample_nchw_f32(_ptr(img), 0 if (nb == 1 and n > 1) else C * H * W, _ptr(grid), n, C, H, W,
                                                    Ho, Wo, _ptr(out), _stream()), "lwg_grid_sample_nchw_f32")
    return out


def uv_merge(src_warp, vis):
    """flowcomposition.py:123-130 for one batch item: (ns,3,H,W), (ns,1,H,W) -> (3,H,W)."""
    ns, _, H, W = src_warp.shape
    out = torch.empty(3, H, W, device=src_warp.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_uv_merge_f32(_ptr(src_warp.contiguous()), _ptr(vis.contiguous()), ns, H, W, _ptr(out), _stream()),
               "lwg_uv_merge_f32")
    return out


def uv_merge_parts(uv_imgs, vis):
    """flowcomposition.py:816-856 (merge_uv_img): (n,3,H,W), (n,1,H,W) -> (1,3,H,W)."""
    n, _, H, W = uv_imgs.shape
    out = torch.empty(1, 3, H, W, device=uv_imgs.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_uv_merge_parts_f32(_ptr(uv_imgs.contiguous()), _ptr(vis.contiguous()), n, H, W, _ptr(out), _stream()),
               "lwg_uv_merge_parts_f32")
    return out


def pack_inputs(a, b, mask, c_pad):
    """cat[a * mask, b] (NCHW) -> NHWC with c_pad channels."""
    n, Ca, H, W = a.shape
    Cb = 0 if b is None else b.shape[1]
    out = torch.empty(n, H, W, c_pad, device=a.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_pack_inputs_f32(_ptr(a.contiguous()), Ca, _ptr(None if b is None else b.contiguous()), Cb,
                                               _ptr(None if mask is None else mask.contiguous()), n, H, W, c_pad, _ptr(out),
                                               _stream()), "lwg_pack_inputs_f32")
    return out


def frames_to_u8(pred, bgr=False):
    """(B,3,S,S) fp32 in [-1,1] -> (B,S,S,3) uint8 with save_cv2_img(normalize=True)'s numerics (cv_utils.py:111-113)."""
    B, C, S, S2 = pred.shape
    assert C == 3 and S == S2
    out = torch.empty(B, S, S, 3, device=pred.device, dtype=torch.uint8)
    if B == 0:               # an empty shard / all-padding chunk (sharding.sharded_synthesize): nothing to launch (a zero-size grid is invalid)
        return out
    _lib.check(_lib.lib().lwg_frames_to_u8(_ptr(pred), B, S, 1 if bgr else 0, _ptr(out, torch.uint8), _stream()), "lwg_frames_to_u8")
    return out


# ---------------------------------------------------------------------------------------------- backward of the convs
def conv2d_wgrad(x0, spec, dy, x1=None, out_hw=None, ycoff=0):
    """dW of the conv described by ``spec`` for inputs x0 (x1) and output gradient dy (laid out like the forward y) ->
    (ntaps * Cin, N) fp32 in the forward panel's K order (see packing.unpack_wgrad)."""
    a = conv_args(x0, spec, dy, x1=x1, out_hw=out_hw, ycoff=ycoff)
    Ktot = spec.ntaps * spec.Cin
    dw = torch.empty(Ktot, spec.N, device=x0.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().lwg_conv2d_wgrad_ws_floats(Ktot, spec.N, a.M), device=x0.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_conv2d_wgrad_nhwc_f32(a, _ptr(dy), _ptr(dw), _ptr(ws), _stream()), "lwg_conv2d_wgrad_nhwc_f32")
    return dw


def conv2d_wgrad_unpacked(x0, spec, dy, dw, transposed, kidx, cin, nout, x1=None, out_hw=None, ycoff=0, db=None):
    """``conv2d_wgrad`` + ``unpack_wgrad`` in one reduction launch: the weight gradient lands at positions kidx of dw (D0,D1,KH,KW).
    db: optional (nout,) tensor that receives the bias gradient (column sums of dy) from the same two launches."""
    a = conv_args(x0, spec, dy, x1=x1, out_hw=out_hw, ycoff=ycoff)
    Ktot = spec.ntaps * spec.Cin
    D0, D1, KH, KW = dw.shape
    assert dw.is_contiguous() and len(kidx) == spec.ntaps
    ws = torch.empty(_lib.lib().lwg_conv2d_wgrad_ws_floats(Ktot, spec.N, a.M), device=x0.device, dtype=torch.float32)
    arr = (ctypes.c_int * spec.ntaps)(*[int(k) for k in kidx])
    if db is not None:
        assert db.is_contiguous() and db.dtype == torch.float32 and db.numel() == nout
    _lib.check(_lib.lib().lwg_conv2d_wgrad_unpacked_f32(a, _ptr(dy), _ptr(ws), _ptr(dw), D0, D1, KH, KW, 1 if transposed else 0, arr, cin, nout,
                                                        None if db is None else _ptr(db), _stream()), "lwg_conv2d_wgrad_unpacked_f32")
    return dw


class PanelCache:
    """The panels of a training step, re-packed by ONE launch per step (lwg_pack_panels_f32) instead of one launch per panel.

    While a cache is installed (``ops.PANEL_CACHE``; trainers.LWGTrainer installs its own for the duration of a step) ``pack_panel``
    hands out a persistent output buffer per (weight storage, packing arguments): the first request packs it with a single launch
    and registers it, later requests return the buffer untouched - it was refreshed from the current weights by ``refresh()`` at the
    start of the step (no weight changes between that point and its last use within a step: Adam(G) runs after G's backward,
    Adam(D) at the end).  The weights are held by reference, so a registered address is never recycled."""

    def __init__(self, params=()):
        """params: the tensors whose STORAGE may be cached from (module parameters / flat optimizer buffers).  A weight assembled per
        call (e.g. the concatenated image + mask regressors of the fused head's backward) lives in a temporary: its address
        changes every step, so it is packed by a single launch each time and never registered."""
        self.out, self.rows, self.keep, self.table, self.blocks = {}, [], [], None, 0
        params = list(params)
        self.storages = {p.untyped_storage().data_ptr() for p in params}
        self.training_storages = {p.untyped_storage().data_ptr() for p in params if p.requires_grad}    # (pack_panel sees detached views)
        # Winograd fragment panels derived from registered GEMM panels (ops._wwino): key (panel address, tap order) -> U; the ones whose
        # weights train are re-derived by ONE launch per step behind the GEMM panels' refresh (lwg_winograd_panels_f32)
        self.src_of, self.wino, self.wino_rows, self.wino_table, self.wino_blocks = {}, {}, [], None, 0
        # frozen weights (the loss networks): packed once - but watched: [weight, tensor version at pack time, cache key]; refresh() re-packs a
        # panel whose weight was written since (load_state_dict / an un-freeze after the first step) instead of training against a stale one
        self.frozen = []

    def cacheable(self, w):
        return w.untyped_storage().data_ptr() in self.storages

    def get(self, w, transposed, kidx, cin, cin_pad, nout, n_pad):
        key = (w.data_ptr(), tuple(w.shape), bool(transposed), kidx, cin, cin_pad, nout, n_pad)
        hit = self.out.get(key)
        if hit is not None:
            return hit, False
        if self.table is not None and torch.cuda.is_current_stream_capturing() and w.untyped_storage().data_ptr() in self.training_storages:
            raise RuntimeError("PanelCache: a panel was requested for the first time inside a hipGraph capture (the table upload is "
                               "not capturable): run one eager step before capturing")
        D0, D1, KH, KW = w.shape
        Kp = (len(kidx) * cin_pad + 31) // 32 * 32
        out = torch.empty(Kp // 4, n_pad, 4, device=w.device, dtype=torch.float32)
        self.out[key] = out
        self.src_of[out.data_ptr()] = w
        self.keep.append(w)
        if w.untyped_storage().data_ptr() in self.training_storages:     # a frozen weight (the loss networks) is packed once, by the caller's launch
            self.rows.append((w.data_ptr(), out.data_ptr(), D1, KH * KW, 1 if transposed else 0, len(kidx), cin, cin_pad, nout, n_pad, Kp, kidx))
            self.table = None
        else:
            self.frozen.append([w, w._version, key])
        return out, True

    def _repack_stale_frozen(self):
        """Frozen weights written in place since their panels were packed: single launches, eager only (a captured step cannot see the host check:
        ``invalidate()`` + a re-capture is the route there, and the trainer re-captures when this raises)."""
        stale = [f for f in self.frozen if f[0]._version != f[1]]
        if not stale:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("PanelCache: a frozen weight changed (load_state_dict / un-freeze) - its panels cannot be re-packed inside a hipGraph "
                               "capture: call refresh() eagerly once (or invalidate()) before capturing")
        for f in stale:
            w, _, key = f
            _, _, transposed, kidx, cin, cin_pad, nout, n_pad = key
            out = self.out[key]
            D0, D1, KH, KW = w.shape
            arr = (ctypes.c_int * len(kidx))(*kidx)
            _lib.check(_lib.lib().lwg_pack_panel_f32(_ptr(w), D0, D1, KH, KW, 1 if transposed else 0, arr, len(kidx), cin, cin_pad, nout, n_pad,
                                                     _ptr(out), _stream()), "lwg_pack_panel_f32")
            for (pp, tap9), U in self.wino.items():
                if pp == out.data_ptr():
                    arr9 = (ctypes.c_int * 9)(*tap9)
                    _lib.check(_lib.lib().lwg_winograd_panel_f32(_ptr(out), _ptr(U), cin_pad, out.shape[1], arr9, _stream()), "lwg_winograd_panel_f32")
            f[1] = w._version

    def invalidate(self):
        """Forget every panel (weights re-allocated or reloaded wholesale): the next requests pack afresh."""
        self.out, self.rows, self.keep, self.table, self.blocks, self.frozen = {}, [], [], None, 0, []
        self.src_of, self.wino, self.wino_rows, self.wino_table, self.wino_blocks = {}, {}, [], None, 0

    def refresh(self):
        """Re-pack every registered panel of a weight that trains (requires_grad when the cache was built) from the current weights: one launch
        on the current stream, one more for the Winograd panels derived from them.  Panels of frozen weights are built once and re-packed only
        when the weight's tensor version moved since (an in-place load_state_dict)."""
        self._repack_stale_frozen()
        if not self.rows:
            return
        if self.table is None:
            descs = (_lib.LwgPackDesc * len(self.rows))()
            first = 0
            for d, (wp, op, D1, KHW, tr, nt, cin, cp, nout, npad, Kp, kidx) in zip(descs, self.rows):
                d.w, d.out = wp, op
                d.D1, d.KHW, d.transposed, d.ntaps, d.cin, d.cin_pad, d.nout, d.n_pad, d.Kp, d.first_block = D1, KHW, tr, nt, cin, cp, nout, npad, Kp, first
                for i, k in enumerate(kidx):
                    d.kidx[i] = k
                first += ((Kp // 4) * npad + 255) // 256
            raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
            self.table = raw.to(self.keep[0].device)
            self.blocks = first
        _lib.check(_lib.lib().lwg_pack_panels_f32(self.table.data_ptr(), len(self.rows), self.blocks, _stream()), "lwg_pack_panels_f32")
        if self.wino_rows:
            if self.wino_table is None:
                descs = (_lib.LwgWinoDesc * len(self.wino_rows))()
                first = 0
                for d, (wp, up, cin, N, tap9) in zip(descs, self.wino_rows):
                    d.wpanel, d.upk, d.Cin, d.N, d.first_block = wp, up, cin, N, first
                    for i in range(9):
                        d.tap9[i] = tap9[i]
                    first += ((N + 63) // 64) * ((cin + 15) // 16)
                raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
                self.wino_table = raw.to(self.keep[0].device)
                self.wino_blocks = first
            _lib.check(_lib.lib().lwg_winograd_panels_f32(self.wino_table.data_ptr(), len(self.wino_rows), self.wino_blocks, _stream()),
                       "lwg_winograd_panels_f32")

    def winograd(self, spec, tap9):
        """The fragment panel of a spec whose GEMM panel is registered here, or None (a per-call temporary: single launch each time).  First
        request: allocated, built by a single launch and - when the source weight trains - registered for the per-step refresh."""
        w = self.src_of.get(spec.w.data_ptr())
        if w is None:
            return None
        key = (spec.w.data_ptr(), tuple(tap9))
        U = self.wino.get(key)
        if U is None:
            trains = w.untyped_storage().data_ptr() in self.training_storages
            if self.wino_table is not None and trains and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("PanelCache: a Winograd panel was requested for the first time inside a hipGraph capture: run one eager step first")
            K4, N, _ = spec.w.shape
            U = torch.empty(16, spec.Cin // 8, 2, N, 4, device=spec.w.device, dtype=torch.float32)
            arr = (ctypes.c_int * 9)(*tap9)
            _lib.check(_lib.lib().lwg_winograd_panel_f32(_ptr(spec.w), _ptr(U), spec.Cin, N, arr, _stream()), "lwg_winograd_panel_f32")
            self.wino[key] = U
            if trains:                                       # frozen weights (the loss networks): built once
                self.wino_rows.append((spec.w.data_ptr(), U.data_ptr(), spec.Cin, N, tuple(tap9)))
                self.wino_table = None
        return U


PANEL_CACHE = None      # a PanelCache while a trainer step runs (trainers.LWGTrainer), else None: every pack_panel call launches
BRANCH_STREAM = None    # a torch.cuda.Stream while a trainer step runs the background network next to the source / transfer streams


def pack_panel(w, transposed, kidx, cin, cin_pad, nout, n_pad):
    """Weight (D0,D1,KH,KW) on the device -> the fp32 GEMM panel [ceil32(ntaps*cin_pad)/4][n_pad][4] in one launch
    (csrc/train_ops.hip lwg_pack_panel_f32; see include/lwg_hip.h for the index convention)."""
    w = w.detach()
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    D0, D1, KH, KW = w.shape
    kidx = tuple(int(k) for k in kidx)
    ntaps = len(kidx)
    Kp = (ntaps * cin_pad + 31) // 32 * 32
    if PANEL_CACHE is not None and PANEL_CACHE.cacheable(w):
        out, fresh = PANEL_CACHE.get(w, transposed, kidx, cin, cin_pad, nout, n_pad)
        if not fresh:
            return out
    else:
        out = torch.empty(Kp // 4, n_pad, 4, device=w.device, dtype=torch.float32)
    arr = (ctypes.c_int * ntaps)(*[int(k) for k in kidx])
    _lib.check(_lib.lib().lwg_pack_panel_f32(_ptr(w), D0, D1, KH, KW, 1 if transposed else 0, arr, ntaps, cin, cin_pad, nout, n_pad,
                                             _ptr(out), _stream()), "lwg_pack_panel_f32")
    return out


def unpack_wgrad(dwk, dw, transposed, kidx, cin, cin_pad, nout):
    """(ntaps*cin_pad, n_pad) weight gradient in kernel K order -> positions kidx of dw (D0,D1,KH,KW), in place."""
    D0, D1, KH, KW = dw.shape
    ntaps = len(kidx)
    assert dwk.is_contiguous() and dw.is_contiguous() and dwk.shape[0] == ntaps * cin_pad
    arr = (ctypes.c_int * ntaps)(*[int(k) for k in kidx])
    _lib.check(_lib.lib().lwg_unpack_wgrad_f32(_ptr(dwk), D0, D1, KH, KW, 1 if transposed else 0, arr, ntaps, cin, cin_pad, nout,
                                               dwk.shape[1], _ptr(dw), _stream()), "lwg_unpack_wgrad_f32")
    return dw


def maxpool2_fwd(x):
    """MaxPool2d(2, 2) on (B,H,W,C) NHWC."""
    B, H, W, C = x.shape
    y = torch.empty(B, H // 2, W // 2, C, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_maxpool2_fwd_nhwc_f32(_ptr(x), _ptr(y), B, H, W, C, _stream()), "lwg_maxpool2_fwd_nhwc_f32")
    return y


def maxpool2_bwd(x, dy):
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().lwg_maxpool2_bwd_nhwc_f32(_ptr(x), _ptr(dy.contiguous()), _ptr(dx), B, H, W, C, _stream()), "lwg_maxpool2_bwd_nhwc_f32")
    return dx


def colsum(x2d_or_nhwc):
    """Sum over every dimension but the last (bias gradient)."""
    x = x2d_or_nhwc.contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty(C, device=x.device, dtype=torch.float32)
    ws = torch.empty(512 * C, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().lwg_colsum_nhwc_f32(_ptr(x), rows, C, _ptr(out), _ptr(ws), _stream()), "lwg_colsum_nhwc_f32")
    return out


def act_bwd(dy, y, act):
    """dy * act'(y) for ReLU / LeakyReLU(0.2) / tanh / sigmoid outputs y."""
    dy, y = dy.contiguous(), y.contiguous()
    out = torch.empty_like(dy)
    _lib.check(_lib.lib().lwg_act_bwd_f32(_ptr(dy), _ptr(y), dy.numel(), act, _ptr(out), _stream()), "lwg_act_bwd_f32")
    return out


def crop_resize(x, box, out_hw, want_valid=True):
    """(N,C,H,W) fp32 NCHW, box (N,4) int64 device tensor (min_x, max_x, min_y, max_y) -> (crops (N,C,OH,OW), valid (N) fp32): the head crops of
    FaceLoss (faceloss.py:384-406) resized with bilinear / align_corners = True, the boxes never read by the host (lwg_crop_resize_bilinear_f32)."""
    N, C, H, W = x.shape
    OH, OW = out_hw
    x = x.contiguous()
    box = box.contiguous()
    y = x.new_empty(N, C, OH, OW)
    valid = x.new_empty(N) if want_valid else None
    _lib.check(_lib.lib().lwg_crop_resize_bilinear_f32(_ptr(x), _ptr(box, torch.int64), _ptr(y), _ptr(valid), N, C, H, W, OH, OW, _stream()),
               "lwg_crop_resize_bilinear_f32")
    return y, valid


def crop_resize_bwd(dy, box, in_hw):
    """Gradient of ``crop_resize`` with respect to the images: (N,C,OH,OW) -> (N,C,H,W)."""
    N, C, OH, OW = dy.shape
    H, W = in_hw
    dx = dy.new_zeros(N, C, H, W)
    _lib.check(_lib.lib().lwg_crop_resize_bilinear_bwd_f32(_ptr(dy.contiguous()), _ptr(box.contiguous(), torch.int64), _ptr(dx), N, C, H, W, OH, OW,
                                                           _stream()), "lwg_crop_resize_bilinear_bwd_f32")
    return dx


def prelu(x, slope, res=None):
    """(.., C) NHWC fp32, slope (C): res + PReLU(x) in one launch (lwg_prelu_f32; the frozen Sphere20a's activations and residual adds)."""
    x = x.contiguous()
    C = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(_lib.lib().lwg_prelu_f32(_ptr(x), _ptr(slope.contiguous()), None if res is None else _ptr(res.contiguous()), x.numel() // C, C, _ptr(y),
                                        _stream()), "lwg_prelu_f32")
    return y


def prelu_bwd(x, slope, dy):
    """dx of ``prelu`` (the slopes are frozen; the residual's gradient is dy)."""
    x, dy = x.contiguous(), dy.contiguous()
    C = x.shape[-1]
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().lwg_prelu_bwd_f32(_ptr(x), _ptr(slope.contiguous()), _ptr(dy), x.numel() // C, C, _ptr(dx), _stream()), "lwg_prelu_bwd_f32")
    return dx


def _nsplit(hw):
    return max(1, min(64, hw // 64))


def _elem_ptr(t, off):
    """Device address of element ``off`` of a contiguous fp32 tensor (the beta / dbeta half of a fused gamma | beta tensor)."""
    return ctypes.c_void_p(t.data_ptr() + 4 * off)


def norm_fwd(x, gamma=None, beta=None, act=ACT_NONE, eps=1e-5, gb=None):
    """y = act(InstanceNorm(x) * (1 + gamma) + beta) on NHWC -> (y, mean, rstd).
    gb: (B,H,W,2C) = gamma | beta as ONE tensor (the output of the fused mlp_gamma | mlp_beta convolution) instead of gamma, beta."""
    B, H, W, C = x.shape
    x = x.contiguous()
    mean, rstd = x.new_empty(B, C), x.new_empty(B, C)
    ns = _nsplit(H * W)
    instnorm_stats(x, mean, rstd, x.new_empty(B * C * ns * 3), eps=eps, nsplit=ns)
    y = torch.empty_like(x)
    if gb is not None:
        assert gamma is None and beta is None and gb.is_contiguous() and gb.shape == (B, H, W, 2 * C)
        gp, bp, gs = _ptr(gb), _elem_ptr(gb, C), 2 * C
    else:
        gp, bp, gs = _ptr(None if gamma is None else gamma.contiguous()), _ptr(None if beta is None else beta.contiguous()), 0
    _lib.check(_lib.lib().lwg_norm_fwd_nhwc_f32(_ptr(x), _ptr(mean), _ptr(rstd), gp, bp, gs, B, H * W, C, act, _ptr(y),
                                                 _stream()), "lwg_norm_fwd_nhwc_f32")
    return y, mean, rstd


def _nsplit_bwd(hw, B, C):
    """Splits of the reduction pass of norm_bwd: ~512 workgroups per launch (one training sample is B = 1: the statistics' 64 splits
    leave three quarters of the CUs idle), at least 32 pixels per split."""
    per_image = max(1, 512 // max(1, B * ((C // 4 + 63) // 64)))
    return max(1, min(per_image, hw // 32, 512))


def norm_bwd(dy, y, x, mean, rstd, gamma=None, act=ACT_NONE, gb=None):
    """Backward of norm_fwd -> (dx, dgamma, dbeta); with gb (the fused gamma | beta tensor) -> (dx, dgb (B,H,W,2C), None)."""
    B, H, W, C = x.shape
    dy = dy.contiguous()
    ns = _nsplit_bwd(H * W, B, C)
    dx = torch.empty_like(x)
    ws = x.new_empty(B * (ns + 1) * C * 2)          # the split records + their fold
    if gb is not None:
        dgb = torch.empty_like(gb)
        _lib.check(_lib.lib().lwg_norm_bwd_nhwc_f32(_ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gb), 2 * C, B, H * W, C, act, ns,
                                                     _ptr(dx), _ptr(dgb), _elem_ptr(dgb, C), _ptr(ws), _stream()), "lwg_norm_bwd_nhwc_f32")
        return dx, dgb, None
    dg = torch.empty_like(x) if gamma is not None else None
    db = torch.empty_like(x) if gamma is not None else None
    _lib.check(_lib.lib().lwg_norm_bwd_nhwc_f32(_ptr(dy), _ptr(y), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), 0, B, H * W, C, act, ns,
                                                 _ptr(dx), _ptr(dg), _ptr(db), _ptr(ws), _stream()), "lwg_norm_bwd_nhwc_f32")
    return dx, dg, db


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, t_dev):
    """In-place Adam update with the step count on the device: t_dev (1,) int32 is incremented, then used (graph-capturable)."""
    _lib.check(_lib.lib().lwg_adam_step_dev_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, _ptr(t_dev, torch.int32),
                                                _stream()), "lwg_adam_step_dev_f32")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, t):
    """In-place Adam update of the flat fp32 buffers p, m, v with gradient g."""
    _lib.check(_lib.lib().lwg_adam_step_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, int(t), _stream()),
               "lwg_adam_step_f32")
