"""Lab: the fused F(2x2, 3x3) Winograd convolution against torch (fp64 on the CPU) and against the direct fp32 MFMA convolution: errors and
launch times.  usage: winolab.py            the probe (tools/probes/winograd_f23.hip -> tools/lab/libwino.so, build command in its header)
       winolab.py LIB.so [...]  the library kernel (ops.conv_precision("winograd")) of each variant library (tools/labvariant.sh NAME
                                conv_winograd.hip -DLWG_WINO_VSTRIDE=96 ...), one process per library"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from ipercore_amd import ops
from ipercore_amd.networks import packing

dev = "cuda:0"
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] != "--lib"):
    import subprocess
    for L in sys.argv[1:]:
        print("==", L, flush=True)
        subprocess.run([sys.executable, __file__, "--lib"], env=dict(os.environ, LWG_WINOLAB_LIB=os.path.abspath(L)))
    sys.exit(0)
USE_LIB = len(sys.argv) == 2
if USE_LIB:
    from ipercore_amd import _lib as _l
    _l.LIB_PATH = os.environ["LWG_WINOLAB_LIB"]
else:
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "lab", "libwino.so"))
    lib.wino_conv3x3_f32.restype = ctypes.c_int
    lib.wino_conv3x3_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)


def wino(x, U, bias, act):
    B, H, W, Cin = x.shape
    N = U.shape[3]
    y = torch.empty(B, H, W, N, device=dev)
    e = lib.wino_conv3x3_f32(x.data_ptr(), U.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W, Cin, N, act, None)
    assert e == 0, e
    return y


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (B, H, W, Cin, N, check) in ((2, 24, 40, 64, 64, True), (1, 17, 31, 32, 128, True), (32, 64, 64, 256, 256, False), (8, 256, 256, 128, 128, False)):
    g = torch.Generator().manual_seed(7)
    w = torch.randn(N, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5
    b = torch.randn(N, generator=g) * 0.1
    x = torch.randn(B, H, W, Cin, generator=g)
    U = torch.einsum("ij,ncjk,lk->ilcn", G, w.double(), G).reshape(16, Cin, N).float()                           # [xi*4+nu][c][n]
    # the probe's fragment panel [16][Cin/8][kh 2][N][kk 4]: channel 8 s + 2 kk + kh
    U = U.view(16, Cin // 8, 4, 2, N).permute(0, 1, 3, 4, 2).contiguous().to(dev)
    xd, bd = x.to(dev), b.to(dev)
    spec = packing.spec_to(packing.pack_conv(w, b, stride=1, pad=1), dev)
    if USE_LIB:
        def wino(xd_, U_, bd_, act_):
            with ops.conv_precision("winograd"):
                return ops.conv2d(xd_, spec, torch.empty(B, H, W, N, device=dev), act=ops.ACT_RELU)
    yw = wino(xd, U, bd, 1)
    yd = ops.conv2d(xd, spec, torch.empty(B, H, W, N, device=dev), act=ops.ACT_RELU)
    torch.cuda.synchronize()
    line = f"B={B} {H}x{W} {Cin}->{N}: max |wino - direct| {float((yw - yd).abs().max()):.2e}"
    if check:
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).relu().permute(0, 2, 3, 1)
        line += f"  vs fp64: wino {float((yw.cpu().double() - ref).abs().max()):.2e}  direct {float((yd.cpu().double() - ref).abs().max()):.2e}  (max |y| {float(ref.abs().max()):.2f})"
    else:
        tw = timeit(lambda: wino(xd, U, bd, 1))
        td = timeit(lambda: ops.conv2d(xd, spec, yd, act=ops.ACT_RELU))
        fl = 2.0 * B * H * W * 9 * Cin * N
        line += f"  time: wino {tw:.3f} ms ({fl / tw / 1e9:.0f} algorithmic TFLOP/s)  direct {td:.3f} ms ({fl / td / 1e9:.0f} TFLOP/s)"
    print(line)
