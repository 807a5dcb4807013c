"""Lab: the fused bf16 transposed convolution (lwg_conv_transpose4_nhwc_bf16) at the last up-sampling layer's shape.
usage: up4lab.py [LIB.so ...]  (each in its own process; prints us / launch, PFLOP/s and a checksum of the output: variants must agree)"""
import sys, os, subprocess, json, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(lib):
    import torch
    from ipercore_amd import _lib
    if lib != "product":
        _lib.LIB_PATH = os.path.abspath(lib)
    from ipercore_amd import ops
    from ipercore_amd.networks import packing
    dev, BF = "cuda:0", torch.bfloat16
    res = {}
    for (B, H, Cin, N) in ((20, 512, 128, 64), (20, 256, 128, 64), (8, 512, 64, 64)):
        g = torch.Generator().manual_seed(5)
        w = (torch.randn(Cin, N, 4, 4, generator=g) * (Cin * 4) ** -0.5).to(BF).float()
        specs = [packing.spec_to(s, dev) for s in packing.pack_conv_transpose(w, 0.1 * torch.randn(N, generator=g))]
        x = torch.randn(B, H, H, Cin, generator=g).to(BF).to(dev)
        y = torch.empty(B, 2 * H, 2 * H, N, device=dev, dtype=BF)
        for _ in range(3):
            ops.conv_transpose2d(x, specs, y, act=ops.ACT_RELU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ops.conv_transpose2d(x, specs, y, act=ops.ACT_RELU)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        flops = 2.0 * B * H * H * Cin * 16 * N
        sha = hashlib.sha256(y[:2].cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
        res[f"{B}x{H}x{H}x{Cin}->{N}"] = {"us": round(us, 1), "PFLOP/s": round(flops / us / 1e9, 3), "sha": sha}
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(sys.argv[2])
    else:
        for lib in (sys.argv[1:] or ["product"]):
            r = subprocess.run([sys.executable, __file__, "--worker", lib], capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            print("==", lib)
            if not line:
                print(r.stdout[-1500:], r.stderr[-3000:])
                continue
            for k, v in json.loads(line[0][7:]).items():
                print(f"  {k:28s} {v}")
