"""Bisect of the background network's gradient outliers (reference bg_inpaintor.py:24-60,
lwg_trainer.py:326-352).  The sub-graph is `TrainableGenerator.forward_bg` + an L1 loss at 512 x 512, full width; the reference is the
oracle's autograd in fp64.  Per parameter, element-wise error / scale of

  CPU (torch)  : fp32 as is; fp32 with the INPUT perturbed by 1e-7 relative (n seeds) - every fp32 association of the same graph is such a
                 perturbation; fp64 with the same perturbation (what the perturbation alone does to the exact gradient)
  GPU (HIP)    : default; forward / dgrad launches without split-K; InstanceNorm (fwd + bwd) replaced by torch's own (lwg_norm_* out);
                 every convolution's WEIGHT gradient replaced by torch's (lwg_conv2d_wgrad_* out); both
  GPU (torch)  : the same graph on torch-ROCm's own kernels (another fp32 association, none of ours)

usage: python tools/diag_bg_grads.py [S=512] [n_perturb=4]   (CPU part runs without a GPU: prints that table alone)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from ipercore_amd import synthetic
from ipercore_amd.networks import generator_param_shapes
from oracle import lwg_oracle as orc

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nf, nres, bgf = [64, 128, 256], 6, [64, 128, 128, 256]
sdn = {k: v for k, v in synthetic.fill_state_dict(generator_param_shapes(nf, nres, bgf), seed=7).items() if k.startswith("bg_net")}
bg_in = torch.tensor(synthetic.uniform_image((1, 1, 4, S, S), 10, "bg_inputs"))
tgt = torch.tensor(synthetic.uniform_image((1, 1, 3, S, S), 500, "tgt"))


def cpu_run(dt, perturb=0.0, seed=0, device="cpu"):
    sd = {k: torch.tensor(v, dtype=dt, device=device, requires_grad=True) for k, v in sdn.items()}
    x = bg_in.to(dt)
    if perturb:
        g = torch.Generator().manual_seed(seed)
        x = x * (1 + perturb * torch.randn(x.shape, generator=g).to(dt))
    out = orc.gen_forward_bg(sd, x.to(device), n_down=len(bgf), n_res=nres)
    (out - tgt.to(device=device, dtype=dt)).abs().mean().backward()
    return {k: v.grad.double().cpu() for k, v in sd.items()}


def table(title, g64, cols):
    gmax = max(v.abs().max().item() for v in g64.values())
    names = list(cols)
    print(f"\n== {title} (element-wise max |g - g64| / max(max |g64|, 1e-3 gmax); gmax {gmax:.3e})")
    print("param".ljust(30) + " ".join(n[:14].rjust(14) for n in names))
    worst = {n: 0.0 for n in names}
    for k in g64:
        if not k.endswith("weight"):
            continue
        ref = g64[k]
        sc = max(ref.abs().max().item(), 1e-3 * gmax)
        row = []
        for n in names:
            e = (cols[n][k] - ref).abs().max().item() / sc
            worst[n] = max(worst[n], e)
            row.append(f"{e:14.2e}")
        print(k[7:].ljust(30) + " ".join(row))
    print("worst".ljust(30) + " ".join(f"{worst[n]:14.2e}" for n in names), flush=True)
    return worst


def main():
    t0 = time.time()
    g64 = cpu_run(torch.float64)
    print(f"fp64 oracle autograd of the background network at {S}x{S}: {time.time() - t0:.1f} s", flush=True)
    cpu = {"t32": cpu_run(torch.float32)}
    for i in range(NP):
        cpu[f"t32 in*1e-7 #{i}"] = cpu_run(torch.float32, 1e-7, i)
    cpu["f64 in*1e-7 #0"] = cpu_run(torch.float64, 1e-7, 0)
    table("CPU, torch", g64, cpu)
    if not torch.cuda.is_available():
        return
    from ipercore_amd import ops
    from ipercore_amd.networks import NetworksFactory, packing, training
    from tests import parity_utils as pu
    DEV = "cuda:0"
    G = NetworksFactory.get_by_name("AttLWB-SPADE", cfg=pu.gen_cfg(nf, nres, bgf), temporal=False)
    full = synthetic.fill_state_dict(generator_param_shapes(nf, nres, bgf), seed=7)
    G.load_state_dict({k: torch.tensor(v) for k, v in full.items()}, strict=True)
    G.to(DEV).train()

    def torch_instance_norm(x, act=0):
        y = F.instance_norm(x.permute(0, 3, 1, 2), eps=1e-5).permute(0, 2, 3, 1)
        return F.relu(y) if act == 1 else y

    def torch_wgrad_conv(x0, spec, dy, x1, kh, kw, cin, N, db=None):
        x = x0 if x1 is None else torch.cat([x0, x1], dim=3)
        pad = -spec.dy[0]
        dw = torch.nn.grad.conv2d_weight(x[..., :cin].permute(0, 3, 1, 2), (N, cin, kh, kw), dy[..., :N].permute(0, 3, 1, 2), stride=spec.stride, padding=pad)
        if db is not None:
            db.copy_(dy[..., :N].sum(dim=(0, 1, 2)))
        return dw

    def torch_wgrad_convT(x0, specs, dy, cin, n, adj_spec=None):
        # d/dw of conv_transpose2d(x, w, stride 2, padding 1), w (Cin, N, 4, 4): torch autograd on the SAME x / dy, fp32, ROCm kernels
        w = torch.zeros(cin, n, 4, 4, device=dy.device, requires_grad=True)
        with torch.enable_grad():
            y = F.conv_transpose2d(x0[..., :cin].permute(0, 3, 1, 2), w, stride=2, padding=1)
        return torch.autograd.grad(y, w, dy[..., :n].permute(0, 3, 1, 2))[0]

    CAP = {}

    def hip_run(no_splitk=False, torch_norm=False, torch_wgrad=False, torch_wgrad_t=False, capture=False, nosplit_if=None):
        for p_ in G.parameters():
            p_.grad = None
        keep = (ops.conv2d, training.instance_norm, packing.wgrad_conv, packing.wgrad_conv_transpose)
        if torch_wgrad_t:
            packing.wgrad_conv_transpose = torch_wgrad_convT
        if capture:                 # the weight-gradient kernels ALONE: their inputs of this run -> fp64 on the CPU against what they returned
            def cap_t(x0, specs, dy, cin, n, adj_spec=None):
                g = keep[3](x0, specs, dy, cin, n, adj_spec=adj_spec)
                CAP[("convT", cin, n, x0.shape[1])] = (x0.detach().cpu(), dy.detach().cpu(), g.detach().cpu())
                return g

            def cap_c(x0, spec, dy, x1, kh, kw, cin, n, db=None):
                g = keep[2](x0, spec, dy, x1, kh, kw, cin, n, db=db)
                if (kh, spec.stride) in ((3, 1), (4, 2), (3, 2)) and len(CAP) < 12:
                    CAP[("conv", cin, n, x0.shape[1], kh, spec.stride, -spec.dy[0])] = (x0.detach().cpu(), dy.detach().cpu(), g.detach().cpu())
                return g
            packing.wgrad_conv_transpose, packing.wgrad_conv = cap_t, cap_c
        if no_splitk:
            ops.conv2d = lambda *a, **k: keep[0](*a, **{**k, "splitk": False})
        if nosplit_if is not None:      # split-K off for ONE class of launches (a[1] = the ConvSpec): where does the extra error enter?
            ops.conv2d = lambda *a, **k: keep[0](*a, **{**k, "splitk": k.get("splitk", False) and not nosplit_if(a[1], a[0])})
        if torch_norm:
            training.instance_norm = torch_instance_norm
        if torch_wgrad:
            packing.wgrad_conv = torch_wgrad_conv
        try:
            with ops.conv_precision("fp32"):
                x = bg_in.view(1, 4, S, S).permute(0, 2, 3, 1).contiguous().to(DEV)
                out = training.TrainableGenerator(G).forward_bg(x)
                (out.permute(0, 3, 1, 2) - tgt.view(1, 3, S, S).to(DEV)).abs().mean().backward()
            torch.cuda.synchronize()
        finally:
            ops.conv2d, training.instance_norm, packing.wgrad_conv, packing.wgrad_conv_transpose = keep
        return {k: p_.grad.double().cpu() for k, p_ in G.named_parameters() if k.startswith("bg_net")}

    gpu = {"hip": hip_run(), "hip no-splitK": hip_run(no_splitk=True), "hip torch-norm": hip_run(torch_norm=True),
           "hip torch-wgrad": hip_run(torch_wgrad=True), "hip t-norm+wg": hip_run(torch_norm=True, torch_wgrad=True),
           "hip all three": hip_run(True, True, True), "hip t-wgradT": hip_run(torch_wgrad_t=True),
           "hip nosplit+wgT": hip_run(no_splitk=True, torch_wgrad_t=True)}
    gpu["torch-ROCm f32"] = cpu_run(torch.float32, device=DEV)
    gpu["t32 (CPU)"] = cpu["t32"]
    gpu["max perturbed"] = {k: max((cpu[n][k] for n in cpu if "in*1e-7" in n), key=lambda g_: (g_ - g64[k]).abs().max().item()) for k in g64}
    table("GPU", g64, gpu)
    # L2 view of the same columns (a flipped ReLU / L1-sign kink lands on a few elements: large element-wise, small in L2; an arithmetic defect is both)
    gmax = max(v.abs().max().item() for v in g64.values())
    print("\n== GPU, relative L2 error per parameter: ||g - g64|| / ||g64||")
    print("param".ljust(30) + " ".join(n[:14].rjust(14) for n in gpu))
    for k in g64:
        if k.endswith("weight"):
            print(k[7:].ljust(30) + " ".join(f"{((gpu[n][k] - g64[k]).norm() / g64[k].norm()).item():14.2e}" for n in gpu))
    # split-K off for one class of launches at a time (L2 error of three parameters: first layer, a residual block, the first transposed convolution)
    classes = {"none (all split)": lambda sp, x: False, "all": lambda sp, x: True,
               "3x3 s1 (res blocks fwd + dgrad)": lambda sp, x: sp.ntaps == 9 and sp.stride == 1 and sp.omul == 1,
               "16-tap s2 (convT dgrad)": lambda sp, x: sp.ntaps == 16,
               "3x3 s2 fwd": lambda sp, x: sp.ntaps == 9 and sp.stride == 2,
               "omul 2 (s2 dgrad parities / convT fwd parities)": lambda sp, x: sp.omul == 2,
               "launches on 64 x 64 maps": lambda sp, x: x.shape[1] == 64,
               "launches on 128 x 128 maps": lambda sp, x: x.shape[1] == 128,
               "launches on >= 256 maps": lambda sp, x: x.shape[1] >= 256}
    print("\n== split-K off for ONE class of launches: relative L2 error of dW (first layer | res block 12.main.0 | convT main.18 | main.9)")
    for name, pred in classes.items():
        gr = hip_run(nosplit_if=pred)
        print(f"  {name:52s} " + "  ".join(f"{((gr[k] - g64[k]).norm() / g64[k].norm()).item():.2e}" for k in
                                          ("bg_net.main.0.weight", "bg_net.main.12.main.0.weight", "bg_net.main.18.weight", "bg_net.main.9.weight")), flush=True)
    # the weight-gradient kernels alone: captured inputs of one default run, fp64 on the CPU
    hip_run(capture=True)
    print("\n== weight-gradient kernels ALONE (inputs captured from the default run; reference: fp64 on the CPU from the same tensors)")
    for key, (x0, dy, g) in CAP.items():
        if key[0] == "convT":
            _, cin, n, _ = key
            w = torch.zeros(cin, n, 4, 4, dtype=torch.float64, requires_grad=True)
            y = F.conv_transpose2d(x0[..., :cin].double().permute(0, 3, 1, 2), w, stride=2, padding=1)
            ref = torch.autograd.grad(y, w, dy[..., :n].double().permute(0, 3, 1, 2))[0]
        else:
            _, cin, n, _, kh, st, pad = key
            ref = torch.nn.grad.conv2d_weight(x0[..., :cin].double().permute(0, 3, 1, 2), (n, cin, kh, kh), dy[..., :n].double().permute(0, 3, 1, 2), stride=st, padding=pad)
        d = g.double() - ref
        print(f"  {str(key):40s} max |d| / max |ref| {d.abs().max().item() / ref.abs().max().item():.2e}   rel L2 {(d.norm() / ref.norm()).item():.2e}   "
              f"dy: {float((dy != 0).float().mean()):.3f} non-zero, max {dy.abs().max().item():.2e}", flush=True)


if __name__ == "__main__":
    main()
