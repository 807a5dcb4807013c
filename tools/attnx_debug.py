"""Lab: lwg_lwb_attention_x_* against its CPU emulation on identical inputs, in cases that isolate the pieces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ipercore_amd import ops
from tests import emu_ops
DEV = "cuda:0"
torch.manual_seed(0)


def run(tag, B, ns, h, w, C, T, dt=torch.float32, kq_scale=1.0, kap_scale=1.0):
    x = torch.randn(B, h, w, C).to(dt)
    Kq = (kq_scale * torch.randn(ns, h, w, C) / np.sqrt(C)).to(dt)
    Vs = torch.randn(ns, h, w, C).to(dt)
    kap = kap_scale * torch.randn(ns, h, w)
    bv = 0.3 * torch.randn(C)
    want = emu_ops.lwb_attention_x(x.double(), Kq.double(), kap.double(), Vs.double(), bv.double(), T.double(), torch.zeros(B, h, w, C).double())
    nrec = ops.attn_records(h, w, C, dt)
    ws = torch.zeros(ops.instnorm_finalize_ws(B, C, nrec), device=DEV)
    got = ops.lwb_attention_x(x.to(DEV), Kq.to(DEV), kap.to(DEV), Vs.to(DEV), bv.to(DEV), T.to(DEV), torch.full((B, h, w, C), float("nan"), device=DEV, dtype=dt), stats=ws)
    torch.cuda.synchronize()
    err = (got.float().cpu().double() - want).abs()
    pp = err.amax(dim=3)
    print(f"{tag:28s} max {err.max().item():.3e} mean {err.mean().item():.3e} nan {int(torch.isnan(got.float()).sum())}  bad pixels {(pp > 1e-3).sum().item()} of {pp.numel()}",
          "first bad", (pp > 1e-3).nonzero()[:4].tolist())
    return got, want


B, ns, h, w, C = 2, 2, 16, 16, 64
r = np.random.RandomState(0)
Tb = torch.full((B, ns, h, w, 2), -2.0)
run("all background", B, ns, h, w, C, Tb)
# identity flow: pixel centres
ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
Ti = torch.stack([(2 * xs + 1) / w - 1, (2 * ys + 1) / h - 1], dim=-1).float()
Tid = Ti[None, None].expand(B, ns, h, w, 2).contiguous()
run("identity ns=1 (V gather)", B, 1, h, w, C, Tid[:, :1].contiguous(), kq_scale=0.0, kap_scale=0.0)
run("identity ns=2 kq=0 kap=0", B, ns, h, w, C, Tid, kq_scale=0.0, kap_scale=0.0)
run("identity ns=2 kap only", B, ns, h, w, C, Tid, kq_scale=0.0)
run("identity ns=2 kq only", B, ns, h, w, C, Tid, kap_scale=0.0)
run("identity ns=2 full", B, ns, h, w, C, Tid)
Tr = torch.tensor(r.uniform(-1.1, 1.1, size=(B, ns, h, w, 2)).astype(np.float32))
run("random flows", B, ns, h, w, C, Tr)
Tm = Tr.clone(); Tm[0, :, :8] = -2.0
run("random + bg half", B, ns, h, w, C, Tm)
run("random C=256", B, ns, h, w, 256, Tr)
run("random C=128", B, ns, h, w, 128, Tr)
run("random bf16 C=64", B, ns, h, w, 64, Tr, dt=torch.bfloat16)
