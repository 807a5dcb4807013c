"""Lab: the 512x512 full-width training graph - GPU (fp32) and the oracle in fp32 against the oracle in fp64, per parameter."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import gpu_checks as gc
from tests import parity_utils as pu
from oracle import lwg_oracle as orc
from ipercore_amd import synthetic
from ipercore_amd.networks import NetworksFactory, generator_param_shapes
from ipercore_amd.networks.training import TrainableGenerator
S, ns = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2
nf, nres, bgf = gc.FULL
DEV = "cuda:0"
G = NetworksFactory.get_by_name("AttLWB-SPADE", cfg=pu.gen_cfg(nf, nres, bgf), temporal=False)
sdn = synthetic.fill_state_dict(generator_param_shapes(nf, nres, bgf), seed=7)
G.load_state_dict({k: torch.tensor(v) for k, v in sdn.items()}, strict=True)
G.to(DEV).train()
bg_in, src_in, tsf_in, Tst = gc._training_inputs(S, ns, nf, nres, bgf, True)
tgt = [torch.tensor(synthetic.uniform_image(s, 500 + i, "tgt")) for i, s in enumerate(((1, 1, 3, S, S), (1, ns, 3, S, S), (1, ns, 1, S, S), (1, 1, 3, S, S), (1, 1, 1, S, S)))]
def run_oracle(dt):
    sd = {k: torch.tensor(v, dtype=dt, requires_grad=True) for k, v in sdn.items()}
    outs = orc.gen_forward_train(sd, bg_in.to(dt), src_in.to(dt), tsf_in.to(dt), Tst.to(dt), n_down=len(nf), n_res=nres, n_bg=len(bgf))
    sum((o - t.to(dt)).abs().mean() for o, t in zip(outs, tgt)).backward()
    return {k: v.grad for k, v in sd.items()}
t0 = time.time(); g64 = run_oracle(torch.float64); print("fp64 oracle", round(time.time() - t0, 1), "s", flush=True)
g32 = run_oracle(torch.float32)
def run_gpu(tag, no_splitk=False):
    from ipercore_amd import ops
    for p_ in G.parameters():
        p_.grad = None
    orig = ops.conv2d
    if no_splitk:
        ops.conv2d = lambda *a, **k: orig(*a, **{**k, "splitk": False})
    try:
        outs = TrainableGenerator(G).forward(bg_in.to(DEV), src_in.to(DEV), tsf_in.to(DEV), Tst.to(DEV))
        sum((o - t.to(DEV)).abs().mean() for o, t in zip(outs, tgt)).backward()
        torch.cuda.synchronize()
    finally:
        ops.conv2d = orig
    gmax = max(v.abs().max().item() for v in g64.values())
    rows = []
    for k, p in G.named_parameters():
        ref = g64[k]; sc = max(ref.abs().max().item(), 1e-3 * gmax)
        rows.append((k, (p.grad.cpu().double() - ref).abs().max().item(), (g32[k].double() - ref).abs().max().item(), ref.abs().max().item(), sc))
    rows.sort(key=lambda r: -r[1] / r[4])
    print(f"== {tag}: gmax {gmax:.4e}; param: abs err GPU-fp32 vs fp64 | abs err oracle-fp32 vs fp64 | max |ref| | rel (GPU) | rel (oracle32)")
    for r in rows[:10]:
        print(f"  {r[0]:36s} {r[1]:.3e} {r[2]:.3e} {r[3]:.3e} {r[1] / r[4]:.5f} {r[2] / r[4]:.5f}")
    print("  max rel over params: GPU", max(r[1] / r[4] for r in rows), "oracle32", max(r[2] / r[4] for r in rows), flush=True)
run_gpu("default")
run_gpu("forward / dgrad without split-K", no_splitk=True)
