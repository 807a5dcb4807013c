#!/usr/bin/env python3
"""bench_personalize.py - BASELINE configs[4]: the personalization step (G + D forward / backward / Adam) at 512x512,
one sample per GPU, data parallel over N MI355X with ONE flat RCCL all-reduce of the gradients per network.

    python bench_personalize.py [--gpus N] [--steps K] [--warmup W] [--size 512]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench_personalize.py --gpus N)

NOT the headline bench (that is bench.py: synthesized frames/s).  A step = LWGTrainer.optimize_parameters()
(reference tools/trainers/lwg_trainer.py:326-352) on one synthetic sample (ns = 2 sources, nt = 1 target): G forward with
only_tsf=False (bg + src with decoder + tsf), LSGAN + L1 + BCE mask + TV losses (--use-vgg / --use-face add the VGG19 and
SphereFace losses on seeded weights: their checkpoints are not available offline), backward, gradient all-reduce (global batch =
N samples, BASELINE configs[4]: 8), Adam; then the discriminator step.  Every convolution (forward, dgrad, wgrad) runs on the hand-written MFMA
kernels; the elementwise glue is PyTorch-ROCm autograd this round (ipercore_amd/networks/training.py).

Prints ONE JSON line on rank 0: samples/s over all ranks, and the achieved conv TFLOP/s from the algorithmic conv flops of
the step (forward + dgrad + wgrad of every ConvFn call, counted live).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

if int(os.environ.get("WORLD_SIZE", "1")) > 1:                  # before the first HIP call (see bench.py)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


from ipercore_amd.launch import self_launch_if_needed  # noqa: E402  (N > 1 without torchrun: spawn the ranks ourselves)

if __name__ == "__main__":
    self_launch_if_needed()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def measure(dev, steps=5, warmup=2, size=512, use_vgg=False, use_face=False, precision="winograd", rank=0, world=1, graph=None,
            panel_cache=None, branch=None, overlap_d=None, _keep=None, dp_schedule=None, _host_probe=True):
    """One process' share of the measurement -> the result dict (rank 0) / None.  ``graph``: None = the trainer's default (the
    static-shape step replayed as a hipGraph when that is supported), False = eager launches."""
    from ipercore_amd import ops, synthetic as syn
    from ipercore_amd.networks import NetworksFactory, generator_param_shapes
    from ipercore_amd.trainers import LWGTrainer, PatchGlobalDiscriminator, TrainOpts

    S, ns = size, 2
    on_gpu = torch.device(dev).type == "cuda"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)        # --device cpu: the CPU test-suite's plumbing run (emulated C ABI)
    nf, nres, bgf = [64, 128, 256], 6, [64, 128, 128, 256]
    G = NetworksFactory.get_by_name("AttLWB-SPADE", cfg=syn.gen_cfg(nf, nres, bgf), temporal=False)
    sd = syn.fill_state_dict(generator_param_shapes(nf, nres, bgf), seed=7)
    G.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=True)
    G.to(dev).train()
    torch.manual_seed(0)
    D = PatchGlobalDiscriminator().to(dev)
    # one synthetic sample per rank (different seeds per rank: data parallel); flows from the real renderer path
    case = syn.build_case(image_size=S, n_frames=1, ns=ns, seed=rank)
    im = syn.make_imitator(case, frame_batch=1, device=dev)
    tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
    tsf8, Tst, _ = im.make_inputs_for_tsf(im.src_info, tgt[0:1], "smooth", t=0)
    u = lambda shape, seed, name: torch.tensor(syn.uniform_image(shape, seed + 100 * rank, name), device=dev)   # noqa: E731
    cond = im.src_info["cond"]
    inp = {"input_G_bg": u((1, 1, 4, S, S), 10, "bg_inputs"),
           "input_G_src": torch.cat([torch.tensor(case.src_img, device=dev)[0], cond], dim=1).unsqueeze(0),
           "input_G_tsf": ops.nhwc_to_nchw(tsf8, channels=6).unsqueeze(0), "Tst": Tst.unsqueeze(1).contiguous(),
           "real_src": torch.tensor(case.src_img, device=dev), "real_tsf": u((1, 1, 3, S, S), 701, "real_tsf"),
           "real_bg": u((1, 3, S, S), 702, "real_bg"), "body_mask": (u((1, ns + 1, 1, S, S), 703, "mask") > 0).float(),
           "head_bbox": torch.tensor([[S * 3 // 8, S * 5 // 8, S // 16, S * 5 // 16]])}        # a head-sized box (min_x, max_x, min_y, max_y)
    del im
    topts = TrainOpts()
    topts.conv_precision = precision
    topts.use_vgg = "VGG19" if use_vgg else "None"
    topts.use_face = bool(use_face)
    topts.allow_seeded_loss_nets = True        # the licensed checkpoints are not available offline: seeded weights, same cost per step
    if graph is not None and hasattr(topts, "use_graph"):
        topts.use_graph = bool(graph)
    if panel_cache is not None:
        topts.use_panel_cache = bool(panel_cache)
    if branch is not None:
        topts.branch_streams = bool(branch)
    if overlap_d is not None:
        topts.overlap_d_step = bool(overlap_d)
    if dp_schedule is not None:
        topts.dp_schedule = dp_schedule
    tr = LWGTrainer(G, D, opts=topts)
    tr.set_input(inp)
    if _keep is not None:
        _keep["trainer"] = tr

    flops, wino_algo = [0.0], [0.0]

    def hook(begin, M, spec, epi=0, info=None):
        if begin:
            flops[0] += 2.0 * M * spec.algo_kn
        elif info is not None and info.get("kind") == "winograd":
            wino_algo[0] += 2.0 * M * spec.algo_kn          # executes 4/9 of these: sixteen products per 2 x 2 outputs instead of 36
    wg = [0.0]
    orig = ops.conv2d_wgrad

    def counted_wgrad(x0, spec, dy, **kw):
        M = dy.shape[0] * dy.shape[1] * dy.shape[2] // (spec.omul ** 2)
        wg[0] += 2.0 * M * spec.algo_kn
        return orig(x0, spec, dy, **kw)
    orig_u = ops.conv2d_wgrad_unpacked

    def counted_wgrad_unpacked(x0, spec, dy, *a, **kw):
        wg[0] += 2.0 * (dy.shape[0] * dy.shape[1] * dy.shape[2] // (spec.omul ** 2)) * spec.algo_kn
        return orig_u(x0, spec, dy, *a, **kw)
    # the algorithmic conv flops of a step are counted on ONE eager step (forward + dgrad launches through the hook, the weight
    # gradients through the wrappers: same 2*M*K*N), outside the timed region - a replayed graph makes no Python calls to count
    prev_hook, want_graph = ops.CONV_HOOK, getattr(tr.opts, "use_graph", False)
    ops.CONV_HOOK, ops.conv2d_wgrad, ops.conv2d_wgrad_unpacked = hook, counted_wgrad, counted_wgrad_unpacked
    tr.opts.use_graph = False                  # the counting step runs eager (a capture would run - and count - its warm-up steps too)
    try:
        tr.optimize_parameters()
    finally:
        ops.CONV_HOOK, ops.conv2d_wgrad, ops.conv2d_wgrad_unpacked = prev_hook, orig, orig_u
        tr.opts.use_graph = want_graph
    per_step = flops[0] + wg[0]
    for _ in range(warmup):
        tr.optimize_parameters()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        lg, ld = tr.optimize_parameters()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    # host share: one step from an idle queue - time until the last launch is enqueued vs until the GPU is done
    host_ms, total_ms = [], []
    for _ in range(3 if _host_probe else 0):
        sync()
        h0 = time.perf_counter()
        tr.optimize_parameters()
        h1 = time.perf_counter()
        sync()
        host_ms.append((h1 - h0) * 1e3)
        total_ms.append((time.perf_counter() - h0) * 1e3)
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank != 0:
        return None
    nG, nD = sum(p.numel() for p in G.parameters()), sum(p.numel() for p in D.parameters())
    tf = per_step / (dt / steps) / 1e12
    executed = per_step - wino_algo[0] * 5.0 / 9.0
    tf_exec = executed / (dt / steps) / 1e12
    return {
        "metric": f"personalization steps (samples)/sec at {S}x{S}, G+D fwd/bwd/Adam, 1 sample per GPU", "value": round(steps * world / dt, 4),
        "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "dtype": "f32" if precision in ("fp32", "winograd") else "f32 (forward / dgrad products as bf16x6 exact split)",
        "conv_engine": {"fp32": "direct fp32 MFMA implicit GEMMs", "winograd": "3x3 / stride 1 forward + data-gradient convolutions as F(2x2,3x3) Winograd (fp32 MFMA), "
                        "the rest and every weight gradient direct", "split": "bf16x6"}[precision],
        "data": "synthetic" + ("" if on_gpu else " (CPU plumbing run over the emulated C ABI: NOT a measurement)"),
        "config": {"workload": f"personalize step {S}x{S} ns=2 nt=1 (BASELINE configs[4]); losses: LSGAN + L1 rec + " + ("VGG19 perceptual" if use_vgg else "L1") + " tsf" + (" + Sphere20a face" if use_face else "") + " + BCE mask + TV",
                   "global_batch": world,
                   "parallelism": f"dp{world}: the flat gradient buffers of G and D all-reduced over RCCL ({nG} + {nD} fp32 gradients = "
                                  f"{(nG + nD) * 4 / 1e6:.1f} MB per step per rank), G's in 4 ranges behind D's forward / backward segment",
                   "step": getattr(tr, "step_mode", "eager launches"),
                   "panel_cache": bool(getattr(tr.opts, "use_panel_cache", False)), "branch_streams": bool(getattr(tr.opts, "branch_streams", False))},
        "conv_gflop_per_step": round(per_step / 1e9, 1), "conv_gflop_executed_per_step": round(executed / 1e9, 1), "conv_tflops_whole_step": round(tf, 2),
        "roofline": {"bound": "mfma", "achieved": round(tf_exec, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf_exec / 157.3, 4),
                     "algorithmic_equivalent_tflops": round(tf, 2),
                     "what": "EXECUTED conv flops of the step (forward + dgrad + wgrad of G, D and the loss networks; a Winograd launch counts 4/9 of its "
                             "2 M K N) / whole-step wall time; algorithmic_equivalent_tflops prices the same step at 2 M K N throughout"},
        "bytes_allreduced_per_step": (nG + nD) * 4 if world > 1 else 0,
        # N > 1: time the compute stream waited for RCCL in the last step (G's exchange runs behind D's forward / backward segment, D's
        # behind Adam(G): trainers.LWGTrainer._run_dp_schedule); N = 1: nothing is exchanged
        "exposed_allreduce_ms_per_step": (0.0 if world == 1 else (None if tr.exposed_allreduce_ms() is None else round(tr.exposed_allreduce_ms(), 3))),
        "allreduce_overlap": (getattr(tr, "allreduce_overlap", None) or
                              ("no collective at N = 1; at N > 1 the captured step runs G's gradient all-reduce (4 ranges, RCCL's stream) "
                               "behind D's forward / backward graph and D's behind Adam(G)" if world == 1 else
                               f"hook-driven: {getattr(tr.optimizer_G, 'overlapped_ranges', None)} ranges of G issued during backward")),
        "single_step_host_enqueue_ms": round(min(host_ms), 2) if host_ms else None, "single_step_total_ms": round(min(total_ms), 2) if total_ms else None,
        "loss_G": round(lg.item(), 4), "loss_D": round(ld.item(), 4)}


def graph_vs_eager_check(dev, size, use_vgg, use_face, precision, steps=3):
    """Self-check of the benched configuration: the same seeded trainer run ``steps`` optimisation steps as captured hipGraph replays and as eager
    launches (fresh networks, the same inputs, the same number of updates): every parameter of G and D afterwards within 5e-5 of its
    magnitude scale of each other (the two forms issue the same kernels; atomics in the weight-gradient slabs and the head-crop backward
    are the only order-dependent sums), losses within 1e-4 relative."""
    hold_g, hold_e = {}, {}
    rg = measure(dev, steps=steps, warmup=0, size=size, use_vgg=use_vgg, use_face=use_face, precision=precision, graph=True, _keep=hold_g, _host_probe=False)
    re_ = measure(dev, steps=steps, warmup=0, size=size, use_vgg=use_vgg, use_face=use_face, precision=precision, graph=False, _keep=hold_e, _host_probe=False)
    tg, te = hold_g["trainer"], hold_e["trainer"]
    lr = tg.optimizer_G.lr
    n_upd = steps + 1                                      # + the flop-counting step of measure()
    dmax, dsum, n = 0.0, 0.0, 0
    for net_g, net_e in ((tg.G, te.G), (tg.D, te.D)):
        for (k, a), (_, b) in zip(net_g.named_parameters(), net_e.named_parameters()):
            d = (a.detach() - b.detach()).abs()
            dmax = max(dmax, d.max().item())
            dsum += d.sum().item()
            n += d.numel()
    dl = max(abs(rg["loss_G"] - re_["loss_G"]) / max(abs(re_["loss_G"]), 1e-6), abs(rg["loss_D"] - re_["loss_D"]) / max(abs(re_["loss_D"]), 1e-6))
    captured = "hipGraph" in rg["config"]["step"] and "failed" not in rg["config"]["step"]
    # Adam moves a weight by <= lr per update whatever the gradient's size (a noise-level gradient can flip the direction of a single weight):
    # a missing or extra update would show as ~lr EVERYWHERE - the bound of tests/gpu_checks.py::check_graph_vs_eager_steps_512_full
    ok = captured and dmax <= 2 * n_upd * lr and dsum / n <= 0.1 * lr and dl <= 2e-3
    return {"result": ("captured == eager (same updates: mean |d theta| <= 0.1 lr, max <= 2 n lr)" if ok else "MISMATCH" if captured else "NOT CAPTURED: " + rg["config"]["step"]),
            "updates_each": n_upd, "max_param_diff_over_lr": dmax / lr, "mean_param_diff_over_lr": dsum / n / lr, "max_rel_loss_diff": dl,
            "captured_step": rg["config"]["step"], "eager_step": re_["config"]["step"]}


def breakdown(dev, size=512, steps=3):
    """Lab view of one eager step: every forward / data-gradient conv launch (ops.CONV_HOOK) and every weight-gradient launch bracketed
    with events -> per shape: launches per step, ms per step, achieved TFLOP/s.  Written to gpurun_out/personalize_breakdown.json."""
    from ipercore_amd import ops
    from ipercore_amd.trainers import LWGTrainer                           # noqa: F401
    import bench
    timer, wtimer = bench.ConvTimer(), bench.ConvTimer()
    hold = {}

    def build():
        hold["r"] = measure(dev, steps=1, warmup=0, size=size, graph=False, _keep=hold)
    build()
    tr = hold["trainer"]
    ops.CONV_HOOK = lambda b, M, spec, epi=0, info=None: timer(b, M, spec, epi, info, 4)
    orig_u, orig_w = ops.conv2d_wgrad_unpacked, ops.conv2d_wgrad

    def wrap(fn):
        def inner(x0, spec, dy, *a, **kw):
            M = dy.shape[0] * dy.shape[1] * dy.shape[2] // (spec.omul ** 2)
            wtimer(True, M, spec, 0, 4)
            out = fn(x0, spec, dy, *a, **kw)
            wtimer(False, M, spec, 0, 4)
            return out
        return inner
    ops.conv2d_wgrad_unpacked, ops.conv2d_wgrad = wrap(orig_u), wrap(orig_w)
    try:
        tr.optimize_parameters()
        torch.cuda.synchronize()
        timer.enabled = wtimer.enabled = True
        for _ in range(steps):
            tr.optimize_parameters()
        torch.cuda.synchronize()
    finally:
        ops.CONV_HOOK, ops.conv2d_wgrad_unpacked, ops.conv2d_wgrad = None, orig_u, orig_w
    out = {}
    for name, t in (("forward_dgrad", timer), ("wgrad", wtimer)):
        rows = t.breakdown()
        for r in rows:
            r["launches"] = r["launches"] / steps
            r["ms"] = round(r["ms"] / steps, 4)
        out[name] = rows
        tot = sum(r["ms"] for r in rows)
        print(f"== {name}: {tot:.2f} ms per step over {sum(r['launches'] for r in rows):.0f} launches")
        for r in rows[:28]:
            print(f"  {r['shape']:44s} n={r['launches']:5.1f} ms={r['ms']:7.3f} TF={r['tflops']:7.1f}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "personalize_breakdown.json"), "w") as fp:
        json.dump(out, fp, indent=1)


def main(argv=None):
    import faulthandler
    faulthandler.enable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--use-vgg", action="store_true", help="VGG19 perceptual transfer loss (deploy.toml:83; seeded weights: "
                                                          "vgg19-dcbb9e9d.pth is not available offline) instead of L1")
    ap.add_argument("--use-face", action="store_true", help="SphereFace (Sphere20a) loss on the head crop (deploy.toml:77-79)")
    ap.add_argument("--precision", choices=("fp32", "split", "winograd"), default="winograd",
                    help="split: forward / data-gradient convs on the bf16x6 kernel (fp32-level accuracy), weight gradients fp32 MFMA; winograd: the 3x3 / "
                         "stride 1 forward and data-gradient convs as F(2x2,3x3) Winograd convolutions (fp32 MFMA), weight gradients direct")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="eager launches instead of the captured step")
    ap.add_argument("--two-pass-loss", action="store_true", help="lab: the frozen loss networks in two passes (target under no_grad) instead of one [fake | target] batch")
    ap.add_argument("--no-panel-cache", dest="panel_cache", action="store_false", help="one pack launch per weight panel (round-1 behaviour)")
    ap.add_argument("--no-branch-streams", dest="branch_streams", action="store_false",
                    help="the background network and the source decoder on the main stream (round-2 default: a second stream)")
    ap.add_argument("--no-overlap-d", dest="overlap_d", action="store_false", help="D's forward / backward after Adam(G), not next to G's backward")
    ap.add_argument("--no-fused-convt", dest="fused_convt", action="store_false", help="lab: transposed-conv forwards as four parity launches")
    ap.add_argument("--no-fused-bias", dest="fused_bias", action="store_false", help="lab: bias gradients by the separate column-sum kernel")
    ap.add_argument("--no-convt-wgrad", dest="convt_wgrad", action="store_false", help="lab: transposed-convolution weight gradients as four parity launches")
    ap.add_argument("--no-s2-dgrad", dest="s2_dgrad", action="store_false", help="lab: 4x4 stride-2 data gradients as four parity launches")
    ap.add_argument("--no-relu-mask", dest="relu_mask", action="store_false", help="lab: every ReLU convolution runs its own act_bwd pass")
    ap.add_argument("--no-kv-pair", dest="kv_pair", action="store_false", help="lab: the fk / fv projections as two 1x1 convolutions")
    ap.add_argument("--no-spade-pair", dest="spade_pair", action="store_false", help="lab: SPADE's gamma / beta convolutions as two launches per pass")
    ap.add_argument("--no-self-check", dest="self_check", action="store_false", help="skip the captured-vs-eager self-check (N = 1, GPU)")
    ap.add_argument("--breakdown", action="store_true", help="lab: per-shape conv times of one eager step (events around every launch)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="torch.distributed backend; nccl = RCCL (the product)")
    ap.add_argument("--device", choices=("cuda", "cpu"), default="cuda",
                    help="cpu: plumbing dry run for the CPU test-suite ONLY (tests/test_bench_launch.py installs the emulated C ABI around "
                         "main(); without it every op raises on CPU tensors - there is no CPU product path)")
    args = ap.parse_args(argv)
    if args.two_pass_loss:
        from ipercore_amd import trainers as _tr
        _tr.LOSS_NETS_ONE_PASS = False
    self_launch_if_needed(sys.argv[1:] if argv is None else argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device == "cuda":
        assert torch.cuda.is_available(), "needs the MI355X"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    if world > 1:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        if args.backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group("gloo", init_method="env://")
    assert args.gpus == world
    if not args.fused_convt:
        from ipercore_amd.networks import training as _tr
        _tr.FUSED_CONVT_FWD = False
    if not args.fused_bias:
        from ipercore_amd.networks import training as _tr
        _tr.FUSED_BIAS_GRAD = False
    if not args.spade_pair:
        from ipercore_amd.networks import training as _tr
        _tr.FUSED_SPADE_PAIR = False
    if not args.convt_wgrad:
        from ipercore_amd.networks import training as _tr
        _tr.FUSED_CONVT_WGRAD = False
    if not args.s2_dgrad:
        from ipercore_amd.networks import training as _tr
        _tr.FUSED_S2_DGRAD = False
    if not args.relu_mask:
        from ipercore_amd.networks import training as _tr
        _tr.FUSED_RELU_MASK = False
    if not args.kv_pair:
        from ipercore_amd.networks import training as _tr
        _tr.FUSED_KV_PAIR = False
    if args.breakdown:
        return breakdown(dev, args.size)
    res = measure(dev, args.steps, args.warmup, args.size, args.use_vgg, args.use_face, args.precision, rank, world,
                  graph=None if args.graph else False, panel_cache=None if args.panel_cache else False,
                  branch=None if args.branch_streams else False, overlap_d=None if args.overlap_d else False,
                  dp_schedule="segmented" if (args.device == "cpu" and world > 1) else None)     # no graphs on the CPU: the same segment schedule, eager
    if rank == 0 and world == 1 and args.self_check and args.graph and args.device == "cuda":
        try:
            res["self_check_detail"] = graph_vs_eager_check(dev, args.size, args.use_vgg, args.use_face, args.precision)
            res["self_check"] = res["self_check_detail"]["result"]
        except Exception as e:       # noqa: BLE001
            res["self_check"] = f"error: {type(e).__name__}: {e}"[:300]
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
