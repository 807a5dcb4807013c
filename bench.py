#!/usr/bin/env python3
"""bench.py - synthesized frames/s of the per-frame path (run_imitator, 512x512, AttLWB-SPADE fp32) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

A step = one pass of the hot path (camera swap, SMPL-H skinning, projection +
rasterization, fused flows, AttLWB-SPADE forward_tsf, head + compositing) over one batch of ``--frame-batch``
synthetic target frames of BASELINE.json configs[1] (512x512, one source/reference pair, ns = 2, random-init
weights of the real architecture, synthetic SMPL-H + T-pose template geometry).  Inputs (SMPL parameters, cached
source state) are resident in HBM before the timed region; source_setup and PNG writing are outside it.  With
N > 1 the clip is frame-sharded (weak scaling: every rank renders K batches) and the timed region ends with the
single RCCL all-gather of the output video tensor.

Prints ONE JSON line on rank 0.  ``roofline`` is for the dominant kernel (the fp32 MFMA implicit-GEMM conv):
algorithmic conv flops of all its launches in the timed region / their HIP-event time, against the 157.3
TFLOP/s fp32 matrix peak of gfx950.  ``cpu_baseline`` times the CPU oracle ("port") on this host for a few
frames of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (same guide); the bf16-operand kernel is L2/HBM bound, far below it


class ConvTimer:
    """Brackets every conv launch with events on torch's current stream (the stream the kernels are launched on)."""

    def __init__(self):
        self.pairs, self.flops, self.bytes, self.enabled, self.meta = [], 0.0, 0.0, False, []
        self._start = None

    def __call__(self, begin, M, spec, epi=0):
        if not self.enabled:
            return
        if begin:
            self._start = torch.cuda.Event(enable_timing=True)
            self._start.record()
        else:
            stop = torch.cuda.Event(enable_timing=True)
            stop.record()
            self.pairs.append((self._start, stop))
            self.flops += 2.0 * M * spec.algo_kn
            # algorithmic bytes of the launch: input read once + weight panel + output written (+ the epilogue's operands)
            out = M * spec.N if epi != 2 else M * spec.N       # SPADE: reads xn (M*N/2) and writes y (M*N/2)
            self.bytes += 4.0 * (M * spec.stride ** 2 * spec.Cin + spec.w.numel() + out + (M * spec.N if epi == 1 else 0))
            self.meta.append((M, spec.N, spec.Cin, spec.ntaps, spec.stride, spec.omul, 2.0 * M * spec.algo_kn))

    def result(self):
        """(busy ms, flops, launches, mean launch ms).  busy = length of the UNION of the launches' [start, stop] intervals
        on the device clock: with one stream that is the sum of the durations; with several streams, launches of
        different streams overlap in time and share the machine, and flops / busy is the aggregate rate."""
        if not self.pairs:
            return 0.0, 0.0, 0, 0.0
        base = self.pairs[0][0]
        iv = sorted((base.elapsed_time(a), base.elapsed_time(b)) for a, b in self.pairs)
        busy, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
        for s0, e0 in iv[1:]:
            if s0 > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s0, e0
            else:
                cur_e = max(cur_e, e0)
        busy += cur_e - cur_s
        mean = sum(e0 - s0 for s0, e0 in iv) / len(iv)
        return busy, self.flops, len(self.pairs), mean

    def breakdown(self):
        """Per distinct conv shape: launches, total ms, achieved TFLOP/s (algorithmic flops / event time)."""
        agg = {}
        for (a, b), m in zip(self.pairs, self.meta):
            key = "M%d N%d Cin%d taps%d s%d up%d" % m[:6]
            e = agg.setdefault(key, [0, 0.0, 0.0])
            e[0] += 1
            e[1] += a.elapsed_time(b)
            e[2] += m[6]
        rows = [{"shape": k, "launches": v[0], "ms": round(v[1], 3), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2)}
                for k, v in agg.items()]
        return sorted(rows, key=lambda r: -r["ms"])


def cpu_baseline(case, n_frames):
    """The oracle (CPU restatement of the reference algorithm) on the host cores, a bounded sample."""
    from tests import parity_utils as pu
    t0 = time.time()
    pu.oracle_source(case)                       # source-side work is not part of the per-frame metric
    t_src = time.time() - t0
    t0 = time.time()
    pu.run_oracle(case, frames=list(range(n_frames)))
    dt = time.time() - t0 - t_src                # run_oracle rebuilds the source state once
    return {"value": round(n_frames / max(dt, 1e-9), 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_frames} frames @{case.S}x{case.S} ns={case.ns} through oracle/lwg_oracle.py (torch-CPU fp32 + "
                      f"OpenMP C rasterizer), {os.cpu_count()} host cpus"}


def with_output(im, smpls, FB, n_frames, t_base):
    """Reported separately: the same per-frame path WITH the output stage - device uint8 conversion, pinned async
    D2H, PNG encoding + file writes on host threads (ipercore_amd/output.py).  Not the headline value."""
    import shutil
    import tempfile
    from ipercore_amd.output import FrameWriter
    d = tempfile.mkdtemp(prefix="lwg_bench_out_")
    try:
        w = FrameWriter(d, prefix="pred_")
        n = min(n_frames, smpls.shape[0]) // FB * FB
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(0, n, FB):
            tsf8, Tst, _ = im.make_inputs_for_tsf(im.src_info, smpls[s:s + FB], "smooth", t=t_base + s)
            w.submit(im.forward(tsf8, Tst)[0], s)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
        paths = w.close()
        dt = time.perf_counter() - t0
        assert len(paths) == n
        return {"value": round(n / dt, 2), "unit": "frames/s", "frames": n, "png_threads": w.workers,
                "gpu_side_frames_per_s": round(n / t_gpu, 2),
                "what": "per-frame path + uint8 conversion on the device + async D2H + PNG (zlib level 1) files on tmpfs/disk"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def split_products(step, im, W, K, FB, ref_outs):
    """Reported separately: the same K steps with every Cin % 32 == 0 convolution on the bf16x6 kernel
    (csrc/conv_igemm_split.hip: fp32 in / out / accumulate, each fp32 product formed from six bf16 MFMAs over an exact
    three-way split of both operands).  Not the headline value; `max_abs_diff_vs_fp32_path` compares the frames."""
    from ipercore_amd import ops
    hook, ops.CONV_HOOK = ops.CONV_HOOK, None
    prev = im.generator.conv_precision
    im.generator.conv_precision = "split"
    try:
        for i in range(min(W, 4)):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [step(i) for i in range(W, W + K)]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        diff = max((a - b).abs().max().item() for a, b in zip(outs, ref_outs))
        return {"value": round(K * FB / dt, 3), "unit": "frames/s", "ms_per_step": round(dt / K * 1e3, 3),
                "max_abs_diff_vs_fp32_path": diff, "frames_range": "[-1, 1]",
                "what": "bf16x6: exact 3-way bf16 split of both fp32 operands, 6 bf16 MFMAs per product, fp32 accumulation"}
    finally:
        im.generator.conv_precision = prev
        ops.CONV_HOOK = hook


def pipelined(step, W, K, FB, n_streams, dev):
    """Reported separately: the same K steps with independent frame batches in flight on several HIP streams, so that one
    batch's launch gaps, kernel tails and HBM-bound kernels overlap another batch's MFMA work.  Not the headline value (the
    per-kernel roofline accounting above needs launches that own the machine)."""
    from ipercore_amd import ops
    hook, ops.CONV_HOOK = ops.CONV_HOOK, None
    try:
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        for i in range(W):
            with torch.cuda.stream(streams[i % n_streams]):
                step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        for i in range(W, W + K):
            with torch.cuda.stream(streams[i % n_streams]):
                outs.append(step(i))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert all(torch.isfinite(o).all() for o in outs[-n_streams:])
        return {"value": round(K * FB / dt, 3), "unit": "frames/s", "streams": n_streams, "ms_per_step": round(dt / K * 1e3, 3)}
    finally:
        ops.CONV_HOOK = hook


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)   # the shader clock needs ~0.2 s of load to settle
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frame-batch", type=int, default=0,
                    help="frames per launch batch; 0 = 8 at 512x512 scaled by (512/size)^2 (the 64x64-feature layers need "
                         ">= 32768 GEMM rows to give every CU two 128x128 tiles), clamped to [2, 64]")
    ap.add_argument("--cpu-frames", type=int, default=3, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--precision", choices=("fp32", "bf16", "split"), default="fp32",
                    help="bf16: BASELINE configs[3] mode - bf16 MFMA operands in the convs (fp32 activations / accumulation); "
                         "the headline metric (configs[1]) is fp32")
    ap.add_argument("--workload", choices=("imitate", "novel_view"), default="imitate",
                    help="novel_view: BASELINE configs[3] poses - create_T_pose_novel_view_smpl(180), global rotation y = 0..360")
    ap.add_argument("--streams", type=int, default=1, help="frame batches in flight on separate HIP streams (see DESIGN.md 5)")
    ap.add_argument("--pipelined-streams", type=int, default=3,
                    help="extra (separately reported) measurement with this many frame batches in flight; 0/1 = skip")
    ap.add_argument("--no-conv-events", action="store_true")
    ap.add_argument("--no-split-extra", dest="split_extra", action="store_false",
                    help="skip the separately reported bf16x6 (exact-split products) measurement")
    ap.add_argument("--output-frames", type=int, default=160, help="frames of the with-output measurement (0 = skip)")
    ap.add_argument("--conv-breakdown", action="store_true", help="write gpurun_out/conv_breakdown.json")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="env://", device_id=dev)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from ipercore_amd import ops, sharding, synthetic as pu      # product path only; the oracle is imported in cpu_baseline()

    S = args.size
    FB = args.frame_batch or max(2, min(64, int(round(8 * (512.0 / S) ** 2))))
    K, W = args.steps, args.warmup
    per_rank = (K + W) * FB
    case = pu.build_case(image_size=S, n_frames=per_rank * world, ns=2)
    if args.workload == "novel_view":
        from ipercore_amd.imitator import create_T_pose_novel_view_smpl
        nv = create_T_pose_novel_view_smpl(180)
        nv[:, 0:3], nv[:, -10:] = case.src_smpl[0, 0:3], case.src_smpl[0, -10:]
        case.tgt_smpls = np.concatenate([nv] * (case.tgt_smpls.shape[0] // 180 + 1), axis=0)[:case.tgt_smpls.shape[0]]
    im = pu.make_imitator(case, frame_batch=FB, device=dev)
    if args.precision != "fp32":
        im.generator.conv_precision = args.precision
        im.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
    tgt = im.prepare_sequence(case.tgt_smpls, "smooth")          # sequence-global pre-pass, every rank identically
    lo, hi = sharding.shard_range(tgt.shape[0], rank, world)
    mine = tgt[lo:hi].contiguous()

    lo_out = rank * K * FB                                   # this rank's block inside the gathered (K * FB * world) video
    timer = ConvTimer()
    ops.CONV_HOOK = None if args.no_conv_events else timer

    def step(i):
        chunk = mine[i * FB:(i + 1) * FB]
        tsf8, Tst, _ = im.make_inputs_for_tsf(im.src_info, chunk, "smooth", t=lo + i * FB)
        return im.forward(tsf8, Tst)[0]

    last = None
    for i in range(W):
        last = step(i)
    torch.cuda.synchronize()
    if world > 1:
        if last is None:                                     # --warmup 0: the collective still gets its untimed first call
            last = step(0)
        # untimed: the first collective of a size class sets up RCCL's channels / buffers - do it once at the timed shape
        wg = sharding.OverlappedGather(FB * world)
        wg.submit(last, 0)
        wg.finish()
        del wg
        dist.barrier()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    if streams:
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        for i in range(W):          # warm the side streams too
            with torch.cuda.stream(streams[i % len(streams)]):
                step(i)
        torch.cuda.synchronize()
    timer.enabled = True
    outs = []
    og = sharding.OverlappedGather(K * FB * world) if world > 1 else None     # each step's frames go to RCCL as they appear
    t0 = time.perf_counter()
    for i in range(W, W + K):
        if streams:
            with torch.cuda.stream(streams[i % len(streams)]):
                outs.append(step(i))
        else:
            outs.append(step(i))
        if og is not None and not streams:
            og.submit(outs[-1], (i - W) * FB)
    if streams:
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        if og is not None:
            for j, o in enumerate(outs):
                og.submit(o, j * FB)
    if og is not None:
        video = og.finish()
        local = video[lo_out:lo_out + K * FB]
    else:
        video = local = torch.cat(outs, dim=0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.enabled = False
    assert video.shape[0] == K * FB * world and torch.isfinite(local).all()

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        conv_ms, conv_flops, n_launch, mean_launch_ms = timer.result()
        frames = K * FB * world
        line = {
            "metric": ("synthesized frames/sec at 512x512 (run_imitator)" if S == 512 else f"synthesized frames/sec at {S}x{S}")
                      + {"fp32": "", "bf16": " [bf16 MFMA conv tiles]", "split": " [bf16x6 exact-split products]"}[args.precision],
            "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 MFMA operands, f32 activations + accumulation",
                      "split": "f32 in/out/accumulate, products as 6 bf16 MFMAs over an exact 3-way split"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"run_imitator {S}x{S} single src/ref pair, AttLWB-SPADE generator fp32 (BASELINE configs[1])"
                       if args.precision == "fp32" else
                       f"per-frame path {S}x{S}, AttLWB-SPADE generator with bf16 MFMA conv tiles (BASELINE configs[3] precision mode)"
                       if args.precision == "bf16" else
                       f"run_imitator {S}x{S} single src/ref pair, AttLWB-SPADE generator, fp32 with bf16x6 exact-split products",
                       "poses": args.workload, "image_size": S, "num_source": 2, "frame_batch": FB, "frames_per_step_per_gpu": FB,
                       "parallelism": f"frame-shard x{world}" + (" + RCCL all-gather of the output video, chunked behind the frame loop"
                                                                  if world > 1 else ""),
                       "batches_in_flight": args.streams,
                       "weights": "random-init (seeded) of the real architecture, 36,276,992 params"},
        }
        if n_launch:
            traffic, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # written by tools/pmc_round.sh (separate --pmc passes)
            if S == 512 and FB == 8 and args.streams == 1 and args.precision == "fp32" and os.path.exists(tpath):
                with open(tpath) as fp:
                    tj = json.load(fp)
                traffic, traffic_src = tj.get("traffic_bytes_per_launch"), "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
            achieved = conv_flops / (conv_ms * 1e-3) / 1e12
            peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
            if args.precision == "split":
                achieved *= 6.0          # executed bf16 MFMA flops: six partial products per algorithmic fp32 product
            line["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                                "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                                "traffic_source": traffic_src,
                                "algorithmic_bytes_per_launch": round(timer.bytes / n_launch, 1),
                                "kernel": {"fp32": "lwg_conv_igemm_kernel (fp32 MFMA implicit GEMM)",
                                           "bf16": "lwg_conv_igemm_bf16_kernel (bf16-operand MFMA implicit GEMM) + fp32 first layers",
                                           "split": "lwg_conv_igemm_split_kernel (bf16x6: achieved = 6 x algorithmic flops, the bf16 "
                                                    "MFMA work actually executed) + fp32 first layers"}[args.precision],
                                "launches": n_launch, "avg_launch_us": round(mean_launch_ms * 1e3, 2), "streams": args.streams,
                                "algorithmic_gflop_per_frame": round(conv_flops / (K * FB) / 1e9, 2),
                                "share_of_step_time": round(conv_ms * 1e-3 / dt, 4)}
        if args.conv_breakdown and n_launch:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "conv_breakdown.json"), "w") as fp:
                json.dump(timer.breakdown(), fp, indent=1)
        if args.pipelined_streams > 1 and args.streams == 1 and world == 1:
            line["pipelined"] = pipelined(step, W, K, FB, args.pipelined_streams, dev)
        if args.split_extra and args.precision == "fp32" and args.streams == 1 and world == 1:
            line["split_products"] = split_products(step, im, W, K, FB, outs)
        if args.output_frames > 0 and world == 1:
            line["with_output"] = with_output(im, mine, FB, args.output_frames, lo)
        if args.cpu_frames > 0 and world == 1:          # the CPU baseline is an N = 1 measurement (rank 0 only)
            small = pu.build_case(image_size=S, n_frames=args.cpu_frames, ns=2)
            line["cpu_baseline"] = cpu_baseline(small, args.cpu_frames)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
