#!/usr/bin/env python3
"""bench.py - synthesized frames/s of the per-frame path (run_imitator, 512x512, AttLWB-SPADE fp32) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

Workload (all N): ONE source/reference pair, a ``--frames``-frame reference clip (300: BASELINE.json configs[2]; the same clip on
one GPU is configs[1]).  A step = one pass of the hot path (camera swap, SMPL-H skinning, projection +
rasterization, fused flows, AttLWB-SPADE forward_tsf, head + compositing) over the WHOLE clip: every rank renders its contiguous
shard (37 / 38 frames at N = 8) in batches of ``--frame-batch`` frames and the output video tensor is assembled by the RCCL
all-gather, issued chunk by chunk behind the frame loop (ipercore_amd/sharding.py).  Total work is fixed as N grows: STRONG
scaling; ``value`` = frames of the clip x K / wall time (max over ranks, barrier + synchronize on both sides).  Inputs (SMPL
parameters, cached source state) are resident in HBM before the timed region; the sequence-global pre-pass (stabilize),
source_setup and PNG writing are outside it.  Synthetic data, random-init (seeded) weights of the real architecture.
``--mode batch`` is the former weak-scaling measurement (a step = one frame batch per rank).

Prints ONE JSON line on rank 0.  ``roofline`` is for the dominant kernel (the MFMA implicit-GEMM conv): algorithmic conv flops of
all its launches in the timed region / their HIP-event time, against the matrix peak of the dtype.  ``cpu_baseline`` times the CPU
oracle ("port") on this host for a few frames of the same workload and, where the reference checkout is importable
(LWG_REFERENCE, default /root/reference - the authoring container, not the GPU box), the reference's own modules ("reference").
Extra objects, each measured in its own loop and never the headline: ``pipelined``, ``split_products``, ``winograd_products``, ``with_output``,
``b1_latency`` (frame_batch = 1), ``novel_view_1024_bf16`` (BASELINE configs[3]), ``personalize_step`` (configs[4]).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


from ipercore_amd.launch import self_launch_if_needed  # noqa: E402  (no torch import: runs before the heavy imports)

if __name__ == "__main__":
    self_launch_if_needed()

# multi-process runs: the host driver supports dmabuf IPC only and the HSA runtime reads this when the first HIP call initialises it -
# so it must be in the environment BEFORE torch touches the device (it is already exported on the GPU boxes; this is the safety net)
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (same guide)
PEAK_HBM_TBS = 8.0
# Default frame batch at 512x512 per precision mode, scaled by (512 / size)^2.  N = 1: the whole clip is ONE launch batch when it fits the budget
# below (300 frames at 512x512 fp32 hold 29 GB of the 288 GB; measured on one box: 498.6 / 500.6 / 501.8 / 507.2 / 510.8 frames/s at 32 / 48 / 64 /
# 100 / 300 frames per batch - fewer launch prologues and tails per clip; bf16 1024x1024: 850 / 857 / 860 / 870 at 20 / 45 / 90 / 180).  N > 1: a
# rank's shard goes in batches of FB_512_SHARDED so that the exchange of one batch overlaps the synthesis of the next (sharding.py).
FB_512 = {"fp32": 384, "split": 384, "winograd": 384, "bf16": 768}
FB_512_SHARDED = {"fp32": 32, "split": 32, "winograd": 32, "bf16": 80}


def default_frame_batch(precision, S, world=1):
    base = (FB_512 if world == 1 else FB_512_SHARDED)[precision]
    return max(2, min(1024 if world == 1 else 64, int(round(base * (512.0 / S) ** 2))))


class ConvTimer:
    """Brackets every conv launch with events on torch's current stream (the stream the kernels are launched on)."""

    def __init__(self):
        self.pairs, self.flops, self.bytes, self.enabled, self.meta, self.kernels = [], 0.0, 0.0, False, [], 0
        self.exec_flops, self.kinds = 0.0, {}
        self._start = None

    def reset(self):
        self.pairs, self.flops, self.bytes, self.meta, self.kernels = [], 0.0, 0.0, [], 0
        self.exec_flops, self.kinds = 0.0, {}

    def __call__(self, begin, M, spec, epi=0, info=None, act_bytes=4):
        if not self.enabled:
            return
        if begin:
            self._start = torch.cuda.Event(enable_timing=True)
            self._start.record()
        else:
            stop = torch.cuda.Event(enable_timing=True)
            stop.record()
            self.pairs.append((self._start, stop))
            info = info or {"kernels": 1, "kind": "direct"}
            self.kernels += info["kernels"]                  # a call whose input exceeds the 32-bit buffer range runs as batch slices
            self.flops += 2.0 * M * spec.algo_kn
            # EXECUTED matrix-pipe flops: the F(2x2, 3x3) Winograd kernel forms 16 products per 2 x 2 outputs where the direct form has 36, the
            # F(4x4, 3x3) kernel 36 per 4 x 4 outputs where the direct form has 144,
            # ... and the F(2x2, 2x2) form of a transposed convolution 36 per 4 x 4 input patch where the direct form has 64
            ex = 2.0 * M * spec.algo_kn * {"winograd": 4.0 / 9.0, "winograd4": 2.25 / 9.0, "winograd_up4": 9.0 / 16.0}.get(info["kind"], 1.0)
            self.exec_flops += ex
            k = self.kinds.setdefault(info["kind"], [0, 0.0, 0.0, [], 0.0, 0])     # calls, algorithmic flops, executed flops, event pairs, bytes, kernels
            k[0] += 1
            k[1] += 2.0 * M * spec.algo_kn
            k[2] += ex
            k[3].append((self._start, stop))
            # algorithmic bytes of the launch: input read once + weight panel + output written (+ the epilogue's operands)
            out = M * spec.N if epi != 2 else M * spec.N       # SPADE: reads xn (M*N/2) and writes y (M*N/2)
            nbytes = float(act_bytes) * (M * spec.stride ** 2 * spec.Cin + out + (M * spec.N if epi == 1 else 0)) + \
                float(act_bytes) * spec.w.numel()
            self.bytes += nbytes
            k[4] += nbytes
            k[5] += info["kernels"]
            self.meta.append((M, spec.N, spec.Cin, spec.ntaps, spec.stride, spec.omul, 2.0 * M * spec.algo_kn, nbytes))

    def result(self):
        """(busy ms, flops, kernel launches, mean kernel-launch ms).  busy = length of the UNION of the bracketed [start, stop] intervals
        on the device clock: with one stream that is the sum of the durations; with several streams, launches of
        different streams overlap in time and share the machine, and flops / busy is the aggregate rate.  A bracket is one
        entry-point call = one kernel launch, or several when the library slices the batch (ops.LAST_CONV_KERNELS): the mean is
        per KERNEL launch, the quantity a rocprofv3 kernel trace averages."""
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
        nk = max(self.kernels, len(self.pairs))
        mean = sum(e0 - s0 for s0, e0 in iv) / nk
        return busy, self.flops, nk, mean

    def by_kind(self):
        """Per kernel family: calls, measured ms (sum of the bracketed durations), algorithmic and executed TFLOP/s."""
        out = {}
        for kind, (calls, alg, ex, pairs, nbytes, kernels) in self.kinds.items():
            ms = sum(a.elapsed_time(b) for a, b in pairs)
            out[kind] = {"calls": calls, "ms": round(ms, 3), "algorithmic_tflops": round(alg / ms / 1e9, 2) if ms > 0 else None,
                         "executed_tflops": round(ex / ms / 1e9, 2) if ms > 0 else None, "kernel_launches": kernels,
                         "algorithmic_bytes_per_launch": round(nbytes / max(kernels, 1), 1)}
        return out

    def governing(self, peak_tflops, peak_tbs=PEAK_HBM_TBS):
        """Fraction of the GOVERNING roof, launch by launch: a launch's roof time is max(flops / matrix peak, algorithmic bytes / HBM
        peak) - the 1x1 and up-sampling layers of the bf16 mode are HBM-bound, the 3x3 layers matrix-bound; sum of roof times over
        sum of measured times.  Also the share of the measured time spent in HBM-governed launches."""
        roof = meas = hbm_meas = 0.0
        for (a, b), m in zip(self.pairs, self.meta):
            t_mfma, t_hbm = m[6] / (peak_tflops * 1e12), m[7] / (peak_tbs * 1e12)
            dt = a.elapsed_time(b) * 1e-3
            roof += max(t_mfma, t_hbm)
            meas += dt
            if t_hbm > t_mfma:
                hbm_meas += dt
        return (roof / meas, hbm_meas / meas) if meas > 0 else (0.0, 0.0)

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


# ---------------------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline(case, n_frames):
    """The oracle (CPU restatement of the reference algorithm) on the host cores, a bounded sample: the per-frame loop only
    (the source state is built first, untimed - source-side work is not part of the per-frame metric)."""
    from oracle import lwg_oracle as orc
    from tests import parity_utils as pu
    model, tables, sd, info = pu.oracle_source(case)
    tgt = orc.stabilize(model, torch.tensor(case.tgt_smpls))
    first_cam = tgt[0:1, 0:3].clone()
    t0 = time.time()
    for t in range(n_frames):
        with torch.no_grad():
            r = orc.imitate_frame(model, tables, sd, info, tgt[t], first_cam, case.S, "smooth")
    dt = time.time() - t0
    assert torch.isfinite(r["pred"]).all()
    out = {"value": round(n_frames / max(dt, 1e-9), 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n_frames} frames @{case.S}x{case.S} ns={case.ns} through oracle/lwg_oracle.py (torch-CPU fp32 + "
                     f"OpenMP C rasterizer), {os.cpu_count()} host cpus"}
    ref = cpu_baseline_reference(case, n_frames)
    if ref is not None:
        out["reference"] = ref
    return out


def cpu_baseline_reference(case, n_frames):
    """The reference's OWN modules (forward_tsf, cal_bc_transform, encode_fim, lbs + grid_sample / compose) timed on this
    host, when its checkout is importable (LWG_REFERENCE, default /root/reference: the authoring container - the GPU box has no
    copy, and then this returns None and the line carries the port only).  The rasterizer call inside SMPLRenderer goes to the
    oracle's C restatement (neural_renderer is not vendored with the reference): that leg is "restatement", everything else "reference"."""
    ref_root = os.environ.get("LWG_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "iPERCore")):
        return None
    try:
        from tests.golden import make_golden as mg       # the import recipe of SURVEY 8c (stubs neural_renderer / cv2 / torchvision)
        mg.install_stubs()
        import torch.nn.functional as F
        from iPERCore.models.networks.generators.attlwb_spade_resunet import AttentionLWBGenerator
        from iPERCore.tools.human_digitalizer.bodynets.batch_smplh import SMPLH
        from iPERCore.tools.human_digitalizer.renders.nmr import SMPLRenderer
        from iPERCore.tools.utils.geometry.cam_pose_utils import WeakPerspectiveCamera
        from ipercore_amd import synthetic
        S, ns = case.S, case.ns
        tmp = synthetic.tmp_asset_dir()
        cfgdir = os.path.join(ref_root, "assets/configs/pose3d")
        render = SMPLRenderer(face_path=synthetic.write_smpl_faces_npy(os.path.join(tmp, "smpl_faces.npy")),
                              fim_enc_path=os.path.join(cfgdir, "mapper_fim_enc.txt"), uv_map_path=os.path.join(cfgdir, "mapper_uv.txt"),
                              part_path=os.path.join(cfgdir, "smpl_part_info.json"), front_path=os.path.join(cfgdir, "front_body.json"),
                              head_path=os.path.join(cfgdir, "head.json"), facial_path=os.path.join(cfgdir, "front_facial.json"),
                              map_name="uv_seg", tex_size=3, image_size=S, fill_back=False, anti_aliasing=True,
                              background_color=(0, 0, 0), has_front=True, top_k=3)
        smplh = SMPLH(model_path=synthetic.write_smplh_pickle(os.path.join(tmp, "smplh_synth.pkl"), seed=0))
        G = AttentionLWBGenerator(mg.gen_cfg(case.num_filters, case.n_res, case.bg_filters), temporal=False).eval()
        G.load_state_dict({k: torch.tensor(v) for k, v in case.state.items()}, strict=True)
        wcam = WeakPerspectiveCamera(smplh)
        t_leg = {"lbs": 0.0, "rasterize (restatement)": 0.0, "encode_fim + cal_bc_transform": 0.0, "grid_sample": 0.0, "forward_tsf": 0.0}
        with torch.no_grad():
            src = smplh.get_details(torch.tensor(case.src_smpl), 0, links_ids=None)
            src_f2pts, src_fim, _ = render.render_fim_wim(src["cam"], src["verts"], smpl_faces=True)
            src_cond, _ = render.encode_fim(fim=src_fim, transpose=True)
            enc, res = G.forward_src(torch.cat([torch.tensor(case.src_img)[0], src_cond], dim=1).unsqueeze(0), only_enc=True)
            uv_img, bg = torch.tensor(case.uv_img), torch.tensor(case.bg_img)
            f_uvs2img = render.get_f_uvs2img(1)
            tgt = wcam.stabilize(torch.tensor(case.tgt_smpls))
            first_cam = tgt[0:1, 0:3].clone()
            t_all = time.time()
            for t in range(n_frames):
                t0 = time.time()
                cam = WeakPerspectiveCamera.cam_swap(src["cam"][0:1], tgt[t:t + 1, 0:3], first_cam, "smooth")
                ref = smplh.get_details(torch.cat([cam, tgt[t:t + 1, 3:-10], src["shape"][0:1]], dim=1), 0, links_ids=None)
                t1 = time.time()
                _, fim, wim = render.render_fim_wim(ref["cam"], ref["verts"], smpl_faces=True)
                t2 = time.time()
                cond, _ = render.encode_fim(fim=fim, transpose=True)
                Tuv2t = render.cal_bc_transform(f_uvs2img.clone(), fim, wim)
                Tst = render.cal_bc_transform(src_f2pts, fim.repeat(ns, 1, 1), wim.repeat(ns, 1, 1, 1))
                t3 = time.time()
                syn = F.grid_sample(uv_img, Tuv2t)
                t4 = time.time()
                img, mask = G.forward_tsf(torch.cat([syn, cond], dim=1), enc, res, Tst.view(1, ns, S, S, 2))
                pred = mask * bg + (1 - mask) * img
                t5 = time.time()
                for k, d in zip(t_leg, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                    t_leg[k] += d
            dt = time.time() - t_all
        assert torch.isfinite(pred).all()
        return {"value": round(n_frames / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "reference",
                "legs_ms_per_frame": {k: round(v / n_frames * 1e3, 1) for k, v in t_leg.items()},
                "frames_per_s_reference_legs_only": round(n_frames / max(sum(v for k, v in t_leg.items() if "restatement" not in k), 1e-9), 4),
                "sample": f"{n_frames} frames @{S}x{S} ns={ns}: the reference's own SMPLH.get_details, SMPLRenderer.encode_fim / "
                          f"cal_bc_transform, F.grid_sample and AttentionLWBGenerator.forward_tsf (torch-CPU fp32, imported from {ref_root}); "
                          "its external rasterizer call = the oracle's C restatement"}
    except Exception as e:                       # the baseline is a courtesy measurement: never fail the bench line over it
        return {"kind": "reference", "error": f"{type(e).__name__}: {e}"}


# ---------------------------------------------------------------------------------------------------------- extra measurements
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


def rerender_check(im, tgt, video, small_fb=1, n_pick=6, cam="smooth"):
    """Self-check of a separately reported measurement, outside its timed loop: ``n_pick`` frames spread over the clip are rendered
    again in launches of ``small_fb`` frames (1 = the reference's calling convention, imitator.py:341) on ONE stream and must equal
    the measured loop's frames bit for bit - every launch shape the loop used against the small-batch kernels."""
    from ipercore_amd import ops
    n = video.shape[0]
    idx = sorted(set(int(round(x)) for x in np.linspace(0, n - 1, n_pick)))
    prev = im.frame_batch, im.streams, ops.CONV_HOOK
    im.frame_batch, im.streams, ops.CONV_HOOK = small_fb, 1, None
    try:
        bad = []
        for t in idx:
            lo = min(t, n - small_fb)
            blk = im.synthesize(tgt[lo:lo + small_fb], cam, t0=lo)
            if not torch.equal(blk[t - lo], video[t]):
                bad.append(t)
        torch.cuda.synchronize()
    finally:
        im.frame_batch, im.streams, ops.CONV_HOOK = prev
    return {"result": "bitwise" if not bad else f"MISMATCH at frames {bad}", "frames": idx,
            "against": f"the same frames rendered in launches of {small_fb} (frame_batch = {small_fb}), one stream"}


def _timed_clips(render, W, K):
    for _ in range(W):
        render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(K):
        out = render()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def split_products(im, render, n, W, K, ref_video):
    """Reported separately: the same clip with every Cin % 32 == 0 convolution on the bf16x6 kernel
    (csrc/conv_igemm_split.hip: fp32 in / out / accumulate, each fp32 product formed from six bf16 MFMAs over an exact
    three-way split of both operands).  Not the headline value; `max_abs_diff_vs_fp32_path` compares the frames."""
    from ipercore_amd import ops
    hook, ops.CONV_HOOK = ops.CONV_HOOK, None
    prev = im.generator.conv_precision
    im.generator.conv_precision = "split"
    try:
        dt, video = _timed_clips(render, W, K)
        diff = (video - ref_video).abs().max().item()
        return {"value": round(K * n / dt, 3), "unit": "frames/s", "ms_per_clip": round(dt / K * 1e3, 3), "clips": K,
                "self_check": "allclose (max |d| <= 2e-3 vs the fp32 path's frames of the same clip)" if diff <= 2e-3 else f"MISMATCH: max |d| = {diff:.3e}",
                "max_abs_diff_vs_fp32_path": diff, "frames_range": "[-1, 1]",
                "what": "bf16x6: exact 3-way bf16 split of both fp32 operands, 6 bf16 MFMAs per product, fp32 accumulation"}
    finally:
        im.generator.conv_precision = prev
        ops.CONV_HOOK = hook


def direct_products(im, render, n, W, K, ref_video):
    """Reported separately: the same clip with EVERY convolution on the direct implicit-GEMM kernel (generator.conv_precision = "fp32", the
    rounds 1-4 default): what the F(2x2,3x3) Winograd engine of the 3x3 / stride 1 layers buys, and how far the frames of the two engines are apart."""
    from ipercore_amd import ops
    hook, ops.CONV_HOOK = ops.CONV_HOOK, None
    prev = im.generator.conv_precision
    im.generator.conv_precision = "fp32"
    try:
        dt, video = _timed_clips(render, W, K)
        diff = (video - ref_video).abs().max().item()
        return {"value": round(K * n / dt, 3), "unit": "frames/s", "ms_per_clip": round(dt / K * 1e3, 3), "clips": K,
                "self_check": "allclose (max |d| <= 1e-4 vs the default engine's frames of the same clip)" if diff <= 1e-4 else f"MISMATCH: max |d| = {diff:.3e}",
                "max_abs_diff_vs_default_engine": diff, "frames_range": "[-1, 1]",
                "what": "every convolution as a direct fp32 MFMA implicit GEMM (lwg_conv_igemm_kernel)"}
    finally:
        im.generator.conv_precision = prev
        ops.CONV_HOOK = hook


def winograd_products(im, render, n, W, K, ref_video):
    """Reported separately: the same clip with the 3x3 / stride 1 convolutions (plain and residual epilogues) as fused F(2x2, 3x3) Winograd
    convolutions on the fp32 matrix pipe (csrc/conv_winograd.hip; 16 multiplies per 2x2 outputs instead of 36, fp32 throughout).  Not the
    headline value: its sums are fp32-grade but not the direct kernel's bits."""
    from ipercore_amd import ops
    hook, ops.CONV_HOOK = ops.CONV_HOOK, None
    prev = im.generator.conv_precision
    im.generator.conv_precision = "winograd"
    try:
        dt, video = _timed_clips(render, W, K)
        diff = (video - ref_video).abs().max().item()
        return {"value": round(K * n / dt, 3), "unit": "frames/s", "ms_per_clip": round(dt / K * 1e3, 3), "clips": K,
                "self_check": "allclose (max |d| <= 1e-4 vs the fp32 path's frames of the same clip)" if diff <= 1e-4 else f"MISMATCH: max |d| = {diff:.3e}",
                "max_abs_diff_vs_fp32_path": diff, "frames_range": "[-1, 1]",
                "what": "F(2x2,3x3) Winograd on v_mfma_f32_32x32x2_f32 for every 3x3 / stride 1 convolution with Cin % 32 == 0 (plain, residual, SPADE "
                        "epilogues, skip concatenations); strided, transposed, 1x1 and first-layer launches stay on the direct kernel"}
    finally:
        im.generator.conv_precision = prev
        ops.CONV_HOOK = hook


def pipelined(im, render, n, W, K, n_streams, tgt=None):
    """Reported separately: the same clip with independent frame batches in flight on several HIP streams (Imitator(streams=n)), so
    that one batch's launch gaps, kernel tails and HBM-bound kernels overlap another batch's MFMA work.  Not the headline value (the
    per-kernel roofline accounting needs launches that own the machine)."""
    from ipercore_amd import ops
    hook, ops.CONV_HOOK = ops.CONV_HOOK, None
    prev, prev_fb = im.streams, im.frame_batch
    im.streams = n_streams
    im.frame_batch = min(im.frame_batch, -(-n // n_streams))         # one batch per stream (the headline runs the clip as one batch)
    try:
        dt, video = _timed_clips(render, W, K)
        assert torch.isfinite(video).all()
        out = {"value": round(K * n / dt, 3), "unit": "frames/s", "streams": n_streams, "frame_batch": im.frame_batch,
               "ms_per_clip": round(dt / K * 1e3, 3), "clips": K}
        if tgt is not None:
            chk = rerender_check(im, tgt, video)
            out["self_check"], out["self_check_detail"] = chk["result"], chk
        return out
    finally:
        im.streams, im.frame_batch = prev, prev_fb
        ops.CONV_HOOK = hook


def b1_latency(im, tgt, timer, n_frames=64):
    """frame_batch = 1 - the reference's calling convention (one frame per Imitator.forward, imitator.py:341): per-frame time when single
    frames are issued back to back (eager launches, ~85 per frame) and the wall time of ONE frame from an idle queue (launch to last
    byte); the conv kernel's roofline fraction in that regime comes from a pass with HIP events around every conv launch (the
    64x64-feature layers then have 4096 GEMM rows: one 128x128 tile row per 8 CUs).  (Rounds 2-3 also replayed the frame as one hipGraph:
    equal to eager launches within 0.5 % - the GPU is the bound - so that path was removed.)"""
    from ipercore_amd import ops
    prev = im.frame_batch
    im.frame_batch = 1
    hook = ops.CONV_HOOK
    try:
        ops.CONV_HOOK = None
        n = min(n_frames, tgt.shape[0])
        im.synthesize(tgt[:8], "smooth")
        idle = []
        for i in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            im.synthesize(tgt[i:i + 1], "smooth", t0=i)
            torch.cuda.synchronize()
            idle.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        im.synthesize(tgt[:n], "smooth")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = {"frame_batch": 1, "frames": n, "mode": "eager launches", "ms_per_frame_back_to_back": round(dt / n * 1e3, 3),
               "frames_per_s": round(n / dt, 2), "ms_one_frame_from_idle_median": round(float(np.median(idle)), 3),
               "ms_one_frame_from_idle_min": round(min(idle), 3)}
        # single frames alternating over 2 / 3 HIP streams (Imitator(streams=k)): a one-frame launch of a 64 x 64 layer is 64-128 workgroups on 256
        # CUs - the frame on the other stream takes the idle half.  Same kernels, same bits; the latency of ONE frame from idle does not change.
        prev_streams = im.streams
        try:
            ref = im.synthesize(tgt[:n], "smooth") if n <= 64 else None
            for k in (2, 3):
                im.streams, im._side_streams = k, None
                im.synthesize(tgt[:8], "smooth")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fr = im.synthesize(tgt[:n], "smooth")
                torch.cuda.synchronize()
                out[f"ms_per_frame_back_to_back_{k}_streams"] = round((time.perf_counter() - t0) / n * 1e3, 3)
                if ref is not None:
                    out[f"bitwise_{k}_streams_vs_1"] = bool(torch.equal(fr, ref))
        finally:
            im.streams, im._side_streams = prev_streams, None
        # the latency engine (generator.conv_precision = "winograd2x2": the F(4x4,3x3) kernel off): a one-frame F(4x4,3x3) launch of a 64 x 64 layer is 32
        # workgroups of 2.25 K stages' worth each; the F(2x2,3x3) kernel's small-launch form gives the same layer 128.  Each engine is batch-invariant in
        # itself (checked here: frames of one-frame launches = the same frames in a batch of 8); the two engines' frames differ at the 1e-5 level.
        prev_prec = im.generator.conv_precision
        if prev_prec == "winograd":
            try:
                ref1 = im.synthesize(tgt[:8], "smooth")
                im.generator.conv_precision = "winograd2x2"
                im.synthesize(tgt[:8], "smooth")
                idle2 = []
                for i in range(10):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    im.synthesize(tgt[i:i + 1], "smooth", t0=i)
                    torch.cuda.synchronize()
                    idle2.append((time.perf_counter() - t0) * 1e3)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fr2 = im.synthesize(tgt[:n], "smooth")
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
                im.frame_batch = 8
                b8 = im.synthesize(tgt[:8], "smooth")
                im.frame_batch = 1
                im.streams, im._side_streams = 2, None
                im.synthesize(tgt[:8], "smooth")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                im.synthesize(tgt[:n], "smooth")
                torch.cuda.synchronize()
                dt2s = time.perf_counter() - t0
                out["latency_engine"] = {"conv_precision": "winograd2x2", "what": "every eligible 3x3 layer on the F(2x2,3x3) kernel (the F(4x4,3x3) kernel off)",
                                         "ms_per_frame_back_to_back": round(dt2 / n * 1e3, 3), "ms_one_frame_from_idle_median": round(float(np.median(idle2)), 3),
                                         "ms_per_frame_back_to_back_2_streams": round(dt2s / n * 1e3, 3),
                                         "bitwise_frame_batch_1_vs_8": bool(torch.equal(fr2[:8], b8)),
                                         "max_abs_diff_vs_default_engine": round((fr2[:8] - ref1).abs().max().item(), 8)}
            finally:
                im.generator.conv_precision = prev_prec
                im.streams, im._side_streams = prev_streams, None
                im.frame_batch = 1
        timer.reset()
        timer.enabled, ops.CONV_HOOK = True, timer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        im.synthesize(tgt[:n], "smooth")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timer.enabled = False
        conv_ms, conv_flops, n_launch, mean_ms = timer.result()
        ach = timer.exec_flops / (conv_ms * 1e-3) / 1e12          # executed matrix-pipe flops (the Winograd launches execute 4/9 of the algorithmic)
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "algorithmic_equivalent_tflops": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                           "launches": n_launch, "avg_launch_us": round(mean_ms * 1e3, 2),
                           "conv_ms_per_frame": round(conv_ms / n, 3), "measured_in": "a pass with events around every conv launch",
                           "share_of_time": round(conv_ms * 1e-3 / dt, 4)}
        return out
    finally:
        im.frame_batch = prev
        ops.CONV_HOOK = hook
        timer.reset()


def novel_view_smpls(case, hands_mean, length=180):
    """BASELINE configs[3] poses as services/run_viewer.py:69-77 builds them: create_T_pose_novel_view_smpl(180) (base_runner.py:11-30),
    the source's shape and body pose (T_pose = False), add_hands_params_to_smpl."""
    from ipercore_amd.imitator import add_hands_params_to_smpl, create_T_pose_novel_view_smpl
    nv = create_T_pose_novel_view_smpl(length)
    nv[:, -10:] = case.src_smpl[0, -10:]
    nv[:, 6:-10] = case.src_smpl[0, 6:-10]
    return add_hands_params_to_smpl(nv, hands_mean).astype(np.float32)


def novel_view_1024_bf16(dev, timer, W, K):
    """BASELINE configs[3] on this GPU, reported beside the headline: 1024x1024, the 180 novel-view poses, bf16 MFMA conv tiles with
    bf16 activation storage, fp32 renderer; a step = the 180-frame clip."""
    from ipercore_amd import ops, synthetic as syn
    S, n = 1024, 180
    case = syn.build_case(image_size=S, n_frames=1, ns=2)
    FB = default_frame_batch("bf16", S)      # the 180 poses as one launch batch (see FB_512)
    im = syn.make_imitator(case, frame_batch=FB, device=dev)
    im.generator.conv_precision = "bf16"
    im.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
    tgt = im.prepare_sequence(novel_view_smpls(case, im.body_rec.np_hands_mean, n), "smooth")
    hook = ops.CONV_HOOK
    try:
        ops.CONV_HOOK = None
        for _ in range(W):
            im.synthesize(tgt, "smooth")
        timer.reset()
        timer.enabled, ops.CONV_HOOK = True, (lambda b, M, spec, epi=0, info=None: timer(b, M, spec, epi, info, 2))     # bf16 tensors: 2 bytes per element
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            video = im.synthesize(tgt, "smooth")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timer.enabled = False
        assert torch.isfinite(video).all()
        chk = rerender_check(im, tgt, video, small_fb=2)         # 1024x1024 bf16: against batches of 2 (check_benched_shapes_1024_bf16's form)
        conv_ms, conv_flops, n_launch, mean_ms = timer.result()
        ach = conv_flops / (conv_ms * 1e-3) / 1e12
        gov, hbm_share = timer.governing(PEAK_BF16_MFMA_TFLOPS)
        nbytes_launch = timer.bytes / max(n_launch, 1)
        timer.enabled, ops.CONV_HOOK = False, None
        prev_streams, im.streams = im.streams, 3             # the same clip with three frame batches in flight on HIP streams
        prev_fb, im.frame_batch = im.frame_batch, min(im.frame_batch, 20)    # nine batches over three streams (60-frame batches on three streams measured 485 frames/s: three 20 GB working sets)
        try:
            im.synthesize(tgt, "smooth")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(K):
                im.synthesize(tgt, "smooth")
            torch.cuda.synchronize()
            piped = K * n / (time.perf_counter() - t1)
        finally:
            im.streams, im.frame_batch = prev_streams, prev_fb
        return {"value": round(K * n / dt, 2), "unit": "frames/s", "frames_per_clip": n, "clips": K, "frame_batch": min(im.frame_batch, n),
                "frame_batch_requested": FB, "image_size": S, "self_check": chk["result"], "self_check_detail": chk,
                "pipelined_3_streams_frames_per_s": round(piped, 2),
                "dtype": "bf16 MFMA operands + bf16 activation storage, f32 accumulation / renderer",
                "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4), "launches": n_launch, "avg_launch_us": round(mean_ms * 1e3, 2),
                             "algorithmic_gflop_per_frame": round(conv_flops / (K * n) / 1e9, 1),
                             "algorithmic_bytes_per_launch": round(nbytes_launch, 1),
                             "hbm_time_at_peak_us_per_launch": round(nbytes_launch / (PEAK_HBM_TBS * 1e12) * 1e6, 2),
                             "frac_of_governing_roof": round(gov, 4), "share_of_conv_time_in_hbm_governed_launches": round(hbm_share, 4),
                             "share_of_time": round(conv_ms * 1e-3 / dt, 4)}}
    finally:
        ops.CONV_HOOK = hook
        timer.reset()


def size_extra(dev, timer, S, W=2, K=2):
    """the metric is quoted "at 256 / 512 / 1024" - the same fp32 per-frame path at another image size in a short loop
    (W warm-up + K timed clips), with the conv kernel's roofline fraction from HIP events.  Reported beside the headline, never as it."""
    from ipercore_amd import ops, synthetic as syn
    n = {256: 300, 1024: 96}.get(S, 96)
    FB = default_frame_batch("fp32", S)
    torch.cuda.empty_cache()                     # (the 1024 x 1024 clip needs ~100 GB of new blocks: start from an empty pool - on a fragmented one the timed
                                                 #  clips of round 6's evidence run spent 80 % of their wall time in the allocator: 66 instead of 245-288 frames/s)
    case = syn.build_case(image_size=S, n_frames=n, ns=2)
    im = syn.make_imitator(case, frame_batch=FB, device=dev)
    tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
    hook = ops.CONV_HOOK
    try:
        ops.CONV_HOOK = None
        for _ in range(W):
            im.synthesize(tgt, "smooth")
        timer.reset()
        timer.enabled, ops.CONV_HOOK = True, (lambda b, M, spec, epi=0, info=None: timer(b, M, spec, epi, info, 4))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            video = im.synthesize(tgt, "smooth")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timer.enabled = False
        assert video.shape[0] == n and torch.isfinite(video).all()
        conv_ms, conv_flops, n_launch, mean_ms = timer.result()
        chk = rerender_check(im, tgt, video)
        ach = timer.exec_flops / (conv_ms * 1e-3) / 1e12          # executed flops (Winograd launches: 4/9 of the algorithmic)
        return {"value": round(K * n / dt, 2), "unit": "frames/s", "image_size": S, "dtype": "f32", "frames_per_clip": n, "clips": K,
                "frame_batch": min(im.frame_batch, n), "self_check": chk["result"], "self_check_detail": chk,
                "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "algorithmic_equivalent_tflops": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                             "launches": n_launch, "avg_launch_us": round(mean_ms * 1e3, 2),
                             "algorithmic_gflop_per_frame": round(conv_flops / (K * n) / 1e9, 2), "share_of_time": round(conv_ms * 1e-3 / dt, 4)}}
    except Exception as e:                       # an extra must never take the headline line with it
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        ops.CONV_HOOK = hook
        timer.reset()
        del im
        torch.cuda.empty_cache()


def personalize_step_extra(steps=10, warmup=4, size=512, timeout_s=900, extra_args=()):
    """BASELINE configs[4] beside the headline: bench_personalize.py in its OWN process (the step captures hipGraphs and owns
    its allocator pools; a failure there must not take the headline line with it) -> its JSON line, or {"error": ...}.
    extra_args: e.g. ("--use-vgg", "--use-face") = the reference's default loss set (deploy.toml:76-102)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench_personalize.py"), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
           "--size", str(size)] + list(extra_args)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": f"bench_personalize.py exited {r.returncode} without a result line", "stderr_tail": r.stderr[-600:]}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def shard_of_8(im, tgt, n_clip, reps=6):
    """Reported separately, and NOT a hardware measurement of 8 GPUs: what ONE rank of BASELINE configs[2] does (300 frames over 8 ranks = shards of
    38 / 37 frames, models/imitator.py:298-299,327-382), timed on this one GPU - the sharded frame loop of ipercore_amd.sharding exactly as a rank
    runs it (chunk plan, post-processing, async all-gather per chunk through RCCL in a ONE-rank group: the collective is issued and waited
    for, the volume is 1/8 of the real one).  compute_ceiling_8gpu_fps = 300 / t_shard is the bound the driver's 8-GPU line cannot exceed."""
    import torch.distributed as dist
    from ipercore_amd import ops, sharding
    hook, ops.CONV_HOOK = ops.CONV_HOOK, None
    prev_fb = im.frame_batch
    out = {"what": "one rank's share of the 300-frame clip at N = 8, run on one GPU (no 8-GPU hardware curve was measured)",
           "shard_frames": max(sharding.shard_counts(n_clip, 8))}
    own_group = False
    try:
        n = out["shard_frames"]
        shard = tgt[:n]
        try:
            if not dist.is_initialized():
                port = 29500 + (os.getpid() % 2000)
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=tgt.device)
                own_group = True
            rccl = True
        except Exception as e:       # noqa: BLE001
            rccl = False
            out["rccl_error"] = f"{type(e).__name__}: {e}"[:200]

        def timed(fn):
            fn()
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return min(ts), sorted(ts)[len(ts) // 2]

        variants = {}
        for fb in (32, n):
            im.frame_batch = fb
            plan = [m for _, m in sharding.chunk_plan(n_clip, 8, fb, sharding.round_frames_of(im))]
            tag = "+".join(str(m) for m in plan)

            def compute_only():
                off = 0
                for m in plan:
                    im.synthesize(shard[off:off + m], "smooth", t0=off)
                    off += m
            variants[f"chunks_{tag}_compute_only_ms"] = [round(v, 3) for v in timed(compute_only)]
            if rccl:
                for post_name, post in (("f32", None), ("u8", ops.frames_to_u8)):
                    def full(post=post):
                        sharding.sharded_synthesize(im, shard, "smooth", prepared=True, post=post, force_collective=True)
                    variants[f"chunks_{tag}_{post_name}_exchange_ms"] = [round(v, 3) for v in timed(full)]
        out["ms_min_median"] = variants
        best = min(v[0] for k, v in variants.items() if "exchange" in k or not rccl)
        out["t_shard_ms"] = round(best, 3)
        out["frames_per_s_per_rank"] = round(n / best * 1e3, 1)
        out["compute_ceiling_8gpu_fps"] = round(n_clip / best * 1e3, 1)
        return out
    finally:
        im.frame_batch = prev_fb
        ops.CONV_HOOK = hook
        if own_group:
            try:
                dist.destroy_process_group()
            except Exception:        # noqa: BLE001
                pass


def _extra(fn, *a, **kw):
    """A separately reported measurement must never take the headline line with it: an exception becomes {"error": ...}."""
    try:
        return fn(*a, **kw)
    except Exception as e:       # noqa: BLE001
        try:
            torch.cuda.synchronize()
        except Exception:        # noqa: BLE001
            pass
        return {"error": f"{type(e).__name__}: {e}"[:500]}


def main(argv=None):
    import faulthandler
    faulthandler.enable()                    # a crash inside a native library leaves a Python traceback on stderr
    # ONE JSON line on stdout, whatever the libraries print (RCCL writes its version banner to stdout when a communicator is created): file
    # descriptor 1 points at stderr for the whole run, the line goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)    # the shader clock needs ~0.2 s of load to settle
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frames", type=int, default=0, help="frames of the reference clip rendered per step (0 = 300, BASELINE configs[2]; "
                                                          "180 for --workload novel_view)")
    ap.add_argument("--mode", choices=("clip", "batch"), default="clip",
                    help="clip: a step = the whole clip, frame-sharded over the ranks (strong scaling, the default); batch: a step = one "
                         "frame batch per rank (weak scaling, the round-1 measurement)")
    ap.add_argument("--frame-batch", type=int, default=0,
                    help="frames per launch batch; 0 = 32 (fp32) / 48 (bf16) at 512x512 scaled by (512/size)^2, clamped to [2, 64]: "
                         "measured in one process 472 / 476 / 478 frames/s at 16 / 24 / 32 (fp32, 512x512) and 723 / 740 / 739 at "
                         "12 / 16 / 20 -> 746 / 754 / 757 (bf16, 1024x1024)")
    ap.add_argument("--gather-dtype", choices=("auto", "f32", "u8"), default="auto",
                    help="N > 1: what the all-gather exchanges.  f32 (the default, 'auto' = f32): the (n,3,S,S) fp32 video Imitator.inference "
                         "returns - the same result tensor as at N = 1 and as the reference's; u8: the (n,S,S,3) uint8 video its PNG writer "
                         "consumes (device-side conversion, a quarter of the bytes on the per-link-bound xGMI ring).  With f32 the u8 form "
                         "is measured too, in its own short loop, and reported as `exchange_u8` - never as `value`")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="torch.distributed backend; nccl = RCCL (the product)")
    ap.add_argument("--device", choices=("cuda", "cpu"), default="cuda",
                    help="cpu: plumbing dry run for the CPU test-suite ONLY (tests/test_bench_launch.py installs the emulated C ABI "
                         "around main(); without it every op raises on CPU tensors - there is no CPU product path)")
    ap.add_argument("--no-overlap-gather", dest="overlap", action="store_false", help="one all-gather after the frame loop")
    ap.add_argument("--chunk-plan", choices=("auto", "batches", "one"), default="auto",
                    help="N > 1: the chunk schedule of a rank's shard.  batches: frame batches of --frame-batch, each exchanged behind the next one's "
                         "synthesis (38 frames -> 24 + 14); one: the shard as ONE chunk (no launch set cut, the exchange exposed); auto: whichever the "
                         "ring model of sharding.choose_chunk_plan predicts faster from this run's measured per-frame time")
    ap.add_argument("--cpu-frames", type=int, default=3, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--precision", choices=("fp32", "bf16", "split", "winograd"), default="winograd",
                    help="bf16: BASELINE configs[3] mode - bf16 MFMA operands and bf16 activation storage in the convs (fp32 accumulation); "
                         "the headline metric (configs[1]) is fp32")
    ap.add_argument("--workload", choices=("imitate", "novel_view"), default="imitate",
                    help="novel_view: BASELINE configs[3] poses - create_T_pose_novel_view_smpl(180), global rotation y = 0..360")
    ap.add_argument("--streams", type=int, default=1, help="frame batches in flight on separate HIP streams (see DESIGN.md 5)")
    ap.add_argument("--pipelined-streams", type=int, default=3,
                    help="extra (separately reported) measurement with this many frame batches in flight; 0/1 = skip")
    ap.add_argument("--no-conv-events", action="store_true")
    ap.add_argument("--no-extras", dest="extras", action="store_false",
                    help="skip every separately reported measurement (pipelined, split_products, with_output, b1_latency, "
                         "novel_view_1024_bf16, personalize_step)")
    ap.add_argument("--no-split-extra", dest="split_extra", action="store_false")
    ap.add_argument("--output-frames", type=int, default=300, help="frames of the with-output measurement (0 = skip; 300 = the whole clip)")
    ap.add_argument("--conv-breakdown", action="store_true", help="write gpurun_out/conv_breakdown.json")
    ap.add_argument("--tiny-arch", action="store_true", help="reduced-width generator (plumbing tests only; never a reported number)")
    ap.add_argument("--no-self-check", dest="self_check", action="store_false")
    ap.add_argument("--only-extras", default="", help="lab: comma-separated names of the separately reported measurements to run (default: all)")
    ap.add_argument("--lab-no-wino4", action="store_true", help="lab A/B: the F(4x4,3x3) kernel off (every eligible 3x3 layer on the F(2x2,3x3) kernel, rounds 5-6's engine)")
    ap.add_argument("--no-sizes-extra", dest="sizes_extra", action="store_false")
    ap.add_argument("--no-exchange-u8", dest="exchange_u8", action="store_false", help="N > 1 with the f32 exchange: skip the extra uint8-exchange loop")
    args = ap.parse_args(argv)
    self_launch_if_needed(sys.argv[1:] if argv is None else argv)      # N > 1 without torchrun: spawn the ranks ourselves

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = args.device == "cuda"
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs the MI355X"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
    else:
        dev = torch.device("cpu")
        args.no_conv_events, args.extras, args.cpu_frames = True, False, 0

        def sync():
            return None
    if world > 1:
        if on_gpu and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0" and rank == 0:
            print("[bench] HSA_ENABLE_IPC_MODE_LEGACY is not 0: RCCL's IPC handles may fail on this driver", file=sys.stderr)
        if args.backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group("gloo", init_method="env://")
        assert dist.get_world_size() == world
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.gather_dtype == "auto":
        args.gather_dtype = "f32"

    from ipercore_amd import ops, sharding, synthetic as pu      # product path only; the oracle is imported in cpu_baseline()
    if args.lab_no_wino4:
        ops.WINO4 = False

    S = args.size
    # frames per launch batch (FB_512 above): N = 1 - the whole clip as one batch; N > 1 - 32 at 512x512 fp32 (the 64x64-feature layers have
    # 64 output tiles per frame: every multiple of 8 frames is a whole number of rounds of the 256 CUs at two workgroups each)
    FB = args.frame_batch or default_frame_batch(args.precision, S, world)
    K, W = args.steps, args.warmup
    clip = args.mode == "clip"
    n_clip = args.frames or (180 if args.workload == "novel_view" else 300)
    n_seq = n_clip if clip else (K + W) * FB * world
    arch = dict(num_filters=[64, 64, 128], n_res=2, bg_filters=[64, 64, 128]) if args.tiny_arch else {}
    case = pu.build_case(image_size=S, n_frames=n_seq, ns=2, **arch)
    im = pu.make_imitator(case, frame_batch=FB, device=dev)
    im.streams = max(1, args.streams)
    if args.workload == "novel_view":
        nv = novel_view_smpls(case, im.body_rec.np_hands_mean, 180)
        case.tgt_smpls = np.concatenate([nv] * (n_seq // 180 + 1), axis=0)[:n_seq]
    im.generator.conv_precision = args.precision
    is_f32 = args.precision in ("fp32", "winograd")           # fp32 tensors and fp32 MFMA arithmetic (BASELINE configs[1] / [2])
    headline = args.precision == "winograd"                   # the product's default engine
    if args.precision != "winograd":                          # the sources were encoded in the default mode: again in this one
        im.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
    FB_requested, FB = FB, im.frame_batch        # (no clamp any more: batches beyond the kernels' 3 GiB buffer range are sliced inside the C entry points)
    if rank == 0:                                                # what tools/pmc_summary.py stamps the PMC traffic file with
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_last_config.json"), "w") as fp:
                json.dump({"frame_batch": FB, "image_size": S, "precision": args.precision, "workload": args.workload, "frames": n_clip}, fp)
        except OSError:
            pass
    tgt = im.prepare_sequence(case.tgt_smpls, "smooth")          # sequence-global pre-pass, every rank identically
    act_bytes = 2 if args.precision == "bf16" else 4

    timer = ConvTimer()
    hook = None if args.no_conv_events else (lambda b, M, spec, epi=0, info=None: timer(b, M, spec, epi, info, act_bytes))
    ops.CONV_HOOK = hook

    def to_u8(x):
        return ops.frames_to_u8(x) if x.shape[0] else torch.empty((0, S, S, 3), device=x.device, dtype=torch.uint8)
    # the exchange format; at N = 1 nothing is exchanged and the fp32 video is the result
    post = to_u8 if (args.gather_dtype == "u8" and world > 1) else None
    fmt = {"post": post}
    stats = {}

    plan_box = {"plan": None, "model": None}       # N > 1: the chunk schedule, chosen after the first warm-up step (below)
    if clip:
        def step(i):
            st = {"sync": sync} if world > 1 else {}
            v = sharding.sharded_synthesize(im, tgt, "smooth", gather=True, overlap=args.overlap, prepared=True, post=fmt["post"], stats=st,
                                            plan=plan_box["plan"])
            stats.setdefault("exposed_gather_s", []).append(st.get("exposed_gather_s"))
            stats.setdefault("compute_s", []).append(st.get("compute_s"))
            stats.update({k: st[k] for k in ("shard", "bytes_received", "chunks", "chunk_lengths") if k in st})
            return v
        frames_per_step = n_clip
    else:
        lo, hi = sharding.shard_range(tgt.shape[0], rank, world)
        mine = tgt[lo:hi].contiguous()

        def step(i):
            chunk = mine[i * FB:(i + 1) * FB]
            tsf8, Tst, _ = im.make_inputs_for_tsf(im.src_info, chunk, "smooth", t=lo + i * FB)
            out = im.forward(tsf8, Tst)[0]
            return out if world == 1 else sharding.all_gather_frames(out, FB * world)
        frames_per_step = FB * world

    last = None
    if world > 1 and clip and args.overlap and args.chunk_plan == "auto":
        # the chunk schedule of the timed steps (sharding.choose_chunk_plan: one chunk per shard - no launch set cut, the whole exchange exposed - or
        # frame batches exchanged behind each other's synthesis), from THIS run's own per-frame time: one untimed step on the default plan, the
        # slowest rank's compute time per frame, the per-link ring model of the exchange; every rank computes the same choice
        step(0)
        step(0)
        t_f = torch.tensor([(stats["compute_s"][-1] or 0.0) / max(1, -(-n_clip // world))], device=dev, dtype=torch.float64)
        dist.all_reduce(t_f, op=dist.ReduceOp.MAX)
        bpf = S * S * 3 * (1 if args.gather_dtype == "u8" else 4)
        plan_box["plan"], plan_box["model"] = sharding.choose_chunk_plan(n_clip, world, FB, sharding.round_frames_of(im), bpf, float(t_f.item()))
    elif world > 1 and clip and args.overlap and args.chunk_plan == "one":
        plan_box["plan"] = [(0, max(sharding.shard_counts(n_clip, world)))]
    for i in range(W):                       # untimed: settles the clock and, for N > 1, sets up RCCL's channels at the timed sizes
        last = step(i)
    if last is None and world > 1:
        last = step(0)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    stats.pop("exposed_gather_s", None)
    stats.pop("compute_s", None)
    timer.reset()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(W, W + K):
        last = step(i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    timer.enabled = False
    assert last.shape[0] == frames_per_step and bool(torch.isfinite(last.float()).all())

    # self-check, outside the timed region: 8 frames spread over the clip rendered again ONE frame per launch (the reference's calling
    # convention, imitator.py:341) must equal the frames of the timed step bit for bit - every launch shape the timed loop used (full
    # frame batches, the tail, at N > 1 the other ranks' shards through the exchange) against the B = 1 kernels
    self_check = None
    if args.self_check and clip and last.shape[0] == n_clip:
        idx = sorted(set(int(round(x)) for x in np.linspace(0, n_clip - 1, 8)))
        prev_fb, prev_streams, prev_hook = im.frame_batch, im.streams, ops.CONV_HOOK
        im.frame_batch, im.streams, ops.CONV_HOOK = 1, 1, None
        try:
            bad = []
            for t in idx:
                one = im.synthesize(tgt[t:t + 1], "smooth", t0=t)
                one = one if post is None else post(one)
                same = torch.equal(one[0], last[t]) if on_gpu else \
                    bool((one[0].float() - last[t].float()).abs().max() <= (1.0 if one.dtype == torch.uint8 else 2e-4))
                if not same:
                    bad.append(t)
            sync()
        finally:
            im.frame_batch, im.streams, ops.CONV_HOOK = prev_fb, prev_streams, prev_hook
        assert not bad, f"self-check failed: frames {bad} of the timed step differ from their frame_batch = 1 rendering"
        self_check = {"result": "bitwise" if on_gpu else "allclose (cpu plumbing run)", "frames": idx, "against": "the same frames rendered one per launch (frame_batch = 1)"}

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    per_rank = None
    if world > 1 and clip:
        mine_stats = {"rank": rank, "shard": list(stats.get("shard", ())), "frames": stats["shard"][1] - stats["shard"][0],
                      "exposed_gather_ms_per_step": round(1e3 * float(np.mean([x for x in stats.get("exposed_gather_s", []) if x is not None] or [0.0])), 3),
                      "compute_ms_per_step": round(1e3 * float(np.mean([x for x in stats.get("compute_s", []) if x is not None] or [0.0])), 3),
                      "bytes_received_per_step": stats.get("bytes_received"), "chunk_lengths": stats.get("chunk_lengths")}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_stats)

    # N > 1, fp32 exchange in the headline: the uint8 exchange (what a PNG-writing consumer needs) in its own short loop, same protocol
    exchange_u8 = None
    if world > 1 and clip and args.gather_dtype == "f32" and args.exchange_u8:
        fmt["post"], prev_hook, ops.CONV_HOOK = to_u8, ops.CONV_HOOK, None
        try:
            Ku = max(2, K // 5)
            v8 = step(0)
            sync()
            dist.barrier()
            sync()
            t1 = time.perf_counter()
            for i in range(Ku):
                v8 = step(i)
            sync()
            dist.barrier()
            sync()
            tu = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(tu, op=dist.ReduceOp.MAX)
            same = bool(torch.equal(v8, to_u8(last))) if last.dtype == torch.float32 else None
            exchange_u8 = {"value": round(Ku * frames_per_step / float(tu.item()), 3), "unit": "frames/s", "steps": Ku,
                           "ms_per_step": round(float(tu.item()) / Ku * 1e3, 3), "exchanged": "(n,S,S,3) uint8 video",
                           "equals_u8_of_the_f32_video": same}
        finally:
            fmt["post"], ops.CONV_HOOK = post, prev_hook

    if rank == 0:
        conv_ms, conv_flops, n_launch, mean_launch_ms = timer.result()
        conv_exec_flops, conv_by_kind = timer.exec_flops, timer.by_kind()
        frames = K * frames_per_step
        prec_tag = {"winograd": "", "fp32": " [every layer on the direct kernel]", "bf16": " [bf16 MFMA conv tiles]", "split": " [bf16x6 exact-split products]"}[args.precision]
        line = {
            "metric": ("synthesized frames/sec at 512x512 (run_imitator)" if S == 512 else f"synthesized frames/sec at {S}x{S}") + prec_tag,
            "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "strong" if clip else "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 MFMA operands + bf16 activation storage, f32 accumulation",
                      "split": "f32 in/out/accumulate, products as 6 bf16 MFMAs over an exact 3-way split",
                      "winograd": "f32"}[args.precision],
            "conv_engine": {"winograd": "fp32 MFMA: 3x3 / stride 1 layers as fused F(4x4,3x3) Winograd convolutions (lwg_conv_winograd4_kernel; Cin < 64: F(2x2,3x3), lwg_conv_winograd_kernel), the transposed "
                                        "convolutions as fused F(2x2,2x2) Winograd convolutions (lwg_convt_winograd_kernel), the strided / first layers as direct "
                                        "implicit GEMMs (lwg_conv_igemm_kernel)",
                            "fp32": "fp32 MFMA: every layer as a direct implicit GEMM (lwg_conv_igemm_kernel)",
                            "bf16": "bf16 MFMA implicit GEMMs", "split": "bf16x6 implicit GEMMs"}[args.precision],
            "data": "synthetic" + (" (tiny architecture, CPU plumbing run: NOT a measurement)" if (args.tiny_arch or not on_gpu) else ""),
            "result_tensor": ("(n,3,S,S) f32 video" if (world == 1 or args.gather_dtype == "f32") else "(n,S,S,3) uint8 video") +
                             (f", all-gathered as {args.gather_dtype}" if world > 1 else ""),
            "config": {"workload": (f"run_imitator {S}x{S} single src/ref pair, {n_clip}-frame reference clip frame-sharded over {world} GPU(s), "
                                    "AttLWB-SPADE generator fp32 (BASELINE configs[1] at N = 1, configs[2] at N = 8)" if clip else
                                    f"run_imitator {S}x{S} single src/ref pair, one {FB}-frame batch per GPU per step (weak scaling)")
                       if is_f32 else
                       f"per-frame path {S}x{S}, AttLWB-SPADE generator, precision mode {args.precision}",
                       "poses": args.workload, "image_size": S, "num_source": 2, "frame_batch": (min(FB, -(-n_clip // world)) if clip else FB), "frame_batch_requested": FB_requested,
                       "frames_per_step": frames_per_step,
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "parallelism": f"frame-shard x{world}" + (f" + RCCL all-gather of the output video ({args.gather_dtype}), " +
                                                                  ("chunked behind the frame loop" if args.overlap else "one collective")
                                                                  if world > 1 else ""),
                       "batches_in_flight": args.streams,
                       "weights": "random-init (seeded) of the real architecture, 36,276,992 params"},
        }
        if per_rank is not None:
            line["config"]["per_rank"] = per_rank
            # self-diagnosing top level for the first 8-GPU run (no hardware curve was measured by the builder): where a step's time went on the
            # slowest rank, what every rank rendered, what RCCL saw, which chunk schedule ran and what the model behind that choice predicted
            line["t_compute_ms"] = max(r["compute_ms_per_step"] for r in per_rank)
            line["t_exposed_gather_ms"] = max(r["exposed_gather_ms_per_step"] for r in per_rank)
            line["frames_per_rank"] = [r["frames"] for r in per_rank]
            line["rccl_world_size"] = dist.get_world_size()
            line["backend"] = dist.get_backend()
            line["chunk_plan"] = per_rank[0].get("chunk_lengths")
            line["chunk_plan_model"] = plan_box["model"]
            line["bytes_received_per_rank_per_step"] = per_rank[0].get("bytes_received_per_step")
        if exchange_u8 is not None:
            line["exchange_u8"] = exchange_u8
        line["self_check"] = self_check["result"] if self_check else None
        if self_check:
            line["self_check_detail"] = self_check
        if n_launch:
            traffic, traffic_src, traffic_by_kernel, tj = None, None, None, {}
            tname = {"winograd": "pmc_traffic.json", "fp32": "pmc_traffic_direct.json", "bf16": "pmc_traffic_bf16.json"}.get(args.precision)
            tpath = os.path.join(ROOT, "profiles", tname) if tname else None       # written by tools/pmc_round.sh (separate --pmc passes)
            if tpath and S == (512 if is_f32 else 1024) and args.streams == 1 and os.path.exists(tpath):
                with open(tpath) as fp:
                    tj = json.load(fp)
                cfg = tj.get("bench_config") or {}
                if cfg.get("frame_batch") == FB and cfg.get("image_size") == S and cfg.get("workload") == args.workload and \
                        cfg.get("precision", "fp32") == args.precision:   # same launches
                    traffic, traffic_src = tj.get("traffic_bytes_per_launch"), f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
                    # like for like (round 6): the counters per KERNEL FAMILY against that family's own algorithmic bytes per kernel launch
                    fam = {"lwg_conv_winograd_kernel": "winograd", "lwg_conv_winograd4_kernel": "winograd4", "lwg_convt_winograd_kernel": "winograd_up4",
                           "lwg_conv_igemm_kernel": "direct"}
                    traffic_by_kernel = {}
                    for kname, rec in (tj.get("by_kernel") or {}).items():
                        bk = conv_by_kind.get(fam.get(kname, kname)) or {}
                        tb, ab = rec.get("traffic_bytes_per_launch"), bk.get("algorithmic_bytes_per_launch")
                        traffic_by_kernel[kname] = {"traffic": tb, "algorithmic_bytes_per_launch": ab, "launches_pmc_pass": rec.get("launches_fetch_pass"),
                                                    "launches_per_step": (bk.get("kernel_launches") or 0) // max(K, 1),
                                                    "traffic_over_algorithmic": round(tb / ab, 3) if tb and ab else None}
            # matrix-pipe flops EXECUTED per second over all conv launches: the direct kernels execute their algorithmic flops, the F(2x2,3x3)
            # Winograd kernel 4/9 of them (16 products per 2 x 2 outputs instead of 36) - a roofline fraction is executed work over the pipe's peak
            achieved = (conv_exec_flops if args.precision == "winograd" else conv_flops) / (conv_ms * 1e-3) / 1e12
            peak = PEAK_FP32_MFMA_TFLOPS if args.precision in ("fp32", "winograd") else PEAK_BF16_MFMA_TFLOPS
            if args.precision == "split":
                achieved *= 6.0          # executed bf16 MFMA flops: six partial products per algorithmic fp32 product
            line["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                                "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                                "traffic_source": traffic_src,
                                "algorithmic_bytes_per_launch": round(timer.bytes / n_launch, 1),
                                "traffic_by_kernel": traffic_by_kernel,
                                # like for like for the DOMINANT kernel (the one `traffic` is the counter figure of): its own algorithmic bytes per launch
                                "traffic_kernel": tj.get("kernel") if traffic is not None else None,
                                "traffic_over_algorithmic": ((traffic_by_kernel or {}).get(tj.get("kernel")) or {}).get("traffic_over_algorithmic")
                                if traffic is not None else None,
                                "kernel": {"fp32": "lwg_conv_igemm_kernel (fp32 MFMA implicit GEMM)",
                                           "bf16": "lwg_conv_igemm_bf16_kernel (bf16 MFMA implicit GEMM, bf16 activations) + fp32-input first layers",
                                           "split": "lwg_conv_igemm_split_kernel (bf16x6: achieved = 6 x algorithmic flops, the bf16 "
                                                    "MFMA work actually executed) + fp32 first layers",
                                           "winograd": "lwg_conv_winograd4_kernel (F(4x4,3x3) on the fp32 MFMA pipe: achieved counts the EXECUTED flops, "
                                                       "2 M 2.25 Cin N on those launches; F(2x2,3x3) launches: 2 M 4 Cin N) + lwg_convt_winograd_kernel (F(2x2,2x2) form of the transposed convolutions: "
                                                       "2 M 9 Cin N per input pixel instead of 16) + lwg_conv_igemm_kernel for the strided / first layers"}[args.precision],
                                "launches": n_launch, "avg_launch_us": round(mean_launch_ms * 1e3, 2), "streams": args.streams,
                                "algorithmic_gflop_per_frame": round(conv_flops * world / frames / 1e9, 2),
                                "share_of_step_time": round(conv_ms * 1e-3 / dt, 4)}
            if args.precision == "winograd":
                wk = conv_by_kind.get("winograd") or {}
                line["roofline"]["algorithmic_equivalent_tflops"] = round(conv_flops / (conv_ms * 1e-3) / 1e12, 2)    # what a direct convolution would have to sustain
                line["roofline"]["executed_gflop_per_frame"] = round(conv_exec_flops * world / frames / 1e9, 2)
                line["roofline"]["by_kernel"] = conv_by_kind
                if wk.get("executed_tflops"):
                    line["roofline"]["winograd_kernel_frac"] = round(wk["executed_tflops"] / peak, 4)
                if (conv_by_kind.get("winograd4") or {}).get("executed_tflops"):
                    line["roofline"]["winograd4_kernel_frac"] = round(conv_by_kind["winograd4"]["executed_tflops"] / peak, 4)
                if (conv_by_kind.get("winograd_up4") or {}).get("executed_tflops"):
                    line["roofline"]["winograd_up4_kernel_frac"] = round(conv_by_kind["winograd_up4"]["executed_tflops"] / peak, 4)
            if args.precision == "bf16":
                line["roofline"]["hbm_time_at_peak_us_per_launch"] = round(timer.bytes / n_launch / (PEAK_HBM_TBS * 1e12) * 1e6, 2)
                gov, hbm_share = timer.governing(PEAK_BF16_MFMA_TFLOPS)
                line["roofline"]["frac_of_governing_roof"] = round(gov, 4)
                line["roofline"]["share_of_conv_time_in_hbm_governed_launches"] = round(hbm_share, 4)
        if args.conv_breakdown and n_launch:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "conv_breakdown.json"), "w") as fp:
                json.dump(timer.breakdown(), fp, indent=1)
        timer.reset()
        if args.extras and world == 1 and args.streams == 1 and clip:
            Ke = max(2, K // 5)
            only = set(filter(None, args.only_extras.split(",")))

            def want(name):
                return not only or name in only

            def render():
                return im.synthesize(tgt, "smooth")
            if args.pipelined_streams > 1 and want("pipelined"):
                line["pipelined"] = _extra(pipelined, im, render, n_clip, 1, Ke, args.pipelined_streams, tgt if args.self_check else None)
            if args.split_extra and is_f32 and want("split_products"):
                line["split_products"] = _extra(split_products, im, render, n_clip, 1, Ke, last)
                if headline:
                    line["direct_products"] = _extra(direct_products, im, render, n_clip, 1, Ke, last)
                else:
                    line["winograd_products"] = _extra(winograd_products, im, render, n_clip, 1, Ke, last)
            if args.output_frames > 0 and want("with_output"):
                # finer batches for the output pipeline: D2H / PNG encoding of batch t overlaps the synthesis of batch t+1, and a 160-frame
                # measurement at 32 frames per batch is mostly pipeline fill and drain (408 vs 430 frames/s at 16)
                line["with_output"] = _extra(with_output, im, tgt, min(FB, 16), args.output_frames, 0)
            if headline and S == 512 and args.sizes_extra and want("sizes"):
                line["sizes"] = {str(S2): size_extra(dev, timer, S2) for S2 in (256, 1024)}
            if headline and S == 512 and clip and want("shard_of_8"):
                line["shard_of_8"] = _extra(shard_of_8, im, tgt, n_clip)
            if headline and S == 512:
                if want("b1_latency"):
                    line["b1_latency"] = _extra(b1_latency, im, tgt, timer)
                ops.CONV_HOOK = hook
                if want("novel_view_1024_bf16"):
                    try:
                        line["novel_view_1024_bf16"] = novel_view_1024_bf16(dev, timer, 1, 3)
                    except Exception as e:
                        line["novel_view_1024_bf16"] = {"error": f"{type(e).__name__}: {e}"}
                if want("personalize_step"):
                    line["personalize_step"] = personalize_step_extra()
                # the reference's DEFAULT loss set (VGG19 perceptual + SphereFace on seeded weights: the licensed checkpoints are not
                # available offline; same cost per step).  The face crop reads its box on the host, so this step runs eager launches.
                if want("personalize_step_vgg_face"):
                    line["personalize_step_vgg_face"] = personalize_step_extra(steps=6, warmup=3, extra_args=("--use-vgg", "--use-face"))
        if args.cpu_frames > 0 and world == 1:          # the CPU baseline is an N = 1 measurement (rank 0 only)
            small = pu.build_case(image_size=S, n_frames=args.cpu_frames, ns=2)
            line["cpu_baseline"] = _extra(cpu_baseline, small, args.cpu_frames)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
