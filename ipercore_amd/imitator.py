"""The Imitator runner (reference iPERCore/models/imitator.py:130-401) on the MI355X.

Same public surface as the reference for the per-frame path: ``Imitator(opt, device)``, ``swap_params``
(:248-256), ``make_inputs_for_tsf`` (:258-325), ``inference(tgt_smpls, cam_strategy, output_dir, prefix, ...)``
(:327-382), ``forward`` (:384-395).  ``inference`` returns what the reference returns: a list of file paths
(``"{prefix}{t:0>8}.png"``) when ``output_dir`` is given, else a list of ``(3, S, S)`` float arrays in [-1, 1].

MI355X-first structure: with ``temporal=False`` (deploy.toml:40) a frame depends only on the cached source
state, its own SMPL parameters and ``first_cam`` (:298-299), so ``inference`` walks the clip in batches of
``frame_batch`` frames; every batch is: one batched skinning call, one projection + rasterization, one fused
flow pass, the generator (MFMA convs at batch B), one head + compositing kernel.  No host sync inside a batch.
Frames of one clip shard over ranks with ``ipercore_amd.sharding`` (one all-gather of the output tensor).

``source_setup`` (:177-246) runs the once-per-source stage on the device (morphology, Canny boundary fill, UV merge,
background network, SIDNet + K/V hoist); ``set_source`` enters with an already prepared UV image / background.
"""
import os

import numpy as np
import torch

from . import ops
from .bodynets import SMPLH
from .flowcomposition import FlowComposition, _force
from .geometry import cam_pose_utils
from .networks import NetworksFactory


def _opt_get(opt, key, default=None):
    if isinstance(opt, dict):
        return opt.get(key, default)
    return getattr(opt, key, default)


class TemporalFIFO(object):
    """Ring of the last ``time_step`` synthesized frames (reference models/imitator.py:18-127): their projected faces
    (for the Ttt flows, flowcomposition.py:569-579) and, per AttLWB site, the hoisted K / V projections of their SIDNet
    features.  The ring lives in ONE preallocated K and V tensor per site laid out [ns sources | time_step slots], so the
    attention kernel reads sources + temporal frames with no per-frame concatenation."""

    def __init__(self, time_step, src_feats, src_f2pts):
        self.time_step, self.index = time_step, 0
        self.ns = src_feats.ns
        self.f2pts = torch.cat([src_f2pts, src_f2pts.new_zeros((time_step,) + tuple(src_f2pts.shape[1:]))], dim=0)
        self.kv = []
        for site in src_feats.kv:                   # (Kq, V, kappa) per attention site: one ring tensor each
            self.kv.append(tuple(torch.cat([t, t.new_zeros((time_step,) + tuple(t.shape[1:]))], dim=0) for t in site))
        self.src_feats = src_feats

    @property
    def nt(self):
        return min(self.index, self.time_step)

    def append(self, f2pts, feats):
        """f2pts (1,nf,3,2) of the frame just synthesized; feats: SourceFeatures of forward_src([pred, cond])."""
        i = self.ns + self.index % self.time_step
        self.f2pts[i:i + 1].copy_(f2pts)
        for ring, new in zip(self.kv, feats.kv):
            for R, t in zip(ring, new):
                R[i:i + 1].copy_(t)
        self.index += 1

    def view(self):
        """(f2pts (ns+nt,nf,3,2), SourceFeatures over ns+nt entries) - slots in ring order, as the reference concatenates."""
        n = self.ns + self.nt
        from .networks.generator import SourceFeatures
        return self.f2pts[:n], SourceFeatures(self.src_feats.enc, self.src_feats.res, [tuple(R[:n] for R in ring) for ring in self.kv], n, False)


class Imitator(object):
    def __init__(self, opt, device=torch.device("cuda:0"), frame_batch=8, streams=1):
        self._opt = opt
        # frame_batch == 1 is the reference's calling convention (one frame per Imitator.forward, imitator.py:341): ~85 eager launches per
        # frame.  (Rounds 2-3 replayed them as one hipGraph: measured equal to eager launches - 2.99 vs 2.97 ms per frame, the GPU is the
        # bound at B = 1 - so that path and its capture / fallback logic were removed in round 4.)
        self._name = "Imitator"
        self.device = torch.device(device)
        self._frame_batch_req = int(frame_batch)   # what the caller asked for; ``frame_batch`` (property) is what runs
        self.streams = max(1, int(streams))       # independent frame batches in flight on separate HIP streams
        self._side_streams = None
        self.src_info = None
        self.first_cam = None
        self.image_size = int(_opt_get(opt, "image_size", 512))
        self.generator = None
        self.temporal = bool(_opt_get(opt, "temporal", False))
        self.time_step = int(_opt_get(opt, "time_step", 1))
        self.temporal_fifo = None
        self.primary_ids = 0                      # which source's camera / shape drives the target (Swapper sets it)
        self._create_networks()

    # ------------------------------------------------------------------ frame batch actually launched
    def max_frame_batch(self):
        """No per-tensor cap reaches the caller any more: a launch whose gathered input would exceed the conv kernels' 32-bit buffer
        offsets (3 GiB) is cut into batch slices INSIDE the C entry points (csrc/lwg_conv_slices.h; a frame is independent of its
        batch, so the values do not change).  What bounds a frame batch now is device memory (~0.2 GB of fp32 activations per
        512 x 512 frame): returns None."""
        return None

    @property
    def frame_batch(self):
        """Frames per launch batch: what the caller asked for."""
        return self._frame_batch_req

    @frame_batch.setter
    def frame_batch(self, value):
        self._frame_batch_req = max(1, int(value))

    def _create_networks(self):
        self.body_rec = SMPLH(model_path=_opt_get(self._opt, "smpl_model_hand")).to(self.device)
        self.weak_cam_swapper = cam_pose_utils.WeakPerspectiveCamera(self.body_rec)
        self.flow_comp = FlowComposition(opt=self._opt).to(self.device)
        self.generator = self._create_generator(_opt_get(_opt_get(self._opt, "neural_render_cfg"), "Generator"))
        self.generator = self.generator.to(self.device)

    def _create_generator(self, cfg):
        net = NetworksFactory.get_by_name(_opt_get(self._opt, "gen_name", "AttLWB-SPADE"), cfg=cfg, temporal=self.temporal)
        meta = _opt_get(self._opt, "meta_data")
        load_path = _opt_get(meta, "personalized_ckpt_path", "") if meta is not None else ""
        if not (load_path and os.path.exists(load_path)):
            load_path = _opt_get(self._opt, "load_path_G", "")
        if load_path and os.path.exists(load_path):
            ckpt = torch.load(load_path, map_location="cpu")
            ckpt = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ckpt.items()}
            missing = net.load_state_dict(ckpt, strict=False)
            if missing.missing_keys or missing.unexpected_keys:
                # the reference loads with strict=False silently (imitator.py:167-168); we say what did not match
                print(f"[ipercore_amd] checkpoint {load_path}: missing {len(missing.missing_keys)} keys, "
                      f"unexpected {len(missing.unexpected_keys)} keys")
        net.eval()
        return net

    # ------------------------------------------------------------------ source state
    @torch.no_grad()
    def source_setup(self, src_path, src_smpl, masks=None, bg_img=None, offsets=0, links_ids=None, visualizer=None):
        """imitator.py:177-246.  ``src_path``: list of image paths (loaded like cv_utils.load_images, :190) or an
        array / tensor (ns,3,S,S) in [-1,1]; ``masks`` (ns,1,S,S) foreground masks or None; ``bg_img`` (3,S,S) or None."""
        dev, S = self.device, self.image_size
        if isinstance(src_path, (list, tuple, str)):
            src_np = load_images(src_path, S)
        else:
            src_np = np.asarray(src_path.detach().cpu() if torch.is_tensor(src_path) else src_path, dtype=np.float32)
        src_img = torch.tensor(src_np[None], dtype=torch.float32, device=dev)            # (1, ns, 3, S, S)
        src_smpl = torch.as_tensor(src_smpl, dtype=torch.float32, device=dev)
        off = torch.as_tensor(np.asarray(offsets), dtype=torch.float32, device=dev) if not torch.is_tensor(offsets) else offsets.to(dev)
        src_info = self.body_rec.get_details(src_smpl, off, links_ids=links_ids)
        ns = src_smpl.shape[0]
        src_info["num_source"] = ns
        if masks is not None:
            src_info["masks"] = 1.0 - torch.as_tensor(np.asarray(masks), dtype=torch.float32, device=dev)
        self.flow_comp.add_rendered_f2verts_fim_wim(src_info, use_morph=True, get_uv_info=True)
        src_info["offsets"], src_info["links_ids"] = off, links_ids
        uv_img, input_G_bg, input_G_src = self.flow_comp.process_source(src_img, src_info, primary_ids=[0])
        counts = src_info.pop("_edge_counts").cpu()                # the one host sync of source_setup
        if int(counts.min()) < 3:
            raise RuntimeError("source silhouette has fewer than 3 boundary pixels (the reference's topk(k=3) raises too)")
        src_info["uv_img"] = uv_img
        src_info["uv_img4"] = ops.nchw_to_nhwc(uv_img.contiguous(), c_pad=4)[0].contiguous()
        if bool(_opt_get(self._opt, "use_inpaintor", False)) or bg_img is not None:
            bg = torch.as_tensor(np.asarray(bg_img), dtype=torch.float32, device=dev)[None, None]
        else:
            bg = self.generator.forward_bg(input_G_bg.contiguous())
        enc, res = self.generator.forward_src(input_G_src.contiguous(), only_enc=True)
        src_info["img"] = src_img
        src_info["bg"] = bg[:, 0].contiguous()
        src_info["feats"] = (enc, res)
        src_info["feats_nhwc"] = enc.lwg_cache
        self.src_info = src_info
        return src_info

    @torch.no_grad()
    def set_source(self, src_smpl, uv_img, bg_img, src_img=None, offsets=0, links_ids=None):
        """Build ``src_info`` from prepared tensors: what source_setup (:177-246) caches after process_source.

        src_smpl (ns,85); uv_img (1,3,S,S); bg_img (1,3,S,S); src_img (1,ns,3,S,S) source images (already morphed).
        Runs the source renders (f2pts, cond), SIDNet (forward_src) and the K/V projections on the device.
        """
        dev = self.device
        src_smpl = torch.as_tensor(src_smpl, dtype=torch.float32, device=dev)
        off = offsets if not isinstance(offsets, np.ndarray) else torch.tensor(offsets, dtype=torch.float32, device=dev)
        src_info = self.body_rec.get_details(src_smpl, off, links_ids=links_ids)
        ns = src_smpl.shape[0]
        src_info["num_source"] = ns
        self.flow_comp.add_rendered_f2verts_fim_wim(src_info, use_morph=False, get_uv_info=False)
        src_info["offsets"], src_info["links_ids"] = off, links_ids
        src_info["uv_img"] = torch.as_tensor(uv_img, dtype=torch.float32, device=dev).contiguous()
        src_info["uv_img4"] = ops.nchw_to_nhwc(src_info["uv_img"], c_pad=4)[0].contiguous()
        src_info["bg"] = torch.as_tensor(bg_img, dtype=torch.float32, device=dev).contiguous()
        if src_img is None:
            raise ValueError("src_img (1, ns, 3, S, S) is required")
        src_img = torch.as_tensor(src_img, dtype=torch.float32, device=dev)
        src_info["img"] = src_img
        S = self.image_size
        input_G_src = self.flow_comp.make_src_inputs(src_img.view(ns, 3, S, S), src_info).view(1, ns, 6, S, S)
        enc, res = self.generator.forward_src(input_G_src, only_enc=True)
        src_info["feats"] = (enc, res)
        src_info["feats_nhwc"] = enc.lwg_cache
        self.src_info = src_info
        return src_info

    # ------------------------------------------------------------------ per-frame path
    def swap_params(self, src_cam, src_shape, tgt_smpl, cam_strategy="smooth"):
        """imitator.py:248-256, for a batch of target frames."""
        cam = self.weak_cam_swapper.cam_swap(src_cam, tgt_smpl[:, 0:3], self.first_cam, cam_strategy)
        return torch.cat([cam, tgt_smpl[:, 3:-10], src_shape.expand(tgt_smpl.shape[0], -1)], dim=1)

    @torch.no_grad()
    def make_inputs_for_tsf(self, src_info, tgt_smpl, cam_strategy="smooth", t=0, primary_ids=0, use_selected_f2pts=False,
                            want_aux=False):
        """imitator.py:258-325 for B frames at once -> (tsf8 NHWC, Tst, ref_info)."""
        if t == 0 and cam_strategy == "smooth" and self.first_cam is None:
            self.first_cam = tgt_smpl[0:1, 0:3].clone()
        ref_smpl = self.swap_params(src_info["cam"][primary_ids:primary_ids + 1],
                                    src_info["shape"][primary_ids:primary_ids + 1], tgt_smpl, cam_strategy)
        ref_info = self.body_rec.get_details(ref_smpl.contiguous(), src_info["offsets"], links_ids=src_info["links_ids"])
        # flowcomposition.py:556-562: selected faces (Swapper) > visible faces only (opt.only_vis) > all projected source faces
        key = "selected_f2pts" if use_selected_f2pts else ("only_vis_f2pts" if self.flow_comp.only_vis else "f2pts")
        tsf8, Tst, aux = self.flow_comp.frame_inputs(ref_info["cam"].contiguous(), ref_info["verts"], src_info["uv_img4"],
                                                     _force(src_info[key]).contiguous(), want_aux=want_aux)
        if aux is not None:
            ref_info.update(aux)
        return tsf8, Tst, ref_info

    @torch.no_grad()
    def forward(self, tsf8, Tst):
        """imitator.py:384-395 on the engine's NHWC input -> (pred (B,3,S,S), mask (B,1,S,S))."""
        pred, mask, _ = self.generator.run_tsf(tsf8, self.src_info["feats_nhwc"], Tst, bg=self.src_info["bg"],
                                               want_pred=True, want_mask=True)
        return pred, mask

    @torch.no_grad()
    def synthesize(self, tgt_smpls, cam_strategy="smooth", t0=0, use_selected_f2pts=False):
        """Frames [t0, t0+n) of an (already stabilised) device tensor (n,85) -> pred (n,3,S,S) on the device."""
        outs = []
        sel = dict(primary_ids=self.primary_ids, use_selected_f2pts=use_selected_f2pts)
        if tgt_smpls.shape[0] == 0:               # an empty shard (clip shorter than the number of ranks): an empty video block
            return torch.empty((0, 3, self.image_size, self.image_size), device=tgt_smpls.device, dtype=torch.float32)
        if self.streams > 1 and tgt_smpls.is_cuda:
            # frames are independent: batches alternate over HIP streams so one batch's kernel tails, launch gaps and
            # HBM-bound kernels overlap another batch's MFMA work (+6 % frames/s at 3 streams on MI355X)
            if self._side_streams is None:
                self._side_streams = [torch.cuda.Stream(device=self.device) for _ in range(self.streams)]
            cur = torch.cuda.current_stream(self.device)
            for st in self._side_streams:
                st.wait_stream(cur)
            for k, s in enumerate(range(0, tgt_smpls.shape[0], self.frame_batch)):
                with torch.cuda.stream(self._side_streams[k % self.streams]):
                    chunk = tgt_smpls[s:s + self.frame_batch]
                    tsf8, Tst, _ = self.make_inputs_for_tsf(self.src_info, chunk, cam_strategy, t=t0 + s, **sel)
                    outs.append(self.forward(tsf8, Tst)[0])
            for st in self._side_streams:
                cur.wait_stream(st)
            for o in outs:
                o.record_stream(cur)
            return torch.cat(outs, dim=0)
        for s in range(0, tgt_smpls.shape[0], self.frame_batch):
            chunk = tgt_smpls[s:s + self.frame_batch]
            tsf8, Tst, _ = self.make_inputs_for_tsf(self.src_info, chunk, cam_strategy, t=t0 + s, **sel)
            outs.append(self.forward(tsf8, Tst)[0])
        return torch.cat(outs, dim=0)

    @torch.no_grad()
    def synthesize_temporal(self, tgt_smpls, cam_strategy="smooth", use_selected_f2pts=False):
        """temporal=True (imitator.py:341-366): a recurrence - frame t attends to the sources and to the last ``time_step``
        synthesized frames, so frames are produced one at a time (this mode does not shard over frames: replicas only)."""
        src = self.src_info
        key = "selected_f2pts" if use_selected_f2pts else ("only_vis_f2pts" if self.flow_comp.only_vis else "f2pts")
        self.temporal_fifo = fifo = TemporalFIFO(self.time_step, src["feats_nhwc"], _force(src[key]).contiguous())
        outs = []
        for t in range(tgt_smpls.shape[0]):
            if t == 0 and cam_strategy == "smooth" and self.first_cam is None:
                self.first_cam = tgt_smpls[0:1, 0:3].clone()
            p = self.primary_ids
            ref_smpl = self.swap_params(src["cam"][p:p + 1], src["shape"][p:p + 1], tgt_smpls[t:t + 1], cam_strategy)
            ref_info = self.body_rec.get_details(ref_smpl.contiguous(), src["offsets"], links_ids=src["links_ids"])
            f2pts_all, feats = fifo.view()
            tsf8, T, aux = self.flow_comp.frame_inputs(ref_info["cam"].contiguous(), ref_info["verts"], src["uv_img4"],
                                                       f2pts_all.contiguous(), want_aux=True)      # Tst | Ttt in one pass
            pred, _, _ = self.generator.run_tsf(tsf8, feats, T, bg=src["bg"], want_pred=True, want_mask=True)
            # post_update (:397-401): SIDNet features of [pred, cond] enter the ring
            cur8 = ops.pack_inputs(pred, aux["cond"], None, 8)
            fifo.append(aux["f2pts"], self.generator.encode_sources(cur8, batched=False, ns=1))
            outs.append(pred)
        return torch.cat(outs, dim=0)

    @torch.no_grad()
    def prepare_sequence(self, tgt_smpls, cam_strategy="smooth"):
        """The sequence-global pre-pass of inference (imitator.py:335-339): to device, stabilise, fix first_cam."""
        self.first_cam = None
        tgt = torch.as_tensor(np.asarray(tgt_smpls), dtype=torch.float32).to(self.device)
        if cam_strategy == "smooth":
            tgt = self.weak_cam_swapper.stabilize(tgt)
            self.first_cam = tgt[0:1, 0:3].clone()
        return tgt

    @torch.no_grad()
    def inference(self, tgt_smpls, cam_strategy="smooth", output_dir="", prefix="pred_", use_selected_f2pts=False,
                  visualizer=None, verbose=True):
        """imitator.py:327-382."""
        tgt = self.prepare_sequence(tgt_smpls, cam_strategy)
        if self.temporal:
            preds = self.synthesize_temporal(tgt, cam_strategy, use_selected_f2pts=use_selected_f2pts)
            if output_dir:
                from .output import FrameWriter
                writer = FrameWriter(output_dir, prefix=prefix)
                writer.submit(preds, 0)
                return writer.close()
            preds = preds.cpu().numpy()
            return [preds[i] for i in range(preds.shape[0])]
        if output_dir:
            # device-side uint8 conversion + pinned async D2H + threaded PNG encoding: the frame loop never waits for disk
            from .output import FrameWriter
            writer = FrameWriter(output_dir, prefix=prefix)
            for s in range(0, tgt.shape[0], self.frame_batch):
                writer.submit(self.synthesize(tgt[s:s + self.frame_batch], cam_strategy, t0=s, use_selected_f2pts=use_selected_f2pts), s)
            return writer.close()
        outputs = []
        for s in range(0, tgt.shape[0], self.frame_batch):
            preds = self.synthesize(tgt[s:s + self.frame_batch], cam_strategy, t0=s, use_selected_f2pts=use_selected_f2pts).cpu().numpy()
            outputs.extend(preds[i] for i in range(preds.shape[0]))
        return outputs


class Viewer(Imitator):
    """Novel-view runner (reference models/imitator.py:403-462): the same per-frame path; ``inference`` has no ``prefix``
    argument and writes ``pred_{t:0>8}.png``."""

    def __init__(self, opt, device=torch.device("cuda:0"), **kw):
        super().__init__(opt, device, **kw)
        self._name = "Viewer"

    @torch.no_grad()
    def inference(self, tgt_smpls, cam_strategy="smooth", output_dir="", visualizer=None, verbose=True):
        return super().inference(tgt_smpls, cam_strategy=cam_strategy, output_dir=output_dir, prefix="pred_",
                                 visualizer=visualizer, verbose=verbose)


class Swapper(Imitator):
    """Part-wise appearance swap between several people (reference models/imitator.py:468-622): every person contributes the
    faces of its ``swap_parts``; the sources of all people feed the attention of ONE target driven by the primary person's
    camera and shape.  Same per-frame kernels; ``inference(..., use_selected_f2pts=True)`` is how run_swapper drives it."""

    def __init__(self, opt, device=torch.device("cuda:0"), **kw):
        super().__init__(opt, device, **kw)
        self._name = "Swapper"

    def _create_networks(self):
        from .flowcomposition import FlowCompositionForSwapper
        self.body_rec = SMPLH(model_path=_opt_get(self._opt, "smpl_model_hand")).to(self.device)
        self.weak_cam_swapper = cam_pose_utils.WeakPerspectiveCamera(self.body_rec)
        self.flow_comp = FlowCompositionForSwapper(opt=self._opt).to(self.device)
        self.generator = self._create_generator(_opt_get(_opt_get(self._opt, "neural_render_cfg"), "Generator")).to(self.device)

    def get_selected_info_by_part_mask(self, swap_masks):
        raise NotImplementedError                          # as in the reference (:489-500)

    def get_selected_info_by_part_name(self, swap_parts, primary_ids=0):
        """:502-546 -> (part ids per person, face ids per person); faces nobody selected go to the primary person."""
        fc = self.flow_comp
        selected_part_ids, selected_face_ids, all_faces = [], [], set()
        for swap_part in swap_parts:
            part_ids, face_ids = set(), set()
            for sub_part in swap_part:
                ids = fc.PART_IDS[sub_part]
                part_ids |= set(ids)
                face_ids |= set(fc.get_selected_fids(ids))
            all_faces |= face_ids
            selected_part_ids.append(sorted(part_ids))
            selected_face_ids.append(sorted(face_ids))
        left = set(fc.all_faces_ids) - all_faces
        if left:
            selected_face_ids[primary_ids] = sorted(set(selected_face_ids[primary_ids]) | left)
        return selected_part_ids, selected_face_ids

    @torch.no_grad()
    def swap_source_setup(self, src_path_list, src_smpl_list, masks_list, bg_img_list=None, offsets_list=0, links_ids_list=None,
                          swap_parts=(["head"], ["body"]), swap_masks=None, primary_ids=0, visualizer=None):
        """:548-621 -> merged src_info (also kept as ``self.src_info``)."""
        assert not (swap_parts is None and swap_masks is None)
        if swap_parts is not None:
            _, selected_face_ids = self.get_selected_info_by_part_name(swap_parts, primary_ids)
        else:
            _, selected_face_ids = self.get_selected_info_by_part_mask(swap_masks)
        n = len(src_path_list)
        pick = lambda lst, i, d=None: d if lst is None or isinstance(lst, (int, float)) else lst[i]     # noqa: E731
        infos = []
        for i in range(n):
            info = self.source_setup(src_path_list[i], src_smpl_list[i], pick(masks_list, i), pick(bg_img_list, i),
                                     offsets=pick(offsets_list, i, 0), links_ids=pick(links_ids_list, i), visualizer=visualizer)
            self.flow_comp.add_rendered_selected_f2pts(info, [selected_face_ids[i]] * info["num_source"])
            infos.append(info)
        self.primary_ids = primary_ids
        self.src_info = self.flow_comp.merge_src_info(infos, primary_ids=primary_ids)
        return self.src_info


class ModelsFactory(object):
    """reference models/base_model.py:8-32."""

    @staticmethod
    def get_by_name(model_name, *args, **kwargs):
        if model_name == "imitator":
            return Imitator(*args, **kwargs)
        if model_name == "viewer":
            return Viewer(*args, **kwargs)
        if model_name == "swapper":
            return Swapper(*args, **kwargs)
        raise ValueError(f"Model {model_name} not recognized.")


def create_T_pose_novel_view_smpl(length=180):
    """services/base_runner.py:11-30: T-pose SMPL rows whose global rotation is R.from_euler("xyz", [180, y, 0]), y = 0..360."""
    from scipy.spatial.transform import Rotation as R
    smpls = np.zeros((length, 85), dtype=np.float32)
    delta = 360 / (length - 1) if length > 1 else 0
    for i in range(length):
        smpls[i, 3:6] = R.from_euler("xyz", [180, delta * i, 0], degrees=True).as_rotvec()
    return smpls


def add_hands_params_to_smpl(smpls, hands_param):
    """services/base_runner.py:33-55: (n,85) + hands (90,) or (n,90) -> (n,156+...)= [cam, pose[:66], hands, shape]."""
    hands_param = np.asarray(hands_param, dtype=np.float32)
    if hands_param.ndim == 1:
        hands_param = np.tile(hands_param, reps=(smpls.shape[0], 1))
    return np.concatenate([smpls[:, 0:3], smpls[:, 3:-10][:, 0:66], hands_param, smpls[:, -10:]], axis=1)


def load_images(paths, image_size):
    """cv_utils.load_images (cv_utils.py:45-66): (ns,3,S,S) RGB in [-1,1].  cv2 is not a dependency: PIL decodes and
    resizes (bilinear), so a resized image is close to, not bit-identical with, cv2.resize."""
    from PIL import Image
    if isinstance(paths, str):
        paths = [paths]
    out = []
    for p in paths:
        im = Image.open(p).convert("RGB")
        if im.size != (image_size, image_size):
            im = im.resize((image_size, image_size), Image.BILINEAR)
        a = np.asarray(im, dtype=np.float32) / 255.0
        out.append((a * 2 - 1).transpose(2, 0, 1))
    return np.stack(out, axis=0)


def to_uint8_hwc(pred_chw):
    """cv_utils.save_cv2_img(normalize=True) numerics (cv_utils.py:100-116): (x+1)/2*255, truncated to uint8."""
    img = np.transpose(pred_chw, (1, 2, 0))
    return ((img + 1) / 2.0 * 255).astype(np.uint8)


def save_image(pred_chw, path):
    """Write one frame as PNG.  The reference converts RGB->BGR and calls cv2.imwrite, which stores RGB on disk; cv2
    is not a dependency here, PIL writes the same RGB pixels."""
    from PIL import Image
    Image.fromarray(to_uint8_hwc(pred_chw), mode="RGB").save(path)
    return path
