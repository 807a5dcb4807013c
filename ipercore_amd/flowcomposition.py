"""Flow composition of the runner (reference iPERCore/models/flowcomposition.py:21-744), per-frame part.

Built in this round (the per-frame path of ``Imitator.inference``):
  ``add_rendered_f2verts_fim_wim`` (:139-204), ``make_tsf_inputs`` (:206-248), ``make_trans_flow`` (:514-582, temporal=False),
  ``make_batch_trans_flow`` (:584-662), ``make_uv_setup`` (:78-85),
  ``make_src_inputs`` (:262-265), and the fused ``frame_inputs`` that the MI355X runner actually calls: ONE
  pass over (fim, wim) producing cond, the UV flow + UV sample, the generator input and all source flows
  (``csrc/flow.hip``) instead of 2 + ns boolean-mask gathers with host syncs.
Once-per-source stage (``Imitator.source_setup``): ``add_rendered_f2verts_fim_wim(use_morph=True)`` (:139-204),
``make_morph_image`` (:335-386: silhouette morphology, Canny boundary, 3-nearest-boundary fill of the uncertain band -
one O(n1 n2) kernel with no (n1, n2) matrix and no nonzero() host sync), ``make_uv_img`` (:87-137), ``make_bg_inputs``
(:250-260), ``make_src_inputs`` (:262-265), ``process_source`` (:452-512) - kernels in ``csrc/source.hip``.
"""
import torch

from . import ops
from .morphology import CannyFilter, morph
from .renders import SMPLRenderer


class FlowComposition(torch.nn.Module):
    # flowcomposition.py:23-39: part name -> indices into the (sorted) body parts of smpl_part_info.json
    PART_IDS = {
        "head": [0], "torso": [1], "left_leg": [2], "right_leg": [3], "left_arm": [4], "right_arm": [5], "left_foot": [6],
        "right_foot": [7], "left_hand": [8], "right_hand": [9], "facial": [10],
        "upper": [1, 4, 5, 8, 9], "lower": [2, 3, 6, 7], "body": [1, 2, 3, 4, 5, 6, 7, 8, 9],
        "all": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9],
    }

    def __init__(self, opt):
        super().__init__()
        self._opt = opt
        self._name = "FlowComposition"
        g = lambda k, d=None: getattr(opt, k, d) if not isinstance(opt, dict) else opt.get(k, d)    # noqa: E731
        self.image_size = int(g("image_size", 512))
        self.only_vis = bool(g("only_vis", False))
        self.conf_erode_ks = int(g("conf_erode_ks", 3))        # deploy.toml:41
        self.out_dilate_ks = int(g("out_dilate_ks", 51))       # deploy.toml:42
        self.bg_ks = int(g("bg_ks", 11))                       # deploy.toml:11
        self.num_source = int(g("num_source", 2))
        self.time_step = int(g("time_step", 1))
        self.render = SMPLRenderer(
            face_path=g("face_path"), fim_enc_path=g("fim_enc_path"), uv_map_path=g("uv_map_path"),
            part_path=g("part_path"), front_path=g("front_path"), head_path=g("head_path"), facial_path=g("facial_path"),
            map_name=g("map_name", "uv_seg"), image_size=self.image_size, fill_back=False, anti_aliasing=True,
            background_color=(0, 0, 0), has_front=True, top_k=3)
        self.register_buffer("grid", self.render.create_meshgrid(image_size=self.image_size))
        self.f2uvs = None
        self.uv_fim = None
        self.uv_wim = None
        self.one_map = None

    # ------------------------------------------------------------------ reference-shaped methods
    def make_uv_setup(self, bs, ns, nt, device):
        if self.f2uvs is None:
            n = bs * max(ns, nt)
            self.uv_fim, self.uv_wim = self.render.render_uv_fim_wim(n)
            self.f2uvs = self.render.get_f_uvs2img(n)
            self.one_map = torch.ones(bs * ns, 1, self.image_size, self.image_size, dtype=torch.float32, device=device)

    @torch.no_grad()
    def add_rendered_f2verts_fim_wim(self, smpl_info, use_morph=False, get_uv_info=True):
        """flowcomposition.py:139-204."""
        f2pts, fim, wim = self.render.render_fim_wim(cam=smpl_info["cam"], vertices=smpl_info["verts"], smpl_faces=True)
        cond, _ = self.render.encode_fim(fim=fim, transpose=True)
        if use_morph:
            rendered_sil = 1 - cond[:, -1:]
            human_sil = 1 - smpl_info["masks"] if "masks" in smpl_info else rendered_sil
            smpl_info["confidant_sil"] = morph(human_sil, ks=self.conf_erode_ks, mode="erode")
            smpl_info["outpad_sil"] = morph(((human_sil + rendered_sil) > 0).float(), ks=self.out_dilate_ks, mode="dilate")
        smpl_info["f2pts"] = f2pts
        smpl_info["only_vis_f2pts"] = _Lazy(lambda: self.render.get_vis_f2pts(f2pts, fim))   # consumed only if only_vis
        smpl_info["cond"], smpl_info["fim"], smpl_info["wim"] = cond, fim, wim
        if get_uv_info:
            obj_f2pts, obj_fim, obj_wim = self.render.render_fim_wim(cam=smpl_info["cam"], vertices=smpl_info["verts"],
                                                                     smpl_faces=False)
            smpl_info["obj_f2pts"] = obj_f2pts
            smpl_info["only_vis_obj_f2pts"] = self.render.get_vis_f2pts(obj_f2pts, obj_fim)
            smpl_info["obj_fim"], smpl_info["obj_wim"] = obj_fim, obj_wim
        return smpl_info

    @torch.no_grad()
    def make_tsf_inputs(self, uv_img, ref_info):
        """flowcomposition.py:206-248 -> (bs, nt, 6, h, w) NCHW."""
        fim, wim = ref_info["fim"], ref_info["wim"]
        bs, _, h, w = uv_img.shape
        nt = fim.shape[0] // bs
        uv4 = ops.nchw_to_nhwc(uv_img.contiguous().float(), c_pad=4)
        outs = []
        for b in range(bs):
            sl = slice(b * nt, (b + 1) * nt)
            tsf8, _, _, _ = ops.flow_compose(fim[sl].contiguous(), wim[sl].contiguous(), self.render.map_fn,
                                             self.render.f_uvs2img, uv4[b],
                                             torch.empty(0, self.render.nf, 3, 2, device=fim.device))
            outs.append(ops.nhwc_to_nchw(tsf8, channels=6))
        return torch.cat(outs, dim=0).view(bs, nt, 6, h, w)

    @torch.no_grad()
    def make_trans_flow(self, bs, ns, nt, src_info, temp_info, ref_info, temporal=True, use_selected_f2pts=False):
        """flowcomposition.py:514-582 -> (Tst (bs,ns,h,w,2), Ttt)."""
        if temporal:
            raise NotImplementedError("Ttt comes out of the runner's fused pass (Imitator.synthesize_temporal -> frame_inputs with the "
                                      "ring's f2pts appended); this reference-shaped helper only builds Tst")
        h = w = self.image_size
        key = "selected_f2pts" if use_selected_f2pts else ("only_vis_f2pts" if self.only_vis else "f2pts")
        src_f2pts = _force(src_info[key])
        n = ns * bs
        Tst = self.render.cal_bc_transform(src_f2pts, ref_info["fim"].repeat(max(ns, nt), 1, 1)[0:n],
                                           ref_info["wim"].repeat(max(ns, nt), 1, 1, 1)[0:n])
        return Tst.view(bs, ns, h, w, 2), None

    @torch.no_grad()
    def make_batch_trans_flow(self, bs, ns, nt, src_info, ref_info, temporal=False, use_selected_f2pts=False):
        """flowcomposition.py:584-662 (training form): Tst (bs, nt, ns, h, w, 2): every target frame of a sample against every
        source of the same sample.  temporal flows (Ttt) are produced by the runner's recurrent path only."""
        if temporal:
            raise NotImplementedError("Ttt for multi-step temporal training is not built (temporal inference is: Imitator(temporal=True))")
        h = w = self.image_size
        key = "selected_f2pts" if use_selected_f2pts else ("only_vis_f2pts" if self.only_vis else "f2pts")
        f2 = _force(src_info[key]).view(bs, 1, ns, -1, 3, 2).expand(bs, nt, ns, -1, 3, 2).reshape(bs * nt * ns, -1, 3, 2).contiguous()
        fim = ref_info["fim"].view(bs, nt, 1, h, w).expand(bs, nt, ns, h, w).reshape(bs * nt * ns, h, w).contiguous()
        wim = ref_info["wim"].view(bs, nt, 1, h, w, 3).expand(bs, nt, ns, h, w, 3).reshape(bs * nt * ns, h, w, 3).contiguous()
        return self.render.cal_bc_transform(f2, fim, wim).view(bs, nt, ns, h, w, 2), None

    def make_src_inputs(self, src_img, src_info):
        """flowcomposition.py:262-265."""
        return torch.cat([src_img, src_info["cond"]], dim=1)

    @torch.no_grad()
    def make_bg_inputs(self, src_img, src_info):
        """flowcomposition.py:250-260 -> (bs*ns, 4, h, w)."""
        bg_mask = src_info["masks"] if "masks" in src_info else src_info["cond"][:, -1:, :, :]
        src_bg_mask = morph(bg_mask.contiguous(), ks=self.bg_ks, mode="erode")
        return torch.cat([src_img * src_bg_mask, src_bg_mask], dim=1)

    @torch.no_grad()
    def make_morph_image(self, src_img, src_info, erode_ks=3, dilate_ks=11, want_debug=False):
        """flowcomposition.py:335-386 -> (bs*ns, 3, h, w).  The reference calls it with erode_ks = dilate_ks = 0."""
        confidant_sil = morph(src_info["confidant_sil"], ks=erode_ks, mode="erode") if erode_ks > 0 else src_info["confidant_sil"]
        outpad_sil = morph(src_info["outpad_sil"], ks=dilate_ks, mode="dilate") if dilate_ks > 0 else src_info["outpad_sil"]
        thin_edges = CannyFilter()(confidant_sil, 0.1, 0.9, True)
        out, counts, top3 = ops.boundary_fill(src_img, confidant_sil, outpad_sil, thin_edges, want_top3=want_debug)
        src_info["_edge_counts"] = counts            # device tensor; source_setup checks it once (topk needs >= 3 edges)
        if want_debug:
            return out, thin_edges, top3
        return out

    @torch.no_grad()
    def make_uv_img(self, src_img, src_info):
        """flowcomposition.py:87-137: (bs, ns, 3, h, w) -> merged UV image (bs, 3, h, w)."""
        bs, ns, _, h, w = src_img.shape
        n = bs * ns
        uv_fim = self.uv_fim[0:1].expand(n, -1, -1).contiguous()
        uv_wim = self.uv_wim[0:1].expand(n, -1, -1, -1).contiguous()
        one_map = torch.ones(n, 1, h, w, dtype=torch.float32, device=src_img.device)
        only_vis_Ts2uv = self.render.cal_bc_transform(src_info["only_vis_obj_f2pts"], uv_fim, uv_wim)
        Ts2uv = self.render.cal_bc_transform(src_info["obj_f2pts"], uv_fim, uv_wim)
        src_warp = ops.grid_sample(src_img.reshape(n, 3, h, w), Ts2uv)
        vis_warp = ops.grid_sample(one_map, only_vis_Ts2uv)
        vis_warp = morph(vis_warp, ks=13, mode="dilate")
        outs = [ops.uv_merge(src_warp[b * ns:(b + 1) * ns], vis_warp[b * ns:(b + 1) * ns]) for b in range(bs)]
        return torch.stack(outs, dim=0)

    @torch.no_grad()
    def process_source(self, src_img, src_info, primary_ids=None):
        """flowcomposition.py:452-512 -> (uv_img (bs,3,h,w), input_G_bg (bs,len(primary),4,h,w), input_G_src (bs,ns,6,h,w))."""
        bs, ns, _, h, w = src_img.shape
        self.make_uv_setup(bs, self.num_source, self.time_step, src_img.device)
        flat = src_img.reshape(bs * ns, 3, h, w).contiguous()
        morph_src_img = self.make_morph_image(flat, src_info, erode_ks=0, dilate_ks=0)
        morph_uv_img = self.make_uv_img(morph_src_img.view(bs, ns, 3, h, w), src_info)
        input_G_src = self.make_src_inputs(morph_src_img, src_info).view(bs, ns, -1, h, w)
        input_G_bg = self.make_bg_inputs(flat, src_info).view(bs, ns, -1, h, w)
        if primary_ids is None:
            primary_ids = [0]        # the reference draws one at random here (np.random.choice); its only caller passes [0]
        return morph_uv_img, input_G_bg[:, primary_ids], input_G_src

    # ------------------------------------------------------------------ fused MI355X per-frame entry
    @torch.no_grad()
    def frame_inputs(self, cam, verts, uv_img4, src_f2pts, want_aux=False):
        """B target frames -> (tsf8 (B,S,S,8) NHWC, Tst (B,ns,S,S,2), aux dict).  Equivalent to
        add_rendered_f2verts_fim_wim(get_uv_info=False) + make_tsf_inputs + make_trans_flow (temporal=False)."""
        fv, f2pts = ops.project_faces(verts, cam, self.render.smpl_faces, want_f2pts=want_aux)
        fim, wim = ops.rasterize_fim_wim(fv, self.image_size)
        tsf8, Tst, cond, tuv = ops.flow_compose(fim, wim, self.render.map_fn, self.render.f_uvs2img, uv_img4, src_f2pts,
                                                want_cond=want_aux, want_tuv=want_aux)
        aux = {"fim": fim, "wim": wim, "f2pts": f2pts, "cond": cond, "Tuv2t": tuv} if want_aux else None
        return tsf8, Tst, aux


class FlowCompositionForSwapper(FlowComposition):
    """flowcomposition.py:747-959: part-wise selection of source faces and the merge of several people's source state."""

    def __init__(self, opt):
        super().__init__(opt)
        self._name = "FlowCompositionForSwapper"
        self.all_faces_ids = list(range(self.render.nf))
        self.part_faces = list(self.render.body_parts.values())

    def get_selected_fids(self, selected_part_ids):
        """:762-780 (face ids sorted: the reference returns them in set order, which only feeds index assignments)."""
        fids = set()
        for i in selected_part_ids:
            fids |= set(int(f) for f in self.part_faces[i])
        return sorted(fids)

    def get_select_left_info(self, part_name="body"):
        """:782-792."""
        selected_part_ids = self.PART_IDS[part_name]
        left_part_ids = [i for i in self.PART_IDS["all"] if i not in selected_part_ids]
        return selected_part_ids, left_part_ids, self.get_selected_fids(selected_part_ids), self.get_selected_fids(left_part_ids)

    @torch.no_grad()
    def add_rendered_selected_f2pts(self, src_info, selected_fids):
        """:794-814: f2pts with every non-selected face moved to -2 (outside any flow)."""
        src_info["selected_obj_f2pts"] = self.render.get_selected_f2pts(src_info["obj_f2pts"], selected_fids)
        src_info["selected_f2pts"] = self.render.get_selected_f2pts(src_info["f2pts"], selected_fids)
        if self.only_vis:
            fim = src_info["fim"]
            src_info["selected_obj_f2pts"] = self.render.get_vis_f2pts(src_info["selected_obj_f2pts"], fim)
            src_info["selected_f2pts"] = self.render.get_vis_f2pts(src_info["selected_f2pts"], fim)

    @torch.no_grad()
    def merge_uv_img(self, src_info_list):
        """:816-856: every person's UV image weighted by where its selected faces land in UV space."""
        S = self.image_size
        one = torch.ones(1, 1, S, S, dtype=torch.float32, device=self.uv_fim.device)
        uv, vis = [], []
        for info in src_info_list:
            Ts2uv = self.render.cal_bc_transform(info["selected_obj_f2pts"][0:1].contiguous(), self.uv_fim[0:1], self.uv_wim[0:1])
            uv.append(info["uv_img"])
            vis.append(ops.grid_sample(one, Ts2uv))
        return ops.uv_merge_parts(torch.cat(uv, dim=0), torch.cat(vis, dim=0))

    @torch.no_grad()
    def merge_src_info(self, src_info_list, primary_ids):
        """:858-959.  Per-source tensors are concatenated along the source axis; offsets / links / background come from the
        primary person; the engine's NHWC feature cache is concatenated the same way."""
        from .networks.generator import SourceFeatures, _FeatList
        cat0 = ("cam", "shape", "pose", "fim", "wim", "f2pts", "obj_f2pts", "selected_f2pts", "selected_obj_f2pts")
        out = {"num_source": sum(i["num_source"] for i in src_info_list)}
        for k in cat0:
            out[k] = torch.cat([_force(i[k]) for i in src_info_list], dim=0)
        out["only_vis_f2pts"] = _Lazy(lambda: torch.cat([_force(i["only_vis_f2pts"]) for i in src_info_list], dim=0))
        out["img"] = torch.cat([i["img"] for i in src_info_list], dim=1)
        prim = src_info_list[primary_ids]
        out["offsets"], out["links_ids"], out["bg"] = prim["offsets"], prim["links_ids"], prim["bg"]
        enc = _FeatList(torch.cat(f, dim=0) for f in zip(*[i["feats"][0] for i in src_info_list]))
        res = _FeatList(torch.cat(f, dim=0) for f in zip(*[i["feats"][1] for i in src_info_list]))
        caches = [i["feats_nhwc"] for i in src_info_list]
        kv = [tuple(None if parts[0] is None else torch.cat(parts, dim=0) for parts in zip(*site))
              for site in zip(*[c.kv for c in caches])]
        cache = SourceFeatures([torch.cat(f, dim=0) for f in zip(*[c.enc for c in caches])],
                               [torch.cat(f, dim=0) for f in zip(*[c.res for c in caches])], kv, out["num_source"], batched=False)
        enc.lwg_cache = res.lwg_cache = cache
        out["feats"], out["feats_nhwc"] = (enc, res), cache
        out["uv_img"] = self.merge_uv_img(src_info_list)
        out["uv_img4"] = ops.nchw_to_nhwc(out["uv_img"].contiguous(), c_pad=4)[0].contiguous()
        return out


class _Lazy:
    """Defers a value nobody may ask for (the reference computes only_vis_f2pts eagerly, with two host syncs)."""

    def __init__(self, fn):
        self.fn, self.val = fn, None

    def get(self):
        if self.val is None:
            self.val = self.fn()
        return self.val


def _force(v):
    return v.get() if isinstance(v, _Lazy) else v
