"""TEST INFRASTRUCTURE ONLY - CPU oracle for the per-frame synthesis path of iPERCore's Liquid Warping GAN.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module,
and only as the checker.  Nothing under ``ipercore_amd/`` imports it; the product path has no CPU fallback.

Each function is a plain fp32 (torch-CPU / numpy) restatement of one reference function and cites the
file:line (relative to the iPERCore checkout) it follows.  Pinning status:

* everything except the rasterizer is pinned against outputs of the reference's own Python, imported in the
  authoring container by ``tests/golden/make_golden.py`` (fixtures in ``tests/golden/*.npz``; checked by
  ``tests/test_oracle_golden.py``);
* ``rasterize_fim_wim`` (``oracle/raster_oracle.c``) is **parity unpinned**: the reference delegates it to
  the third-party CUDA package ``neural_renderer`` (iPERDance/neural_renderer@e5f54f7, requirements/build.txt:3)
  whose source is absent; it restates the published upstream algorithm and is anchored on the reference's
  call sites and on the identity-warp property (see the C file's header).
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblwg_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "liblwg_oracle.so"])
        _LIB = ctypes.CDLL(path)
        _LIB.lwg_oracle_rasterize_fim_wim.restype = None
        _LIB.lwg_oracle_rasterize_fim_wim.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
            ctypes.c_void_p, ctypes.c_void_p]
    return _LIB


# --------------------------------------------------------------------------------------------------
# a2: SMPL-H linear blend skinning
# --------------------------------------------------------------------------------------------------
def quat_to_rotmat(q):
    """(N,4) wxyz -> (N,3,3), renormalising first (tools/utils/geometry/rotations.py:355-375)."""
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    m = torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                     2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                     2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1)
    return m.view(-1, 3, 3)


def rotvec_to_rotmat(rv):
    """(N,3) axis-angle -> (N,3,3) through a quaternion; angle = ||rv + 1e-8|| (rotations.py:318-332)."""
    ang = torch.norm(rv + 1e-8, p=2, dim=1).unsqueeze(-1)
    axis = rv / ang
    half = ang * 0.5
    return quat_to_rotmat(torch.cat([torch.cos(half), torch.sin(half) * axis], dim=1))


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """smplx/lbs.py:137-227 (pose2rot=True).  Returns verts (B,V,3), posed joints (B,J,3)."""
    B = max(betas.shape[0], pose.shape[0])
    nj = J_regressor.shape[0]
    v_shaped = v_template + torch.einsum("bl,mkl->bmk", betas, shapedirs)          # lbs.py:185, :250-271
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)                          # lbs.py:190, :230-247
    R = rotvec_to_rotmat(pose.reshape(-1, 3)).view(B, nj, 3, 3)                     # lbs.py:201, :378-406
    pose_feature = (R[:, 1:] - torch.eye(3)).reshape(B, -1)
    v_posed = torch.matmul(pose_feature, posedirs).view(B, -1, 3) + v_shaped        # lbs.py:203-215
    # batch_rigid_transform, lbs.py:321-375
    rel = J.clone()
    rel[:, 1:] = rel[:, 1:] - J[:, parents[1:]]
    T = torch.zeros(B, nj, 4, 4)
    T[:, :, :3, :3] = R
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1.0
    chain = [T[:, 0]]
    for j in range(1, nj):
        chain.append(torch.matmul(chain[int(parents[j])], T[:, j]))
    G = torch.stack(chain, dim=1)
    posed_joints = G[:, :, :3, 3]
    Jh = F.pad(J.unsqueeze(-1), [0, 0, 0, 1])
    A = G - F.pad(torch.matmul(G, Jh), [3, 0, 0, 0, 0, 0, 0, 0])
    Tv = torch.matmul(lbs_weights.unsqueeze(0).expand(B, -1, -1), A.view(B, nj, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1)], dim=2)
    verts = torch.matmul(Tv, vh.unsqueeze(-1))[:, :, :3, 0]
    return verts, posed_joints


class SMPLHModel:
    """Buffers of bodynets/batch_smplh.py:36-131 + smplx/body_models.py:200-296, from the pickle dict."""

    def __init__(self, data):
        f32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))                # noqa: E731
        self.v_template = f32(data["v_template"])
        self.shapedirs = f32(data["shapedirs"])
        pd = np.asarray(data["posedirs"])
        self.posedirs = f32(np.reshape(pd, [-1, pd.shape[-1]]).T)
        self.J_regressor = f32(data["J_regressor"])
        parents = torch.tensor(np.asarray(data["kintree_table"][0]).astype(np.int64))
        parents[0] = -1
        self.parents = parents
        self.lbs_weights = f32(data["weights"])
        self.hands_mean = f32(np.concatenate([data["hands_meanl"], data["hands_meanr"]]))


def link(verts, linked_ids):
    """bodynets/base_smpl.py:28-50."""
    out = verts.clone()
    ids = torch.as_tensor(linked_ids).long()
    if ids.dim() == 2:
        out[:, ids[:, 0]] = verts[:, ids[:, 1]]
    else:
        for i in range(verts.shape[0]):
            on = ids[i, :, 2] == 1
            out[i, ids[i, on, 0]] = verts[i, ids[i, on, 1]]
    return out


def smplh_get_details(model, theta, offsets=0, links_ids=None):
    """bodynets/base_smpl.py:107-142 over bodynets/batch_smplh.py:137-180."""
    theta = torch.as_tensor(theta, dtype=torch.float32)
    cam, pose, shape = theta[:, 0:3], theta[:, 3:-10].contiguous(), theta[:, -10:].contiguous()
    if pose.shape[1] == 72:
        pose = torch.cat([pose[:, 0:66], model.hands_mean.repeat(pose.shape[0], 1)], dim=1)
    verts, j3d = lbs(shape, pose, model.v_template + torch.as_tensor(offsets, dtype=torch.float32),
                     model.shapedirs, model.posedirs, model.J_regressor, model.parents, model.lbs_weights)
    if links_ids is not None:
        verts = link(verts, links_ids)
    j2d = cam[:, None, 0:1] * (j3d[:, :, :2] + cam[:, None, 1:])                    # base_smpl.py:7-18
    return {"theta": theta, "cam": cam, "pose": theta[:, 3:-10].contiguous(), "shape": shape,
            "verts": verts, "j2d": j2d, "j3d": j3d}


def cam_swap(src_cam, ref_cam, first_cam, strategy="smooth"):
    """tools/utils/geometry/cam_pose_utils.py:17-50."""
    if strategy == "smooth":
        cam = src_cam.clone()
        cam[:, 1:] += ref_cam[:, 1:] - first_cam[:, 1:]
        cam[:, 0] = cam[:, 0] * ref_cam[:, 0] / first_cam[:, 0]
        return cam
    if strategy == "ref_txty":
        cam = src_cam.clone()
        cam[:, 1:] = ref_cam[:, 1:]
        return cam
    return src_cam if strategy == "source" else ref_cam


# --------------------------------------------------------------------------------------------------
# a3/a4: projection + rasterizer
# --------------------------------------------------------------------------------------------------
EYE_Z = -(1.0 / np.tan(np.radians(30.0)) + 1.0)                                       # renders/nmr.py:225


def look_at(vertices, eye):
    """Upstream neural_renderer.look_at (at = 0, up = +y): rotate/translate into the eye frame."""
    eye = torch.tensor(eye, dtype=torch.float32)
    up = torch.tensor([0.0, 1.0, 0.0])
    z = F.normalize(-eye, dim=0, eps=1e-5)
    x = F.normalize(torch.cross(up, z, dim=0), dim=0, eps=1e-5)
    y = F.normalize(torch.cross(z, x, dim=0), dim=0, eps=1e-5)
    r = torch.stack([x, y, z], dim=0)
    return torch.matmul(vertices - eye, r.t())


def project_faces(cam, verts, faces):
    """renders/nmr.py:34-52,:326-336: s(xy+t) keeping z, flip y, look_at, gather -> (B,nf,3,3)."""
    s = cam[:, 0].view(-1, 1, 1)
    t = cam[:, 1:3].view(cam.shape[0], 1, -1)
    proj = torch.cat([s * (verts[:, :, :2] + t), verts[:, :, 2:3]], dim=2).clone()
    proj[:, :, 1] *= -1
    v = look_at(proj, [0.0, 0.0, float(EYE_Z)])
    return v[:, torch.as_tensor(faces).long()]                                         # nr.vertices_to_faces


def rasterize_fim_wim(faces_v, image_size, near=0.1, far=100.0):
    """(B,nf,3,3) -> fim (B,S,S) int32, wim (B,S,S,3) fp32 via oracle/raster_oracle.c."""
    fv = np.ascontiguousarray(np.asarray(faces_v, dtype=np.float32))
    B, nf = fv.shape[0], fv.shape[1]
    S = int(image_size)
    fim = np.empty((B, S, S), dtype=np.int32)
    wim = np.empty((B, S, S, 3), dtype=np.float32)
    _lib().lwg_oracle_rasterize_fim_wim(fv.ctypes.data, B, nf, S, ctypes.c_float(near), ctypes.c_float(far),
                                        fim.ctypes.data, wim.ctypes.data)
    return torch.from_numpy(fim), torch.from_numpy(wim)


def render_fim_wim(cam, verts, faces, image_size):
    """renders/nmr.py:319-342 -> f2pts (B,nf,3,2) (y un-flipped), fim, wim."""
    fv = project_faces(cam, verts, faces)
    fim, wim = rasterize_fim_wim(fv.numpy(), image_size)
    f2pts = fv[:, :, :, 0:2].clone()
    f2pts[:, :, :, 1] *= -1
    return f2pts, fim, wim


def render_uv_fim_wim(f_img2uvs, bs, image_size):
    """renders/nmr.py:344-358."""
    f = torch.as_tensor(f_img2uvs, dtype=torch.float32).repeat(bs, 1, 1, 1).clone()
    f[:, :, :, 1] *= -1
    return rasterize_fim_wim(f.numpy(), image_size)


# --------------------------------------------------------------------------------------------------
# a5/a7/a8/a9: codes and flows
# --------------------------------------------------------------------------------------------------
def encode_fim(map_fn, fim, transpose=True):
    """renders/nmr.py:390-401 - fim == -1 picks the LAST row by negative indexing."""
    enc = torch.as_tensor(map_fn)[fim.long()]
    return enc.permute(0, 3, 1, 2) if transpose else enc


def cal_bc_transform(src_f2pts, dst_fims, dst_wims):
    """renders/nmr.py:713-757: T[p] = sum_k wim[p,k] * src_f2pts[fim[p],k,:], background -> (-2,-2)."""
    B, S = dst_fims.shape[0], dst_fims.shape[1]
    T = -2 * torch.ones((B, S * S, 2), dtype=torch.float32)
    for i in range(B):
        idx = dst_fims[i].long().reshape(-1)
        w = dst_wims[i].reshape(-1, 3)
        on = idx != -1
        T[i, on] = (src_f2pts[i][idx[on]] * w[on][:, :, None]).sum(dim=1)
    return T.view(B, S, S, 2)


def make_tsf_inputs(uv_img, f_uvs2img, cond, fim, wim):
    """models/flowcomposition.py:206-248 for nt == 1: cat[grid_sample(uv_img, Tuv2t), cond] -> (bs,6,h,w)."""
    bs = cond.shape[0]
    Tuv2t = cal_bc_transform(torch.as_tensor(f_uvs2img).repeat(bs, 1, 1, 1), fim, wim)
    syn = F.grid_sample(uv_img.expand(bs, -1, -1, -1), Tuv2t, mode="bilinear", padding_mode="zeros",
                        align_corners=False)
    return torch.cat([syn, cond], dim=1), Tuv2t


def make_trans_flow(src_f2pts, ref_fim, ref_wim):
    """models/flowcomposition.py:514-567 (bs=1, temporal=False): Tst (1,ns,h,w,2)."""
    ns = src_f2pts.shape[0]
    T = cal_bc_transform(src_f2pts, ref_fim.repeat(ns, 1, 1), ref_wim.repeat(ns, 1, 1, 1))
    return T.view(1, ns, T.shape[1], T.shape[2], 2)


# --------------------------------------------------------------------------------------------------
# a10/a11/a12/a15: AttLWB-SPADE generator, functional over a state_dict
# --------------------------------------------------------------------------------------------------
def _conv(sd, name, x, stride=1, pad=1):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=pad)


def _convT(sd, name, x):
    return F.conv_transpose2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=2, padding=1)


def lwb_transform(x, T):
    """generators/attlwb_spade_resunet.py:175-191: flow resize (align_corners=True) then grid_sample."""
    h, w = x.shape[-2:]
    if T.shape[1] != h or T.shape[2] != w:
        T = F.interpolate(T.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    return F.grid_sample(x, T, mode="bilinear", padding_mode="zeros", align_corners=False)


def attention_lwb(sd, p, tsf_x, src_x, Tst, temp_x=None, Ttt=None):
    """SelfAttentionLWB.forward :208-252 + SelfAttentionBlock :106-139 + SPADE :80-93; with temporal features
    (temp_x (bs*nt,c,h,w), Ttt (bs,nt,H,W,2)) their warped K / V are appended along the source axis (:232-243)."""
    bs, ns, H, W, _ = Tst.shape
    h, w = tsf_x.shape[-2:]
    warp = lwb_transform(src_x, Tst.reshape(bs * ns, H, W, 2))
    K = _conv(sd, p + ".fk", warp, pad=0).view(bs, ns, -1, h, w)
    V = _conv(sd, p + ".fv", warp, pad=0).view(bs, ns, -1, h, w)
    if temp_x is not None and Ttt is not None:
        nt = Ttt.shape[1]
        twarp = lwb_transform(temp_x, Ttt.reshape(bs * nt, H, W, 2))
        K = torch.cat([K, _conv(sd, p + ".fk", twarp, pad=0).view(bs, nt, -1, h, w)], dim=1)
        V = torch.cat([V, _conv(sd, p + ".fv", twarp, pad=0).view(bs, nt, -1, h, w)], dim=1)
    q = _conv(sd, p + ".fq", tsf_x, pad=0)
    logits = (K * q.unsqueeze(1)).sum(dim=2, keepdim=True) / math.sqrt(K.shape[2])
    alpha = torch.softmax(logits, dim=1)
    x = (alpha * V).sum(dim=1)
    normalized = F.instance_norm(tsf_x, eps=1e-5)
    actv = F.relu(_conv(sd, p + ".spade.mlp_shared.0", x))
    gamma = _conv(sd, p + ".spade.mlp_gamma", actv)
    beta = _conv(sd, p + ".spade.mlp_beta", actv)
    return normalized * (1 + gamma) + beta


def fuse_lwb(sd, p, tsf_x, src_x, Tst, kind):
    """The non-attention Liquid Warping Blocks: ``AddLWB`` / ``AvgLWB`` (generators/lwb_resunet.py:77-152: sum / mean over
    [tsf_x, warped sources]) and ``SoftGateLWB`` (generators/lwb_softgate_resunet.py:77-123: tsf_x + gate(tsf_x) * sum|mean of the
    warped sources, gate = sigmoid(conv3(relu(conv3(tsf_x))))).  kind in {"add", "avg", "sg_add", "sg_avg"}."""
    bs, ns, H, W, _ = Tst.shape
    h, w = tsf_x.shape[-2:]
    warp = lwb_transform(src_x, Tst.reshape(bs * ns, H, W, 2)).view(bs, ns, -1, h, w)
    if kind == "add":
        return tsf_x + warp.sum(dim=1)
    if kind == "avg":
        return torch.cat([tsf_x.unsqueeze(1), warp], dim=1).mean(dim=1)
    fused = warp.sum(dim=1) if kind == "sg_add" else warp.mean(dim=1)
    gate = torch.sigmoid(_conv(sd, p + ".gate_conv.2", F.relu(_conv(sd, p + ".gate_conv.0", tsf_x))))
    return tsf_x + gate * fused


def _res_block(sd, p, x):
    """ResidualBlock :14-25."""
    return x + _conv(sd, p + ".main.2", F.relu(_conv(sd, p + ".main.0", x)))


def gen_forward_src(sd, src_inputs, n_down=3, n_res=6):
    """BaseAttentionLWBGenerator.forward_src(only_enc=True) :450-478."""
    bs, ns, _, h, w = src_inputs.shape
    x = src_inputs.view(bs * ns, -1, h, w)
    enc = []
    for i in range(n_down):
        x = F.relu(_conv(sd, f"src_net.encoders.layers.{i}.0", x, stride=2))
        enc.append(x)
    res = []
    for i in range(n_res):
        x = _res_block(sd, f"src_net.res_blocks.{i}", x)
        res.append(x)
    return enc, res


def gen_forward_src_full(sd, src_inputs, n_down=3, n_res=6):
    """forward_src(only_enc=False) :450-478: features + the SIDNet decoder (Decoder :291-314) and regressors (:376-384)."""
    bs, ns, _, h, w = src_inputs.shape
    enc, res = gen_forward_src(sd, src_inputs, n_down, n_res)
    x = res[-1]
    for i in range(n_down):
        x = F.relu(_convT(sd, f"src_net.decoders.layers.{i}.0", x))
    img = torch.tanh(_conv(sd, "src_net.img_reg.0", x, pad=2))
    mask = torch.sigmoid(_conv(sd, "src_net.att_reg.0", x, pad=2))
    return enc, res, img.view(bs, ns, 3, h, w), mask.view(bs, ns, 1, h, w)


def gen_forward_train(sd, bg_inputs, src_inputs, tsf_inputs, Tst, n_down=3, n_res=6, n_bg=4):
    """AttentionLWBGenerator.forward(..., only_tsf=False) :633-699 with temporal=False, bs = 1: differentiable w.r.t. sd."""
    bs, nt = tsf_inputs.shape[:2]
    bg = gen_forward_bg(sd, bg_inputs, n_down=n_bg, n_res=n_res)
    enc, res, s_img, s_mask = gen_forward_src_full(sd, src_inputs, n_down, n_res)
    imgs, masks = [], []
    for t in range(nt):
        img, mask = gen_forward_tsf(sd, tsf_inputs[:, t], enc, res, Tst[:, t], n_down, n_res)
        imgs.append(img)
        masks.append(mask)
    return bg, s_img, s_mask, torch.stack(imgs, dim=1), torch.stack(masks, dim=1)


def gen_forward_tsf(sd, tsf_inputs, src_enc_outs, src_res_outs, Tst, n_down=3, n_res=6, temp_enc_outs=None,
                    temp_res_outs=None, Ttt=None, lwb="att"):
    """BaseAttentionLWBGenerator.forward_tsf :480-535 -> (tsf_img, tsf_mask); temp_* / Ttt: the temporal attention inputs.
    lwb != "att": the same stream with the AddLWB / AvgLWB / SoftGateLWB blocks (lwb_resunet.py:414-455,
    lwb_softgate_resunet.py:414-465; those generators ignore the temporal inputs)."""
    x = tsf_inputs
    enc = []
    tmp = temp_enc_outs is not None and Ttt is not None

    def block(p, x_, src, tmp_feats, i_):
        if lwb == "att":
            return attention_lwb(sd, p, x_, src, Tst, tmp_feats[i_] if tmp else None, Ttt if tmp else None)
        return fuse_lwb(sd, p, x_, src, Tst, lwb)

    for i in range(n_down):
        x = F.relu(_conv(sd, f"tsf_net_enc.layers.{i}.0", x, stride=2))
        x = block(f"enc_attlwbs.{i}", x, src_enc_outs[i], temp_enc_outs, i)
        enc.append(x)
    for i in range(n_res):
        x = _res_block(sd, f"res_blocks.{i}", x)
        x = block(f"res_attlwbs.{i}", x, src_res_outs[i], temp_res_outs, i)
    for i in range(n_down):                                                      # SkipDecoder :316-357
        x = F.relu(_convT(sd, f"tsf_net_dec.upconvs.{i}.0", x))
        if i != n_down - 1:
            x = F.relu(_conv(sd, f"tsf_net_dec.skippers.{i}.0", torch.cat([enc[n_down - 2 - i], x], dim=1)))
    img = torch.tanh(_conv(sd, "tsf_img_reg.0", x, pad=2))
    mask = torch.sigmoid(_conv(sd, "tsf_att_reg.0", x, pad=2))
    return img, mask


def gen_forward_bg(sd, bg_inputs, n_down=4, n_res=6):
    """AttentionLWBGenerator.forward_bg :615-631 over ResNetInpaintor (generators/bg_inpaintor.py:24-60).

    n_down = len(num_filters); Sequential indices follow the construction order of bg_inpaintor.py:31-57.
    """
    bs, ns, _, h, w = bg_inputs.shape
    x = bg_inputs.view(bs * ns, -1, h, w)

    def cv(name, t, stride, pad):
        return F.conv2d(t, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=pad)

    i = 0
    x = F.relu(F.instance_norm(cv(f"bg_net.main.{i}", x, 1, 3), eps=1e-5))
    i += 3
    for _ in range(n_down - 1):
        x = F.relu(F.instance_norm(cv(f"bg_net.main.{i}", x, 2, 1), eps=1e-5))
        i += 3
    for _ in range(n_res):
        y = F.relu(F.instance_norm(cv(f"bg_net.main.{i}.main.0", x, 1, 1), eps=1e-5))
        x = x + F.instance_norm(cv(f"bg_net.main.{i}.main.3", y, 1, 1), eps=1e-5)
        i += 1
    for _ in range(n_down - 1):
        x = F.conv_transpose2d(x, sd[f"bg_net.main.{i}.weight"], sd.get(f"bg_net.main.{i}.bias"), stride=2, padding=1)
        x = F.relu(F.instance_norm(x, eps=1e-5))
        i += 3
    x = torch.tanh(cv(f"bg_net.main.{i}", x, 1, 3))
    return x.view(bs, ns, 3, h, w)


def compose(tsf_img, tsf_mask, bg_img):
    """models/imitator.py:393."""
    return tsf_mask * bg_img + (1 - tsf_mask) * tsf_img


def to_uint8_bgr(pred_chw):
    """tools/utils/filesio/cv_utils.py:100-116 (normalize=True): CHW RGB [-1,1] -> HWC BGR uint8, truncation."""
    img = np.transpose(np.asarray(pred_chw), (1, 2, 0))[:, :, ::-1]
    return ((img + 1) / 2.0 * 255).astype(np.uint8)


# --------------------------------------------------------------------------------------------------
# whole frame (a1..a13), used by parity tests and bench.py's cpu_baseline
# --------------------------------------------------------------------------------------------------
def imitate_frame(model, tables, sd, src_info, tgt_smpl, first_cam, image_size, cam_strategy="smooth", ref_override=None):
    """One iteration of Imitator.inference (models/imitator.py:341-395, temporal=False, cam "smooth").

    tables: dict(smpl_faces, map_fn, f_uvs2img).  src_info: dict(cam, shape, offsets, links_ids, uv_img, bg,
    f2pts (ns,nf,3,2), feats=(enc list, res list)).  Returns dict with pred and every intermediate.
    """
    tgt = torch.as_tensor(tgt_smpl, dtype=torch.float32).view(1, -1)
    cam = cam_swap(src_info["cam"][0:1], tgt[:, 0:3], first_cam, cam_strategy)
    ref_smpl = torch.cat([cam, tgt[:, 3:-10], src_info["shape"][0:1]], dim=1)
    ref = smplh_get_details(model, ref_smpl, src_info.get("offsets", 0), src_info.get("links_ids"))
    own_verts, own_cam = ref["verts"], ref["cam"]
    if ref_override is not None:          # tests: rasterize the caller's (bit-identical) vertices, see parity_utils
        ref["cam"], ref["verts"] = ref_override
    f2pts, fim, wim = render_fim_wim(ref["cam"], ref["verts"], tables["smpl_faces"], image_size)
    cond = encode_fim(tables["map_fn"], fim)
    tsf_inputs, Tuv2t = make_tsf_inputs(src_info["uv_img"], tables["f_uvs2img"], cond, fim, wim)
    # flowcomposition.py:556-562: opt.only_vis swaps in the visible-faces-only source projection (nmr.py:639-681)
    Tst = make_trans_flow(src_info["only_vis_f2pts"] if src_info.get("only_vis") else src_info["f2pts"], fim, wim)
    enc, res = src_info["feats"]
    img, mask = gen_forward_tsf(sd, tsf_inputs, enc, res, Tst, n_down=len(enc), n_res=len(res))
    pred = compose(img, mask, src_info["bg"])
    return {"pred": pred, "mask": mask, "img": img, "tsf_inputs": tsf_inputs, "Tst": Tst, "fim": fim, "wim": wim,
            "cond": cond, "verts": ref["verts"], "f2pts": f2pts, "cam": ref["cam"], "Tuv2t": Tuv2t, "own_verts": own_verts, "own_cam": own_cam,
            "src_own_verts": src_info.get("own_verts"), "src_fim": src_info.get("fim")}


class TemporalFIFO:
    """models/imitator.py:18-127: ring of the last ``time_step`` synthesized frames (f2pts + their SIDNet features)."""

    def __init__(self, time_step):
        self.time_step, self.index = time_step, 0
        self.f2pts, self.enc, self.res = [None] * time_step, [None] * time_step, [None] * time_step

    @property
    def nt(self):
        return min(self.index, self.time_step)

    def append(self, f2pts, enc, res):
        i = self.index % self.time_step
        self.f2pts[i], self.enc[i], self.res[i] = f2pts, enc, res
        self.index += 1

    def tensors(self):
        n = self.nt
        f2pts = torch.cat(self.f2pts[:n], dim=0)
        enc = [torch.cat([self.enc[k][l] for k in range(n)], dim=0) for l in range(len(self.enc[0]))]
        res = [torch.cat([self.res[k][l] for k in range(n)], dim=0) for l in range(len(self.res[0]))]
        return f2pts, enc, res


def imitate_sequence_temporal(model, tables, sd, src_info, tgt_smpls, image_size, time_step=1, cam_strategy="smooth"):
    """Imitator.inference with temporal=True (models/imitator.py:341-366): frame t attends to the sources AND to the last
    ``time_step`` synthesized frames (features of forward_src on [pred, cond], post_update :397-401), warped by Ttt
    (flowcomposition.py:569-579).  Returns the list of preds."""
    fifo = TemporalFIFO(time_step)
    tgt = torch.as_tensor(tgt_smpls, dtype=torch.float32)
    if cam_strategy == "smooth":
        tgt = stabilize(model, tgt)
    first_cam = tgt[0:1, 0:3].clone()
    n_down, n_res = len(src_info["feats"][0]), len(src_info["feats"][1])
    preds = []
    for t in range(tgt.shape[0]):
        cam = cam_swap(src_info["cam"][0:1], tgt[t:t + 1, 0:3], first_cam, cam_strategy)
        ref_smpl = torch.cat([cam, tgt[t:t + 1, 3:-10], src_info["shape"][0:1]], dim=1)
        ref = smplh_get_details(model, ref_smpl, src_info.get("offsets", 0), src_info.get("links_ids"))
        f2pts, fim, wim = render_fim_wim(ref["cam"], ref["verts"], tables["smpl_faces"], image_size)
        cond = encode_fim(tables["map_fn"], fim)
        tsf_inputs, _ = make_tsf_inputs(src_info["uv_img"], tables["f_uvs2img"], cond, fim, wim)
        Tst = make_trans_flow(src_info["f2pts"], fim, wim)
        enc, res = src_info["feats"]
        if t == 0:
            img, mask = gen_forward_tsf(sd, tsf_inputs, enc, res, Tst, n_down, n_res)
        else:
            tf2pts, tenc, tres = fifo.tensors()
            Ttt = make_trans_flow(tf2pts, fim, wim)
            img, mask = gen_forward_tsf(sd, tsf_inputs, enc, res, Tst, n_down, n_res, tenc, tres, Ttt)
        pred = compose(img, mask, src_info["bg"])
        cur = torch.cat([pred, cond], dim=1).unsqueeze(1)
        e2, r2 = gen_forward_src(sd, cur, n_down, n_res)
        fifo.append(f2pts, e2, r2)
        preds.append(pred)
    return preds


# --------------------------------------------------------------------------------------------------
# sequence-global pre-pass of Imitator.inference: WeakPerspectiveCamera.stabilize
# --------------------------------------------------------------------------------------------------
def _turning_points(y):
    """cam_pose_utils.py:130-153."""
    idx = [0] + [i for i in range(1, len(y) - 1) if (y[i] - y[i - 1]) * (y[i + 1] - y[i]) < 0]
    return idx + [len(y) - 1]


def jump_intervals(final_foot_y, up_thr=0.2, down_thr=0.1):
    """cam_pose_utils.py:155-208 -> [(start, end), ...]."""
    y = final_foot_y
    ground = y[0]
    pts = _turning_points(y)
    out, start, in_jump = [], None, False
    for a, b in zip(pts[:-1], pts[1:]):
        rise = y[b] - y[a]
        if rise < 0 and abs(rise) > up_thr:
            in_jump = True
            below = [f for f in range(a, b) if y[f] < ground]
            start = below[0] if below else a
        elif in_jump:
            if y[b] < y[start] and abs(y[b] - y[start]) > down_thr:
                continue
            in_jump = False
            out.append((start, b))
            start = None
    if in_jump:
        out.append((start, len(y) - 1))
    return out


def stabilize(model, smpls):
    """cam_pose_utils.py:52-99 (foot height via cam_pose_utils.py:101-128 = SMPLH.forward without offsets)."""
    smpls = torch.as_tensor(smpls, dtype=torch.float32)
    cam, pose, shape = smpls[:, 0:3], smpls[:, 3:-10], smpls[:, -10:]
    shape = shape[0:1].repeat(pose.shape[0], 1)
    full = pose if pose.shape[1] != 72 else torch.cat([pose[:, 0:66], model.hands_mean.repeat(pose.shape[0], 1)], dim=1)
    verts, _ = lbs(shape, full, model.v_template, model.shapedirs, model.posedirs, model.J_regressor, model.parents,
                   model.lbs_weights)
    foot_y = verts[:, :, 1].max(dim=1)[0]
    cam_y = cam[:, 2]
    new_y = cam_y[0] + (-foot_y + foot_y[0])
    for s, e in jump_intervals((foot_y + cam_y).numpy()):
        new_y[s:e + 1] = torch.min(cam_y[s:e + 1], new_y[s:e + 1])
    new_cam = torch.zeros_like(cam)
    new_cam[:, 0] = 1
    new_cam[:, 2] = new_y
    return torch.cat([new_cam, pose, shape], dim=1)


# --------------------------------------------------------------------------------------------------
# a15: once-per-source image stage (Imitator.source_setup -> FlowComposition.process_source)
# Pinned by tests/golden/golden_source_v1.npz (generated by tests/golden/make_golden_source.py from the reference's own
# FlowComposition / morph / CannyFilter code, with cv2's two kernel-construction calls restated - see that script).
# --------------------------------------------------------------------------------------------------
def morph(mask, ks, mode="erode"):
    """tools/utils/morphology/morph_ops.py:7-37."""
    n_ks, pad_s = ks ** 2, ks // 2
    kernel = torch.ones(1, 1, ks, ks, dtype=torch.float32)
    if mode == "erode":
        out = F.conv2d(F.pad(mask, [pad_s] * 4, value=1.0), kernel)
        return (out == n_ks).float()
    out = F.conv2d(F.pad(mask, [pad_s] * 4, value=0.0), kernel)
    return (out >= 1).float()


def canny_kernels():
    """canny_ops.py:9-36 gaussian / sobel (float64 -> float32 like the reference's weight assignment) and the 8
    directional kernels of :39-68.  cv2 is absent here, so the directional kernels are restated as what
    getRotationMatrix2D + warpAffine + the |k| == 1 mask produce for 0..315 degrees: +1 at the centre, -1 at the
    neighbour E, NE, N, NW, W, SW, S, SE (image y down; positive angle = counter-clockwise)."""
    g1 = np.linspace(-1, 1, 3)
    x, y = np.meshgrid(g1, g1)
    d = (x ** 2 + y ** 2) ** 0.5
    g = np.exp(-(d - 0) ** 2 / (2 * 1 ** 2)) / (2 * np.pi * 1 ** 2)
    g = g / np.sum(g)
    r = np.linspace(-1, 1, 3)
    x, y = np.meshgrid(r, r)
    den = x ** 2 + y ** 2
    den[:, 1] = 1
    sob = x / den
    dirs = np.zeros((8, 3, 3))
    for i, (dx, dy) in enumerate(((1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1))):
        dirs[i, 1, 1] = 1
        dirs[i, 1 + dy, 1 + dx] = -1
    t = lambda a: torch.tensor(a, dtype=torch.float64).float()      # noqa: E731
    return t(g)[None, None], t(sob)[None, None], t(sob.T)[None, None], t(dirs)[:, None]


def _conv3_np(img, k9):
    """3x3 cross-correlation, zero padding, taps accumulated in row-major order with one rounding per multiply and
    per add (numpy fp32: IEEE on every host).  Equals torch's CPU conv2d on the authoring container bit for bit
    (golden fixture) without depending on which oneDNN kernel another host's CPU selects."""
    n, c, H, W = img.shape
    p = np.pad(img, ((0, 0), (0, 0), (1, 1), (1, 1)))
    acc = np.zeros_like(img, dtype=np.float32)
    for t in range(9):
        acc = acc + np.float32(k9[t]) * p[:, :, t // 3:t // 3 + H, t % 3:t % 3 + W]
    return acc


def canny(img, low, high):
    """canny_ops.py:137-212 CannyFilter.forward(img (B,1,H,W), low, high, hysteresis=True) -> thin edges {0,1}.
    Same operations in the same order as the reference, evaluated with numpy fp32 (every +,*,/,sqrt individually and
    correctly rounded) so the result does not depend on the host CPU's vector ISA."""
    kg, kx, ky, kd = (k.numpy() for k in canny_kernels())
    x = img.numpy().astype(np.float32)
    blurred = _conv3_np(x, kg.reshape(9))
    gx, gy = _conv3_np(blurred, kx.reshape(9)), _conv3_np(blurred, ky.reshape(9))
    mag = np.sqrt(gx * gx + gy * gy)
    with np.errstate(divide="ignore", invalid="ignore"):
        ori = np.arctan(gy / gx) * np.float32(360 / np.pi) + np.float32(180)
        ori = np.round(ori / np.float32(45)) * np.float32(45)
        pos_idx = np.mod(ori / np.float32(45), np.float32(8))
    directional = np.stack([_conv3_np(mag, kd[i].reshape(9)) for i in range(8)], axis=1)[:, :, 0]
    thin = mag.copy()
    for pos_i in range(4):
        neg_i = pos_i + 4
        oriented = (pos_idx == pos_i) | (pos_idx == neg_i)
        is_max = np.minimum(directional[:, pos_i], directional[:, neg_i]) > 0.0
        thin[(~is_max[:, None]) & oriented] = 0.0
    lowm, highm = thin > np.float32(low), thin > np.float32(high)
    thin = lowm * np.float32(0.5) + highm * np.float32(0.5)
    weak = thin == 0.5
    weak_is_high = (_conv3_np(thin.astype(np.float32), np.full(9, 1.25, np.float32)) > 1) & weak
    return torch.tensor((highm * 1 + weak_is_high * 1).astype(np.float32))


def top_k_nearest(uncertain_pts, boundary_pts, top_k=3, chunk=4096):
    """flowcomposition.py:268-293 cal_top_k_ids: squared distances (int64), the k smallest per row.  The reference's
    topk(sorted=False) leaves the choice among equal distances to the backend; here ties go to the lowest boundary
    index (composite key), one valid instance of that behaviour.  Returns (weights (n1,k) fp32, ids (n1,k), vals)."""
    n2 = boundary_pts.shape[0]
    vals, ids = [], []
    ar = torch.arange(n2, dtype=torch.int64)[None]
    for s in range(0, uncertain_pts.shape[0], chunk):
        u = uncertain_pts[s:s + chunk]
        d = ((u[:, None, :] - boundary_pts[None, :, :]) ** 2).sum(dim=-1)
        key = d * n2 + ar
        k, _ = key.topk(k=top_k, dim=-1, largest=False, sorted=True)
        vals.append(k // n2)
        ids.append(k % n2)
    vals = torch.cat(vals) if vals else torch.zeros(0, top_k, dtype=torch.int64)
    ids = torch.cat(ids) if ids else torch.zeros(0, top_k, dtype=torch.int64)
    v = vals.float()
    return v / torch.sum(v, dim=1, keepdim=True), ids, vals


def make_morph_image(src_img, confidant_sil, outpad_sil):
    """flowcomposition.py:295-386 with erode_ks = dilate_ks = 0 (the reference's call, :482-483).
    Returns (morph_img (n,3,h,w), thin_edges, tie_mask (n,h,w) bool: pixels whose 3rd and 4th nearest boundary
    distances coincide - there the reference's own result depends on the topk backend)."""
    n, _, h, w = src_img.shape
    thin = canny(confidant_sil, 0.1, 0.9)
    uncertain_sil = outpad_sil * (1 - confidant_sil)
    outs, ties = [], torch.zeros(n, h, w, dtype=torch.bool)
    for i in range(n):
        b_pts = thin[i, 0].nonzero(as_tuple=False)
        u_pts = uncertain_sil[i, 0].nonzero(as_tuple=False)
        img = src_img[i] * confidant_sil[i]
        if u_pts.shape[0]:
            weights, ids, vals = top_k_nearest(u_pts, b_pts, 3)
            if b_pts.shape[0] > 3:
                _, _, v4 = top_k_nearest(u_pts, b_pts, 4)
                ties[i, u_pts[:, 0], u_pts[:, 1]] = v4[:, 2] == v4[:, 3]
            nn = b_pts[ids.reshape(-1)]
            rgbs = src_img[i][:, nn[:, 0], nn[:, 1]].view(3, -1, 3).permute(1, 0, 2)       # (n1, 3, k)
            img[:, u_pts[:, 0], u_pts[:, 1]] = torch.matmul(rgbs, weights.unsqueeze(-1)).squeeze(-1).permute(1, 0)
        outs.append(img)
    return torch.stack(outs, dim=0), thin, ties


def get_vis_f2pts(f2pts, fims, face_k_nearest):
    """renders/nmr.py:639-681."""
    out = []
    for i in range(f2pts.shape[0]):
        vis = torch.zeros_like(f2pts[i]) - 2.0
        face_ids = fims[i].unique()[1:].long()
        ids = torch.as_tensor(face_k_nearest)[face_ids].unique()
        vis[ids] = f2pts[i][ids]
        out.append(vis)
    return torch.stack(out, dim=0)


def make_uv_img(src_img, obj_f2pts, only_vis_obj_f2pts, uv_fim, uv_wim):
    """flowcomposition.py:87-137: src_img (bs,ns,3,h,w); f2pts (bs*ns,nf,3,2); uv maps (1,h,w[,3]) -> (bs,3,h,w)."""
    bs, ns, _, h, w = src_img.shape
    n = bs * ns
    fim, wim = uv_fim[0:1].repeat(n, 1, 1), uv_wim[0:1].repeat(n, 1, 1, 1)
    only_vis_T = cal_bc_transform(only_vis_obj_f2pts, fim, wim)
    T = cal_bc_transform(obj_f2pts, fim, wim)
    src_warp = F.grid_sample(src_img.reshape(n, 3, h, w), T, align_corners=False).view(bs, ns, -1, h, w)
    vis_warp = F.grid_sample(torch.ones(n, 1, h, w), only_vis_T, align_corners=False)
    vis_warp = morph(vis_warp, ks=13, mode="dilate").view(bs, ns, -1, h, w)
    vis_sum = torch.sum(vis_warp[:, 1:], dim=1)
    temp = torch.sum(src_warp[:, 1:] * vis_warp[:, 1:], dim=1) / (vis_sum + 1e-5)
    front_invisible = (1 - vis_warp[:, 0]) * (vis_sum >= 1).float()
    return src_warp[:, 0] * (1 - front_invisible) + temp * front_invisible


def process_source(src_img, cond, fim, obj_f2pts, obj_fim, uv_fim, uv_wim, face_k_nearest, masks=None,
                   conf_erode_ks=3, out_dilate_ks=51, bg_ks=11):
    """flowcomposition.py:139-204 (use_morph branch) + :452-512 for bs = 1, primary_ids = [0].
    src_img (1,ns,3,h,w); cond (ns,3,h,w); masks: background masks (ns,1,h,w) (= 1 - foreground) or None.
    Returns dict(uv_img, input_G_bg (1,1,4,h,w), input_G_src (1,ns,6,h,w), morph_img, thin_edges, tie_mask, ...)."""
    _, ns, _, h, w = src_img.shape
    rendered_sil = 1 - cond[:, -1:]
    human_sil = 1 - masks if masks is not None else rendered_sil
    confidant = morph(human_sil, conf_erode_ks, "erode")
    outpad = morph(((human_sil + rendered_sil) > 0).float(), out_dilate_ks, "dilate")
    flat = src_img.view(ns, 3, h, w)
    morph_img, thin, ties = make_morph_image(flat, confidant, outpad)
    only_vis_obj = get_vis_f2pts(obj_f2pts, obj_fim, face_k_nearest)
    uv_img = make_uv_img(morph_img.view(1, ns, 3, h, w), obj_f2pts, only_vis_obj, uv_fim, uv_wim)
    input_G_src = torch.cat([morph_img, cond], dim=1).view(1, ns, 6, h, w)
    bg_mask = masks if masks is not None else cond[:, -1:]
    src_bg_mask = morph(bg_mask, bg_ks, "erode")
    input_G_bg = torch.cat([flat * src_bg_mask, src_bg_mask], dim=1).view(1, ns, 4, h, w)[:, [0]]
    return {"uv_img": uv_img, "input_G_bg": input_G_bg, "input_G_src": input_G_src, "morph_img": morph_img,
            "thin_edges": thin, "tie_mask": ties, "confidant_sil": confidant, "outpad_sil": outpad,
            "only_vis_obj_f2pts": only_vis_obj}


# ------------------------------------------------------------------------------------------------ Swapper
PART_IDS = {     # models/flowcomposition.py:23-39
    "head": [0], "torso": [1], "left_leg": [2], "right_leg": [3], "left_arm": [4], "right_arm": [5], "left_foot": [6],
    "right_foot": [7], "left_hand": [8], "right_hand": [9], "facial": [10],
    "upper": [1, 4, 5, 8, 9], "lower": [2, 3, 6, 7], "body": [1, 2, 3, 4, 5, 6, 7, 8, 9], "all": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9],
}


def select_faces_by_part_name(part_faces, nf, swap_parts, primary_ids=0):
    """Swapper.get_selected_info_by_part_name (models/imitator.py:502-546): ``part_faces`` = list of face-id lists in the
    renderer's body-part order; faces no person selected join the primary person's."""
    selected, union = [], set()
    for parts in swap_parts:
        fids = set()
        for name in parts:
            for i in PART_IDS[name]:
                fids |= set(int(f) for f in part_faces[i])
        union |= fids
        selected.append(fids)
    left = set(range(nf)) - union
    if left:
        selected[primary_ids] |= left
    return [sorted(s) for s in selected]


def get_selected_f2pts(f2pts, selected_fids):
    """nmr.py:601-637: (n, nf, 3, 2) with the faces outside ``selected_fids[i]`` set to -2."""
    out = torch.full_like(f2pts, -2.0)
    for i in range(f2pts.shape[0]):
        ids = torch.as_tensor(selected_fids[i], dtype=torch.long)
        out[i, ids] = f2pts[i, ids]
    return out


def merge_uv_img(uv_imgs, selected_obj_f2pts_first, uv_fim, uv_wim):
    """FlowCompositionForSwapper.merge_uv_img (models/flowcomposition.py:816-856): uv_imgs [(1,3,h,w)] per person,
    selected_obj_f2pts_first [(1,nf,3,2)] = each person's first source; uv_fim / uv_wim (1,h,w) / (1,h,w,3)."""
    h, w = uv_fim.shape[-2:]
    one = torch.ones(1, 1, h, w)
    vis = [F.grid_sample(one, cal_bc_transform(f, uv_fim[0:1], uv_wim[0:1]), mode="bilinear", padding_mode="zeros", align_corners=False)
           for f in selected_obj_f2pts_first]
    imgs, vis = torch.cat(uv_imgs, dim=0), torch.cat(vis, dim=0)
    norm = vis / (vis.sum(dim=0, keepdim=True) + 1e-7)
    return (imgs * norm).sum(dim=0, keepdim=True)


# ------------------------------------------------------------------------------------------------ discriminators (row a16)
def patch_discriminator(sd, prefix, x, n_layers):
    """models/networks/discriminators/patch_dis.py:8-70 (norm_type = "instance", no sigmoid) from a state_dict:
    ``{prefix}model.{0, 2, 5, ...}``: n_layers 4x4 stride-2 convs then two of stride 1, InstanceNorm + LeakyReLU(0.2) between them."""
    idx = [0] + [2 + 3 * (n - 1) for n in range(1, n_layers + 1)]
    idx.append(idx[-1] + 3)
    for i, k in enumerate(idx):
        x = F.conv2d(x, sd[f"{prefix}model.{k}.weight"], sd[f"{prefix}model.{k}.bias"], stride=2 if i < n_layers else 1, padding=1)
        if i == len(idx) - 1:
            break
        if i > 0:
            x = F.instance_norm(x, eps=1e-5)
        x = F.leaky_relu(x, 0.2)
    return x


def crop_img(imgs, rects, fact=2):
    """multi_scale_dis.py:21-44: boxes (min_x, max_x, min_y, max_y) cropped and resized to (H / fact, W / fact); degenerate ones dropped."""
    H, W = imgs.shape[-2:]
    crops = [F.interpolate(imgs[i:i + 1, :, y0:y1, x0:x1], size=(H // fact, W // fact), mode="bilinear", align_corners=True)
             for i, (x0, x1, y0, y1) in enumerate(torch.as_tensor(rects).tolist()) if x0 != x1 and y0 != y1]
    return torch.cat(crops, dim=0) if crops else []


def discriminator_forward(name, sd, x, bg_x, body_rects, head_rects, n_layers, use_aug_bg):
    """multi_scale_dis.py:82-107 (patch_global: [global, bg]), :152-191 (patch_global_local: [bg, global, local]) and :236-284
    (patch_global_body_head: [bg, global, body, head]) -> (outs, reduce_tensor(outs) :9-18)."""
    pd = lambda p, t: patch_discriminator(sd, p, t, n_layers)                                          # noqa: E731
    outs = [pd("global_model.", x)]
    if bg_x is not None and use_aug_bg:
        outs = outs + [pd("bg_model.", bg_x)] if name == "patch_global" else [pd("bg_model.", bg_x)] + outs
    crops = {"patch_global": (), "patch_global_local": (("local_model.", body_rects, 2),),
             "patch_global_body_head": (("body_model.", body_rects, 2), ("head_model.", head_rects, 4))}[name]
    for prefix, rects, fact in crops:
        c = crop_img(x, rects, fact)
        if len(c) != 0:
            outs.append(pd(prefix, c))
    return outs, sum(o.mean() for o in outs) / len(outs)


# ------------------------------------------------------------------------------------------------ SMPL (24 joints), trainers
def batch_rodrigues(theta):
    """bodynets/batch_smpl.py:73-109: R = cos*I + (1 - cos) r r^T + sin*skew(r), angle = |theta + 1e-8|, r = theta / angle."""
    angle = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    r = theta / angle
    cos, sin = torch.cos(angle).unsqueeze(-1), torch.sin(angle).unsqueeze(-1)
    rx, ry, rz = r[:, 0], r[:, 1], r[:, 2]
    zero = torch.zeros_like(rx)
    skew = torch.stack([zero, -rz, ry, rz, zero, -rx, -ry, rx, zero], dim=1).view(-1, 3, 3)
    outer = r.unsqueeze(2) * r.unsqueeze(1)
    return cos * torch.eye(3).unsqueeze(0) + (1 - cos) * outer + sin * skew


def smpl24_get_details(model, theta, offsets=0):
    """SMPL.forward + BaseSMPL.get_details (batch_smpl.py:332-436, base_smpl.py:107-142) for a parameter dict as in
    ipercore_amd.synthetic.smpl_model_dict: verts, 19 COCO+ joints regressed from the posed verts, their projection."""
    dense = lambda a: np.asarray(a.todense()) if hasattr(a, "todense") else np.asarray(a)      # noqa: E731
    f = lambda a: torch.tensor(np.ascontiguousarray(dense(a), dtype=np.float32))               # noqa: E731
    theta = torch.as_tensor(theta, dtype=torch.float32)
    cam, pose, beta = theta[:, 0:3], theta[:, 3:-10], theta[:, -10:]
    N = theta.shape[0]
    v_template, shapedirs = f(model["v_template"]), f(model["shapedirs"]).reshape(-1, 10).t()
    Jr, posedirs = f(model["J_regressor"]).t(), f(model["posedirs"]).reshape(-1, 207).t()
    weights, coco = f(model["weights"]), f(model["cocoplus_regressor"]).t()
    parents = dense(model["kintree_table"])[0].astype(np.int64)
    v_shaped = (beta @ shapedirs).view(N, -1, 3) + v_template + offsets
    J = torch.stack([v_shaped[:, :, k] @ Jr for k in range(3)], dim=2)
    Rs = batch_rodrigues(pose.reshape(-1, 3)).view(N, 24, 3, 3)
    v_posed = ((Rs[:, 1:] - torch.eye(3)).reshape(N, 207) @ posedirs).view(N, -1, 3) + v_shaped

    def make_A(R, t):
        return torch.cat([torch.cat([R, t.unsqueeze(-1)], dim=2), torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(N, 1, 4)], dim=1)
    results = [make_A(Rs[:, 0], J[:, 0])]
    for i in range(1, 24):
        results.append(results[parents[i]] @ make_A(Rs[:, i], J[:, i] - J[:, parents[i]]))
    G = torch.stack(results, dim=1)                                                # (N, 24, 4, 4)
    Jh = torch.cat([J, torch.zeros(N, 24, 1)], dim=2).unsqueeze(-1)
    A = G - F.pad(G @ Jh, (3, 0))
    T = (weights @ A.view(N, 24, 16)).view(N, -1, 4, 4)
    verts = (T @ torch.cat([v_posed, torch.ones(N, v_posed.shape[1], 1)], dim=2).unsqueeze(-1))[:, :, :3, 0]
    joints = torch.stack([verts[:, :, k] @ coco for k in range(3)], dim=2)
    j2d = cam[:, None, 0:1] * (joints[:, :, :2] + cam[:, None, 1:])
    return {"theta": theta, "cam": cam, "pose": pose, "shape": beta, "verts": verts, "j3d": joints, "j2d": j2d}


# ------------------------------------------------------------------------------------------------ textured rendering
# PARITY UNPINNED: neural_renderer (requirements/build.txt: iPERDance/neural_renderer@e5f54f7) is not vendored with the
# reference and the reference holds no output of nr.rasterize / nr.lighting.  These restate the package's published
# algorithm (forward_texture_sampling, lighting) as SMPLRenderer.render calls them (renders/nmr.py:271-290); the image is taken
# on the pixel grid of rasterize_face_index_map for the same faces (render() hands both the same pre-flipped vertices).
def nr_lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1), color_directional=(1, 1, 1),
                direction=(0, 1, 0)):
    bs, nf = faces.shape[:2]
    light = torch.zeros(bs, nf, 3)
    if intensity_ambient != 0:
        light = light + intensity_ambient * torch.tensor(color_ambient, dtype=torch.float32)[None, None, :]
    if intensity_directional != 0:
        f = faces.reshape(bs * nf, 3, 3)
        n = F.normalize(torch.cross(f[:, 0] - f[:, 1], f[:, 2] - f[:, 1], dim=1), eps=1e-5).reshape(bs, nf, 3)
        cos = torch.relu((n * torch.tensor(direction, dtype=torch.float32)[None, None, :]).sum(dim=2))
        light = light + intensity_directional * torch.tensor(color_directional, dtype=torch.float32)[None, None, :] * cos[:, :, None]
    return textures * light[:, :, None, None, None, :]


def texture_sample(fim, wim, faces_v, textures, eps=1e-3, background_color=(0.0, 0.0, 0.0)):
    """fim (B,S,S) int, wim (B,S,S,3), faces_v (B,nf,3,3), textures (B,nf,T,T,T,3) -> rgb (B,S,S,3)."""
    B, S = fim.shape[:2]
    T = textures.shape[2]
    out = torch.tensor(background_color, dtype=torch.float32).expand(B, S, S, 3).clone()
    for b in range(B):
        m = fim[b] >= 0
        fi = fim[b][m].long()
        w = wim[b][m]                                                    # (P,3)
        z = faces_v[b, fi][:, :, 2]                                      # (P,3)
        zp = 1.0 / (w / z).sum(dim=1, keepdim=True)
        t = (w * (T - 1) * (zp / z)).clamp(0.0, (T - 1) - eps)
        ti = t.to(torch.int64)
        tf = t - ti.to(torch.float32)
        tex = textures[b, fi].reshape(fi.shape[0], T * T * T, 3)
        acc = torch.zeros(fi.shape[0], 3)
        for pn in range(8):
            ww = torch.ones(fi.shape[0])
            idx = torch.zeros(fi.shape[0], dtype=torch.int64)
            for k in range(3):
                hi = (pn >> k) & 1
                ww = ww * (tf[:, k] if hi else 1.0 - tf[:, k])
                idx = idx * T + (ti[:, k] + hi).clamp_max(T - 1)
            acc = acc + ww[:, None] * tex[torch.arange(fi.shape[0]), idx]
        out[b][m] = acc
    return out


def nr_rasterize(faces_v, textures, image_size, anti_aliasing=True, near=0.1, far=100.0, eps=1e-3, background_color=(0.0, 0.0, 0.0)):
    """-> images (B,3,S,S)."""
    S = image_size * 2 if anti_aliasing else image_size
    fim, wim = rasterize_fim_wim(faces_v.numpy() if torch.is_tensor(faces_v) else faces_v, S, near, far)
    rgb = texture_sample(fim, wim, torch.as_tensor(faces_v), textures, eps, background_color).permute(0, 3, 1, 2)
    return F.avg_pool2d(rgb, 2) if anti_aliasing else rgb
