"""Deterministic synthetic inputs for tests and bench.py.

The licensed SMPL-H pickle, ``smpl_faces.npy`` and the generator checkpoint are not distributable, so every
measurement uses: the T-pose template ``v`` of ``mapper_uv.txt`` as ``v_template``, random SMPL-H blend
tensors of the documented scales, seeded poses around the upright canonical view (global rotation pi about
x, reference ``services/base_runner.py:26``), and seeded generator weights.  Everything derives from
``numpy.random.RandomState`` keyed by (seed, name) so the reference-side golden script, the oracle and the
HIP path all see bit-identical inputs without storing them.
"""
import os
import pickle
import zlib

import numpy as np

from .geometry import mesh

NUM_VERTS = 6890
NUM_FACES = 13776
NUM_JOINTS_SMPLH = 52


def _rs(seed, name):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def _softmax(x, axis):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def smplh_model_dict(seed=0, topo=None):
    """A synthetic SMPL-H parameter dict with the pickle schema the reference reads
    (smplx/body_models.py:200-296, bodynets/batch_smplh.py:105-131)."""
    topo = topo or mesh.load_topology()
    nj = NUM_JOINTS_SMPLH
    parents = np.zeros(nj, dtype=np.int64)
    r = _rs(seed, "parents")
    for j in range(1, nj):
        parents[j] = r.randint(max(0, j - 4), j)          # topologically sorted: parents[j] < j
    kintree = np.stack([parents, np.arange(nj, dtype=np.int64)], axis=0)
    kintree[0, 0] = 4294967295 % (2 ** 31)                # root marker (overwritten by -1 in the loader)
    return {
        "v_template": topo["v"].astype(np.float64),
        "shapedirs": (0.01 * _rs(seed, "shapedirs").standard_normal((NUM_VERTS, 3, 10))),
        "posedirs": (0.001 * _rs(seed, "posedirs").standard_normal((NUM_VERTS, 3, (nj - 1) * 9))),
        "J_regressor": _softmax(_rs(seed, "J_regressor").standard_normal((nj, NUM_VERTS)), axis=1),
        "weights": _softmax(_rs(seed, "weights").standard_normal((NUM_VERTS, nj)), axis=1),
        "kintree_table": kintree,
        "f": topo["faces_uv"].astype(np.uint32),
        "hands_meanl": np.zeros(45), "hands_meanr": np.zeros(45),
        "hands_componentsl": np.eye(45), "hands_componentsr": np.eye(45),
    }


def smpl_model_dict(seed=0, topo=None, sparse=False):
    """A synthetic 24-joint SMPL parameter dict with the pickle schema bodynets/batch_smpl.py:283-330 reads (incl. the 19-point
    ``cocoplus_regressor``); ``sparse``: J_regressor / cocoplus_regressor as scipy CSC matrices, as in smpl_model.pkl."""
    topo = topo or mesh.load_topology()
    nj = 24
    parents = np.zeros(nj, dtype=np.int64)
    r = _rs(seed, "smpl24/parents")
    for j in range(1, nj):
        parents[j] = r.randint(max(0, j - 4), j)
    kintree = np.stack([parents, np.arange(nj, dtype=np.int64)], axis=0)
    kintree[0, 0] = 4294967295 % (2 ** 31)
    J = _softmax(_rs(seed, "smpl24/J_regressor").standard_normal((nj, NUM_VERTS)), axis=1)
    coco = _softmax(_rs(seed, "smpl24/cocoplus").standard_normal((19, NUM_VERTS)), axis=1)
    if sparse:
        import scipy.sparse as sp
        J, coco = sp.csc_matrix(J), sp.csc_matrix(coco)
    return {
        "v_template": topo["v"].astype(np.float64),
        "shapedirs": (0.01 * _rs(seed, "smpl24/shapedirs").standard_normal((NUM_VERTS, 3, 10))),
        "posedirs": (0.001 * _rs(seed, "smpl24/posedirs").standard_normal((NUM_VERTS, 3, (nj - 1) * 9))),
        "J_regressor": J, "cocoplus_regressor": coco,
        "weights": _softmax(_rs(seed, "smpl24/weights").standard_normal((NUM_VERTS, nj)), axis=1),
        "kintree_table": kintree,
        "f": topo["faces_uv"].astype(np.uint32),
    }


def write_smpl_pickle(path, seed=0):
    with open(path, "wb") as fp:
        pickle.dump(smpl_model_dict(seed, sparse=True), fp, protocol=2)
    return path


def write_smplh_pickle(path, seed=0):
    with open(path, "wb") as fp:
        pickle.dump(smplh_model_dict(seed), fp, protocol=2)
    return path


def write_smpl_faces_npy(path, topo=None):
    """``smpl_faces.npy`` stand-in: the faces of mapper_uv.txt (SMPL winding), SURVEY section 0.9."""
    topo = topo or mesh.load_topology()
    np.save(path, topo["faces_uv"].astype(np.int32))
    return path


def _rotvec_compose_x_pi(extra):
    """Rotation vector of R_x(pi) * R(extra) (small extra), via quaternions; returns (3,) float64."""
    qa = np.array([0.0, 1.0, 0.0, 0.0])                                # pi about x
    ang = np.linalg.norm(extra)
    qb = np.array([1.0, 0, 0, 0]) if ang < 1e-12 else np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * extra / ang])
    w = qa[0] * qb[0] - qa[1:] @ qb[1:]
    v = qa[0] * qb[1:] + qb[0] * qa[1:] + np.cross(qa[1:], qb[1:])
    n = np.linalg.norm(v)
    theta = 2 * np.arctan2(n, w)
    return v / n * theta


def smpl_sequence(n_frames, seed=0, pose_dim=72, pose_scale=0.2):
    """(n,3+pose_dim+10) fp32 SMPL params: cam jitter, pi-about-x global rotation, small joint rotations."""
    r = _rs(seed, f"smpls{pose_dim}")
    out = np.zeros((n_frames, 3 + pose_dim + 10), dtype=np.float32)
    for t in range(n_frames):
        out[t, 0] = r.uniform(0.7, 0.9)
        out[t, 1] = r.uniform(-0.1, 0.1)
        out[t, 2] = r.uniform(-0.35, -0.25)
        pose = pose_scale * r.standard_normal(pose_dim)
        pose[0:3] = _rotvec_compose_x_pi(0.3 * pose[0:3])
        out[t, 3:3 + pose_dim] = pose
        out[t, -10:] = 0.5 * r.standard_normal(10)
    return out


def param_array(name, shape, seed=0):
    """Seeded value for one generator tensor: N(0,1)/sqrt(fan_in) for kernels, 0.1*N(0,1) for biases."""
    r = _rs(seed, "param:" + name)
    shape = tuple(int(s) for s in shape)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (r.standard_normal(shape) / np.sqrt(max(fan_in, 1))).astype(np.float32)
    return (0.1 * r.standard_normal(shape)).astype(np.float32)


def fill_state_dict(shapes, seed=0):
    """{name: shape} -> {name: fp32 ndarray}, independent of torch's RNG and of construction order."""
    return {k: param_array(k, shapes[k], seed) for k in sorted(shapes)}


def adversarial_param_array(name, shape, seed=0):
    """A seeded tensor with the statistics a TRAINED checkpoint shows and N(0, 1/fan_in) does not (the published generator weights are
    not distributable): heavy-tailed kernels (Student-t, 3 degrees of freedom), input-channel scales spread over two decades, one dominant
    input channel per layer, a same-sign (DC) component in every filter, and positive biases - so post-ReLU activations carry DC offsets and
    a few large-norm channels.  Same overall gain per layer as ``param_array`` (the network neither dies nor saturates).  Used by the
    Winograd engine's adversarial parity checks (the transforms' cancellation scales with |d| |g| of a patch, not of the result)."""
    r = _rs(seed, "adv:" + name)
    shape = tuple(int(s) for s in shape)
    if len(shape) < 2:
        return (0.25 + 0.25 * r.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    w = r.standard_t(3.0, size=shape) / np.sqrt(3.0)
    if len(shape) == 4 and shape[1] >= 32:
        sc = 10.0 ** r.uniform(-1.0, 1.0, size=shape[1])
        sc[r.randint(shape[1])] *= 8.0
        w = w * (sc / np.sqrt(np.mean(sc ** 2))).reshape(1, -1, 1, 1)
    w = w / np.sqrt(max(fan_in, 1)) + 1.0 / max(fan_in, 1)
    return w.astype(np.float32)


def adversarial_state_dict(shapes, seed=0):
    return {k: adversarial_param_array(k, shapes[k], seed) for k in sorted(shapes)}


def uniform_image(shape, seed, name):
    return _rs(seed, name).uniform(-1.0, 1.0, size=shape).astype(np.float32)


def tmp_asset_dir():
    d = os.environ.get("LWG_TMPDIR") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "lwg_synth")
    os.makedirs(d, exist_ok=True)
    return d


# ---- one synthetic clip: the inputs of bench.py, smoke() and the parity tests ----
class AttrDict(dict):
    __getattr__ = dict.__getitem__


def gen_cfg(num_filters, n_res, bg_filters):
    return AttrDict(name="AttLWB-SPADE",
                    BGNet=AttrDict(norm_type="instance", cond_nc=4, n_res_block=n_res, num_filters=list(bg_filters)),
                    SIDNet=AttrDict(norm_type="None", cond_nc=6, n_res_block=n_res, num_filters=list(num_filters)),
                    TSFNet=AttrDict(norm_type="instance", cond_nc=6, n_res_block=n_res, num_filters=list(num_filters)))


def build_case(image_size=512, num_filters=(64, 128, 256), n_res=6, bg_filters=(64, 128, 128, 256), n_frames=8, ns=2,
               seed=0):
    S = int(image_size)
    from .networks import generator_param_shapes
    shapes = generator_param_shapes(num_filters, n_res, bg_filters)
    case = AttrDict(
        S=S, ns=ns, n_frames=n_frames, num_filters=list(num_filters), n_res=n_res, bg_filters=list(bg_filters),
        smplh=smplh_model_dict(seed=seed),
        state=fill_state_dict(shapes, seed=seed + 7),
        src_smpl=smpl_sequence(ns, seed=seed + 11, pose_dim=72),
        tgt_smpls=smpl_sequence(n_frames, seed=seed + 12, pose_dim=72),
        uv_img=uniform_image((1, 3, S, S), seed + 6, "uv_img"),
        bg_img=uniform_image((1, 3, S, S), seed + 5, "bg_img"),
        src_img=uniform_image((1, ns, 3, S, S), seed + 4, "src_img"),
    )
    case.opt = AttrDict(image_size=S, gen_name="AttLWB-SPADE", temporal=False, only_vis=False, map_name="uv_seg",
                        smpl_model_hand=case.smplh, neural_render_cfg=AttrDict(Generator=gen_cfg(num_filters, n_res, bg_filters)))
    return case


def make_imitator(case, frame_batch=8, device="cuda:0"):
    import torch
    from .imitator import Imitator
    im = Imitator(case.opt, device=torch.device(device), frame_batch=frame_batch)
    im.generator.load_state_dict({k: torch.tensor(v) for k, v in case.state.items()}, strict=True)
    im.generator.to(im.device)
    im.set_source(case.src_smpl, case.uv_img, case.bg_img, src_img=case.src_img)
    return im
