"""Host-side SMPL mesh topology tables (numpy; built once per process, never per frame).

Mirrors the table builders of the reference's ``iPERCore/tools/utils/geometry/mesh.py``
(``load_obj`` :50-106, ``create_uvsampler`` :185-224, ``compute_barycenter`` :227-244,
``get_f2vts`` :246-271, ``cal_face_k_nearest``/``find_part_k_nearest_faces`` :274-320,
``get_part_ids`` :356-377, ``front_mapping`` :431-452, ``create_mapping`` :477-540).

These tables are *indices and small fp32 constants*; the HIP kernels consume them as
device buffers.  Arithmetic is kept in the reference's fp32 evaluation order so the
tables are bit-identical (pinned by ``tests/test_mesh_tables.py`` against hashes taken
from the reference's own functions).
"""
import itertools
import json
import os

import numpy as np

_TOPOLOGY_NPZ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                             "assets", "smpl_topology.npz")


class ObjMesh(dict):
    """Dict with the reference's ``load_obj`` keys: vertices, faces, vts, vns, faces_vts, faces_vns."""


def load_obj(obj_file):
    """Parse the ``v``/``vt``/``vn``/``f a/b/c`` records of a Wavefront OBJ (reference mesh.py:50-106).

    Indices are converted to 0-based int32.  Records other than v/vt/vn/f are ignored.
    """
    rec = {"v": [], "vt": [], "vn": []}
    tri_v, tri_vt, tri_vn = [], [], []
    with open(obj_file, "r") as fp:
        for raw in fp:
            tok = raw.split()
            if not tok:
                continue
            key = tok[0]
            if key == "v" or key == "vn":
                rec[key].append((tok[1], tok[2], tok[3]))
            elif key == "vt":
                rec[key].append((tok[1], tok[2]))
            elif key == "f":
                corners = [c.split("/") for c in tok[1:4]]
                tri_v.append([c[0] for c in corners])
                if len(corners[0]) > 1:
                    tri_vt.append([c[1] for c in corners])
                    tri_vn.append([c[2] for c in corners])

    def _idx(rows):
        if not rows:
            return np.zeros((0, 3), dtype=np.int32)
        return np.asarray(rows, dtype=np.int32) - 1

    return ObjMesh(
        vertices=np.asarray(rec["v"], dtype=np.float32).reshape(-1, 3),
        faces=_idx(tri_v),
        vts=np.asarray(rec["vt"], dtype=np.float32).reshape(-1, 2),
        vns=np.asarray(rec["vn"], dtype=np.float32).reshape(-1, 3),
        faces_vts=_idx(tri_vt),
        faces_vns=_idx(tri_vn),
    )


def load_topology(npz_path=None):
    """Load the packed topology asset (``tools/make_topology_asset.py``): both OBJ meshes + part lists."""
    path = npz_path or _TOPOLOGY_NPZ
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} missing - run `python tools/make_topology_asset.py` where the iPERCore "
            "assets/configs/pose3d files are available")
    z = np.load(path, allow_pickle=False)
    out = {k: z[k] for k in z.files}
    return out


def obj_from_topology(topo, which):
    """Build an ObjMesh ('uv' = mapper_uv.txt, 'fim' = mapper_fim_enc.txt) from the packed asset."""
    assert which in ("uv", "fim")
    return ObjMesh(vertices=topo["v"].astype(np.float32), faces=topo[f"faces_{which}"].astype(np.int32),
                   vts=topo["vt"].astype(np.float32), vns=np.zeros((0, 3), np.float32),
                   faces_vts=topo[f"faces_vts_{which}"].astype(np.int32),
                   faces_vns=np.zeros((0, 3), np.int32))


def _as_obj(path_or_obj):
    return load_obj(path_or_obj) if isinstance(path_or_obj, str) else path_or_obj


def _double_sided(faces):
    return np.concatenate((faces, faces[:, ::-1]), axis=0)


def get_f2vts(uv_map_path_or_obj_info, fill_back=False, z=1):
    """Per-face UV-atlas corner coordinates in [-1,1]^2 plus a constant z: (F,3,3) (mesh.py:246-271).

    v' = 1 - v (image rows grow downwards), then (u,v') * 2 - 1, all in fp32.
    """
    obj = _as_obj(uv_map_path_or_obj_info)
    uv = np.array(obj["vts"], dtype=np.float32, copy=True)
    uv[:, 1] = 1 - uv[:, 1]
    uv = uv * 2 - 1
    uvz = np.concatenate([uv, np.zeros((uv.shape[0], 1), dtype=np.float32) + z], axis=-1)
    tri = obj["faces_vts"]
    if fill_back:
        tri = _double_sided(tri)
    return uvz[tri]


def compute_barycenter(f2vts):
    """(F,3,C) -> (F,C): v2 + 0.5 (v0-v2) + 0.5 (v1-v2)  (mesh.py:227-244; NOT the centroid)."""
    c = f2vts[:, 2]
    return c + 0.5 * (f2vts[:, 0] - c) + 0.5 * (f2vts[:, 1] - c)


def create_uvsampler(uv_mapping_path="data/uv_mappings.txt", tex_size=2, fill_back=False):
    """(F, T*T, 2) texel sample positions in [-1,1] per face (mesh.py:185-224)."""
    ab = np.arange(tex_size, dtype=np.float32) / (tex_size - 1)
    grid = np.stack([p for p in itertools.product(*[ab, ab])])          # (T*T, 2)
    obj = _as_obj(uv_mapping_path)
    uv = np.array(obj["vts"], dtype=np.float32, copy=True)
    uv[:, 1] = 1 - uv[:, 1]
    tri = obj["faces_vts"]
    if fill_back:
        tri = _double_sided(tri)
    tri_uv = uv[tri]
    origin = tri_uv[:, 2]
    e0 = tri_uv[:, 0] - tri_uv[:, 2]
    e1 = tri_uv[:, 1] - tri_uv[:, 2]
    pts = np.dstack([e0, e1]).dot(grid.T) + origin.reshape(-1, 2, 1)      # (F, 2, T*T)
    pts = np.clip(pts, a_min=0.0, a_max=1.0)
    return np.transpose(pts, (0, 2, 1)) * 2 - 1


def cal_face_k_nearest(fbc, nearest_k=10):
    """Indices of the k nearest barycentres (squared L2, expanded form, fp64 accumulator; mesh.py:274-295)."""
    n = fbc.shape[0]
    sq = np.sum(fbc ** 2, axis=1)
    d = np.zeros((n, n))
    d += np.reshape(sq, (1, n))
    d += np.reshape(sq, (n, 1))
    d -= 2 * np.dot(fbc, fbc.T)
    return np.argsort(d, axis=-1)[:, 0:nearest_k]


def find_part_k_nearest_faces(f2vts, parts, k=20):
    """(F,k) int64: for every face its k nearest faces *within the same body part* (mesh.py:298-320)."""
    fbc = compute_barycenter(f2vts)
    out = np.empty((fbc.shape[0], k), dtype=np.int64)
    for _, ids in parts.items():
        local = cal_face_k_nearest(fbc[ids], nearest_k=k)
        out[ids, :] = np.asarray(ids, dtype=np.int64)[local]
    return out


def _read_face_list(path_or_list):
    if isinstance(path_or_list, str):
        with open(path_or_list, "r") as fp:
            return list(json.load(fp)["face"])
    return [int(i) for i in path_or_list]


def get_part_ids(nf, part_info, fill_back=False):
    """{part_name: [face ids]} in sorted-name order; asserts the parts cover all faces (mesh.py:356-377).

    ``part_info`` is the JSON path or an already-parsed ``{name: {"face": [...]}}`` / ``{name: ids}`` dict.
    """
    if isinstance(part_info, str):
        with open(part_info, "r") as fp:
            data = json.load(fp)
    else:
        data = part_info
    half = nf // 2
    parts, seen = {}, set()
    for name in sorted(data.keys()):
        val = data[name]
        ids = list(val["face"]) if isinstance(val, dict) else [int(i) for i in val]
        if fill_back:
            ids = ids + [f + half for f in ids]
        parts[name] = ids
        seen |= set(ids)
    assert len(seen) == nf, "nf_counter = {}, nf = {}".format(len(seen), nf)
    return parts


def front_mapping(nf, front_face_info, fill_back=False):
    """(nf,1) indicator of the listed faces and a zero bg row (mesh.py:431-452)."""
    ids = _read_face_list(front_face_info)
    if fill_back:
        ids = ids + [f + nf // 2 for f in ids]
    table = np.zeros((nf, 1), dtype=np.float32)
    table[ids] = 1.0
    return table, np.zeros((1, 1), dtype=np.float32)


def par_mapping(nf, part_info, fill_back=False):
    parts = get_part_ids(nf, part_info, fill_back=fill_back)
    ndim = len(parts) + 1
    table = np.zeros((nf, ndim), dtype=np.float32)
    for i, name in enumerate(parts.keys()):
        table[parts[name], i] = 1.0
    bg = np.zeros((1, ndim), dtype=np.float32)
    bg[0, -1] = 1
    return table, bg


def create_mapping(map_name, obj_info,
                   part_path="assets/configs/pose3d/smpl_part_info.json",
                   front_path="assets/configs/pose3d/front_body.json",
                   facial_path="assets/configs/pose3d/front_facial.json",
                   head_path="assets/configs/pose3d/head.json",
                   contain_bg=True, fill_back=False):
    """Face-id -> code table, background row appended LAST (so fim == -1 indexes it) (mesh.py:477-540)."""
    f2vts = get_f2vts(obj_info, fill_back=fill_back, z=0)
    nf = f2vts.shape[0]
    if map_name == "uv":
        table, bg = compute_barycenter(f2vts)[:, 0:2], np.array([[-1, -1]], dtype=np.float32)
    elif map_name == "seg":
        table, bg = np.ones((nf, 1), dtype=np.float32), np.array([[0]], dtype=np.float32)
    elif map_name == "uv_seg":
        table, bg = compute_barycenter(f2vts), np.array([[0, 0, 1]], dtype=np.float32)
    elif map_name == "par":
        table, bg = par_mapping(nf, part_path, fill_back=fill_back)
    elif map_name == "front":
        table, bg = front_mapping(nf, front_path, fill_back=fill_back)
    elif map_name == "facial":
        table, bg = front_mapping(nf, facial_path, fill_back=fill_back)
    elif map_name == "head":
        table, bg = front_mapping(nf, head_path, fill_back=fill_back)
    elif map_name == "ids":
        table, bg = np.arange(0, 1, 1 / nf, dtype=np.float32), np.array([[-1]], dtype=np.float32)
    else:
        raise ValueError("map name error {}".format(map_name))
    if contain_bg:
        table = np.concatenate([table, bg], axis=0)
    return table
