"""Bit-exact pin of the host-side topology tables (ipercore_amd/geometry/mesh.py) against hashes of the
buffers built by the reference's own SMPLRenderer.__init__ (renders/nmr.py:128-225)."""
import hashlib

import numpy as np

from ipercore_amd.geometry import mesh


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build_tables(topo, top_k=3, tex_size=3):
    uv, fim = mesh.obj_from_topology(topo, "uv"), mesh.obj_from_topology(topo, "fim")
    parts = mesh.get_part_ids(13776, {str(n): topo["part_" + str(n)] for n in topo["part_names"]})
    f_img2uvs = mesh.get_f2vts(fim, z=1)
    return {
        "smpl_faces": topo["faces_uv"].astype(np.int32),
        "obj_faces": topo["faces_fim"].astype(np.int32),
        "map_fn": mesh.create_mapping("uv_seg", fim, contain_bg=True).astype(np.float32),
        "front_map_fn": mesh.create_mapping("head", fim, head_path=topo["head"], contain_bg=True).astype(np.float32),
        "f_img2uvs": f_img2uvs.astype(np.float32),
        "face_k_nearest": mesh.find_part_k_nearest_faces(f_img2uvs, parts, k=top_k).astype(np.int64),
        "f_uvs2img": mesh.get_f2vts(uv, z=1)[:, :, 0:2].astype(np.float32),
        "img2uv_sampler": mesh.create_uvsampler(uv, tex_size=tex_size).astype(np.float32),
    }


def test_tables_bit_exact(golden, topo):
    t = build_tables(topo)
    for name, arr in t.items():
        assert tuple(arr.shape) == tuple(golden["table_shape/" + name]), name
        assert str(arr.dtype) == str(golden["table_dtype/" + name]), name
        assert sha(arr) == str(golden["table_sha/" + name]), name
    assert np.array_equal(t["map_fn"][[0, 1, 2, 13775, 13776]], golden["table_head/map_fn"])
    assert np.array_equal(t["map_fn"][-1], np.array([0, 0, 1], np.float32))
    assert np.array_equal(t["smpl_faces"][0:3], [[1, 2, 0], [0, 2, 3], [2, 1, 4]])


def test_topology_counts(topo):
    # in-repo data that pin indexing
    assert topo["v"].shape == (6890, 3) and topo["vt"].shape == (7576, 2)
    assert topo["faces_uv"].shape == (13776, 3) and topo["faces_vts_fim"].max() == 7575
    assert len(topo["front_body"]) == 2783 and len(topo["front_facial"]) == 1324 and len(topo["head"]) == 2620
    swapped = (topo["faces_uv"] != topo["faces_fim"]).any(axis=1).sum()
    assert swapped == 872
