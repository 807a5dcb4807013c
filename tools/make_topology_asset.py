#!/usr/bin/env python3
"""Pack the SMPL mesh topology config files of an iPERCore checkout into one compressed npz.

    python tools/make_topology_asset.py [/path/to/iPERCore/assets/configs/pose3d]

Inputs (text/JSON config assets, read where they lie; nothing is copied verbatim):
    mapper_uv.txt, mapper_fim_enc.txt (Wavefront OBJ), smpl_part_info.json,
    front_body.json, head.json, front_facial.json
Output: ipercore_amd/assets/smpl_topology.npz  (indices int32, coordinates fp32) - what the
GPU box (which has no /root/reference) uses for tests, smoke() and bench.py.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ipercore_amd.geometry.mesh import load_obj  # noqa: E402


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/assets/configs/pose3d"
    uv = load_obj(os.path.join(src, "mapper_uv.txt"))
    fim = load_obj(os.path.join(src, "mapper_fim_enc.txt"))
    assert np.array_equal(uv["vertices"], fim["vertices"]) and np.array_equal(uv["vts"], fim["vts"])
    out = {
        "v": uv["vertices"], "vt": uv["vts"],
        "faces_uv": uv["faces"], "faces_vts_uv": uv["faces_vts"],
        "faces_fim": fim["faces"], "faces_vts_fim": fim["faces_vts"],
    }
    with open(os.path.join(src, "smpl_part_info.json")) as fp:
        parts = json.load(fp)
    names = sorted(parts.keys())
    out["part_names"] = np.array(names)
    for n in names:
        out["part_" + n] = np.asarray(parts[n]["face"], dtype=np.int32)
    for key, fn in (("front_body", "front_body.json"), ("head", "head.json"), ("front_facial", "front_facial.json")):
        with open(os.path.join(src, fn)) as fp:
            out[key] = np.asarray(json.load(fp)["face"], dtype=np.int32)
    dst = os.path.join(ROOT, "ipercore_amd", "assets", "smpl_topology.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: v.shape for k, v in out.items() if k != "part_names"})


if __name__ == "__main__":
    main()
