"""Function-level drop-in for the `neural_renderer` calls the reference makes (renders/nmr.py:267-387):

    import ipercore_amd.nr as nr        # or: sys.modules["neural_renderer"] = ipercore_amd.nr

Only the functions on the per-frame path and source_setup are built; the textured /
silhouette / depth renderers raise.  All take and return tensors on the same CUDA device.
"""
import torch

from . import ops


def look_at(vertices, eye, at=None, up=None):
    """nr.look_at as called at nmr.py:285,312,333,383: eye on the -z axis, at = origin, up = +y, for which the
    rotation is the identity and the transform is a translation by -eye."""
    eye_t = torch.as_tensor(eye, dtype=torch.float32, device=vertices.device)
    if float(eye_t[0]) != 0.0 or float(eye_t[1]) != 0.0 or float(eye_t[2]) >= 0.0 or at is not None or up is not None:
        raise NotImplementedError("only the reference's camera (eye = [0, 0, -d], at = 0, up = +y) is supported")
    return vertices - eye_t


def vertices_to_faces(vertices, faces):
    """(bs,nv,3), (bs,nf,3) int -> (bs,nf,3,3)."""
    bs, nv = vertices.shape[:2]
    idx = faces.long() + (torch.arange(bs, device=vertices.device) * nv)[:, None, None]
    return vertices.reshape(bs * nv, 3)[idx]


def rasterize_face_index_map_and_weight_map(faces, image_size, anti_aliasing=False, near=0.1, far=100, eps=1e-3):
    if anti_aliasing:
        raise NotImplementedError("the reference never rasterizes index maps with anti_aliasing (nmr.py:316,337,356)")
    return ops.rasterize_fim_wim(faces.float().contiguous(), image_size, near, far)


def rasterize_face_index_map(faces, image_size, anti_aliasing=False, near=0.1, far=100, eps=1e-3):
    return rasterize_face_index_map_and_weight_map(faces, image_size, anti_aliasing, near, far, eps)[0]


def _next_row(name):
    def f(*a, **k):
        raise NotImplementedError(f"neural_renderer.{name} is not built: not on the Imitator path, and the package that defines it is not vendored (no output to pin it to; DESIGN.md 7)")
    return f


rasterize = _next_row("rasterize")
rasterize_silhouettes = _next_row("rasterize_silhouettes")
rasterize_depth = _next_row("rasterize_depth")
lighting = _next_row("lighting")
