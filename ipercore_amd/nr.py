"""Function-level drop-in for the `neural_renderer` calls the reference makes (renders/nmr.py:267-387):

    import ipercore_amd.nr as nr        # or: sys.modules["neural_renderer"] = ipercore_amd.nr

The functions on the per-frame path and source_setup are built and pinned; ``rasterize`` (textured) and
``lighting`` restate the package's published algorithm without a reference output to pin them to; silhouette / depth raise.  All take and return tensors on the same CUDA device.
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


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1), color_directional=(1, 1, 1),
             direction=(0, 1, 0)):
    """neural_renderer.lighting as called at nmr.py:258-266: per-face light = ambient + directional * relu(n . d), applied to the
    (bs,nf,T,T,T,3) textures.  PARITY UNPINNED (package not vendored): the published formula, normals from cross(v0 - v1, v2 - v1)."""
    bs, nf = faces.shape[:2]
    dev = faces.device
    light = torch.zeros(bs, nf, 3, device=dev)
    col = lambda c: torch.as_tensor(c, dtype=torch.float32, device=dev).reshape(-1, 3).expand(bs, 3) if torch.as_tensor(c).dim() <= 1 \
        else torch.as_tensor(c, dtype=torch.float32, device=dev)                                                   # noqa: E731
    if intensity_ambient != 0:
        light = light + intensity_ambient * col(color_ambient)[:, None, :]
    if intensity_directional != 0:
        f = faces.reshape(bs * nf, 3, 3)
        normals = torch.nn.functional.normalize(torch.cross(f[:, 0] - f[:, 1], f[:, 2] - f[:, 1], dim=1), eps=1e-5).reshape(bs, nf, 3)
        d = col(direction)
        cos = torch.relu((normals * d[:, None, :]).sum(dim=2))
        light = light + intensity_directional * (col(color_directional)[:, None, :] * cos[:, :, None])
    return textures * light[:, :, None, None, None, :]


def rasterize(faces, textures, image_size=256, anti_aliasing=True, near=0.1, far=100, eps=1e-3, background_color=(0, 0, 0)):
    """neural_renderer.rasterize as called at nmr.py:286-287 -> images (bs,3,S,S): index / weight maps at S (2 S with
    anti_aliasing, then a 2x2 average), perspective-correct trilinear texture sampling (csrc/raster.hip lwg_texture_sample_f32).
    PARITY UNPINNED, see include/lwg_hip.h.  The image is on the pixel grid of rasterize_face_index_map for the same faces."""
    S = image_size * 2 if anti_aliasing else image_size
    faces = faces.float().contiguous()
    fim, wim = ops.rasterize_fim_wim(faces, S, near, far)
    rgb = ops.texture_sample(fim, wim, faces, textures.float(), eps=eps, background_color=background_color).permute(0, 3, 1, 2)
    if anti_aliasing:
        rgb = torch.nn.functional.avg_pool2d(rgb, kernel_size=2)
    return rgb.contiguous()


rasterize_silhouettes = _next_row("rasterize_silhouettes")
rasterize_depth = _next_row("rasterize_depth")
