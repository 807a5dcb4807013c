"""SMPL-H body model behind the reference's ``SMPLH`` API, skinning on the MI355X (csrc/lbs.hip).

Drop-in for ``iPERCore.tools.human_digitalizer.bodynets.SMPLH`` (batch_smplh.py:15-180) with the ``BaseSMPL``
helpers (base_smpl.py:21-142): same constructor ``SMPLH(model_path)``, same pickle schema
(smplx/body_models.py:200-296: v_template, shapedirs, posedirs, J_regressor, kintree_table, weights, f,
hands_meanl/r, hands_componentsl/r), same methods ``forward`` / ``get_details`` / ``skinning`` / ``split`` /
``link`` and the ``np_hands_mean`` property used by ``base_runner.add_hands_params_to_smpl``.
The whole frame batch is skinned by one C-ABI call (``lwg_smpl_lbs_f32``); there is no CPU path.
"""
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import ops


def _dense(a):
    if hasattr(a, "todense"):
        a = a.todense()
    return np.asarray(a)


class SMPLH(nn.Module):
    NUM_BODY_JOINTS = 21
    NUM_HAND_JOINTS = 15
    NUM_JOINTS = NUM_BODY_JOINTS + 2 * NUM_HAND_JOINTS      # 51 + root = 52 rotations

    def __init__(self, model_path, use_pca=False, num_pca_comps=6, dtype=torch.float32, **kwargs):
        super().__init__()
        if isinstance(model_path, dict):
            data = model_path
        else:
            with open(model_path, "rb") as fp:
                data = pickle.load(fp, encoding="latin1")
        self.use_pca = use_pca
        self.num_pca_comps = num_pca_comps
        f32 = lambda a: torch.tensor(np.ascontiguousarray(_dense(a), dtype=np.float32))    # noqa: E731
        self.faces = _dense(data["f"])
        self.register_buffer("faces_tensor", torch.tensor(self.faces.astype(np.int64)))
        self.register_buffer("v_template", f32(data["v_template"]))
        self.register_buffer("shapedirs", f32(data["shapedirs"])[:, :, :10].contiguous())
        pd = _dense(data["posedirs"])
        self.register_buffer("posedirs", f32(np.reshape(pd, [-1, pd.shape[-1]]).T))       # (P, V*3)
        self.register_buffer("J_regressor", f32(data["J_regressor"]))
        parents = torch.tensor(_dense(data["kintree_table"])[0].astype(np.int64))
        parents[0] = -1
        self.register_buffer("parents", parents)
        self.register_buffer("parents_i32", parents.to(torch.int32))
        self.register_buffer("lbs_weights", f32(data["weights"]))
        self.np_hands_meanl = _dense(data["hands_meanl"]).astype(np.float32)
        self.np_hands_meanr = _dense(data["hands_meanr"]).astype(np.float32)
        self.register_buffer("hands_meanl", torch.tensor(self.np_hands_meanl))
        self.register_buffer("hands_meanr", torch.tensor(self.np_hands_meanr))
        self.register_buffer("hands_mean", torch.tensor(self.np_hands_mean))
        self.register_buffer("left_hand_components", f32(_dense(data["hands_componentsl"])[:num_pca_comps]))
        self.register_buffer("right_hand_components", f32(_dense(data["hands_componentsr"])[:num_pca_comps]))

    @property
    def np_hands_mean(self):
        return np.concatenate([self.np_hands_meanl, self.np_hands_meanr], axis=0)

    # ---- BaseSMPL helpers -------------------------------------------------------------------------
    def _device_model(self):
        return {"v_template": self.v_template, "shapedirs": self.shapedirs, "posedirs": self.posedirs,
                "J_regressor": self.J_regressor, "parents": self.parents_i32, "lbs_weights": self.lbs_weights}

    def _full_pose(self, theta):
        bs = theta.shape[0]
        if theta.shape[1] == 72:                                            # batch_smplh.py:156-158
            theta = torch.cat([theta[:, 0:66], self.hands_mean.repeat(bs, 1)], dim=1)
        if self.use_pca:                                                    # batch_smplh.py:160-169
            lh = torch.matmul(theta[:, -12:-6], self.left_hand_components)
            rh = torch.matmul(theta[:, -6:], self.right_hand_components)
            theta = torch.cat([theta[:, :-12], lh, rh], dim=1)
        return theta.contiguous()

    def _links_2d(self, links_ids, device):
        """links_ids (n, 2|3) -> (n, 2) int32 (from, to) pairs shared by the whole batch (base_smpl.py:44-45: every row is
        applied).  A per-batch (B, nv, 3) tensor whose rows are all the same sample - what FlowComposition.forward builds with
        ``links_ids.expand(bs, ns, nv, c)`` (flowcomposition.py:695-701) - reduces to the rows whose has_linked flag is 1
        (base_smpl.py:47-49); genuinely different per-sample links return None (``forward`` then skins row by row)."""
        ids = torch.as_tensor(links_ids)
        if ids.dim() == 3:
            if ids.shape[0] > 1 and not bool((ids == ids[0:1]).all()):
                return None
            ids = ids[0]
            if ids.shape[1] >= 3:
                ids = ids[ids[:, 2] == 1]
        elif ids.dim() != 2:
            raise ValueError(f"links_ids must be (n, 2|3) or (B, n, 3), got {tuple(ids.shape)}")
        return ids[:, 0:2].to(device=device, dtype=torch.int32).contiguous()

    @staticmethod
    def _offsets_rows(offsets, B, device):
        """offsets: 0 / (nv, 3) / (bs, nv, 3) -> None, (nv, 3) or (B, nv, 3) on the device.  The reference adds a (bs, nv, 3)
        tensor to a (bs * ns, nv, 3) template by broadcasting (smplx/lbs.py:176, flowcomposition.py:703): bs = 1 is the dataset
        sample's collated (1, nv, 3); bs > 1 rows are repeated for the ns / nt frames of their sample."""
        if isinstance(offsets, np.ndarray):
            offsets = torch.tensor(offsets)
        if not torch.is_tensor(offsets) or offsets.numel() <= 1:
            return None
        off = offsets.to(device=device, dtype=torch.float32)
        if off.dim() == 3:
            if off.shape[0] == 1:
                off = off[0]
            elif off.shape[0] != B:
                if B % off.shape[0] != 0:
                    raise ValueError(f"offsets batch {off.shape[0]} does not divide the {B} skinned rows")
                off = off.repeat_interleave(B // off.shape[0], dim=0)
        return off.contiguous()

    @torch.no_grad()
    def forward(self, beta, theta, offsets=0, links_ids=None, get_skin=False, cam=None):
        """batch_smplh.py:137-180 -> (vertices (B,6890,3), joints (B,52,3), full_pose)."""
        full_pose = self._full_pose(theta.float())
        B, dev = full_pose.shape[0], full_pose.device
        off = self._offsets_rows(offsets, B, dev)
        beta = beta.float().contiguous()
        cam = None if cam is None else cam.float().contiguous()
        links = None if links_ids is None else self._links_2d(links_ids, dev)
        if links_ids is not None and links is None:
            # per-sample links (B, nv, 3) that really differ: one skinning call per row, each with its own flagged pairs
            ids = torch.as_tensor(links_ids)
            rows = []
            for b in range(B):
                lb = ids[b][ids[b][:, 2] == 1][:, 0:2].to(device=dev, dtype=torch.int32).contiguous()
                ob = off if (off is None or off.dim() == 2) else off[b].contiguous()
                rows.append(ops.smpl_lbs(self._device_model(), full_pose[b:b + 1], beta[b:b + 1], None if cam is None else cam[b:b + 1],
                                         ob, lb if lb.shape[0] else None))
            verts, j3d = torch.cat([r[0] for r in rows]), torch.cat([r[1] for r in rows])
            j2d = None if cam is None else torch.cat([r[2] for r in rows])
        else:
            verts, j3d, j2d = ops.smpl_lbs(self._device_model(), full_pose, beta, cam, off,
                                           links if (links is not None and links.shape[0]) else None)
        self._last_j2d = j2d
        return verts, j3d, full_pose

    def link(self, verts, linked_ids):
        """base_smpl.py:28-50 (2-D ids)."""
        ids = self._links_2d(linked_ids, verts.device)
        out = verts.clone()
        if ids is None:                                                     # per-sample pairs (base_smpl.py:46-49)
            l3 = torch.as_tensor(linked_ids).to(verts.device)
            for b in range(verts.shape[0]):
                sel = l3[b][l3[b][:, 2] == 1].long()
                out[b, sel[:, 0]] = verts[b, sel[:, 1]]
            return out
        ids = ids.long()
        out[:, ids[:, 0]] = verts[:, ids[:, 1]]
        return out

    def split(self, theta):
        return {"cam": theta[:, 0:3], "pose": theta[:, 3:-10].contiguous(), "shape": theta[:, -10:].contiguous(),
                "theta": theta}

    def skinning(self, theta, offsets=0, links_ids=None):
        cam, pose, shape = theta[:, 0:3], theta[:, 3:-10].contiguous(), theta[:, -10:].contiguous()
        verts, _, _ = self.forward(beta=shape, theta=pose, offsets=offsets, links_ids=links_ids, get_skin=True)
        return {"cam": cam, "pose": pose, "shape": shape, "verts": verts, "theta": theta}

    def get_details(self, theta, offsets=0, links_ids=None):
        """base_smpl.py:107-142: verts, posed joints and their weak-perspective projection."""
        cam, pose, shape = theta[:, 0:3], theta[:, 3:-10].contiguous(), theta[:, -10:].contiguous()
        verts, j3d, _ = self.forward(beta=shape, theta=pose, offsets=offsets, links_ids=links_ids, get_skin=True,
                                     cam=cam.contiguous())
        return {"theta": theta, "cam": cam, "pose": pose, "shape": shape, "verts": verts, "j2d": self._last_j2d,
                "j3d": j3d}


class SMPL(SMPLH):
    """The 24-joint SMPL of the reference's trainers (bodynets/batch_smpl.py:283-436, used by FlowCompositionForTrainer,
    tools/trainers/base.py:95-97): same linear blend skinning kernel with nj = 24, theta = 72 axis-angle values, and the 19
    COCO+ keypoints regressed from the POSED vertices (``cocoplus_regressor``) as ``j3d`` / ``j2d``.
    Rotations: the reference builds them with Rodrigues' formula on theta / norm(theta + 1e-8) (batch_smpl.py:73-109); the LBS
    kernel goes through the quaternion of the same axis / angle (the SMPL-H route) - equal up to fp32 rounding (checked against
    the reference class to 1e-5 on the vertices)."""
    NUM_JOINTS = 24

    def __init__(self, model_path, rotate=False, **kwargs):
        nn.Module.__init__(self)
        if rotate:
            raise NotImplementedError("rotate_base (batch_smpl.py:174-187) is not used by the trainers")
        if isinstance(model_path, dict):
            data = model_path
        else:
            with open(model_path, "rb") as fp:
                data = pickle.load(fp, encoding="latin1")
        f32 = lambda a: torch.tensor(np.ascontiguousarray(_dense(a), dtype=np.float32))    # noqa: E731
        self.use_pca = False
        self.faces = _dense(data["f"])
        self.register_buffer("faces_tensor", torch.tensor(self.faces.astype(np.int64)))
        self.register_buffer("v_template", f32(data["v_template"]))
        self.register_buffer("shapedirs", f32(data["shapedirs"])[:, :, :10].contiguous())
        pd = _dense(data["posedirs"])
        self.register_buffer("posedirs", f32(np.reshape(pd, [-1, pd.shape[-1]]).T))
        self.register_buffer("J_regressor", f32(data["J_regressor"]))
        parents = torch.tensor(_dense(data["kintree_table"])[0].astype(np.int64))
        parents[0] = -1
        self.register_buffer("parents", parents)
        self.register_buffer("parents_i32", parents.to(torch.int32))
        self.register_buffer("lbs_weights", f32(data["weights"]))
        self.register_buffer("joint_regressor", f32(data["cocoplus_regressor"]).t().contiguous())      # (6890, 19)

    def _full_pose(self, theta):
        assert theta.shape[1] == 72, "SMPL takes 24 x 3 axis-angle values (quaternion / 6-D / matrix inputs are not built)"
        return theta.contiguous()

    @torch.no_grad()
    def forward(self, beta, theta, offsets=0, links_ids=None, get_skin=False, cam=None):
        """batch_smpl.py:332-436 -> (verts (B,6890,3), COCO+ joints (B,19,3), theta)."""
        verts, _, full_pose = super().forward(beta, theta, offsets=offsets, links_ids=links_ids, get_skin=True, cam=None)
        joints = torch.einsum("bvc,vj->bjc", verts, self.joint_regressor)          # library GEMM: 19 keypoints from the posed mesh
        self._last_j2d = None if cam is None else cam[:, None, 0:1] * (joints[:, :, :2] + cam[:, None, 1:3])     # base_smpl.py:7-18
        return verts, joints, full_pose
