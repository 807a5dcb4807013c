"""Weak-perspective camera handling of the runner (reference tools/utils/geometry/cam_pose_utils.py:8-208).

``cam_swap`` runs per frame batch (tiny elementwise work on (B,3) rows, done with tensor ops on the device);
``stabilize`` is the sequence-global pre-pass of ``Imitator.inference`` (imitator.py:339): it needs the foot
height of EVERY frame, which comes from the batched HIP skinning, followed by the jump detection on the host
(pure Python over per-frame scalars, as in the reference).
"""
import numpy as np
import torch


class WeakPerspectiveCamera(object):
    def __init__(self, smpl):
        self.smpl = smpl
        self.infer_smpl_batch_size = 50
        self.jump_up_threshold = 0.2
        self.jump_down_threshold = 0.1

    @staticmethod
    def cam_swap(src_cam, ref_cam, first_cam=None, strategy="smooth"):
        """cam_pose_utils.py:17-50.  src_cam (1 or B,3), ref_cam (B,3), first_cam (1,3) -> (B,3)."""
        if strategy == "smooth":
            cam = src_cam.expand(ref_cam.shape[0], -1).clone()
            cam[:, 1:] += ref_cam[:, 1:] - first_cam[:, 1:]
            cam[:, 0] = cam[:, 0] * ref_cam[:, 0] / first_cam[:, 0]
        elif strategy == "ref_txty":
            cam = src_cam.expand(ref_cam.shape[0], -1).clone()
            cam[:, 1:] = ref_cam[:, 1:]
        elif strategy == "source":
            cam = src_cam.expand(ref_cam.shape[0], -1)
        else:
            cam = ref_cam
        return cam

    def stabilize(self, smpls):
        """cam_pose_utils.py:52-99: cam -> (1, 0, ground + foot denoise), shape -> shape of frame 0."""
        cam, pose, shape = smpls[:, 0:3], smpls[:, 3:-10], smpls[:, -10:]
        new_cam = torch.zeros_like(cam)
        new_cam[:, 0] = 1
        cam_y = cam[:, 2]
        ground_y = cam_y[0]
        shape = shape[0:1, :].repeat(pose.shape[0], 1)
        foot_y = self.infer_smpl_foot_y(pose, shape)
        final_foot_y = (foot_y + cam_y).detach().cpu().numpy()
        jump_info_list, _ = self.get_jump_mask(final_foot_y)
        new_cam_y = ground_y + (-foot_y + foot_y[0])
        for start_idx, end_idx in jump_info_list:
            jump_part = cam_y[start_idx:end_idx + 1].clone()
            new_cam_y[start_idx:end_idx + 1] = torch.min(jump_part, new_cam_y[start_idx:end_idx + 1])
        new_cam[:, 2] = new_cam_y
        return torch.cat([new_cam, pose, shape], dim=1)

    @torch.no_grad()
    def infer_smpl_foot_y(self, pose, shape):
        """cam_pose_utils.py:101-128: max vertex y per frame (y points down), batches of 50 frames."""
        n, bs = pose.shape[0], self.infer_smpl_batch_size
        out = []
        for i in range(int(np.ceil(n / bs))):
            verts, _, _ = self.smpl(shape[i * bs:(i + 1) * bs].contiguous(), pose[i * bs:(i + 1) * bs].contiguous(),
                                    get_skin=True)
            out.append(verts[:, :, 1].max(dim=1)[0])
        return torch.cat(out, dim=0)

    @staticmethod
    def get_checkpoints(y):
        """Indices where the slope of y changes sign, plus both ends (cam_pose_utils.py:130-153)."""
        n = len(y)
        pts = [0]
        for i in range(1, n - 1):
            if (y[i] - y[i - 1]) * (y[i + 1] - y[i]) < 0:
                pts.append(i)
        pts.append(n - 1)
        return pts

    def get_jump_mask(self, final_foot_y):
        """cam_pose_utils.py:155-208 -> ([(start, end), ...], mask (n,))."""
        n = final_foot_y.shape[0]
        jumps = []
        ground_y = final_foot_y[0]
        pts = self.get_checkpoints(final_foot_y)
        jumping, start = False, None
        for k in range(1, len(pts)):
            cur, prev = pts[k], pts[k - 1]
            y_cur, y_prev = final_foot_y[cur], final_foot_y[prev]
            if y_cur - y_prev < 0 and abs(y_cur - y_prev) > self.jump_up_threshold:
                jumping = True
                start = None
                for f in range(prev, cur):           # a take-off frame above ground level is noise
                    if final_foot_y[f] < ground_y:
                        start = f
                        break
                if start is None:
                    start = prev
            elif jumping:
                if y_cur < final_foot_y[start] and abs(y_cur - final_foot_y[start]) > self.jump_down_threshold:
                    continue
                jumping = False
                jumps.append((start, cur))
                start = None
        if jumping:                                   # the clip ends mid-air
            jumps.append((start, n - 1))
        mask = np.zeros((n,))
        for s, e in jumps:
            mask[s:e + 1] = 1
        return jumps, mask
