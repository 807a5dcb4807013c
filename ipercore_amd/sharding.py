"""Frame sharding of one source/reference pair over the GPUs of a node.

The reference runs ``Imitator.inference`` on a single GPU (models/imitator.py:131).  With ``temporal=False``
frame t depends only on the cached source state, ``tgt_smpls[t]`` and ``first_cam`` (imitator.py:298-299),
so a clip partitions exactly: every rank runs the sequence-global pre-pass (stabilize) identically, renders a
contiguous block of frames and the output video tensor is assembled with ONE RCCL all-gather over xGMI
(``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).  No collective touches the
data path of a frame.
"""
import torch
import torch.distributed as dist


def shard_range(n_frames, rank, world_size):
    """Contiguous block [lo, hi) of rank ``rank``: the first ``n % world`` ranks get one extra frame."""
    base, extra = divmod(int(n_frames), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_counts(n_frames, world_size):
    return [shard_range(n_frames, r, world_size)[1] - shard_range(n_frames, r, world_size)[0] for r in range(world_size)]


def all_gather_frames(local, n_frames, group=None):
    """local: (n_local, ...) block of this rank (as given by shard_range) -> (n_frames, ...) on every rank.

    Blocks are padded to the largest shard so a single ``all_gather_into_tensor`` moves everything."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    counts = shard_counts(n_frames, world)
    assert local.shape[0] == counts[dist.get_rank(group)], (local.shape, counts)
    cap = max(counts)
    if local.shape[0] != cap:
        pad = local.new_zeros((cap - local.shape[0],) + tuple(local.shape[1:]))
        local = torch.cat([local, pad], dim=0)
    out = local.new_empty((world * cap,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if all(c == cap for c in counts):
        return out
    return torch.cat([out[r * cap:r * cap + counts[r]] for r in range(world)], dim=0)


class OverlappedGather(object):
    """The same all-gather issued chunk by chunk WHILE the frame loop runs: every frame batch is handed to RCCL as soon as it is
    synthesized (``async_op`` collectives run on RCCL's own stream over xGMI), so only the last chunk's transfer is exposed
    instead of the whole video (8 ranks x 160 frames at 512^2: 3.5 GB received per rank, ~30 ms on the per-link-bound ring vs
    ~2 ms for one 8-frame chunk).  Every rank must submit the same chunk schedule: offsets 0, m, 2m, .. of its own shard, the
    last chunk of a short shard zero-padded by ``submit``."""

    def __init__(self, n_frames, group=None):
        self.n, self.group = int(n_frames), group
        self.world = dist.get_world_size(group)
        self.counts = shard_counts(self.n, self.world)
        self.starts = [shard_range(self.n, r, self.world)[0] for r in range(self.world)]
        self.cap = max(self.counts)
        self.pending = []

    def submit(self, frames, offset, length=None):
        """frames: this rank's frames [offset, offset + k) of its shard, k <= length; ``length`` = the chunk length every rank uses
        (default k): shorter chunks are zero-padded so the collective has one size on all ranks."""
        m = frames.shape[0] if length is None else int(length)
        if frames.shape[0] != m:
            frames = torch.cat([frames, frames.new_zeros((m - frames.shape[0],) + tuple(frames.shape[1:]))], dim=0)
        frames = frames.contiguous()
        out = frames.new_empty((self.world * m,) + tuple(frames.shape[1:]))
        work = dist.all_gather_into_tensor(out, frames, group=self.group, async_op=True)
        self.pending.append((int(offset), m, out, work, frames))

    def finish(self):
        """Wait for every chunk and assemble the (n_frames, ...) video in frame order."""
        video = None
        for offset, m, out, work, _ in self.pending:
            work.wait()
            if video is None:
                video = out.new_empty((self.n,) + tuple(out.shape[1:]))
            for r in range(self.world):
                k = min(m, self.counts[r] - offset)
                if k > 0:
                    video[self.starts[r] + offset:self.starts[r] + offset + k] = out[r * m:r * m + k]
        self.pending = []
        return video


def sharded_synthesize(imitator, tgt_smpls, cam_strategy="smooth", gather=True, group=None, overlap=True):
    """Every rank: prepare the whole sequence, synthesize its block, all-gather the (N,3,S,S) video tensor
    (``overlap``: chunk by chunk behind the frame loop, see OverlappedGather; False: one collective at the end)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    tgt = imitator.prepare_sequence(tgt_smpls, cam_strategy)
    n = tgt.shape[0]
    lo, hi = shard_range(n, rank, world)
    if not (gather and overlap and world > 1):
        local = imitator.synthesize(tgt[lo:hi], cam_strategy, t0=lo)
        return all_gather_frames(local, n, group) if gather else local
    og = OverlappedGather(n, group)
    fb = max(1, int(getattr(imitator, "frame_batch", 8)))
    proto = None
    for off in range(0, og.cap, fb):
        m = min(fb, og.cap - off)
        a, b = lo + off, min(lo + off + m, hi)
        if b > a:
            proto = frames = imitator.synthesize(tgt[a:b], cam_strategy, t0=a)
        else:                                           # a shard one frame shorter than the longest: an all-padding last chunk
            frames = proto[:0]
        og.submit(frames, off, length=m)
    return og.finish()
