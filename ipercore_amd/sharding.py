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


def sharded_synthesize(imitator, tgt_smpls, cam_strategy="smooth", gather=True, group=None):
    """Every rank: prepare the whole sequence, synthesize its block, all-gather the (N,3,S,S) video tensor."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    tgt = imitator.prepare_sequence(tgt_smpls, cam_strategy)
    lo, hi = shard_range(tgt.shape[0], rank, world)
    local = imitator.synthesize(tgt[lo:hi], cam_strategy, t0=lo)
    return all_gather_frames(local, tgt.shape[0], group) if gather else local
