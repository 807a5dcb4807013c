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


def all_gather_frames(local, n_frames, group=None, force=False):
    """local: (n_local, ...) block of this rank (as given by shard_range) -> (n_frames, ...) on every rank.

    Blocks are padded to the largest shard so a single ``all_gather_into_tensor`` moves everything."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
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
    instead of the whole video (8 ranks x 300 frames at 512^2 fp32: 0.83 GB received per rank, ~8 ms on the per-link-bound ring
    vs ~0.6 ms for one 8-frame chunk).  Every rank must submit the same chunk schedule: offsets 0, m, 2m, .. of its own shard,
    the last chunk of a short shard zero-padded by ``submit`` (an all-padding chunk for a shard that ended earlier)."""

    def __init__(self, n_frames, group=None):
        self.n, self.group = int(n_frames), group
        self.world = dist.get_world_size(group)
        self.counts = shard_counts(self.n, self.world)
        self.starts = [shard_range(self.n, r, self.world)[0] for r in range(self.world)]
        self.cap = max(self.counts)
        self.pending = []
        self.bytes_received = 0           # per rank, all chunks (incl. its own block and the padding)
        self.exposed_s = None             # finish(sync=...): wall time between "compute done" and "video assembled"
        self.compute_done_t = None        # finish(sync=...): perf_counter() when the compute stream had drained

    def submit(self, frames, offset, length=None):
        """frames: this rank's frames [offset, offset + k) of its shard, 0 <= k <= length; ``length`` = the chunk length every rank
        uses (default k): shorter chunks are zero-padded so the collective has one size on all ranks."""
        m = frames.shape[0] if length is None else int(length)
        if frames.shape[0] != m:
            frames = torch.cat([frames, frames.new_zeros((m - frames.shape[0],) + tuple(frames.shape[1:]))], dim=0)
        frames = frames.contiguous()
        out = frames.new_empty((self.world * m,) + tuple(frames.shape[1:]))
        work = dist.all_gather_into_tensor(out, frames, group=self.group, async_op=True)
        self.bytes_received += out.numel() * out.element_size()
        self.pending.append((int(offset), m, out, work, frames))

    def finish(self, sync=None):
        """Wait for every chunk and assemble the (n_frames, ...) video in frame order.  ``sync``: a callable that drains the compute
        stream (torch.cuda.synchronize); given, ``exposed_s`` records how long the gather ran past the end of the computation."""
        import time
        t0 = None
        if sync is not None:
            sync()
            t0 = self.compute_done_t = time.perf_counter()
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
        if sync is not None:
            sync()
            self.exposed_s = time.perf_counter() - t0
        return video


def chunk_plan(n_frames, world_size, frame_batch, round_frames=1):
    """The chunk schedule every rank follows for a clip of ``n_frames``: [(offset in the shard, chunk length)], the same list on
    all ranks (collectives must match), sized by the LONGEST shard; rank r's frames of a chunk are
    [lo_r + off, min(lo_r + off + m, hi_r)) - possibly fewer than m, possibly none.

    Full frame batches first, the remainder last (the last chunk is the one whose exchange is exposed, so it should be the short
    one).  ``round_frames``: the number of frames that fill the chip once on the coarsest layers (8 at 512x512: the 64x64-feature
    layers have 64 128x128 output tiles per frame against 256 CUs x 2 workgroups).  A remainder SHORTER than one such round would be
    a launch set that never fills the machine; when the shard allows it, the last full batch gives up whole rounds so that the tail
    holds at least one: 38 frames at frame batch 32 -> 24 + 14 instead of 32 + 6 (same number of tile rounds in total - ceil(4.75)
    - but no sub-round launch set; the exposed exchange grows from 6 to 14 frames, ~0.3 ms on the uint8 video at 8 ranks).
    Balanced chunks (13 + 13 + 12) were tried in round 2 and dropped: 13 frames are 3.25 rounds."""
    cap = max(shard_counts(n_frames, world_size))
    fb = max(1, int(frame_batch))
    rf = max(1, int(round_frames))
    plan = [(off, min(fb, cap - off)) for off in range(0, cap, fb)]
    if len(plan) >= 2 and plan[-1][1] < rf and fb > rf:
        off, m = plan[-2]
        tail = plan[-1][1]
        give = -(-(rf - tail) // rf) * rf                  # whole rounds moved from the last full batch into the tail
        if m - give >= rf:
            plan[-2] = (off, m - give)
            plan[-1] = (off + m - give, tail + give)
    return plan


def choose_chunk_plan(n_frames, world_size, frame_batch, round_frames, bytes_per_frame, t_frame_s, link_bytes_per_s=150e9, split_penalty=0.022):
    """Pick between the overlapped plan (``chunk_plan`` at ``frame_batch``: every chunk but the last is exchanged behind the next chunk's
    synthesis) and ONE chunk per shard (no launch set is cut: ``split_penalty`` = the measured cost of cutting a 38-frame shard into 24 + 14,
    46.3 vs 45.3 ms in bench.py's shard_of_8 - but the whole exchange is exposed) with a per-link-bound ring model of the all-gather: a rank
    receives (world - 1) blocks of a chunk over its slowest link, ~150 GB/s on xGMI (7 links x ~153 GB/s per GPU, point to point: the guide).
    Returns (plan, model) - model holds both estimates, for the bench line.  An ESTIMATE until an 8-GPU run replaces it."""
    cap = max(shard_counts(n_frames, world_size))
    split = chunk_plan(n_frames, world_size, frame_batch, round_frames)
    one = [(0, cap)]

    def exposed(m):
        return (world_size - 1) * m * bytes_per_frame / link_bytes_per_s
    t_one = cap * t_frame_s + exposed(cap)
    # two in-order queues: the compute stream renders chunk after chunk, RCCL's stream exchanges chunk i once it is rendered and chunk i - 1's
    # exchange is done; the clip is assembled when the last exchange ends
    t_c_end = t_x_end = 0.0
    for _, m in split:
        t_c_end += m * t_frame_s * (1.0 + (split_penalty if len(split) > 1 else 0.0))
        t_x_end = max(t_x_end, t_c_end) + exposed(m)
    t_split = t_x_end
    model = {"one_chunk_s": t_one, "chunked_s": t_split, "chunked_plan": [m for _, m in split], "link_GBps_assumed": link_bytes_per_s / 1e9,
             "bytes_per_frame": bytes_per_frame, "t_frame_s_assumed": t_frame_s}
    return (one if (len(split) > 1 and t_one < t_split) else split), model


def round_frames_of(imitator):
    """Frames per full tile round of the coarsest, (S/8)^2 x 256-channel layers: (S/8)^2 / 128 row tiles x 2 column tiles of 128 x 128
    per frame against 512 workgroup slots (256 CUs x 2) -> 8 at 512x512, 32 at 256x256, 2 at 1024x1024."""
    S = int(getattr(imitator, "image_size", 512))
    return max(1, (512 * 64) // max(1, (S // 8) ** 2))


def sharded_synthesize(imitator, tgt_smpls, cam_strategy="smooth", gather=True, group=None, overlap=True, prepared=False, post=None,
                       stats=None, force_collective=False, plan=None):
    """Every rank: prepare the whole sequence, synthesize its block, all-gather the video tensor
    (``overlap``: chunk by chunk behind the frame loop, see OverlappedGather; False: one collective at the end).
    prepared: ``tgt_smpls`` is already the output of ``imitator.prepare_sequence`` (the sequence-global pre-pass, identical on every
    rank).  post: a per-chunk transform applied before the exchange - ``ops.frames_to_u8`` turns the (n,3,S,S) fp32 video into the
    (n,S,S,3) uint8 one the PNG writer consumes: a quarter of the bytes on the xGMI ring.  stats: a dict that receives this rank's
    shard, the bytes it received and (when ``stats["sync"]`` is a callable) the exposed gather time.  force_collective: issue the
    collectives even in a one-rank group (the RCCL check on a single GPU).  plan: [(offset, length)] chunk schedule, the same on every
    rank (``choose_chunk_plan``); default ``chunk_plan`` at the imitator's frame batch."""
    import time
    t_entry = time.perf_counter()
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    tgt = tgt_smpls if prepared else imitator.prepare_sequence(tgt_smpls, cam_strategy)
    n = tgt.shape[0]
    lo, hi = shard_range(n, rank, world)
    fin = (lambda x: x) if post is None else post
    if stats is not None:
        stats.update(rank=rank, world=world, shard=(lo, hi))
    multi = world > 1 or (force_collective and dist.is_initialized())
    if not (gather and overlap and multi):
        local = fin(imitator.synthesize(tgt[lo:hi], cam_strategy, t0=lo))
        return all_gather_frames(local, n, group, force=force_collective) if gather else local
    og = OverlappedGather(n, group)
    if plan is None:
        plan = chunk_plan(n, world, getattr(imitator, "frame_batch", 8), round_frames_of(imitator))
    for off, m in plan:
        a = min(lo + off, hi)
        b = min(lo + off + m, hi)
        # an empty slice (this shard ended before the longest one, or holds no frame at all) still yields a correctly shaped
        # (0, ...) block from the imitator: the collective then carries padding only for this rank
        og.submit(fin(imitator.synthesize(tgt[a:b], cam_strategy, t0=a)), off, length=m)
    video = og.finish(sync=None if stats is None else stats.get("sync"))
    if stats is not None:
        stats.update(bytes_received=og.bytes_received, exposed_gather_s=og.exposed_s, chunks=len(plan), chunk_lengths=[m for _, m in plan],
                     compute_s=None if og.compute_done_t is None else og.compute_done_t - t_entry)
        stats.pop("sync", None)
    return video
