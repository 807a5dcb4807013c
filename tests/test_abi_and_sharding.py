"""CPU checks: the C-ABI library loads and exports every symbol include/lwg_hip.h declares (no compute calls),
and the frame-sharding helpers are correct under a world_size-2 gloo group."""
import ctypes
import os
import socket
import subprocess
import sys

import pytest
import torch

from ipercore_amd import _lib, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    names = _lib.header_symbols()
    assert len(names) >= 17 and "lwg_conv2d_nhwc_f32" in names and "lwg_rasterize_fim_wim_f32" in names
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/lwg_hip.h but not exported"
    assert set(_lib._SIGS) == set(names)
    assert _lib.lib().lwg_abi_version() == 10
    assert _lib.lib().lwg_rasterize_ws_bytes(2, 13776, 512) == 2 * 13776 * 88 + 2 * 256 * 4 + 2 * 256 * 13776 * 4   # records + boxes, then 16 x 16 bins of 32 pixels: cursors + worst-case lists


def test_conv_args_struct_layout_matches_header():
    # compile a tiny C program against the header and compare sizeof/offsets with the ctypes mirror
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "lwg_hip.h"
    int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(LwgConvArgs), offsetof(LwgConvArgs, w),
        offsetof(LwgConvArgs, y), offsetof(LwgConvArgs, res), offsetof(LwgConvArgs, dy), offsetof(LwgConvArgs, dx),
        sizeof(LwgWinoDesc), offsetof(LwgWinoDesc, upk), offsetof(LwgWinoDesc, first_block), offsetof(LwgWinoDesc, tap9)); return 0; }
    '''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    A = _lib.LwgConvArgs
    Wd = _lib.LwgWinoDesc
    assert got == [ctypes.sizeof(A), A.w.offset, A.y.offset, A.res.offset, A.dy.offset, A.dx.offset,
                   ctypes.sizeof(Wd), Wd.upk.offset, Wd.first_block.offset, Wd.tap9.offset]


def test_invalid_arguments_are_rejected_on_the_host():
    """Every entry point validates its arguments before any launch and returns hipErrorInvalidValue (1): no GPU is touched,
    so the error contract of include/lwg_hip.h can be checked here.  (The reference raises from torch / its CUDA extension;
    the drop-in surfaces these codes as RuntimeError through _lib.check.)"""
    L = _lib.lib()
    bad = 0xdead0000                     # a non-NULL pointer that is never dereferenced: the shape checks fail first
    a = _lib.LwgConvArgs()
    assert L.lwg_winograd_panel_f32(None, None, 64, 64, None, None) == 1 and L.lwg_winograd_panels_f32(None, 0, 0, None) == 1
    assert L.lwg_crop_resize_bilinear_f32(None, None, None, None, 1, 3, 8, 8, 4, 4, None) == 1
    assert L.lwg_prelu_f32(None, None, None, 4, 64, None, None) == 1 and L.lwg_prelu_f32(bad, bad, None, 4, 6, bad, None) == 1      # C % 4
    assert L.lwg_prelu_bwd_f32(bad, bad, None, 4, 64, bad, None) == 1                                                               # no dy
    assert L.lwg_conv_slice_count(None) == 0
    assert L.lwg_conv2d_winograd4_f32(None, None) == 1 and L.lwg_conv2d_winograd4_f32(ctypes.byref(a), None) == 1
    assert L.lwg_winograd4_panel_f32(None, None, 64, 64, None, None) == 1 and L.lwg_winograd4_panel_f32(bad, bad, 48, 64, (ctypes.c_int * 9)(), None) == 1     # Cin % 32
    assert L.lwg_conv2d_winograd_ws_floats(None) == 0 and L.lwg_conv2d_winograd_f32_ws(None, None, None) == 1
    assert L.lwg_conv2d_nhwc_f32(None, None) == 1
    assert L.lwg_conv2d_nhwc_f32(a, None) == 1                                   # NULL tensors
    a.x0, a.w, a.y = bad, bad, bad
    a.B, a.H, a.W, a.C0, a.OH, a.OW, a.M, a.N, a.YH, a.YW, a.YC = 1, 8, 8, 64, 8, 8, 64, 48, 8, 8, 48
    a.ntaps, a.stride, a.omul = 9, 1, 1
    assert L.lwg_conv2d_nhwc_f32(a, None) == 1                                   # N not a multiple of 64
    a.N = a.YC = 64
    a.ntaps = 99
    assert L.lwg_conv2d_nhwc_f32(a, None) == 1                                   # more taps than LWG_MAX_TAPS
    a.ntaps, a.C0 = 9, 24
    assert L.lwg_conv2d_nhwc_f32_split(a, None) == 1                                                 # Cin % 32 != 0
    a.C0 = 64
    assert L.lwg_conv2d_nhwc_bf16(a, None) == 1                                                      # fp32 tensors handed to the bf16 kernel
    a.xdt = a.ydt = _lib.DT_BF16
    a.C0 = 96
    assert L.lwg_conv2d_nhwc_bf16(a, None) == 1                                                      # Cin % 64 != 0
    a.C0, a.YC = 64, 68
    assert L.lwg_conv2d_nhwc_bf16(a, None) == 1                                                      # YC % 8 != 0 (16-byte bf16 stores)
    a.YC, a.xdt = 64, _lib.DT_F32
    a.epi = 1
    a.res = bad
    assert L.lwg_conv2d_nhwc_f32(a, None) == 1                                                       # fp32 -> bf16 is EPI_NONE only
    a.epi, a.res, a.ydt = 0, None, _lib.DT_F32
    assert L.lwg_lwb_attention_bf16(bad, bad, bad, bad, bad, bad, bad, 1, 2, 8, 8, 32, 8, 0, None) == 1  # C = 32 is fp32-only
    assert L.lwg_instnorm_stats_nhwc_bf16(bad, 1, 64, 96, 1e-5, bad, bad, bad, 4, None) == 1          # C not in {64,128,256}
    assert L.lwg_head_compose_bf16(bad, bad, None, 0, 1, 64, 128, None, bad, None, None) == 1         # C != 64
    assert L.lwg_lwb_attention_f32(bad, bad, bad, bad, bad, bad, bad, 1, 2, 8, 8, 48, 8, 0, None) == 1   # C not in {32,64,128,256}
    assert L.lwg_lwb_attention_bwd_f32(bad, bad, bad, bad, bad, bad, bad, bad, bad, bad, 1, 9, 8, 8, 64, 8, 0, None) == 1   # ns > 8
    assert L.lwg_lwb_fuse_f32(bad, None, None, bad, bad, 1, 2, 8, 8, 64, 8, 0, 1.0, 1.0, None) == 1       # NULL sources
    assert L.lwg_rasterize_fim_wim_f32(bad, 1, 16, 4096, 0.1, 100.0, bad, bad, bad, None) == 1            # S > 2048
    assert L.lwg_head_compose_f32(bad, bad, None, 0, 1, 64, 60, bad, None, None, None) == 1               # C % 8 != 0 / pred without bg
    a.xdt = a.ydt = _lib.DT_F32
    a.C0, a.C1, a.N, a.YC, a.ntaps, a.stride, a.omul, a.epi, a.res = 64, 0, 64, 64, 4, 1, 1, 0, None
    assert L.lwg_conv_transpose4_nhwc_f32(a, None) == 1                                                 # omul != 2: not a parity-(0,0) description
    assert L.lwg_thin_conv_f32(bad, bad, 1, 64, 64, 3, bad, None) == 1                                    # ks not in {5, 7}
    assert L.lwg_thin_conv_f32(bad, bad, 1, 64, 60, 7, bad, None) == 1                                    # C % 8 != 0
    kidx = (ctypes.c_int * 2)(0, 99)
    assert L.lwg_pack_panel_f32(bad, 64, 64, 3, 3, 0, kidx, 2, 64, 64, 64, 64, bad, None) == 1            # tap index outside the kernel
    with pytest.raises(RuntimeError):
        _lib.check(1, "lwg_conv2d_nhwc_f32")


def test_splitk_plan_is_a_host_function():
    """lwg_conv2d_ws_floats (csrc/conv_igemm.hip lwg_conv_split_plan) decides on the host which launches run split-K: only the
    64x64-tile regime (< 300 128x128 tiles and < 512 64x64 tiles), whole 32-channel chunks, >= 8 K-steps per slice, <= 8 slices,
    never a fused-epilogue or small-Cin launch."""
    L = _lib.lib()

    def slices(M, N, Cin, ntaps, epi=0):
        a = _lib.LwgConvArgs()
        a.M, a.N, a.C0, a.ntaps, a.epi = M, N, Cin, ntaps, epi
        n = L.lwg_conv2d_ws_floats(a)
        assert n % (M * N) == 0
        return n // (M * N)
    assert L.lwg_conv2d_ws_floats(None) == 0
    assert slices(1024, 512, 256, 16) == 8            # D 256 -> 512, 4x4 s2 at 64^2
    assert slices(961, 512, 512, 16) == 8             # D 512 -> 512, 4x4 s1 (M = 31^2): 16 chunks -> 2 per slice
    assert slices(4096, 256, 256, 9) == 4             # res block at 64^2: 256 tiles -> 4 slices of 2 chunks
    assert slices(256, 256, 384, 9) == 6              # 12 chunks, capped at 8 slices -> 2 chunks each
    assert slices(512, 128, 64, 9) == 2
    assert slices(64, 256, 256, 1) == 0               # 1x1: 8 K-steps in all -> one slice
    assert slices(16384, 128, 64, 9) == 0             # 512 tiles of 64x64: filled already
    assert slices(65536, 256, 256, 9) == 0            # 128x128-tile regime
    assert slices(1024, 512, 256, 16, epi=1) == 0 and slices(1024, 512, 256, 16, epi=2) == 0      # fused epilogues run whole
    assert slices(4096, 64, 8, 9) == 0                # small-Cin path


def test_ops_refuse_cpu_tensors():
    from ipercore_amd import ops
    with pytest.raises(RuntimeError):
        ops.encode_fim(torch.zeros(1, 8, 8, dtype=torch.int32), torch.zeros(5, 3))


def test_shard_range_partitions():
    for n in (1, 7, 8, 300, 301):
        for w in (1, 2, 4, 8):
            spans = [sharding.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            c = sharding.shard_counts(n, w)
            assert max(c) - min(c) <= 1 and sum(c) == n
    assert sharding.shard_counts(300, 8) == [38, 38, 38, 38, 37, 37, 37, 37]


def test_choose_chunk_plan_ring_model():
    """sharding.choose_chunk_plan (bench.py --chunk-plan auto): the N = 8 clip with the round-5 shard_of_8 time per frame (45.3 ms / 38):
    the fp32 exchange (0.83 GB received per rank) keeps the overlapped 24 + 14 schedule, the uint8 exchange (a quarter of it) takes the shard as
    one chunk; a shard that fits one frame batch has nothing to choose; the plan given to sharded_synthesize is what runs."""
    t_f = 45.3e-3 / 38
    plan, model = sharding.choose_chunk_plan(300, 8, 32, 8, 3 * 512 * 512 * 4, t_f)
    assert plan == [(0, 24), (24, 14)] and model["chunked_plan"] == [24, 14] and model["chunked_s"] < model["one_chunk_s"]
    # the model's arithmetic: one chunk = compute + the whole exchange exposed at (world - 1) blocks over one ~150 GB/s link
    assert abs(model["one_chunk_s"] - (38 * t_f + 7 * 38 * 3 * 512 * 512 * 4 / 150e9)) < 1e-9
    plan8, model8 = sharding.choose_chunk_plan(300, 8, 32, 8, 3 * 512 * 512, t_f)
    assert plan8 == [(0, 38)] and model8["one_chunk_s"] < model8["chunked_s"]
    plan1, _ = sharding.choose_chunk_plan(64, 8, 32, 8, 3 * 512 * 512 * 4, t_f)
    assert plan1 == [(0, 8)]

    class Fake:
        frame_batch = 2

        def prepare_sequence(self, s, cam):
            return torch.as_tensor(s)

        def synthesize(self, chunk, cam, t0=0):
            self.calls.append(int(chunk.shape[0]))
            return chunk[:, None].expand(-1, 3) * 1.0
    f = Fake()
    f.calls = []
    out = sharding.sharded_synthesize(f, torch.arange(5, dtype=torch.float32), plan=[(0, 5)])     # one rank, no group: the plan is moot, one call
    assert out.shape == (5, 3) and f.calls == [5]


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ipercore_amd import sharding
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
n = int(sys.argv[2])
full = torch.arange(n * 3 * 4 * 4, dtype=torch.float32).view(n, 3, 4, 4)
lo, hi = sharding.shard_range(n, rank, world)
out = sharding.all_gather_frames(full[lo:hi].clone(), n)
assert out.shape == full.shape and torch.equal(out, full), (rank, out.shape)
# the "runner" protocol: a stand-in imitator whose frame t is a function of (t, first frame) only
class Fake:
    def prepare_sequence(self, s, cam): return torch.as_tensor(s)
    def synthesize(self, chunk, cam, t0=0): return chunk[:, None, None, None].expand(-1, 3, 2, 2) * 2 + 1
seq = torch.arange(n, dtype=torch.float32)
for overlap in (True, False):
    vid = sharding.sharded_synthesize(Fake(), seq, overlap=overlap)
    assert vid.shape == (n, 3, 2, 2) and torch.equal(vid[:, 0, 0, 0], seq * 2 + 1), (overlap, vid[:, 0, 0, 0])
# chunked overlap with a frame batch that does not divide the shard, and a shard one frame short
class Fake3(Fake):
    frame_batch = 3
vid = sharding.sharded_synthesize(Fake3(), seq, overlap=True)
assert torch.equal(vid[:, 0, 0, 0], seq * 2 + 1)
# an explicit chunk schedule (bench.py --chunk-plan): the shard as ONE chunk, and an uneven two-chunk plan - same video, the plan's chunk count
cap = max(sharding.shard_counts(n, world))
for plan in ([(0, cap)], [(0, 1), (1, cap - 1)]):
    st = {"sync": lambda: None}
    vid = sharding.sharded_synthesize(Fake3(), seq, overlap=True, plan=plan, stats=st)
    assert torch.equal(vid[:, 0, 0, 0], seq * 2 + 1) and st["chunks"] == len(plan) and st["compute_s"] >= 0.0, (plan, st)
og = sharding.OverlappedGather(n)
lo, hi = sharding.shard_range(n, rank, world)
for off in range(0, og.cap, 2):
    m = min(2, og.cap - off)
    og.submit(full[lo + off:min(lo + off + m, hi)].clone(), off, length=m)
assert torch.equal(og.finish(), full)
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


@pytest.mark.parametrize("n", [7, 8])
def test_all_gather_frames_gloo_world2(tmp_path, n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(n)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


_CLIP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ipercore_amd import sharding
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
class Fake:
    image_size = 2
    def __init__(self, fb): self.frame_batch = fb; self.calls = []
    def prepare_sequence(self, s, cam): return torch.as_tensor(s)
    def synthesize(self, chunk, cam, t0=0):
        self.calls.append((int(t0), int(chunk.shape[0])))
        if chunk.shape[0] == 0:
            return torch.empty((0, 3, 2, 2))
        return chunk[:, None, None, None].expand(-1, 3, 2, 2) * 2 + 1
for n, fb in [(int(a), int(b)) for a, b in (c.split(":") for c in sys.argv[2].split(","))]:
    seq = torch.arange(n, dtype=torch.float32)
    lo, hi = sharding.shard_range(n, rank, world)
    for overlap in (True, False):
        im, st = Fake(fb), {}
        vid = sharding.sharded_synthesize(im, seq, overlap=overlap, prepared=True, stats=st)
        assert vid.shape == (n, 3, 2, 2) and torch.equal(vid[:, 0, 0, 0], seq * 2 + 1), (n, fb, overlap, vid[:, 0, 0, 0])
        assert st["shard"] == (lo, hi) and st["world"] == world
        assert sum(k for _, k in im.calls) == hi - lo                      # every frame of the shard rendered exactly once
        if overlap:
            plan = sharding.chunk_plan(n, world, fb)
            assert st["chunks"] == len(plan) == len(im.calls)               # one collective per planned chunk on EVERY rank
            assert st["bytes_received"] == sum(m for _, m in plan) * world * 3 * 2 * 2 * 4
    # the per-chunk transform (the uint8 video): applied before the exchange, also to the all-padding chunks
    u8 = sharding.sharded_synthesize(Fake(fb), seq, prepared=True, post=lambda x: x.permute(0, 2, 3, 1).to(torch.uint8))
    assert u8.dtype == torch.uint8 and u8.shape == (n, 2, 2, 3) and torch.equal(u8[:, 0, 0, 0].float(), (seq * 2 + 1) % 256)
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def _run_gloo(tmp_path, script_text, world, *argv):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(script_text)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()


def test_clip_schedule_gloo_world2_and_3(tmp_path):
    """BASELINE configs[2]'s schedule (reference loop: services/run_imitator.py:19-84 -> Imitator.inference): a 300-frame clip in
    contiguous shards, 8-frame chunks behind the frame loop; plus the degenerate clips - fewer frames than ranks (an empty shard),
    a last chunk that is all padding on the short shards, a frame batch that does not divide the shard."""
    _run_gloo(tmp_path, _CLIP_WORKER, 2, "300:8,7:3,1:8,5:1")
    _run_gloo(tmp_path, _CLIP_WORKER, 3, "300:8,2:8,7:1,301:8")


def test_clip_schedule_gloo_world8_300_frames(tmp_path):
    """The configuration itself: 8 ranks, 300 frames -> shards of 38/38/38/38/37/37/37/37; at frame batch 8 five chunks of
    (8, 8, 8, 8, 6) frames, at bench.py's frame batch 16 three of (16, 16, 6); the last one carries one padding frame on the four
    short shards."""
    assert sharding.shard_counts(300, 8) == [38, 38, 38, 38, 37, 37, 37, 37]
    assert sharding.chunk_plan(300, 8, 8) == [(0, 8), (8, 8), (16, 8), (24, 8), (32, 6)]
    assert sharding.chunk_plan(300, 8, 16) == [(0, 16), (16, 16), (32, 6)]
    assert sharding.chunk_plan(0, 4, 8) == [] and sharding.chunk_plan(3, 8, 8) == [(0, 1)]
    # round-aware plan (what sharded_synthesize uses: 8 frames fill the chip once on the 64x64-feature layers at 512x512): a remainder
    # shorter than one round takes whole rounds from the last full batch - no launch set that never fills the machine
    assert sharding.chunk_plan(300, 8, 32, 8) == [(0, 24), (24, 14)]
    assert sharding.chunk_plan(300, 8, 16, 8) == [(0, 16), (16, 8), (24, 14)]
    assert sharding.chunk_plan(300, 8, 8, 8) == [(0, 8), (8, 8), (16, 8), (24, 8), (32, 6)]          # nothing to give: batches ARE one round
    assert sharding.chunk_plan(300, 4, 32, 8) == [(0, 32), (32, 32), (64, 11)]                          # tail >= one round: untouched
    assert sharding.chunk_plan(300, 2, 32, 8)[-1] == (128, 22)
    for n, w, fb, rf in [(300, 8, 32, 8), (301, 7, 16, 8), (37, 3, 12, 8), (9, 1, 8, 8), (300, 8, 64, 32)]:
        plan = sharding.chunk_plan(n, w, fb, rf)
        assert sum(m for _, m in plan) == max(sharding.shard_counts(n, w)) and all(0 < m <= fb for _, m in plan)
        assert [o for o, _ in plan] == [sum(m for _, m in plan[:i]) for i in range(len(plan))]
    class _Im:
        image_size = 512
    assert sharding.round_frames_of(_Im()) == 8
    _Im.image_size = 1024
    assert sharding.round_frames_of(_Im()) == 2
    _Im.image_size = 256
    assert sharding.round_frames_of(_Im()) == 32
    _run_gloo(tmp_path, _CLIP_WORKER, 8, "300:8,5:8")


_GRAD_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ipercore_amd.trainers import allreduce_grads
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
ps = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2, 2, 2))]
for i, p in enumerate(ps[:2]):
    p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
allreduce_grads(ps)                      # ps[2] has no gradient: skipped
assert torch.allclose(ps[0].grad, torch.full((3, 5), 1.5)) and torch.allclose(ps[1].grad, torch.full((7,), 3.0)) and ps[2].grad is None
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_allreduce_grads_gloo_world2(tmp_path):
    """The personalization step's data-parallel exchange: one flat all-reduce, mean over the ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "g.py"
    script.write_text(_GRAD_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


_FLAT_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from ipercore_amd.trainers import FlatAdam
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(), torch.nn.Linear(32, 4))
unused = torch.nn.Linear(8, 8)                      # never reached by the loss: its range must still be exchanged
mod = torch.nn.ModuleList([net, unused])
opt = FlatAdam(mod, lr=1e-3)
x = torch.randn(5, 16, generator=torch.Generator().manual_seed(10 + rank))
for trial in range(2):                              # the second pass re-arms the same hooks
    opt.zero_grad()
    gs = torch.autograd.grad(net(x).pow(2).sum(), list(net.parameters()))        # this rank's gradient, computed aside
    local = torch.zeros_like(opt.grad)
    off = 0
    for p_ in opt.params:
        if any(p_ is q for q in net.parameters()):
            g_ = gs[[i for i, q in enumerate(net.parameters()) if q is p_][0]]
            local[off:off + p_.numel()] = g_.reshape(-1)
        off += p_.numel()
    opt.arm(None, n_buckets=3)
    net(x).pow(2).sum().backward()
    in_flight = len(opt._issued)                    # ranges handed to the collective DURING backward
    opt.allreduce(None)
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    assert torch.allclose(opt.grad, sum(both) / world, atol=1e-7), (opt.grad - sum(both) / world).abs().max()
    assert opt.overlapped_ranges == 3 and in_flight >= 1, (opt.overlapped_ranges, in_flight)
    assert net[0].weight.grad.data_ptr() == opt.grad.data_ptr()          # still views of the flat buffer
# un-armed: one collective over the whole buffer
opt.zero_grad(); net(x).pow(2).sum().backward(); opt._gather(); local = opt.grad.clone(); opt.allreduce(None)
assert unused.weight.grad is None and net[0].weight.grad.data_ptr() == opt.grad.data_ptr()
both = [torch.zeros_like(local) for _ in range(world)]; dist.all_gather(both, local)
assert torch.allclose(opt.grad, sum(both) / world, atol=1e-7)
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_flat_adam_overlapped_allreduce_gloo_world2(tmp_path):
    """FlatAdam.arm(): the flat gradient buffer goes out in a few large ranges as soon as backward has filled them
    (post-accumulate hooks + async collectives); the result equals the plain mean over the ranks."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "f.py"
    script.write_text(_FLAT_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


_DP_SCHEDULE_WORKER = r"""
import os, sys, torch, torch.distributed as dist
import unittest.mock as um
sys.path.insert(0, sys.argv[1])
from ipercore_amd import synthetic
from ipercore_amd.networks import NetworksFactory, generator_param_shapes
from ipercore_amd.trainers import LWGTrainer, PatchGlobalDiscriminator, TrainOpts
from tests import emu_ops


class _Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


emu_ops.install(_Patch())
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
S_, ns, nf, nres, bgf = 32, 2, [64, 64, 128], 1, [64, 64, 128]
sdn = synthetic.fill_state_dict(generator_param_shapes(nf, nres, bgf), seed=7)
u = lambda shape, seed, name: torch.tensor(synthetic.uniform_image(shape, seed + 1000 * rank, name))      # a different sample per rank
inp = {"input_G_bg": u((1, 1, 4, S_, S_), 10, "bg_inputs"), "input_G_src": u((1, ns, 6, S_, S_), 8, "src_inputs"),
       "input_G_tsf": u((1, 1, 6, S_, S_), 9, "tsf_inputs"), "Tst": u((1, 1, ns, S_, S_, 2), 11, "Tst"),
       "real_src": u((1, ns, 3, S_, S_), 700, "real_src"), "real_tsf": u((1, 1, 3, S_, S_), 701, "real_tsf"),
       "real_bg": u((1, 3, S_, S_), 702, "real_bg"), "body_mask": (u((1, ns + 1, 1, S_, S_), 703, "mask") > 0).float()}


def run(schedule, steps=2):
    G = NetworksFactory.get_by_name("AttLWB-SPADE", cfg=synthetic.gen_cfg(nf, nres, bgf), temporal=False)
    G.load_state_dict({k: torch.tensor(v) for k, v in sdn.items()}, strict=True)
    G.train()
    torch.manual_seed(2)
    D = PatchGlobalDiscriminator(ndf=32, n_layers=3)
    o = TrainOpts.l1_transfer()
    o.use_graph, o.dp_schedule = False, schedule
    tr = LWGTrainer(G, D, opts=o)
    tr.set_input({k: v.clone() for k, v in inp.items()})
    losses = []
    with um.patch.object(torch.Tensor, "is_cuda", new_callable=um.PropertyMock, return_value=True), \
            um.patch.object(torch.cuda, "is_available", return_value=False):
        for _ in range(steps):
            lg, ld = tr.optimize_parameters()
            losses.append((float(lg), float(ld)))
    return tr, losses


tr_h, l_h = run("hooks")             # the eager step: hook-driven range all-reduces during backward
tr_s, l_s = run("segmented")         # the captured step's schedule, its segments as eager closures
assert "segmented" in tr_s.step_mode and tr_s.allreduce_overlap, tr_s.step_mode
for a, b in zip(l_h, l_s):
    assert abs(a[0] - b[0]) <= 1e-5 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-5 * abs(a[1]), (l_h, l_s)
for oa, ob in ((tr_h.optimizer_G, tr_s.optimizer_G), (tr_h.optimizer_D, tr_s.optimizer_D)):
    assert torch.allclose(oa.flat, ob.flat, atol=1e-6), (oa.flat - ob.flat).abs().max()       # same averaged gradients, same updates
    assert oa.t == ob.t == 2
    both = [torch.zeros_like(ob.flat) for _ in range(world)]
    dist.all_gather(both, ob.flat)
    assert torch.equal(both[0], both[1]), "ranks diverged"                                  # every rank applied the SAME mean gradient
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_segmented_dp_schedule_gloo_world2(tmp_path):
    """The data-parallel schedule of the captured personalization step (LWGTrainer._run_dp_schedule: G's gradient all-reduce issued async
    behind D's forward / backward segment, D's behind Adam(G)) with its segments as eager closures over the emulated C ABI, two gloo
    ranks with different samples: same losses and weights as the hook-driven eager step, ranks stay identical.  Reference behaviour
    being matched: DistributedDataParallel's overlapped all-reduce, iPERCore/services/train.py:89-95."""
    _run_gloo(tmp_path, _DP_SCHEDULE_WORKER, 2)


def test_no_wide_buffer_store_with_register_soffset():
    """gfx950 (round 6): a 12- / 16-byte buffer store with a REGISTER in its scalar-offset field followed at once by a VALU write of its
    first data register stored corrupted data now and then - the compiler plans the wait state between the two only for a constant scalar offset.  Every
    kernel source that issues raw buffer stores is compiled to ISA here and must not contain such a store (pass offsets belong in the vector offset)."""
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "ipercore_amd", "csrc")
    srcs = [f for f in sorted(glob.glob(os.path.join(csrc, "*.hip"))) if "raw_buffer_store" in open(f).read()]
    assert srcs, "the Winograd kernels store through buffer instructions"
    pat = re.compile(r"buffer_store_(dwordx[34]|format_xyzw?)\s+v\[[0-9:]+\],\s*[^,]+,\s*s\[[0-9:]+\],\s*s[0-9]+\b")
    with tempfile.TemporaryDirectory() as d:
        for f in srcs:
            out = os.path.join(d, os.path.basename(f) + ".s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + csrc, "-I" + os.path.join(root, "include"),
                                   "-S", "--cuda-device-only", "-o", out, f], stderr=subprocess.DEVNULL)
            text = open(out).read()
            assert "buffer_store_dwordx4" in text, f
            bad = [ln.strip() for ln in text.splitlines() if pat.search(ln)]
            assert not bad, (os.path.basename(f), bad[:3])
