"""bench.py as the driver starts it: ``python bench.py --gpus N`` with no WORLD_SIZE must spawn its own ranks (torch.distributed.run)
and print ONE parsable JSON line from rank 0; under torchrun it must still run in-process.  Exercised on CPU with the gloo backend
and the emulated C ABI (tests/emu_ops.py) installed AROUND bench.main() by a wrapper script - bench.py itself has no CPU path: with
``--device cpu`` and no emulation every op raises.  Reference launcher being matched: scripts/train/dist_train.py:97-109."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WRAPPER = r'''
import sys
sys.path.insert(0, {root!r})
import bench
bench.self_launch_if_needed()            # N > 1 and no WORLD_SIZE: re-executes THIS wrapper under torch.distributed.run
from tests import emu_ops


class _Patch:                             # the two monkeypatch methods emu_ops.install uses
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


emu_ops.install(_Patch())
bench.main()
'''

_ARGS = ["--device", "cpu", "--backend", "gloo", "--tiny-arch", "--size", "64", "--frames", "7", "--frame-batch", "2", "--steps", "1",
         "--warmup", "0"]


def _run(tmp_path, extra, env_extra=None, launcher=None):
    script = tmp_path / "bench_wrapper.py"
    script.write_text(_WRAPPER.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="2", **(env_extra or {}))
    cmd = (launcher or [sys.executable]) + [str(script)] + _ARGS + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_gloo(tmp_path):
    """Plain ``python <bench> --gpus 2`` (WORLD_SIZE unset): two ranks, a 7-frame clip in shards of 4 + 3; the headline exchanges the
    fp32 (n,3,S,S) video (the result tensor of N = 1 and of the reference), the uint8 exchange is measured beside it."""
    line = _run(tmp_path, ["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["scaling"] == "strong"
    assert line["value"] > 0 and line["unit"] == "frames/s" and line["config"]["frames_per_step"] == 7
    per_rank = line["config"]["per_rank"]
    assert [p["rank"] for p in per_rank] == [0, 1] and [p["frames"] for p in per_rank] == [4, 3]
    assert per_rank[0]["shard"] == [0, 4] and per_rank[1]["shard"] == [4, 7]
    # --chunk-plan auto (the default): frame batch 2 over the longest shard (4) as [2, 2], or the shard as one chunk - whichever the ring model
    # predicts faster from this run's own per-frame time (on CPU tensors a frame takes ~1 s: the cut's 2.2 % outweighs any exchange -> one chunk)
    assert per_rank[0]["chunk_lengths"] in ([2, 2], [4]) and line["chunk_plan"] == per_rank[0]["chunk_lengths"]
    # (3,S,S) fp32 blocks of 4 frames from 2 ranks, in one or two chunks
    assert per_rank[0]["bytes_received_per_step"] == 2 * 2 * 2 * 3 * 64 * 64 * 4
    # the self-diagnosing top level of an N > 1 line
    assert line["rccl_world_size"] == 2 and line["backend"] == "gloo" and line["frames_per_rank"] == [4, 3]
    assert line["t_compute_ms"] > 0 and line["t_exposed_gather_ms"] >= 0 and line["bytes_received_per_rank_per_step"] == per_rank[0]["bytes_received_per_step"]
    model = line["chunk_plan_model"]
    assert model["chunked_plan"] == [2, 2] and model["one_chunk_s"] > 0 and model["chunked_s"] > 0
    assert (model["one_chunk_s"] < model["chunked_s"]) == (line["chunk_plan"] == [4])
    assert all(p["compute_ms_per_step"] > 0 for p in per_rank)
    assert "(f32)" in line["config"]["parallelism"] and "f32 video, all-gathered as f32" in line["result_tensor"]
    assert line["self_check"] is not None and "NOT a measurement" in line["data"]
    u8 = line["exchange_u8"]
    assert u8["value"] > 0 and u8["equals_u8_of_the_f32_video"] is True


def test_bench_under_torchrun_and_u8_gather(tmp_path):
    """The documented multi-GPU launch (``python -m torch.distributed.run ... bench.py --gpus 2``) keeps working; uint8 exchange on request."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    line = _run(tmp_path, ["--gpus", "2", "--gather-dtype", "u8", "--chunk-plan", "batches"], launcher=launcher)
    assert line["n_gpus"] == 2 and "(u8)" in line["config"]["parallelism"] and "exchange_u8" not in line
    assert line["chunk_plan"] == [2, 2] and line["chunk_plan_model"] is None
    assert "uint8 video, all-gathered as u8" in line["result_tensor"]
    assert line["config"]["per_rank"][1]["bytes_received_per_step"] == 2 * 2 * 2 * 64 * 64 * 3


def test_bench_single_process_line(tmp_path):
    """N = 1: no process group, no exchange format applied, the fp32 video is the result."""
    line = _run(tmp_path, ["--gpus", "1"])
    assert line["n_gpus"] == 1 and line["config"]["world_size"] == 1 and "per_rank" not in line["config"]
    assert line["config"]["frame_batch"] == 2 and line["config"]["frame_batch_requested"] == 2


# the training graph's CUDA-only guards (ConvFn.forward etc.) are answered "yes" for CPU tensors, as tests/test_abi_and_sharding.py's
# data-parallel worker does: the emulated ABI stands behind every op
_WRAPPER_PERS = _WRAPPER.replace("import bench\nbench.self_launch_if_needed()", "import bench_personalize as bench\nbench.self_launch_if_needed()") \
    .replace("bench.main()", "import torch\nimport unittest.mock as um\n"
             "with um.patch.object(torch.Tensor, 'is_cuda', new_callable=um.PropertyMock, return_value=True), \\\n"
             "        um.patch.object(torch.cuda, 'is_available', return_value=False):\n    bench.main()")


def test_bench_personalize_eight_ranks_gloo(tmp_path):
    """BASELINE configs[4] as the driver would start it at N = 8 (``python bench_personalize.py --gpus 8``, ranks self-launched), on
    CPU tensors with the gloo backend and the emulated C ABI: the FULL-width generator and discriminator (so the exchanged gradient
    buffers are the real 36,276,992 + 6,962,625 fp32 values = 173.0 MB per step per rank) at 64 x 64, the data-parallel SEGMENT schedule
    (G's all-reduce behind D's forward / backward, D's behind Adam(G): LWGTrainer._run_dp_schedule) with its exposed time reported.
    Reference: DistributedDataParallel, iPERCore/services/train.py:45-51,89-95.  Not a measurement - the line says so."""
    script = tmp_path / "pers_wrapper.py"
    script.write_text(_WRAPPER_PERS.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1")
    cmd = [sys.executable, str(script), "--gpus", "8", "--device", "cpu", "--backend", "gloo", "--size", "64", "--steps", "1", "--warmup", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == 8 and line["scaling"] == "weak"
    assert line["bytes_allreduced_per_step"] == (36276992 + 6962625) * 4 and round(line["bytes_allreduced_per_step"] / 1e6, 1) == 173.0
    assert line["exposed_allreduce_ms_per_step"] is not None and line["exposed_allreduce_ms_per_step"] >= 0.0
    assert "segmented" in line["config"]["step"] and "behind D's forward / backward" in line["allreduce_overlap"]
    assert "NOT a measurement" in line["data"] and line["value"] > 0
    assert abs(line["loss_G"]) < 1e4 and abs(line["loss_D"]) < 1e4


def test_default_frame_batch_and_kernel_launch_count():
    """N = 1: the clip of every benched configuration is ONE launch batch; N > 1: batches of 32 (fp32 at 512) so that the exchange of one batch
    overlaps the next.  The launch accounting counts the kernel launches behind a call the library slices over the batch: the library itself says
    how many (lwg_conv_slice_count = the rule of csrc/lwg_conv_slices.h; no Python copy of it)."""
    import bench
    from ipercore_amd import _lib, ops
    assert bench.default_frame_batch("fp32", 512) >= 300 and bench.default_frame_batch("fp32", 1024) >= 96
    assert bench.default_frame_batch("fp32", 256) >= 300 and bench.default_frame_batch("bf16", 1024) >= 180
    assert bench.default_frame_batch("fp32", 512, world=8) == 32 and bench.default_frame_batch("bf16", 1024, world=8) == 20

    def launches(B, H, W, C0, C1=0, bf16=False):
        a = _lib.LwgConvArgs()
        a.B, a.H, a.W, a.C0, a.C1 = B, H, W, C0, C1
        a.xdt = _lib.DT_BF16 if bf16 else _lib.DT_F32
        return _lib.lib().lwg_conv_slice_count(a)

    assert launches(32, 512, 512, 64) == 1                        # 2.1 GiB
    assert launches(300, 64, 64, 256) == 1                        # the residual blocks of a 300-frame batch: 1.2 GiB
    assert launches(300, 512, 512, 64) == 7                       # 67 MB per frame: 47 frames per slice
    assert launches(300, 256, 256, 128, 128) == 4                 # the larger of the two concatenated inputs governs
    assert launches(47, 512, 512, 64) == 1 and launches(48, 512, 512, 64) == 2
    assert launches(180, 512, 512, 128, bf16=True) == 4           # bf16: half the bytes per element
    assert launches(2, 4096, 4096, 64) == 0 and launches(0, 8, 8, 64) == 0      # one frame does not fit / empty batch: rejected by the entry points


def test_bench_default_frame_batches(tmp_path):
    """No --frame-batch: N = 1 renders the clip as ONE launch batch (reported: the clip length, the request beside it); N = 2 keeps the
    shard in batches of the sharded default (here larger than the 4-frame shard)."""
    line = _run(tmp_path, ["--gpus", "1", "--frame-batch", "0"])
    assert line["config"]["frame_batch"] == 7 and line["config"]["frame_batch_requested"] == 1024
    line = _run(tmp_path, ["--gpus", "2", "--frame-batch", "0", "--no-exchange-u8"])
    assert line["config"]["frame_batch"] == 4 and line["config"]["frame_batch_requested"] == 64
    assert [p["frames"] for p in line["config"]["per_rank"]] == [4, 3]
