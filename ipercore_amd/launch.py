"""Self-launch of the multi-GPU entry points: one process per GPU under ``torch.distributed.run`` (RCCL = backend "nccl").

The reference's launcher spawns its ranks itself (scripts/train/dist_train.py:97-109: one ``python -m torch.distributed.launch`` per
run); here ``python bench.py --gpus N`` / ``python bench_personalize.py --gpus N`` started WITHOUT torchrun (WORLD_SIZE unset - the
shape of the driver's N = 1 command) re-execute themselves with N ranks on a free local port.  No torch import: this runs before the
heavy imports of the caller."""
import os
import socket
import subprocess
import sys


def requested_gpus(argv):
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


def self_launch_if_needed(argv=None, script=None):
    """N > 1 and no WORLD_SIZE: run ``script argv`` under torch.distributed.run with N local ranks and exit with its code (rank 0 of
    the child prints the ONE JSON line on the inherited stdout).  Under torchrun, or with N = 1, returns and the caller runs in-process."""
    argv = list(sys.argv[1:] if argv is None else argv)
    n = requested_gpus(argv)
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver supports dmabuf IPC only (RCCL's handles fail otherwise)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(sys.argv[0])] + argv
    sys.exit(subprocess.call(cmd, env=env))
