import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", device_id=dev)
from ipercore_amd import sharding
og = sharding.OverlappedGather(40)
# world 1: force the collective path anyway
full = torch.randn(40, 3, 64, 64, device=dev)
for off in range(0, 40, 8):
    x = full[off:off + 8] * 1.0          # produced on the current stream right before the submit
    og.submit(x, off)
v = og.finish(); torch.cuda.synchronize()
assert torch.equal(v, full)
print("nccl overlapped gather ok (world 1)")
dist.destroy_process_group()
