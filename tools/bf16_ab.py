"""Lab: bench.py's bf16 1024^2 novel-view run (BASELINE configs[3]) with the fused up4 + head launch on / off, A/B/A/B in ONE process order per call
(python tools/bf16_ab.py [steps]); prints frames/s and the conv family's fraction of the bf16 roof for each."""
import io
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess

steps = sys.argv[1] if len(sys.argv) > 1 else "4"
code = ("import sys; sys.path.insert(0, %r); from ipercore_amd import ops; ops.BF16_UP4_HEAD = %s; import bench; "
        "bench.main(['--precision', 'bf16', '--size', '1024', '--workload', 'novel_view', '--steps', %r, '--warmup', '2', '--no-extras', '--cpu-frames', '0'])")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for fused in (False, True):
        r = subprocess.run([sys.executable, "-c", code % (root, fused, steps)], capture_output=True, text=True, cwd=root)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print("fused" if fused else "two launches", "FAILED", r.stderr[-800:])
            continue
        d = json.loads(line[-1])
        rf = d.get("roofline", {})
        print(("fused up4+head " if fused else "two launches   "), d["value"], "frames/s  conv frac", rf.get("frac"), "governing", rf.get("frac_of_governing_roof"),
              "self_check", d.get("self_check"), flush=True)
