import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
sys.argv = ["convlab", "--split", "--iters", "3", "--shapes", sys.argv[1], "tools/lab/pp_ts.so"]
os.environ["LWG_SPLIT_PP"] = "1"
import tools.convlab as cl
cl.main()
h = ctypes.CDLL(os.path.abspath("tools/lab/pp_ts.so"))
buf = (ctypes.c_ulonglong * 32)()
assert h.lwg_lab_read_pp_ts(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 4).astype(np.int64)
print("wave: mfma_phase  staging_phase  barrier_after_mfma  barrier_after_staging   (cycle totals of WG 0, main loop)")
for w in range(8):
    print(w, a[w], "sum", a[w].sum())
