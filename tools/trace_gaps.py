#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace CSV: how busy was the GPU between the first and the last kernel of the trace's tail?
Prints the union of the kernel intervals, the sum of their durations (> union when streams overlap), the idle time inside the
window and a histogram of the idle gaps.  Usage: trace_gaps.py kernel_trace.csv [last_fraction=0.5]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
iv = iv[int(len(iv) * (1 - frac)):]
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in iv)
print(f"kernels {len(iv)}  window {1e-6 * (t1 - t0):.2f} ms  union busy {1e-6 * busy:.2f} ms ({100 * busy / (t1 - t0):.1f} %)  sum of durations {1e-6 * tot:.2f} ms"
      f"  idle {1e-6 * (t1 - t0 - busy):.2f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 2000), (2000, 5000), (5000, 10000), (10000, 50000), (50000, 10 ** 12)):
    g = [x for x in gaps if lo <= x < hi]
    print(f"  gaps {lo / 1e3:5.0f}-{hi / 1e3 if hi < 10 ** 9 else float('inf'):5.0f} us: {len(g):6d}  total {1e-6 * sum(g):7.2f} ms")
short = sorted(((e - s), n) for s, e, n in iv)
n_short = sum(1 for d, _ in short if d < 5000)
print(f"  kernels shorter than 5 us: {n_short} ({1e-6 * sum(d for d, _ in short if d < 5000):.2f} ms)")
