#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 kernel_stats CSV as ms per step.  Usage: prof_summary.py CSV STEPS [TOP]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("at::native::", "")[:90]
    print("%8.2f ms/step %5.1f%% %6s  %s" % (int(r["TotalDurationNs"]) / steps / 1e6, float(r["Percentage"]), r["Calls"], n))
print("total kernel ms per step: %.2f" % (tot / steps / 1e6))
