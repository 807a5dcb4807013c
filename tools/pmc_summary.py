#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: mean of each counter per dispatch (+ mean duration when the kernel
trace CSV is present).  Usage: pmc_summary.py DIR [out.md]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(.*$", "", n)
    m = re.match(r"void (.*)", n)
    return (m.group(1) if m else n)[:90]


def main():
    d = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    counters = sorted({c for k in agg.values() for c in k})
    lines = ["| kernel | dispatches | avg us | " + " | ".join(counters) + " |", "|---|---|---|" + "---|" * len(counters)]
    for k in sorted(agg, key=lambda k: -sum(dur.get(k, [0]))):
        n = max(len(v) for v in agg[k].values())
        us = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
        lines.append(f"| `{k}` | {n} | {us:.1f} | " + " | ".join(f"{sum(agg[k][c]) / max(len(agg[k][c]), 1):.4g}" for c in counters) + " |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
