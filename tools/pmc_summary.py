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


def traffic_json(fetch_dir, write_dir, out_path, kernel_substr="lwg_conv_igemm_kernel"):
    """Per-launch HBM-side traffic of one kernel family from the FETCH_SIZE / WRITE_SIZE passes (separate rocprofv3
    runs).  Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB;
    on gfx950 FETCH_SIZE reports half of the bytes of 16-B/lane coalesced reads, so it is doubled; WRITE_SIZE is used
    as reported (uncalibrated)."""
    import json

    def mean_kb(d, counter):
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
        return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)

    def mean_kb_of(d, counter, sub):
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
        return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)

    fk, nf = mean_kb(fetch_dir, "FETCH_SIZE")
    wk, nw = mean_kb(write_dir, "WRITE_SIZE")
    by_kernel = {}
    for sub in ("lwg_conv_winograd4_kernel", "lwg_conv_winograd_kernel", "lwg_convt_winograd_kernel", "lwg_conv_igemm_kernel", "lwg_conv_bf16_hr2_kernel", "lwg_lwb_attn_x"):
        f1, n1 = mean_kb_of(fetch_dir, "FETCH_SIZE", sub)
        w1, n2 = mean_kb_of(write_dir, "WRITE_SIZE", sub)
        if n1 and n2:
            by_kernel[sub] = {"launches_fetch_pass": n1, "launches_write_pass": n2, "fetch_size_kib_per_launch_raw": f1, "write_size_kib_per_launch_raw": w1,
                              "traffic_bytes_per_launch": (2.0 * f1 + w1) * 1024.0}
    cfg = None
    cfg_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_last_config.json")
    if os.path.exists(cfg_path):          # written by the bench run the counters were collected on (the last one)
        with open(cfg_path) as fp:
            cfg = json.load(fp)
    out = {"kernel": kernel_substr, "command": os.environ.get("LWG_PMC_CMD", ""), "bench_config": cfg, "launches_fetch_pass": nf, "launches_write_pass": nw,
           "fetch_size_kib_per_launch_raw": fk, "write_size_kib_per_launch_raw": wk,
           "fetch_correction": 2.0,
           "traffic_bytes_per_launch": None if fk is None or wk is None else (2.0 * fk + wk) * 1024.0,
           "by_kernel": by_kernel}
    with open(out_path, "w") as fp:
        json.dump(out, fp, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
        traffic_json(*sys.argv[2:6])            # fetch dir, write dir, out.json[, kernel name substring]
    else:
        main()
