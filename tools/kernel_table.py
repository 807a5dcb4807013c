#!/usr/bin/env python3
"""Per-kernel table from one rocprofv3 --kernel-trace --stats run and the FETCH_SIZE / WRITE_SIZE PMC passes:
average duration, memory-side traffic per launch (FETCH x2: gfx950 correction for 16-B/lane reads; WRITE as reported) and
the bandwidth it corresponds to.  Usage: kernel_table.py STATS_CSV PMC_FETCH_DIR PMC_WRITE_DIR [out.md]"""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(.*$", "", n)
    m = re.match(r"void (.*)", n)
    return m.group(1) if m else n


def agg(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    stats, fdir, wdir = sys.argv[1:4]
    fe, wr = agg(fdir, "FETCH_SIZE"), agg(wdir, "WRITE_SIZE")
    rows = [(short(r["Name"]), float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["Percentage"])) for r in csv.DictReader(open(stats))]
    lines = ["| kernel | calls | avg us | % GPU time | fetch MB/launch (x2) | write MB/launch | memory-side TB/s |", "|---|---|---|---|---|---|---|"]
    for k, us, c, pct in rows:
        if not k.startswith("lwg_"):
            continue
        f, w = 2 * fe.get(k, 0.0) / 1024, wr.get(k, 0.0) / 1024
        lines.append(f"| `{k[:60]}` | {c} | {us:.1f} | {pct:.2f} | {f:.1f} | {w:.1f} | {(f + w) / us:.2f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
