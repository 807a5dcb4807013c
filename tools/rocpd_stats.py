#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (the default --kernel-trace output of ROCm 7.2) as a per-kernel
stats table: calls, total/avg/min/max duration, share of GPU time.  Usage: rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"void (.*)", name)
    return (m.group(1) if m else name)[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            f"from kernels group by {namecol} order by sum(end-start) desc"))
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % GPU time |", "|---|---|---|---|---|---|---|"]
    for n, c, tot, avg, mn, mx in rows:
        lines.append(f"| `{short(n)}` | {c} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.2f} |")
    out = "\n".join(lines) + f"\n\ntotal kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
