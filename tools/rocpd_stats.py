#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) as a per-kernel table:
calls, total ms, avg us, share.  Usage: python tools/rocpd_stats.py results.db [--md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(path, md=False, top=30):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), min(d.end-d.start), max(d.end-d.start) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    sep = " | " if md else "  "
    hdr = ["share%", "total_ms", "calls", "avg_us", "min_us", "max_us", "kernel"]
    print(("| " if md else "") + sep.join(hdr))
    if md:
        print("|" + "---|" * len(hdr))
    for name, n, tot, mn, mx in rows[:top]:
        vals = [f"{100.0 * tot / total:6.2f}", f"{tot / 1e6:9.3f}", f"{n:6d}", f"{tot / n / 1e3:9.2f}", f"{mn / 1e3:8.2f}",
                f"{mx / 1e3:9.2f}", short(name)]
        print(("| " if md else "") + sep.join(vals))


if __name__ == "__main__":
    main(sys.argv[1], md="--md" in sys.argv)
