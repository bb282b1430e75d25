#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the same table `--stats` prints:
per-kernel calls, total / average / min / max duration and share of GPU time.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [steps] > profiles/summary.csv
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"void\s+(.*)", name)
    return (m.group(1) if m else name)[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute("select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                     "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kd, ks))
    rows = list(rows)
    total = sum(r[2] for r in rows)
    print("kernel,calls,total_us,avg_us,min_us,max_us,percent" + (",us_per_step" if steps else ""))
    for name, n, tot, mn, mx in rows:
        line = '"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (short(name), n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3,
                                                    100.0 * tot / total)
        if steps:
            line += ",%.1f" % (tot / 1e3 / steps)
        print(line)
    print('"TOTAL",%d,%.1f,,,,100.0' % (sum(r[1] for r in rows), total / 1e3) + (",%.1f" % (total / 1e3 / steps) if steps else ""))


if __name__ == "__main__":
    main()
