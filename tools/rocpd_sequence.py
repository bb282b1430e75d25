#!/usr/bin/env python
"""Kernel launch sequence of ONE training step from a rocprofv3 (rocpd sqlite) kernel trace: the launches between the last two
launches of `--marker` (default: the fused optimizer's update kernel), in start order, with durations and the idle gap in front of
each.  Usage: python tools/rocpd_sequence.py x_results.db [--marker adam_kernel] > step_sequence.txt"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"void\s+(.*)", name)
    return (m.group(1) if m else name)[:120]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--marker", default="adam_kernel")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
    marks = [i for i, r in enumerate(rows) if a.marker in r[0]]
    if len(marks) < 2:
        raise SystemExit("marker kernel %r launched %d times" % (a.marker, len(marks)))
    lo, hi = marks[-2] + 1, marks[-1] + 1
    step = rows[lo:hi]
    t0 = rows[lo - 1][2]
    print("# %d launches, %.1f us wall, %.1f us of kernels" % (len(step), (step[-1][2] - t0) / 1e3,
                                                                sum(r[2] - r[1] for r in step) / 1e3))
    prev_end = t0
    for name, s, e in step:
        print("%8.1f  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short(name)))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
