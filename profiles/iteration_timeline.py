#!/usr/bin/env python
"""Kernel-by-kernel timeline of ONE GAUSSIAN-state iteration out of a rocprofv3 kernel trace of
profiles/iteration_breakdown.py: every dispatch in start order with the idle gap in front of it, and the totals
(library kernels, torch glue kernels, gaps).  Usage: python profiles/iteration_timeline.py <results.db> <out.md>"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    name = re.sub(r"<.*", "", name)
    return name[:70]


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "mlp_pack_kernel" in r[0]]          # first launch of an iteration (either state)
    if len(marks) < 3:
        print("not enough iterations in the trace", cols); return
    # iteration_breakdown.py runs 8 sizing + 4 warm-up + 16 timed + 8 event-profiled iterations: take one from the middle of the
    # timed block (the profiled ones carry an event record around every launch: ~10 us of idle per kernel)
    k = len(marks) - 16 if len(marks) >= 30 else len(marks) - 3
    a, b = marks[k], marks[k + 1]
    it = rows[a:b]
    t_next = rows[b][1]
    lib = lambda n: "trase" in n
    lines, gaps, t_lib, t_other = [], 0.0, 0.0, 0.0
    prev_end = None
    for name, s, e in it:
        gap = 0.0 if prev_end is None else max(0.0, (s - prev_end) / 1e3)
        gaps += gap
        d = (e - s) / 1e3
        if lib(name): t_lib += d
        else: t_other += d
        lines.append(f"| {short(name)} | {'lib' if lib(name) else 'torch'} | {gap:.1f} | {d:.1f} |")
        prev_end = max(e, prev_end or e)
    tail_gap = max(0.0, (t_next - prev_end) / 1e3)
    span = (t_next - it[0][1]) / 1e3
    with open(out, "w") as f:
        f.write("# One training iteration, kernel by kernel (rocprofv3 --kernel-trace of profiles/iteration_breakdown.py)\n\n")
        f.write(f"span (first dispatch -> first dispatch of the next iteration) {span:.1f} us; {len(it)} dispatches; library kernels "
                f"{t_lib:.1f} us; torch kernels {t_other:.1f} us ({sum(1 for r in it if not lib(r[0]))} dispatches); idle gaps between "
                f"dispatches {gaps:.1f} us + {tail_gap:.1f} us before the next iteration\n\n| kernel | whose | gap before us | duration us |\n|---|---|---:|---:|\n")
        f.write("\n".join(lines) + "\n")
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main()
