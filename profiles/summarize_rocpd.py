#!/usr/bin/env python
"""Turns a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table that is committed
under profiles/.  Usage: python profiles/summarize_rocpd.py <results.db> <out.md> [title]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:90]


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats; durations in microseconds; total GPU kernel time "
                f"{total / 1e3:.1f} us over {sum(r[1] for r in rows)} dispatches\n\n")
        f.write("| kernel | calls | total us | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | grid_x | wg_x |\n")
        f.write("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            f.write(f"| {short(r[0])} | {r[1]} | {r[2] / 1e3:.1f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | "
                    f"{100.0 * r[2] / total:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main()
