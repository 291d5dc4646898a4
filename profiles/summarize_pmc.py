#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 PMC counters from rocpd sqlite files (run on the GPU box, the
databases are too large to copy back).  Usage: summarize_pmc.py out.md db1 [db2 ...]"""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "")[:70]


def main():
    out, dbs = sys.argv[1], sys.argv[2:]
    lines = []
    allagg = {}        # kernel -> counter -> average per launch
    for db in dbs:
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
        if not cols:
            lines.append(f"{db}: no counters_collection view")
            continue
        name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
        cnt_col = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x][0]
        val_col = "value" if "value" in cols else [x for x in cols if "value" in x][0]
        disp = "dispatch_id" if "dispatch_id" in cols else None
        q = (f"select {name_col}, {cnt_col}, count(*), sum({val_col}) from counters_collection "
             f"group by {name_col}, {cnt_col}")
        agg = {}
        for kname, cname, n, tot in c.execute(q):
            agg.setdefault(short(kname), {})[cname] = (n, tot)
        for k, d in agg.items():
            if k.startswith("trase::"):
                allagg.setdefault(k, {}).update({cn: v[1] / v[0] for cn, v in d.items()})
        lines.append(f"## {db}\n")
        counters = sorted({cn for d in agg.values() for cn in d})
        lines.append("| kernel | dispatches | " + " | ".join(counters) + " |")
        lines.append("|---|---:|" + "---:|" * len(counters))
        for k, d in sorted(agg.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
            if not k.startswith("trase::"):
                continue
            n = max(v[0] for v in d.values())
            lines.append(f"| {k} | {n} | " + " | ".join(f"{d[cn][1] / d[cn][0]:.4g}" if cn in d else "-" for cn in counters) + " |")
        lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    import json
    open(out[:-3] + ".json" if out.endswith(".md") else out + ".json", "w").write(json.dumps(allagg, indent=1, sort_keys=True))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
