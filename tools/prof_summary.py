#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace): per-kernel calls / total / average, grouped by grid size.
Usage: python tools/prof_summary.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("teo::", "").replace("unsigned short", "bf16")
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("""select name, grid_x, grid_y, grid_z, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0,
                          min(end-start)/1000.0, max(vgpr_count), max(accum_vgpr_count), max(lds_size)
                          from kernels group by name, grid_x, grid_y, grid_z order by 6 desc""").fetchall()
    total = sum(r[5] for r in rows)
    lines = ["| kernel | grid (threads) | calls | total ms | avg us | min us | % | vgpr | agpr | lds |", "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows[:40]:
        lines.append(f"| `{short(r[0])}` | {r[1]}x{r[2]}x{r[3]} | {r[4]} | {r[5] / 1000:.2f} | {r[6]:.2f} | {r[7]:.2f} | "
                     f"{100 * r[5] / total:.1f} | {r[8]} | {r[9]} | {r[10]} |")
    out = "\n".join(lines) + f"\n\ntotal kernel time {total / 1000:.2f} ms over {sum(r[4] for r in rows)} dispatches\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
