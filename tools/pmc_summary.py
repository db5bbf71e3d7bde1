#!/usr/bin/env python3
"""Summarise PMC counters of a rocprofv3 rocpd database per kernel name.  Usage: pmc_summary.py <db> [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("no counters view; tables:", [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()])
    sys.exit(0)
cols = [d[0] for d in cur.execute(f"select * from {view} limit 1").description]
q = f"select kernel_name, counter_name, count(*), sum(value), avg(value) from {view} where kernel_name like ? group by kernel_name, counter_name order by kernel_name"
for r in cur.execute(q, (f"%{filt}%",)):
    print(f"{r[0][:70]:70s} {r[1]:28s} n={r[2]:5d} sum={r[3]:.4g} avg={r[4]:.4g}")
