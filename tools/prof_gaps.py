#!/usr/bin/env python3
"""Gaps between consecutive kernels of one prefill pass in a rocprofv3 kernel-trace database.
Usage: python tools/prof_gaps.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# a prefill pass: from the first prefill rmsnorm (grid of L rows) to the fp32-logit lm_head GEMV that follows it
idx = [i for i, r in enumerate(rows) if "rope_kv_append_vec_kernel" in r[0]]
if not idx:
    print("no prefill kernels found"); sys.exit(0)
# take the LAST pass: find the last run of 32 rope kernels
last = idx[-1]
first = idx[-32] if len(idx) >= 32 else idx[0]
lo = first - 2                      # rmsnorm + qkv gemm before the first rope
hi = last
while hi + 1 < len(rows) and "gemv_kernel" not in rows[hi][0]:
    hi += 1
win = rows[lo:hi + 1]
span = (win[-1][2] - win[0][1]) / 1e3
busy = sum(r[2] - r[1] for r in win) / 1e3
gaps = [(win[i + 1][1] - win[i][2]) / 1e3 for i in range(len(win) - 1)]
gaps_pos = [g for g in gaps if g > 0]
print(f"prefill window: {len(win)} kernels, span {span:.1f} us, kernel time {busy:.1f} us, gaps {sum(gaps_pos):.1f} us "
      f"(mean {sum(gaps_pos) / max(1, len(gaps_pos)):.2f} us, max {max(gaps):.1f} us)")
by = {}
for i, g in enumerate(gaps):
    k = win[i][0][:40] + " -> " + win[i + 1][0][:40]
    by.setdefault(k, []).append(g)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  {sum(v):8.1f} us total, {sum(v) / len(v):6.2f} us mean x{len(v):3d}  {k}")
# the decode graph for comparison: gaps between consecutive kernels of the last 2000 kernels
tail = rows[-4000:-200]
g2 = [(tail[i + 1][1] - tail[i][2]) / 1e3 for i in range(len(tail) - 1)]
print(f"decode tail: mean gap {sum(g2) / len(g2):.2f} us over {len(g2)} boundaries")
