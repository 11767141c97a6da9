"""GPU busy time against wall time inside the inner solves of a rocprofv3 kernel trace: the windows are the runs
of k_cg_update launches (one per eigen-solve of the trace); prints one JSON line per window.

  python scripts/trace_busy.py <..._kernel_trace.csv>
"""
import csv
import json
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
idx = [i for i, e in enumerate(ev) if "k_cg_update" in e[2]]
groups, cur = [], idx[:1]
for a, b in zip(idx, idx[1:]):
    if ev[b][0] - ev[a][1] > 0.3e9:
        groups.append(cur)
        cur = []
    cur.append(b)
if cur:
    groups.append(cur)
for g in groups:
    i0, i1 = g[0], g[-1]
    start = np.array([e[0] for e in ev[i0:i1 + 1]])
    end = np.array([e[1] for e in ev[i0:i1 + 1]])
    gaps = np.maximum(start[1:] - np.maximum.accumulate(end)[:-1], 0)
    dur = end - start
    print(json.dumps({
        "wall_s": round((end.max() - start[0]) / 1e9, 4), "busy_s": round(dur.sum() / 1e9, 4), "kernels": int(len(dur)),
        "cg_iterations": len(g), "gap_sum_s": round(gaps.sum() / 1e9, 4), "gap_median_us": round(float(np.median(gaps)) / 1e3, 2),
        "gaps_over_20us": int((gaps > 20000).sum()), "gaps_over_20us_sum_s": round(gaps[gaps > 20000].sum() / 1e9, 4),
        "kernels_under_10us": int((dur < 10000).sum()), "kernels_under_10us_sum_s": round(dur[dur < 10000].sum() / 1e9, 4)}))
