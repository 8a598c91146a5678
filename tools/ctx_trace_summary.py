#!/usr/bin/env python3
"""Summary of a rocprofv3 kernel trace of tools/ctx_trace.py: per kernel family the average duration, and for the
solve kernel (k_lm) the share of its lifetime during which a search grid (k_match) of ANOTHER queue was running -- its
workgroups wait for each other inside the kernel, so a foreign grid that holds the SIMDs stretches it."""
import csv, glob, sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            fam = "k_match" if "k_match" in name else ("k_lm" if "k_lm" in name else None)
            if fam:
                rows.append((fam, int(r["Queue_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[2])
# drop the warm-up (everything before the last third of the trace is mixed with the serial warm-up aligns)
t_lo = rows[len(rows) // 3][2]
rows = [r for r in rows if r[2] >= t_lo]
dur = defaultdict(list)
for fam, q, s, e in rows:
    dur[fam].append((e - s) / 1e3)
for fam, d in dur.items():
    d.sort()
    print(f"{fam:8s} n={len(d):6d} avg {sum(d) / len(d):7.2f} us  median {d[len(d) // 2]:7.2f}  p90 {d[int(len(d) * 0.9)]:7.2f}")
match = [(s, e, q) for fam, q, s, e in rows if fam == "k_match"]
tot = ov = 0
j0 = 0
for fam, q, s, e in rows:
    if fam != "k_lm":
        continue
    tot += e - s
    while j0 < len(match) and match[j0][1] < s - 200000:
        j0 += 1
    covered = []
    for ms, me, mq in match[j0:]:
        if ms > e:
            break
        if mq != q and me > s:
            covered.append((max(ms, s), min(me, e)))
    covered.sort()
    cur_s = cur_e = None
    for a, b in covered:
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                ov += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None:
        ov += cur_e - cur_s
print(f"k_lm lifetime overlapped by another queue's k_match: {100.0 * ov / max(tot, 1):.1f} %")
span = (max(r[3] for r in rows) - min(r[2] for r in rows)) / 1e3
n_lm = len(dur["k_lm"])
print(f"span {span / 1e3:.2f} ms, {n_lm} k_lm launches -> {n_lm / 5 / (span / 1e6):.0f} aligns/s under the profiler (5 launches per align)")
