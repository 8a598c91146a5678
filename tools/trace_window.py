#!/usr/bin/env python3
"""A window of a rocprofv3 kernel trace as it ran: start (us, relative to the window), duration, queue, kernel.
usage: python tools/trace_window.py <trace dir> [first row as a fraction of the trace = 0.5] [rows = 60]"""
import csv, glob, re, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40],
                         int(r["Queue_Id"])))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):  # (--memory-copy-trace: copy-engine transfers)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?"), -1))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
i0 = int(len(rows) * frac)
qids = sorted({r[3] for r in rows})
t0 = rows[i0][0]
for s, e, name, q in rows[i0:i0 + n]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {'q%d' % qids.index(q) if q >= 0 else '--'}  {name[:60]}")
