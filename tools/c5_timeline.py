#!/usr/bin/env python3
"""Device timeline of one C5 frame from a rocprofv3 kernel trace of `bench.py --config C5`: frames are cut at
k_fe_stats (the first kernel of a frame's front end); per kernel family the median start offset inside the frame, the
median duration and the queue it ran on; then the frame's busy time (union of all kernels) against its length -- the
rest is the device waiting for the host (launch, read-back, thread hand-off).
usage: python tools/c5_timeline.py <trace dir>"""
import csv, glob, re, sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
            fam = (m.group(1) + (m.group(2) or "")) if m else name[:30]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam, int(r["Queue_Id"])))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_fe_stats")]
frames = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
frames = frames[len(frames) // 4:]  # past the warm-up and the first keyframe growth
lens = sorted(g[0][0] - f[0][0] for f, g in zip(frames[:-1], frames[1:]))
print(f"{len(frames)} frames; frame period on the device: median {lens[len(lens) // 2] / 1e3:.1f} us, p10 {lens[len(lens) // 10] / 1e3:.1f}, p90 {lens[len(lens) * 9 // 10] / 1e3:.1f}")
# the typical frame: the kernel sequence most frames share
sig = defaultdict(list)
for f in frames:
    sig[tuple(r[2] for r in f)].append(f)
best = max(sig.values(), key=len)
print(f"most common kernel sequence: {len(best)} of {len(frames)} frames, {len(best[0])} launches")
n = len(best[0])
print(f"{'kernel':44s} {'queue':>5s} {'start':>8s} {'dur':>7s} {'gap before':>10s}  (us, medians)")
def med(v):
    v = sorted(v)
    return v[len(v) // 2]
qids = sorted({r[3] for f in best for r in f})
for k in range(n):
    st = med([f[k][0] - f[0][0] for f in best]) / 1e3
    du = med([f[k][1] - f[k][0] for f in best]) / 1e3
    # gap to the latest end among the earlier kernels of the frame
    gp = med([f[k][0] - max([r[1] for r in f[:k]] or [f[k][0]]) for f in best]) / 1e3
    print(f"{best[0][k][2][:44]:44s} {qids.index(best[0][k][3]):5d} {st:8.1f} {du:7.1f} {gp:10.1f}")
busy = []
for f in best:
    iv = sorted((r[0], r[1]) for r in f)
    tot, cs, ce = 0, None, None
    for a, b in iv:
        if ce is None or a > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    tot += ce - cs
    busy.append(tot)
print(f"device busy per frame (union of kernels): median {med(busy) / 1e3:.1f} us")
# where does the frame's first kernel (k_fe_stats: upload + statistics) run: beside the previous frame's align (sent ahead,
# lom_odometry_hint_next) or behind it?
solve = sorted((r[0], r[1]) for r in rows if r[2].startswith("k_lm") or r[2].startswith("k_match"))
import bisect
st = [a for a, b in solve]
inside = 0
all_stats = [r for r in rows if r[2].startswith("k_fe_stats")]
for r in all_stats:
    i = bisect.bisect_right(st, r[0]) - 1
    # inside an align: a solve kernel is running, or the next one starts within 3 us (the gap between two of them)
    if i >= 0 and (solve[i][1] > r[0] or (i + 1 < len(solve) and solve[i + 1][0] - r[0] < 3000)):
        inside += 1
print(f"k_fe_stats launches that start while an align's kernels run: {inside} of {len(all_stats)}")
