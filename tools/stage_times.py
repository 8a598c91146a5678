"""Average the per-stage wall times LOM_DEBUG_TIMING=1 prints on stderr (one block per processCloud).
usage: LOM_DEBUG_TIMING=1 python bench.py --config C5 2> log; python tools/stage_times.py log [skip_frames | -last_frames]
(bench.py's C5 run: ~1,000 untimed frames of a two-frame loop that brings the GPU up to its clocks, the warm-up, then the
timed sequence -- `-200` averages the timed 200 only)"""
import collections
import sys

skip = int(sys.argv[2]) if len(sys.argv) > 2 else 20
acc = collections.OrderedDict()
frames = 0
cur = {}
for line in open(sys.argv[1]):
    parts = line.rsplit(None, 2)
    if len(parts) == 3 and parts[2] == "us":
        name = parts[0].strip()
        try:
            us = float(parts[1])
        except ValueError:
            continue
        if name.startswith("processCloud total"):
            frames += 1
            if frames > max(skip, 0):
                cur["total"] = us
                for k, v in cur.items():
                    acc.setdefault(k, []).append(v)
            cur = {}
        else:
            cur[name] = cur.get(name, 0.0) + us
for k, v in acc.items():
    if skip < 0:
        v = v[skip:]
    print(f"{k:18s} {sum(v) / len(v):8.1f} us  (n={len(v)})")
