#!/bin/bash
# C5 frame time with and without one environment switch, alternating runs on ONE box:
#   tools/ab_env_c5.sh <tag> <VAR=value> [runs]     -> gpurun_out/<tag>/ab.txt
set -u
TAG=${1:-ab}; SW=${2:-LOM_NO_CLEANUP_BEHIND_ALIGN=1}; RUNS=${3:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
: > "$OUT/ab.txt"
for i in $(seq 1 $RUNS); do
  for side in default switch; do
    if [ $side = switch ]; then export "$SW"; else unset "${SW%%=*}"; fi
    timeout -k 10 200 python3 "$ROOT/bench.py" --config C5 --no-cpu-baseline --steps 200 > "$OUT/line.json" 2>> "$OUT/err.txt" || { echo "run failed" >> "$OUT/ab.txt"; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('$OUT/line.json').read().strip().splitlines()[-1])
print('run $i %-8s %s  ms_per_frame %.4f' % ('$side', '$SW' if '$side'=='switch' else '(default)', d['ms_per_step']))" >> "$OUT/ab.txt"
  done
done
cat "$OUT/ab.txt"
