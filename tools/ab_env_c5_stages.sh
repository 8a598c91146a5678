#!/bin/bash
# per-stage wall times of processCloud on C5 (LOM_DEBUG_TIMING=1) with and without one environment switch, alternating:
#   tools/ab_env_c5_stages.sh <tag> <VAR=value> [runs]     -> gpurun_out/<tag>/stages.txt
set -u
TAG=${1:-abst}; SW=${2:-LOM_NO_CLEANUP_BEHIND_ALIGN=1}; RUNS=${3:-2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
: > "$OUT/stages.txt"
for i in $(seq 1 $RUNS); do
  for side in default switch; do
    if [ $side = switch ]; then export "$SW"; else unset "${SW%%=*}"; fi
    echo "== run $i, $([ $side = switch ] && echo $SW || echo default)" >> "$OUT/stages.txt"
    LOM_DEBUG_TIMING=1 timeout -k 10 200 python3 "$ROOT/bench.py" --config C5 --no-cpu-baseline --steps 200 > "$OUT/line.json" 2> "$OUT/log.txt" || { echo "run failed" >> "$OUT/stages.txt"; tail -5 "$OUT/log.txt" >> "$OUT/stages.txt"; exit 1; }
    python3 "$ROOT/tools/stage_times.py" "$OUT/log.txt" -200 >> "$OUT/stages.txt"
  done
done
rm -f "$OUT/log.txt"
cat "$OUT/stages.txt"
