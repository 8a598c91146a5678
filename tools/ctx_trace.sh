#!/bin/bash
# kernel traces of 1 / 2 / 3 / 4 concurrent callers, shared and partitioned: tools/ctx_trace.sh <tag> -> gpurun_out/<tag>/ctx_trace_summary.txt
set -u
TAG=${1:-ctx}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "1 shared" "2 shared" "3 shared" "4 shared" "2 partitioned" "3 partitioned" "4 partitioned"; do
  set -- $cfg
  d="$OUT/raw_$1_$2"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$d" -o kt -- python3 "$ROOT/tools/ctx_trace.py" $1 $2 > "$OUT/log_$1_$2.txt" 2>&1
  echo "== $1 callers, $2 (rc=$?)" >> "$OUT/ctx_trace_summary.txt"
  python3 "$ROOT/tools/ctx_trace_summary.py" "$d" >> "$OUT/ctx_trace_summary.txt" 2>&1
  rm -rf "$d"
done
cat "$OUT/ctx_trace_summary.txt"
