#!/bin/bash
# C3 k_match counters in separate passes: tools/pmc_c3_match.sh <tag> -> gpurun_out/<tag>/pmc_c3_*.txt
set -u
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --config ${2:-C3} --no-cpu-baseline --no-extras --steps 10 --warmup 2"
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/raw_$name" -o p -- $B >> "$OUT/log.txt" 2>&1
  echo "$name rc=$?" >> "$OUT/log.txt"
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/raw_$name" k_match > "$OUT/pmc_${2:-C3}_$name.txt" 2>/dev/null
  rm -rf "$OUT/raw_$name"
}
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
