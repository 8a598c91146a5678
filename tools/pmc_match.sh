#!/bin/bash
# k_match counters in separate passes on one configuration:  tools/pmc_match.sh <tag> [C2|C3|C4]  -> gpurun_out/<tag>/pmc_<cfg>_<pass>.txt
# (SQ issue / wait, the vector-memory path: TA, TCP, TD, and the L2 side)
set -u
TAG=${1:-pmc}
CFG=${2:-C3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/raw_$name" -o p -- python3 "$ROOT/bench.py" --config $CFG --no-cpu-baseline --no-extras --steps 10 --warmup 2 >> "$OUT/log.txt" 2>&1
  echo "$name rc=$?" >> "$OUT/log.txt"
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/raw_$name" k_match > "$OUT/pmc_${CFG}_$name.txt" 2>/dev/null
  rm -rf "$OUT/raw_$name"
}
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU


pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum

grep "rc=" "$OUT/log.txt"
