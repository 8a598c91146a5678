#!/bin/bash
# Runs on the GPU box (through gpurun): the bench lines and the rocprofv3 evidence that goes into profiles/.
#   tools/gpu_profile.sh <tag> [lines|stats|counters|all]     -> gpurun_out/<tag>/...
# (one gpurun call lasts at most 20 minutes: `lines` + `stats` fit one call, `counters` another)
# rocprofv3 runs from /tmp (its own temp files), the program itself after `--`, counters in separate passes.
set -u
TAG=${1:-prof}
PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*" >> "$OUT/log.txt"; timeout -k 10 "$@" >> "$OUT/log.txt" 2>&1; echo "rc=$?" >> "$OUT/log.txt"; }

if [ "$PART" = all ] || [ "$PART" = lines ]; then
# 1. bench lines (default = C2 with the CPU baseline; C3, C4, C5 without)
timeout -k 10 300 python3 "$ROOT/bench.py" > "$OUT/bench_default_line.json" 2> "$OUT/bench_default.err"; echo "default rc=$?"
for c in C3 C4 C5; do
  timeout -k 10 300 python3 "$ROOT/bench.py" --config $c --no-cpu-baseline > "$OUT/bench_${c}_line.json" 2> "$OUT/bench_$c.err"; echo "$c rc=$?"
done
fi
if [ "$PART" = all ] || [ "$PART" = stats ]; then
# 2. kernel stats of the same commands
# (the default command itself, extras included: BENCH_rNN.json's numbers -- C2 line, C3 and C5 blocks -- must be reproducible
#  from this one trace; then each configuration alone)
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_default" -o kt -- python3 "$ROOT/bench.py" --no-cpu-baseline
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_C2only" -o kt -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_C5" -o kt -- python3 "$ROOT/bench.py" --config C5 --no-cpu-baseline --steps 200
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_C3" -o kt -- python3 "$ROOT/bench.py" --config C3 --no-cpu-baseline --no-extras --steps 50
find "$OUT" -name "*kernel_stats.csv" | head
fi
if [ "$PART" = all ] || [ "$PART" = counters ]; then
# 3. counters, separate passes (short runs: every dispatch is serialised under --pmc)
B="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- $B
run 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- $B
run 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -o p -- $B
run 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --output-format csv -d "$OUT/pmc_tcc" -o p -- $B
B3="python3 $ROOT/bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq_c3" -o p -- $B3
run 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_c3" -o p -- $B3
run 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_c3" -o p -- $B3
python3 "$ROOT/tools/collect_traffic.py" "$OUT/pmc_fetch_c3" "$OUT/pmc_write_c3" "$OUT/traffic_c3.json" "${LOM_COMMIT:-?}" C3 > "$OUT/traffic_c3_summary.txt" 2>&1
python3 "$ROOT/tools/collect_traffic.py" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/traffic.json" "${LOM_COMMIT:-?}" > "$OUT/traffic_summary.txt" 2>&1
# 4. phase stamps of k_lm
timeout -k 10 120 python3 "$ROOT/tools/lm_debug.py" > "$OUT/lm_stamps.txt" 2>&1; echo "lm_debug rc=$?"
# summaries
for k in k_match k_lm k_bi_claim k_bi_colscan k_bi_scatter k_bi_group k_bi_flagscan k_bi_place; do
  for d in pmc_fetch pmc_write pmc_sq pmc_tcc; do
    python3 "$ROOT/tools/pmc_summary.py" "$OUT/$d" $k >> "$OUT/pmc_summary_$k.txt" 2>/dev/null
  done
done
for k in k_match k_lm; do python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_sq_c3" $k > "$OUT/pmc_c3_summary_$k.txt" 2>/dev/null; done
# the raw counter dumps are large: keep the summaries
rm -rf "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_sq "$OUT"/pmc_tcc "$OUT"/pmc_sq_c3 "$OUT"/pmc_fetch_c3 "$OUT"/pmc_write_c3
"$ROOT/tools/microbench/policy" > "$OUT/policy_microbench.txt" 2>&1; "$ROOT/tools/microbench/exec_skip" > "$OUT/exec_skip.txt" 2>&1
fi
