#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline object.  Run on the GPU box from the
# repo root:   bash profiles/collect.sh [tag]      (outputs under gpurun_out/prof_<tag>/)
# Pass 1: kernel trace + stats of the default bench command.  Passes 2..: one PMC group each
# (counters are collected in their own runs, never together with API tracing), on a shorter run.
set -e -o pipefail
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
  python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/stats.log" 2>&1
grep '^{"metric"' "$OUT/stats.log" > "$OUT/bench_under_profiler.json" || true
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -o p -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc$i.log" 2>&1
  echo "pmc pass $i ($grp) done"
done
find "$OUT" -name '*.csv' | sort
