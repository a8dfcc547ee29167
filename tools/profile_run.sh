#!/bin/bash
# rocprofv3 evidence for profiles/: per-kernel durations (serialised launches and the default overlapped schedule) and three PMC
# passes (SQ, FETCH_SIZE, WRITE_SIZE: separate runs, as MI355X_MICROARCH.md prescribes) of the headline workload.
#   bash tools/profile_run.sh <tag>          -> gpurun_out/<tag>_*.csv
set -u
TAG=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof_$TAG
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py --no-cpu-baseline --dense-only --no-roofline-pass"
stats() {   # $1 = name, rest = bench flags
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG/$name -o $name -- python3 $BENCH "$@" > $OUT/prof_$TAG/$name.log 2>&1
  cp $(find $OUT/prof_$TAG/$name -name "${name}_kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_$name.csv && echo "kernel stats: $name"
}
stats serial --serial --steps 20 --warmup 5
stats default --steps 20 --warmup 5
pmc() {     # $1 = prefix, rest = counters
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/prof_$TAG/pmc -o $name -- python3 $BENCH --serial --steps 3 --warmup 1 > $OUT/prof_$TAG/pmc_$name.log 2>&1
  cp $(find $OUT/prof_$TAG/pmc -name "${name}_counter_collection.csv" | head -1) $OUT/prof_$TAG/${name}_counter_collection.csv && echo "pmc pass: $name"
}
pmc sq SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python3 $REPO/tools/pmc_summary.py $OUT/prof_$TAG $OUT/${TAG}_pmc_summary.csv > /dev/null && echo "pmc summary written"
