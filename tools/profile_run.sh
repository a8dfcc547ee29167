#!/bin/bash
# rocprofv3 evidence for profiles/: per-kernel durations (serialised launches and the default overlapped schedule) and three PMC
# passes (SQ, FETCH_SIZE, WRITE_SIZE: separate runs, as MI355X_MICROARCH.md prescribes) of the headline workload.
#   bash tools/profile_run.sh <tag>          -> gpurun_out/<tag>_*.csv
#   DTYPE=f32 SUFFIX=_f32 bash tools/profile_run.sh <tag>   the same for another arithmetic (files get the suffix)
set -u
TAG=${1:-r02}
DTYPE=${DTYPE:-}
SUFFIX=${SUFFIX:-}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
TAGDIR=$TAG$SUFFIX
mkdir -p $OUT/prof_$TAGDIR
cd /tmp && export TMPDIR=/tmp
BENCH="$REPO/bench.py --no-cpu-baseline --dense-only --no-roofline-pass ${DTYPE:+--dtype $DTYPE}"
stats() {   # $1 = name, rest = bench flags
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAGDIR/$name -o $name -- python3 $BENCH "$@" > $OUT/prof_$TAGDIR/$name.log 2>&1
  cp $(find $OUT/prof_$TAGDIR/$name -name "${name}_kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_$name$SUFFIX.csv && echo "kernel stats: $name"
}
if [ -z "${PMC_ONLY:-}" ]; then      # PMC_ONLY=1: counters only (an experiment's build, named by UGN_LIB)
stats serial --serial --steps 20 --warmup 5
stats default --steps 20 --warmup 5
fi
pmc() {     # $1 = prefix, rest = counters
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/prof_$TAGDIR/pmc -o $name -- python3 $BENCH --serial --steps 3 --warmup 1 > $OUT/prof_$TAGDIR/pmc_$name.log 2>&1
  cp $(find $OUT/prof_$TAGDIR/pmc -name "${name}_counter_collection.csv" | head -1) $OUT/prof_$TAGDIR/${name}_counter_collection.csv && echo "pmc pass: $name"
}
pmc sq SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
# (+ the FETCH / WRITE bytes per launch as the record bench.py's `roofline.traffic` comes from: copy it to profiles/ with the summaries)
python3 $REPO/tools/pmc_summary.py $OUT/prof_$TAGDIR $OUT/${TAG}_pmc_summary$SUFFIX.csv 0 $OUT/roofline_traffic_${DTYPE:-f32x3}.json \
  ${FRAMES:-1800} ${SETS:-72} > /dev/null && echo "pmc summary written"
# LDS bank conflicts per kernel (profiles/<tag>_lds_conflicts.csv): conflict cycles / LDS active cycles, LDS active / CU busy
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CU_CYCLES
python3 - "$OUT/prof_$TAGDIR/lds_counter_collection.csv" "$OUT/${TAG}_lds_conflicts$SUFFIX.csv" <<'PY' && echo "lds summary written"
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("ugn_wino::", "").split("(")[0]
    agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
m = lambda v: sum(v) / len(v) if v else 0.0
rows = [(m(c["SQ_BUSY_CU_CYCLES"]), n, c) for n, c in agg.items() if m(c["SQ_LDS_IDX_ACTIVE"]) > 1e6]
with open(sys.argv[2], "w") as f:
    f.write("kernel,lds_bank_conflict_per_active_cycle,lds_active_per_cu_busy_cycle,lds_instructions_per_launch\n")
    for _, n, c in sorted(rows, key=lambda t: -t[0]):
        f.write('"%s",%.3f,%.3f,%.0f\n' % (n, m(c["SQ_LDS_BANK_CONFLICT"]) / m(c["SQ_LDS_IDX_ACTIVE"]),
                                          m(c["SQ_LDS_IDX_ACTIVE"]) / m(c["SQ_BUSY_CU_CYCLES"]), m(c["SQ_INSTS_LDS"])))
PY
