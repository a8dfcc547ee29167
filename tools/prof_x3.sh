#!/bin/bash
# SQ counters + kernel trace of tools/bench_x3.py --only-x3 (per-kernel MFMA busy, clock, wait share, VALU per MFMA, LDS conflicts)
#   bash tools/prof_x3.sh <tag> [bench_x3 flags]   -> gpurun_out/<tag>_x3_pmc.csv, gpurun_out/<tag>_x3_lds.csv
set -u
TAG=${1:-x3}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/prof_$TAG/pmc -o sq -- python3 $REPO/tools/bench_x3.py --only-x3 --reps 3 "$@" > $OUT/prof_$TAG/sq.log 2>&1
cp $(find $OUT/prof_$TAG/pmc -name "sq_counter_collection.csv" | head -1) $OUT/prof_$TAG/sq_counter_collection.csv
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/prof_$TAG/pmc2 -o lds -- python3 $REPO/tools/bench_x3.py --only-x3 --reps 3 "$@" > $OUT/prof_$TAG/lds.log 2>&1
cp $(find $OUT/prof_$TAG/pmc2 -name "lds_counter_collection.csv" | head -1) $OUT/prof_$TAG/lds_counter_collection.csv
python3 - $OUT/prof_$TAG $OUT/${TAG}_x3_pmc.csv <<'PY'
import collections, csv, sys
d, out = sys.argv[1], sys.argv[2]
def load(prefix):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("%s/%s_counter_collection.csv" % (d, prefix))):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[n]["_dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return agg
m = lambda v: sum(v) / len(v) if v else float("nan")
sq, lds = load("sq"), load("lds")
with open(out, "w") as f:
    f.write("kernel,us,clock_GHz,mfma_busy_pct,wave_wait_pct,valu_per_mfma,lds_conflict_per_active,lds_active_per_busy\n")
    for k, c in sorted(sq.items(), key=lambda kv: -m(kv[1]["_dur"])):
        if m(c["SQ_INSTS_MFMA"]) < 1: continue
        du = m(c["_dur"]) / 1e3
        clk = m(c["GRBM_GUI_ACTIVE"]) / 8 / (du * 1e3)
        busy = m(c["SQ_BUSY_CU_CYCLES"])
        l = lds.get(k, {})
        f.write('"%s",%.1f,%.2f,%.1f,%.1f,%.2f,%.3f,%.3f\n' % (k, du, clk, m(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (busy * 4) * 100,
                m(c["SQ_WAIT_ANY"]) / m(c["SQ_WAVE_CYCLES"]) * 100, m(c["SQ_INSTS_VALU"]) / m(c["SQ_INSTS_MFMA"]),
                m(l.get("SQ_LDS_BANK_CONFLICT", [])) / max(1.0, m(l.get("SQ_LDS_IDX_ACTIVE", [1]))), m(l.get("SQ_LDS_IDX_ACTIVE", [])) / max(1.0, m(l.get("SQ_BUSY_CU_CYCLES", [1])))))
print(open(out).read())
PY
