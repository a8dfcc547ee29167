#!/bin/bash
# Same-box A/B of two library builds on the whole C3 step: per-kernel table of the serialised pass + the timed step, one gpurun call.
#   bash tools/ab_step.sh ugaitnet_amd/libugaitnet_hip_r03.so ugaitnet_amd/libugaitnet_hip.so  -> gpurun_out/ab_step_{a,b}.*
A=$1; B=$2
UGN_LIB=$(pwd)/$A python bench.py --no-cpu-baseline --dense-only --steps 20 --warmup 5 --kernel-table gpurun_out/ab_step_a.csv > gpurun_out/ab_step_a.json 2>/dev/null
UGN_LIB=$(pwd)/$B python bench.py --no-cpu-baseline --dense-only --steps 20 --warmup 5 --kernel-table gpurun_out/ab_step_b.csv > gpurun_out/ab_step_b.json 2>/dev/null
UGN_LIB=$(pwd)/$A python bench.py --no-cpu-baseline --dense-only --steps 20 --warmup 5 --no-roofline-pass > gpurun_out/ab_step_a2.json 2>/dev/null
UGN_LIB=$(pwd)/$B python bench.py --no-cpu-baseline --dense-only --steps 20 --warmup 5 --no-roofline-pass > gpurun_out/ab_step_b2.json 2>/dev/null
python - <<PY
import csv, json
a = {r["label"].split("]")[0]: float(r["total_us_per_step"]) for r in csv.DictReader(open("gpurun_out/ab_step_a.csv"))}
b = {r["label"].split("]")[0]: float(r["total_us_per_step"]) for r in csv.DictReader(open("gpurun_out/ab_step_b.csv"))}
for k in a:
    print("%-50s %8.1f %8.1f %+7.1f" % (k[:50], a[k], b.get(k, float("nan")), b.get(k, float("nan")) - a[k]))
print("sum of the serialised pass: %.1f -> %.1f us" % (sum(a.values()), sum(b.values())))
for f in ("a", "b", "a2", "b2"):
    d = json.load(open("gpurun_out/ab_step_%s.json" % f)); print(f, "timed step", d["ms_per_step"], "ms")
PY
