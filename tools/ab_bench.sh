#!/bin/bash
# Same-box A/B of library builds on the whole C3 step: for each round and each variant (a suffix of libugaitnet_hip_<suffix>.so, or
# "default"), one bench.py process (dense step only: timed region + the serialised per-kernel pass), interleaved A B A B.
#   tools/ab_bench.sh ROUNDS TAG variant [variant ...]       -> gpurun_out/<TAG>_<variant>_<round>.json / .csv, summary on stdout
set -u
ROUNDS=$1; TAG=$2; shift 2
mkdir -p gpurun_out
for r in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    if [ "$v" = default ]; then lib=""; else lib="$PWD/ugaitnet_amd/libugaitnet_hip_$v.so"; fi
    UGN_LIB="$lib" python bench.py --no-cpu-baseline --dense-only --steps 20 --warmup 5 ${AB_ARGS:-} \
      --kernel-table "gpurun_out/${TAG}_${v}_${r}.csv" > "gpurun_out/${TAG}_${v}_${r}.json" 2> "gpurun_out/${TAG}_${v}_${r}.err" || echo "FAILED $v round $r"
  done
done
python - "$TAG" "$ROUNDS" "$@" <<'PY'
import csv, json, sys
tag, rounds, variants = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
rows = {}
for v in variants:
    for r in range(1, rounds + 1):
        try:
            d = json.load(open("gpurun_out/%s_%s_%d.json" % (tag, v, r)))
            rows.setdefault("ms_per_step (timed)", {}).setdefault(v, []).append(d["ms_per_step"])
            rows.setdefault("serial_step_us", {}).setdefault(v, []).append(d["roofline"]["serial_step_us"])
            for line in csv.reader(open("gpurun_out/%s_%s_%d.csv" % (tag, v, r))):
                if line and line[0] != "label" and not line[0].startswith("#"):
                    rows.setdefault(line[0][:70], {}).setdefault(v, []).append(float(line[3]))
        except Exception as e:
            print("missing", v, r, e)
for k, d in rows.items():
    if k in ("ms_per_step (timed)", "serial_step_us") or max(max(x) for x in d.values()) >= 20.0:
        print("%-72s" % k + "  ".join("%s %s" % (v, "/".join("%.1f" % t if t > 20 else "%.3f" % t for t in d.get(v, []))) for v in variants))
PY
