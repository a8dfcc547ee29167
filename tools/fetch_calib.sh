#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of three LDS-DMA streams with known bytes (tools/fetch_calib.hip) -> gpurun_out/<tag>_fetch_calibration.txt
#   bash tools/fetch_calib.sh <tag>
set -u
TAG=${1:-r04}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof_$TAG/calib
[ -x $REPO/tools/_bin/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o $REPO/tools/_bin/fetch_calib $REPO/tools/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_$TAG/calib -o calib -- $REPO/tools/_bin/fetch_calib > $OUT/prof_$TAG/calib.log 2>&1
python3 - "$(find $OUT/prof_$TAG/calib -name 'calib_counter_collection.csv' | head -1)" "$(find $OUT/prof_$TAG/calib -name 'calib_kernel_trace.csv' | head -1)" > $OUT/${TAG}_fetch_calibration.txt <<'PY'
import collections, csv, sys
N = 2 << 30
want = {"0": ("full: every byte, 1 KiB contiguous per wave instruction", N, N),
        "1": ("half: bytes [0,64) of every 128-B line (64-B segments)", N // 2, N),
        "2": ("halves: [0,64) then [64,128) of every line, 32 KiB apart in time", N, N)}
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("FETCH_SIZE calibration on LDS-DMA streams (global_load_lds_dwordx4), 2 GiB buffer, rocprofv3 --pmc FETCH_SIZE; counter unit: KiB")
print("%-72s %12s %12s %12s %10s %10s" % ("stream", "requested MB", "lines MB", "FETCH raw MB", "raw/req", "raw/lines"))
for k, vals in sorted(agg.items()):
    mode = k.split("<")[1].split(">")[0].strip() if "<" in k else "?"
    name, req, lines = want.get(mode, (k, N, N))
    raw = sum(vals) / len(vals) * 1024.0        # FETCH_SIZE counts kilobytes
    us = sum(dur[k]) / max(len(dur[k]), 1)
    print("%-72s %12.1f %12.1f %12.1f %10.3f %10.3f   (%.0f us)" % (name, req / 1e6, lines / 1e6, raw / 1e6, raw / req, raw / lines, us))
PY
cat $OUT/${TAG}_fetch_calibration.txt
