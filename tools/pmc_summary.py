"""Summarise rocprofv3 --pmc passes (csv) per kernel: duration, clock, MFMA busy, wait share, VALU per MFMA, HBM bytes.

usage: pmc_summary.py <dir with sq_/fetch_/write_ csv files> <out.csv> [min_grid_workgroups [traffic.json frames sets]]
With the last three arguments the FETCH + WRITE bytes per launch of every 3x3 kernel are also written as the record bench.py reads
its `roofline.traffic` from (profiles/roofline_traffic_<dtype>.json): `frames` / `sets` = frame-level / set-level images of a launch
(C3 at 24 clips: 1800 / 72; the 64x64 layer has no set-level twin).
FETCH_SIZE is doubled (gfx950 tallies 16 B/lane reads at half, MI355X_MICROARCH.md); units of FETCH/WRITE_SIZE are KiB.
Only launches with at least `min_grid_workgroups` workgroups are averaged (drops the set-level launches of the dense run)."""
import collections
import csv
import sys

d, out = sys.argv[1], sys.argv[2]
min_wg = int(sys.argv[3]) if len(sys.argv) > 3 else 0


def load(prefix):
    rows = list(csv.DictReader(open("%s/%s_counter_collection.csv" % (d, prefix))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        if wgs < min_wg:
            continue
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[name]["_dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return agg


sq, fe, wr = load("sq"), load("fetch"), load("write")
mean = lambda v: sum(v) / len(v) if v else float("nan")
lines = []
for k, c in sq.items():
    busy_cu = mean(c["SQ_BUSY_CU_CYCLES"])
    dur_us = mean(c["_dur"]) / 1e3
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    clk = mean(c["GRBM_GUI_ACTIVE"]) / 8 / (dur_us * 1e3) if dur_us > 0 else float("nan")
    mfma_busy = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (busy_cu * 4) * 100 if busy_cu else float("nan")   # 4 SIMDs per CU
    wait = mean(c["SQ_WAIT_ANY"]) / mean(c["SQ_WAVE_CYCLES"]) * 100 if mean(c["SQ_WAVE_CYCLES"]) else float("nan")
    vpm = mean(c["SQ_INSTS_VALU"]) / mean(c["SQ_INSTS_MFMA"]) if mean(c["SQ_INSTS_MFMA"]) else float("nan")
    f = mean(fe.get(k, {}).get("FETCH_SIZE", [])) * 1024 * 2 / 1e6
    w = mean(wr.get(k, {}).get("WRITE_SIZE", [])) * 1024 / 1e6
    lines.append((dur_us * len(c["_dur"]), k, dur_us, len(c["_dur"]), clk, mfma_busy, wait, vpm, f, w))
lines.sort(reverse=True)
with open(out, "w") as fh:
    fh.write("kernel,us_profiled,launches,clock_GHz,mfma_busy_pct,wave_wait_pct,valu_per_mfma,fetch_MB_x2_corrected,write_MB\n")
    for _, k, du, n, clk, mb, wt, vpm, f, w in lines:
        fh.write('"%s",%.1f,%d,%.2f,%.1f,%.1f,%.2f,%.1f,%.1f\n' % (k, du, n, clk, mb, wt, vpm, f, w))
print(open(out).read())
if len(sys.argv) > 6:
    import json
    import os
    import re
    frames, sets = int(sys.argv[5]), int(sys.argv[6])
    recs = []
    for _, k, du, n, clk, mb, wt, vpm, f, w in lines:
        m = re.match(r"(conv_\w+|wgrad_\w+|wino\w*)<(\d+), (\d+), (\d+)", k)
        if not m or f != f or w != w:
            continue
        images = frames if int(m.group(4)) == 64 else frames + sets
        recs.append(dict(rocprof_kernel=k, images_per_launch=images, hbm_bytes_per_launch=int(round((f + w) * 1e6, -5)),
                         source="%s: FETCH_SIZE x2 (whole 128-B lines: profiles/r04_fetch_calibration.txt) %.1f MB + WRITE_SIZE %.1f MB per "
                                "launch at %d images (separate --pmc passes of `bench.py --serial`)" % (os.path.basename(out), f, w, images)))
    json.dump(dict(kernels=recs), open(sys.argv[4], "w"), indent=1)
    print("traffic record:", sys.argv[4], len(recs), "kernels")
