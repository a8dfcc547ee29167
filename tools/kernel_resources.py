#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel in one .hip source (compiles the device side only; no GPU needed).

    python tools/kernel_resources.py ugaitnet_amd/csrc/conv3x3_wino.hip [more.hip ...]
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "--no-gpu-bundle-output", "-w"]


def resources(src):
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "k.co")
        subprocess.run([HIPCC] + FLAGS + os.environ.get("UGN_EXTRA_HIPCC_FLAGS", "").split() + ["-c", src, "-o", co], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    rows = []
    for blk in notes.split("  - .agpr_count:")[1:]:
        get = lambda k: re.search(r"\.%s:\s+'?([^'\s]+)" % k, blk).group(1)
        name = subprocess.run(["c++filt", get("symbol")[:-3]], capture_output=True, text=True).stdout.strip()
        name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("ugn_wino::", "")
        name = re.sub(r"\(.*", "", name)
        rows.append((name, int(get("vgpr_count")), int(get("vgpr_spill_count")), int(get("sgpr_count")), int(get("sgpr_spill_count")),
                     int(get("private_segment_fixed_size")), int(get("group_segment_fixed_size"))))
    return rows


if __name__ == "__main__":
    for src in sys.argv[1:]:
        print(src)
        for r in sorted(resources(src)):
            print("  %-50s vgpr %3d (spill %d)  sgpr %3d (spill %d)  scratch %4d B  static LDS %6d B" % r)
