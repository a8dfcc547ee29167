"""Edge semantics of the x3 arithmetic, measured: what `ugn_x3_conv3x3_fwd_multi` returns for +-inf, NaN, FLT_MAX,
values at the top of bf16's range, |x| in [2^-126, 2^-100] and fp32 subnormals, beside the library's direct fp32-MFMA kernel
(`ugn_conv3x3_fwd`) on the same tensors.  One probe per image: the image is zero except ONE element = v at pixel (8, 8), channel 3; the
filter is zero except the centre tap w[1, 1, 3, co] = wv for every co.  The output at (8, 8) is then LeakyReLU(v * wv) in exact
arithmetic and every other output 0 (an fp32 convolution adds exact zeros to it), so what the two kernels do with ONE special operand
is read off directly.  Also splits every probe into its planes (`ugn_x3_split`) and reports whether x0 + x1 + x2 == x.

    python tools/x3_edges.py          (prints one line per probe; tests/test_x3_gpu.py::test_x3_edge_semantics pins the outcome)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

FLT_MAX = np.float32(3.4028234663852886e38)
BF16_MAX = np.float32(3.3895313892515355e38)           # 0x7f7f0000
BF16_ROUNDS_UP = np.float32(3.3961775292304995e38)     # 0x7f7f8000: half way between the largest bf16 and 2^128 (ties to even -> inf)


def probes():
    f = np.float32
    out = [("+inf", f(np.inf), f(0.5)), ("-inf", f(-np.inf), f(0.5)), ("nan", f(np.nan), f(0.5)),
           ("flt_max", FLT_MAX, f(0.5)), ("-flt_max", -FLT_MAX, f(0.5)),
           ("bf16_max", BF16_MAX, f(0.5)), ("below_tie_to_inf", np.nextafter(BF16_ROUNDS_UP, f(0)), f(0.5)),
           ("tie_to_inf", BF16_ROUNDS_UP, f(0.5)), ("3.0e38", f(3.0e38), f(0.5)), ("1.0", f(1.0) + f(2.0 ** -23), f(1.0))]
    for e in (-100, -105, -110, -112, -116, -120, -124, -126):
        out.append(("2^%d*(1+2^-23+2^-9)" % e, f(2.0 ** e) * (f(1.0) + f(2.0 ** -23) + f(2.0 ** -9)), f(1.0)))
    out += [("subnormal 2^-130", f(2.0 ** -130), f(1.0)), ("subnormal 2^-149", f(2.0 ** -149), f(1.0)),
            ("tiny product 2^-70*2^-70", f(2.0 ** -70), f(2.0 ** -70)), ("w = inf", f(1.0), f(np.inf)), ("w = flt_max", f(1.0), FLT_MAX)]
    return out


def run(dev=None, hw=16, cin=64, cout=128):
    from ugaitnet_amd import ops, x3
    dev = dev or torch.device("cuda:0")
    rows = []
    for name, v, wv in probes():
        x = np.zeros((1, hw, hw, cin), np.float32)
        x[0, 8, 8, 3] = v
        w = np.zeros((3, 3, cin, cout), np.float32)
        w[1, 1, 3, :] = wv
        xt, wt = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
        o3 = torch.empty((1, hw, hw, cout), device=dev)
        x3.conv3x3_fwd_multi([xt], [x3.pack(wt, False)], cout, False, [o3])
        od = ops.conv3x3_fwd(xt, ops.pack3x3(wt), False)
        od = od[0] if isinstance(od, (tuple, list)) else od
        planes = x3.split(torch.from_numpy(np.array([v, wv], np.float32)).to(dev)).cpu().numpy().view(np.uint16).astype(np.uint32) << 16
        p = planes.view(np.float32).astype(np.float64)
        with np.errstate(all="ignore"):
            exact = [bool(np.float32(p[0, k] + p[1, k] + p[2, k]) == np.float32((v, wv)[k])) for k in (0, 1)]
            want = np.float64(v) * np.float64(wv)
            want = want if want > 0 else 0.3 * want
        a3, ad = o3.cpu().numpy(), od.cpu().numpy()
        rows.append(dict(name=name, v=float(v), w=float(wv), want=float(want), x3=float(a3[0, 8, 8, 0]), direct=float(ad[0, 8, 8, 0]),
                         x3_elsewhere_clean=bool(np.all(np.delete(a3.reshape(-1, cout), 8 * hw + 8, axis=0) == 0)),
                         direct_elsewhere_clean=bool(np.all(np.delete(ad.reshape(-1, cout), 8 * hw + 8, axis=0) == 0)),
                         x3_all_channels_equal=bool(np.all(a3[0, 8, 8] == a3[0, 8, 8, 0]) or np.all(np.isnan(a3[0, 8, 8]))),
                         split_exact_x=exact[0], split_exact_w=exact[1], planes_x=[float(t) for t in p[:, 0]]))
    return rows


if __name__ == "__main__":
    for r in run():
        rel = abs(r["x3"] - r["want"]) / abs(r["want"]) if np.isfinite(r["want"]) and r["want"] != 0 and np.isfinite(r["x3"]) else float("nan")
        print("%-26s v=%-14.8g w=%-12.6g exact %-14.8g | x3 %-14.8g (rel err %.2e) direct fp32 %-14.8g | split exact x=%s w=%s planes %s | "
              "other outputs zero: x3 %s direct %s" % (r["name"], r["v"], r["w"], r["want"], r["x3"], rel, r["direct"], r["split_exact_x"],
                                                        r["split_exact_w"], ["%.6g" % t for t in r["planes_x"]], r["x3_elsewhere_clean"],
                                                        r["direct_elsewhere_clean"]))
