"""The arithmetic claim of the x3 kernels (ugaitnet_amd/csrc/x3_common.h), restated in numpy so that it is checked WITHOUT a GPU:
an fp32 value is exactly the sum of three bf16 values, and a dot product formed from six of the nine partial products of the split
operands, accumulated in fp32, is as close to the fp64 result as the same accumulation of all nine (the exact products) -- and closer
than a sequential fp32 FMA chain.  (tests/test_x3_gpu.py::test_x3_split_planes checks the HIP split against this statement bit for
bit; tools/x3_accuracy.py measures the kernels themselves against the fp32-MFMA kernels of the library.)"""
import numpy as np
import pytest


def bf16_rne(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    x0 = bf16_rne(x)
    r = (x - x0).astype(np.float32)
    x1 = bf16_rne(r)
    x2 = (r - x1).astype(np.float32)
    return x0, x1, x2


def test_split_is_exact_over_the_fp32_range():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.uniform(-30, 30, 20000), rng.uniform(-1, 1, 20000),
                        [0.0, -0.0, 1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9, 1.0 + 2.0 ** -23, 3.0e38, -1.0e-30]]).astype(np.float32)
    x0, x1, x2 = split3(x)
    assert np.array_equal(x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64), x.astype(np.float64))
    assert np.array_equal(bf16_rne(x2), x2)                       # the third plane IS a bf16 value: nothing is left over
    nz = x != 0
    assert np.all(np.abs(x1[nz]) <= 2.0 ** -8 * np.abs(x[nz])) and np.all(np.abs(x2[nz]) <= 2.0 ** -16 * np.abs(x[nz]))


PAIRS6 = [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)]          # csrc/x3_common.h prod_w / prod_x: smallest first
PAIRS9 = [(2, 2), (1, 2), (2, 1)] + PAIRS6


def _accumulate(A, B, pairs, k_block=32):
    """fp32 accumulator; per block of 32 k and per partial product: the block's products summed exactly, added with ONE rounding
    (the model of an MFMA with fp32 accumulate)."""
    m, k = A[0].shape
    acc = np.zeros(m, np.float32)
    for k0 in range(0, k, k_block):
        for i, j in pairs:
            acc = (acc.astype(np.float64) + (A[i][:, k0:k0 + k_block].astype(np.float64) * B[j][:, k0:k0 + k_block]).sum(1)).astype(np.float32)
    return acc


@pytest.mark.parametrize("k", [288, 1152])
def test_six_partial_products_are_as_good_as_all_nine(k):
    rng = np.random.default_rng(k)
    m = 2048
    a = (rng.standard_normal((m, k)) * 0.05).astype(np.float32)          # filters
    b = rng.uniform(-0.5, 0.5, (m, k)).astype(np.float32)               # activations
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    scale = np.abs(a.astype(np.float64) * b).sum(1)
    A, B = split3(a), split3(b)
    e6 = np.abs(_accumulate(A, B, PAIRS6) - ref) / scale
    e9 = np.abs(_accumulate(A, B, PAIRS9) - ref) / scale
    acc = np.zeros(m, np.float32)                                         # a sequential fp32 FMA chain: one rounding per term
    for t in range(k):
        acc = (acc.astype(np.float64) + a[:, t].astype(np.float64) * b[:, t]).astype(np.float32)
    e_fma = np.abs(acc - ref) / scale
    # the three dropped products are below the accumulation's rounding: six and nine agree to a few per cent, both beat the FMA chain
    assert e6.max() <= 1.05 * e9.max() + 1e-9 and e6.mean() <= 1.02 * e9.mean() + 1e-10, (e6.max(), e9.max(), e6.mean(), e9.mean())
    assert e6.mean() < e_fma.mean() and e6.max() < e_fma.max(), (e6.mean(), e_fma.mean())
    assert e6.max() < 1.5e-7
    # ... and three products (what two planes would give) are NOT enough: an order of magnitude worse
    e3 = np.abs(_accumulate(A, B, [(0, 1), (1, 0), (0, 0)]) - ref) / scale
    assert e3.mean() > 5 * e6.mean()


def test_kernel_labels_name_every_template_argument():
    """bench.py finds a launch's PMC traffic record (profiles/roofline_traffic_f32x3.json) and the judge finds its rocprofv3 row by
    the kernel name ugaitnet_amd/x3.py puts in the label: the label must carry as many template arguments as the kernel has, and the
    committed traffic record must use the same spelling."""
    import json
    import os
    import re
    root = os.path.join(os.path.dirname(__file__), "..")
    host = open(os.path.join(root, "ugaitnet_amd", "x3.py")).read()
    for src, kernel in (("conv3x3_x3.hip", "conv_x3_kernel"), ("wgrad3x3_x3.hip", "wgrad_x3_kernel"), ("wgrad3x3_x3.hip", "wgrad_x3s_kernel")):
        text = open(os.path.join(root, "ugaitnet_amd", "csrc", src)).read()
        m = re.search(r"template <([^>]*)>\s*__global__[^\n]*\bvoid %s\(" % kernel, text)
        assert m, kernel
        nargs = m.group(1).count(",") + 1
        labels = re.findall(r'"%s<([^>]*)>"' % kernel, host)
        assert labels and all(lab.count(",") + 1 == nargs for lab in labels), (kernel, nargs, labels)
        rec = json.load(open(os.path.join(root, "profiles", "roofline_traffic_f32x3.json")))["kernels"]
        mine = [r["rocprof_kernel"] for r in rec if r["rocprof_kernel"].startswith(kernel + "<")]
        assert mine and all(n.count(",") + 1 == nargs for n in mine), (kernel, nargs, mine)
