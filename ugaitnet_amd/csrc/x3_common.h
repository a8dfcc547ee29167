// "x3": fp32 3x3 convolutions on the bf16 matrix pipe through the operands' exact three-way split (round 5).
//
// Why.  v_mfma_f32_*_f32 issues at the fp32 VECTOR rate (157 TFLOP/s dense) and holds the SIMD's vector issue while it runs, so the
// Winograd fp32 kernels sit where matrix time and everything else ADD UP (DESIGN_HISTORY.md section B: 0.57-0.67 of that peak since
// round 2).  The bf16 pipe is 16x faster and runs beside the VALU.  An IEEE fp32 value is EXACTLY the sum of three bf16 values
//        x = x0 + x1 + x2,   x0 = bf16(x),  x1 = bf16(x - x0),  x2 = x - x0 - x1        (24 = 8 + 8 + 8 significant bits;
// bf16 has fp32's exponent range, so there is no block exponent and nothing to overflow), and a product is the sum of the nine
// partial products xi * wj, each of them exact in the MFMA's fp32 accumulator.  The kernels here keep the six largest,
//        x*w ~ x0 w0 + x0 w1 + x1 w0 + x1 w1 + x0 w2 + x2 w0,
// and drop x1 w2 + x2 w1 + x2 w2 <= 2^-23 |x w|: below the rounding of the fp32 accumulation they are added into.  Emulated in numpy
// (tests/test_x3_arithmetic.py, K = 288 ... 4608, error against fp64 relative to sum|x w|): six products 0.7e-7 max -- the same to
// three digits with all NINE (exact products) -- against 1.8-2.5e-7 for a sequential fp32 FMA chain.  Measured on the GPU against the
// fp64 oracle, side by side with the library's fp32-MFMA kernels on the same inputs (tools/x3_accuracy.py, profiles/r05_x3_accuracy.txt,
// asserted in tests/test_x3_gpu.py): forward 0.8-0.9x, data gradients 0.6-1.3x, weight gradients 1.0-2.2x their error, every kernel
// inside the bars of the fp32 kernels it stands in for.  Six bf16 MFMAs cost 6/16 of one fp32 MFMA of the same shape: 2.7x the fp32
// matrix rate for fp32-grade results, with fp32 tensors in HBM (nothing about the data layout, the HBM-bound kernels or the saved
// tensors changes; the split happens in registers while a tile is staged into LDS).
#pragma once
#include "common.h"

namespace ugn_mm {
int persistent_wgs();      // conv3x3_mm.hip: ugn_set_persistent_wgs (default 256 = one persistent workgroup per CU)
}

namespace ugn_x3 {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_bf(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}

// two fp32 values -> three dwords of packed bf16 pairs (a in the low half): a = a0 + a1 + a2 exactly, round to nearest even.
// Seven vector instructions per pair: v_cvt_pk_bf16_f32 rounds both values at once, and the residual x - bf16(x) is ONE
// v_dot2c_f32_bf16 per value (pair . (-1, 0) + x, resp. pair . (0, -1) + x: products and sum exact) instead of unpack + subtract.
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float fl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf2 cvt_pk(float a, float b) {
  const fl2 v = {a, b};
  return __builtin_convertvector(v, bf2);
}
__device__ __forceinline__ unsigned pk_bf16(float a, float b) { return __builtin_bit_cast(unsigned, cvt_pk(a, b)); }
__device__ __forceinline__ void split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  // (-1, 0), (0, -1) as REGISTER operands: as immediates hipcc encodes 0x0000bf80 as the inline constant -1.0, which the hardware
  //  does not read as that bf16 pair (every result wrong by orders of magnitude, tests/test_x3_gpu.py::test_x3_split_planes)
  unsigned nl = 0x0000bf80u, nh = 0xbf800000u;
  asm volatile("" : "+s"(nl), "+s"(nh));
  const bf2 neg_lo = __builtin_bit_cast(bf2, nl), neg_hi = __builtin_bit_cast(bf2, nh);
  const bf2 h = cvt_pk(a, b);
  float ra = __builtin_amdgcn_fdot2_f32_bf16(h, neg_lo, a, false), rb = __builtin_amdgcn_fdot2_f32_bf16(h, neg_hi, b, false);   // exact
  const bf2 m = cvt_pk(ra, rb);
  ra = __builtin_amdgcn_fdot2_f32_bf16(m, neg_lo, ra, false);            // exact, <= 8 significant bits left
  rb = __builtin_amdgcn_fdot2_f32_bf16(m, neg_hi, rb, false);
  p0 = __builtin_bit_cast(unsigned, h);
  p1 = __builtin_bit_cast(unsigned, m);
  p2 = pk_bf16(ra, rb);
}
// eight consecutive channels (two float4) -> one 16-byte MFMA k group per plane
__device__ __forceinline__ void split8(const float4& u, const float4& v, uint4& p0, uint4& p1, uint4& p2) {
  split2(u.x, u.y, p0.x, p1.x, p2.x);
  split2(u.z, u.w, p0.y, p1.y, p2.y);
  split2(v.x, v.y, p0.z, p1.z, p2.z);
  split2(v.z, v.w, p0.w, p1.w, p2.w);
}

// the partial products, smallest first: (filter plane, pixel plane).  NP = 6 (the default arithmetic) drops the first three of the
// nine -- x1 w2, x2 w1, x2 w2 <= 2^-23 |x w| together; NP = 9 keeps them: the EXACT product of the operands, 1.5x the matrix time,
// selectable per call (`products` argument of the ugn_x3_* entry points) so that the claim "the dropped ones are below the rounding
// of the accumulation" is checked on the hardware itself (tests/test_x3_gpu.py::test_six_products_against_all_nine).
constexpr int kProducts = 6;
__host__ __device__ constexpr int prod9_w(int i) { return i == 0 ? 2 : i == 1 ? 1 : i == 2 ? 2 : i == 3 ? 0 : i == 4 ? 2 : i == 5 ? 1 : i == 6 ? 0 : i == 7 ? 1 : 0; }
__host__ __device__ constexpr int prod9_x(int i) { return i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : i == 3 ? 2 : i == 4 ? 0 : i == 5 ? 1 : i == 6 ? 1 : i == 7 ? 0 : 0; }
template <int NP> __host__ __device__ constexpr int prod_w(int i) { return prod9_w(i + 9 - NP); }
template <int NP> __host__ __device__ constexpr int prod_x(int i) { return prod9_x(i + 9 - NP); }
static_assert(prod_w<6>(0) == 0 && prod_x<6>(0) == 2 && prod_w<6>(5) == 0 && prod_x<6>(5) == 0 && prod_w<9>(0) == 2 && prod_x<9>(0) == 2, "product order");

constexpr int kMaxJobs = 6;      // jobs per launch: the frame-level layer and the set-level twin of up to three modalities
constexpr int kGrid = 256;       // persistent workgroups (one per CU)

// Work items of a launch are handed out per XCD: workgroup b runs on XCD b % 8 (round-robin dispatch), XCD k owns the contiguous
// item range [k * per, (k + 1) * per) and its 32 workgroups stride over it, so the items in flight on one XCD are neighbouring
// regions of the same images and their halo overlap is served by that XCD's L2.
struct ItemRange {
  int item, end, stride;
};
__device__ __forceinline__ ItemRange xcd_items(int nitems) {
  const int nx = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int per = (nitems + 7) >> 3;
  const int end = (xcd + 1) * per < nitems ? (xcd + 1) * per : nitems;
  return ItemRange{xcd * per + slot, end, nx};
}

}  // namespace ugn_x3
