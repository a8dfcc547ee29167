// Device-side batch assembly: the sample decode of the reference's generator
// (data/mj_dataGeneratorMMUWYHsingle_repetitions.py:300-318 `__load_dd`: int16 optical flow / compressFactor * 0.1 with the
// optional magnitude clipping of :303-306, uint8 gray / depth / 255 - 0.5, silhouettes / 255), its gaitset re-layout
// (:746-753: [60,60,T] -> [25,60,60,C], channel c of frame l = raw plane C*l + c) and the expansion into rows whose
// modality is either a copy of a base sample (flag 1) or the constant `noise` (flag 0) (:732-737, :776-806).  The raw
// int16 / uint8 samples are uploaded once; the fp32 clip tensors, 4-8x larger and `expand` times repeated, are produced
// in HBM.  Pure byte-moving work: one workgroup per (output row, image line), transposed through LDS so that both the raw
// read ([x][t], t fastest) and the fp32 write ([l][..][x][c]) are coalesced.
#include "common.h"

namespace {

constexpr int HW = 60, L = 25;

template <typename T, int C>
__global__ __launch_bounds__(256) void assemble_kernel(const T* __restrict__ raw, const int32_t* __restrict__ src_row,
                                                       float* __restrict__ x_out, float* __restrict__ flag_out, float divisor,
                                                       float offset, float post_mul, float clip_max, float clip_min,
                                                       float noise) {
  constexpr int TT = L * C;                 // raw planes per pixel
  __shared__ float sv[HW * TT];             // decoded line [x][t]
  const int row = blockIdx.y, y = blockIdx.x, tid = threadIdx.x;
  const int src = src_row[row];
  float* dst = x_out + (size_t)row * L * HW * HW * C;
  if (y == 0 && tid == 0) flag_out[row] = src >= 0 ? 1.f : 0.f;
  if (src < 0) {   // absent or disabled modality: the whole row is `noise`
    for (int e = tid; e < L * HW * C; e += 256) {
      const int l = e / (HW * C), rem = e - l * (HW * C);
      dst[((size_t)l * HW + y) * HW * C + rem] = noise;
    }
    return;
  }
  const T* line = raw + ((size_t)src * HW + y) * HW * TT;
  for (int e = tid; e < HW * TT; e += 256) {
    float v = (float)line[e];
    if (clip_max > 0.f && fabsf(v) > clip_max) v = 1e-8f;
    if (clip_min > 0.f && fabsf(v) < clip_min) v = 1e-8f;
    v = v / divisor;
    if (post_mul != 1.f) v = v * post_mul;
    sv[e] = v - offset;
  }
  __syncthreads();
  for (int e = tid; e < L * HW * C; e += 256) {
    const int l = e / (HW * C), rem = e - l * (HW * C);
    const int x = rem / C, c = rem - x * C;
    dst[((size_t)l * HW + y) * HW * C + rem] = sv[x * TT + l * C + c];
  }
}

// Row gather / scatter of the mask-skipping step (GaitCore(skip_masked=True)): a modality's encoder runs on the clips whose flag is 1
// only, so their rows are gathered into a dense batch on the way in and the branch outputs / gradients scattered back on the way out.
// t is [outer][rows][row_floats] fp32 (outer = 1 for the clip tensors, 62 for the [62,B,256] features); pure byte-moving, 16 bytes per
// lane, one workgroup per (outer, gathered row) striding over the row.
template <bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                        const int64_t* __restrict__ idx, int nidx, int src_rows, int dst_rows, size_t row_f4) {
  const int o = blockIdx.y, i = blockIdx.x;
  const int64_t r = idx[i];
  const float4* s = src + ((size_t)o * src_rows + (SCATTER ? (size_t)i : (size_t)r)) * row_f4;
  float4* d = dst + ((size_t)o * dst_rows + (SCATTER ? (size_t)r : (size_t)i)) * row_f4;
  for (size_t e = threadIdx.x; e < row_f4; e += 256) d[e] = s[e];
}

}  // namespace

static int move_rows(const float* src, const int64_t* idx, float* dst, int outer, int src_rows, int dst_rows, int nidx, size_t row_floats,
                     bool scatter, void* stream, const char* what) {
  UGN_REQUIRE(src && idx && dst && outer > 0 && nidx > 0 && src_rows > 0 && dst_rows > 0, "%s: null pointer or empty shape", what);
  UGN_REQUIRE(row_floats % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "%s: rows must be whole, 16-byte aligned float4s", what);
  UGN_REQUIRE(outer <= 65535, "%s: outer extent %d too large", what, outer);
  const dim3 grid(nidx, outer), block(256);
  if (scatter)
    hipLaunchKernelGGL(move_rows_kernel<true>, grid, block, 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, idx, nidx, src_rows,
                       dst_rows, row_floats / 4);
  else
    hipLaunchKernelGGL(move_rows_kernel<false>, grid, block, 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, idx, nidx, src_rows,
                       dst_rows, row_floats / 4);
  UGN_CHECK_LAUNCH(what);
  return 0;
}

extern "C" int ugn_gather_rows(const float* src, const int64_t* idx, float* dst, int outer, int src_rows, int nidx, size_t row_floats,
                               void* stream) {
  return move_rows(src, idx, dst, outer, src_rows, nidx, nidx, row_floats, false, stream, "ugn_gather_rows");
}

extern "C" int ugn_scatter_rows(const float* src, const int64_t* idx, float* dst, int outer, int dst_rows, int nidx, size_t row_floats,
                                void* stream) {
  return move_rows(src, idx, dst, outer, nidx, dst_rows, nidx, row_floats, true, stream, "ugn_scatter_rows");
}

extern "C" int ugn_assemble_modality(const void* raw, int is_int16, const int32_t* src_row, int nrows, int channels,
                                     float divisor, float offset, float post_mul, float clip_max, float clip_min, float noise,
                                     float* x_out, float* flag_out, void* stream) {
  UGN_REQUIRE(src_row && x_out && flag_out && nrows > 0, "ugn_assemble_modality: null pointer or nrows <= 0");
  UGN_REQUIRE(channels == 1 || channels == 2, "ugn_assemble_modality: channels must be 1 or 2 (got %d)", channels);
  UGN_REQUIRE(divisor != 0.f, "ugn_assemble_modality: divisor is 0");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(HW, nrows), block(256);
#define UGN_ASM(T_, C_)                                                                                                 \
  hipLaunchKernelGGL((assemble_kernel<T_, C_>), grid, block, 0, st, (const T_*)raw, src_row, x_out, flag_out, divisor, \
                     offset, post_mul, clip_max, clip_min, noise)
  if (is_int16) {
    if (channels == 2) UGN_ASM(int16_t, 2); else UGN_ASM(int16_t, 1);
  } else {
    if (channels == 2) UGN_ASM(uint8_t, 2); else UGN_ASM(uint8_t, 1);
  }
#undef UGN_ASM
  UGN_CHECK_LAUNCH("assemble_modality");
  return 0;
}
