// Pieces shared by the Winograd F(2x2,3x3) forward / data-gradient kernels (conv3x3_wino.hip, conv3x3_wino_tall.hip).
#pragma once
#include "common.h"

namespace ugn_wino {

enum { EPI_LRELU = 0, EPI_LRELU_POOL = 1, EPI_DGRAD = 2 };

// One launch serves up to kMaxJobs convolutions of the same shape -- the frame-level layer and its set-level twin of the
// global branch (which alone would leave most CUs idle), for every modality of the model: the three encoders run the same
// ten layer shapes, so a 3-modality step issues one launch per layer instead of six.  Items [start[j], start[j+1]) belong
// to job j.
struct WinoJob {
  const float* in;
  const uint8_t* in_idx;
  const float* upk;
  float* out;
  uint8_t* out_idx;
  const float* act;
  const float* addend;
  float* raw_out;
  // "routed" addend (epilogue flag 8, needs act): the set-max gradient of the layer's output, formed on the fly as
  // (act == smax_m[clip]) ? smax_g[clip] : 0 with clip = image / frames  (smax_g = dL/dm / #maxima)
  const float* smax_m;
  const float* smax_g;
  int frames;
};

constexpr int kGrid = 256;   // persistent workgroups (one per CU: the kernels use the whole LDS)
constexpr int kMaxJobs = 6;

// The job table travels in the kernel-argument segment; a job is looked up per ITEM (wave-uniform: scalar loads).
struct WinoJobs {
  WinoJob job[kMaxJobs];
  int start[kMaxJobs + 1];   // first item of job j; entries from the job count onwards hold the total item count
};

__device__ __forceinline__ int wino_job_of(const WinoJobs& jt, int it) {
  int jb = 0;
#pragma unroll
  for (int j = 1; j < kMaxJobs; ++j) jb += it >= jt.start[j] ? 1 : 0;
  return jb;
}

// filter layouts (decided from the GEMM dimensions alone, so that packer and kernels agree):
//   tall  : 32 output channels            -> conv3x3_wino_tall.hip
//   wide  : >= 64 output and K channels   -> wino_kernel, wide variant
//   narrow: everything else               -> wino_kernel, narrow variant
__host__ __device__ constexpr bool wino_tall(int kc, int nc) { return nc == 32; }
// (128 -> 64 channels at 16x16 stays narrow: wide would leave 600 frame-level items for 256 persistent workgroups, 2.34
//  rounds of which the last is a third full; narrow makes 1200 items = 4.7 rounds, and 48 instead of 24 for the set-level twin)
__host__ __device__ constexpr bool wino_wide(int kc, int nc) { return nc >= 64 && kc >= 64 && !(kc == 128 && nc == 64); }
// bf16 operands: the data gradient of a POOLED 64 -> 64 layer (a4 / b2) runs narrow -- the wide pooled variant sits at 254
// registers in fp32 and spills with the zero-padded operand pairs of the bf16 form
__host__ __device__ constexpr bool wino_wide_ex(int kc, int nc, bool bf, bool pooled_dgrad) {
  return wino_wide(kc, nc) && !(bf && pooled_dgrad && kc == 64 && nc == 64);
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  // lane l: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D reg r of lane l = D[i = 4*(l>>4) + r][j = l&15]
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// bf16-operand variant (template flag BF of the kernels; SURVEY 8(d) "C5": bf16 operands in the MFMA, fp32 accumulate, fp32
// tensors in HBM).  A lane owns 4 (narrow) or 2 (wide / tall) channels of a group: their transformed values are rounded to
// bf16 and fed to ONE v_mfma_f32_16x16x16_bf16 (k-slot i of lane kq = the lane's i-th channel; unused slots hold zeros on both
// operands) in place of 4 / 2 v_mfma_f32_16x16x4_f32; the filters are packed as bf16 in the same element order
// (ugn_wino_pack mode bit 4), which halves their LDS-DMA traffic.  D layout as mfma16.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
  bf16x2_t r;
  r[0] = (__bf16)a;
  r[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ f32x4 mfma_bf16(uint32_t a01, uint32_t a23, uint32_t b01, uint32_t b23, f32x4 c) {
  const uint2 a = make_uint2(a01, a23), b = make_uint2(b01, b23);
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4_t, a), __builtin_bit_cast(bf16x4_t, b), c, 0, 0, 0);
}

// global -> LDS, 16 B per lane, no VGPR destination (LDS address = M0 + lane * 16).  Inline asm: with the builtin hipcc
// waits vmcnt(0) before the next ds_read (it cannot tell the DMA target from the buffers being read), which exposes the
// whole global latency.  Completion is awaited by the caller (s_waitcnt vmcnt) before the barrier that publishes the data.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst_uniform)
               : "memory");
}

// same, 4 B per lane (LDS address = M0 + lane * 4)
__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst_uniform)
               : "memory");
}

// even pixel columns of a halo row first, then the odd ones (bank layout, see conv3x3_wino.hip)
__device__ __forceinline__ constexpr int colpos(int col) { return (col & 1) ? 9 + (col >> 1) : (col >> 1); }

// 256 B of zeros in HBM: the LDS-DMA source for halo lanes outside the image (defined in conv3x3_wino.hip)
const float* zero_block();

// tall variant (conv3x3_wino_tall.hip); kind: 0 forward, 1 data gradient
bool tall_supported(int kind, int hw, int kc, int unpool_or_pool);
int launch_tall(int kind, const WinoJob* jobs, const int* n, int njobs, int hw, int kc, int unpool_or_pool, bool bf, hipStream_t st);

// host: the table of a launch (per_img items per image); returns the item count
inline int make_job_table(WinoJobs& jt, const WinoJob* jobs, const int* n, int njobs, int per_img) {
  int total = 0;
  for (int j = 0; j < kMaxJobs; ++j) {
    jt.job[j] = jobs[j < njobs ? j : njobs - 1];
    jt.start[j] = total;
    if (j < njobs) total += n[j] * per_img;
  }
  jt.start[kMaxJobs] = total;
  for (int j = njobs; j < kMaxJobs; ++j) jt.start[j] = total;
  return total;
}

}  // namespace ugn_wino
