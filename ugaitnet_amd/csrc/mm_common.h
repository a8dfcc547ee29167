// "H2" tensors and the kernels that multiply them on the f16 matrix pipe (round 3).
//
// Why.  v_mfma_f32_*_f32 issues at the fp32 VECTOR rate (64 FLOP/clk/SIMD) and blocks the SIMD's vector issue while it runs,
// so the Winograd fp32 kernels top out where MFMA time + everything else ADD UP.  The f16 matrix pipe is
// 16x faster and runs beside the VALU.  An fp32 value split into two halves
//        x = H + L,   H = f16(x),   L = f16(x - H)            (x - H is exact in fp32; 22 significant bits survive)
// multiplies as  a*b = aH*bH + aH*bL + aL*bH  (+ aL*bL ~ 2^-22 |ab|, dropped): three v_mfma_f32_32x32x16_f16 with fp32
// accumulation.  Measured on MI355X (tools/probe_split.hip, K = 288 ... 4608): error 0.87e-7 of sum|ab| against 1.0-1.5e-7 for
// the exact fp32 MFMA chain -- fp32-class results at 3/16 of the fp32 matrix time, and no transform arithmetic at all when the
// halves are what is STORED: the producer's epilogue splits once, every consumer feeds ds_read_b128 results to the MFMA.
//
// Format.  A tensor [pixels][C] is stored as f16 bit patterns [pixels][2][C]: plane 0 = H, plane 1 = L (4 bytes per element,
// the size and pixel pitch of the fp32 tensor it replaces), plus an H2Meta {e, amax}:  true value = (H + L) * 2^-e.
// f16 has 5 exponent bits, so the power-of-two block scale matters: the probe's 1e-6-magnitude operands lose everything
// unscaled (3.8e-2) and nothing prescaled (8e-7).  `e` is chosen by the PRODUCER from a rigorous bound of its output
// (max|in| * L1 norm of the filter for a convolution), so nothing can overflow; `amax` = max |stored value|, gathered by the
// producer with one atomicMax per wave and item (order-independent: results stay bitwise reproducible), tells the next
// kernel the true range of its input, so the looseness of a bound never accumulates over layers.
// Stored maxima sit at or below 2^15 (f16 max 65504); elements down to 2^-18 of the tensor's bound keep all 22 bits, smaller
// ones degrade gracefully to an absolute error of 2^-40 of the bound (f16 subnormals are not flushed by v_cvt or the MFMA).
#pragma once
#include "common.h"

namespace ugn_mm {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct H2Meta {
  int e;            // stored = true * 2^e
  unsigned amax;    // bits of max |stored| (a non-negative float: unsigned order = float order); zeroed once per step
};
struct WMeta {
  int e;            // packed filter halves = w * 2^e
  float l1;         // max over outputs of sum |w| over everything an output sums: |conv(x)| <= max|x| * l1
};

constexpr int kStoredLog2 = 15;    // stored values stay below 2^15

__device__ __forceinline__ float h2_true_amax(int e, unsigned amax_bits) { return ldexpf(__uint_as_float(amax_bits), -e); }

// exponent that brings a tensor bounded by b below 2^15 (0 for an all-zero or non-finite bound)
__device__ __forceinline__ int h2_exp_for_bound(float b) {
  if (!(b > 0.f) || !(b < 3.0e38f)) return 0;
  int x;
  (void)frexpf(b, &x);            // b = m * 2^x, 0.5 <= m < 1
  int e = kStoredLog2 - x;
  return e > 100 ? 100 : (e < -100 ? -100 : e);
}

__device__ __forceinline__ void h2_split(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;                     // v_cvt_f16_f32, round to nearest even
  lo = (_Float16)(v - (float)hi);       // exact residual, rounded once
}
__device__ __forceinline__ unsigned h2_bits(_Float16 a) { return (unsigned)__builtin_bit_cast(unsigned short, a); }
__device__ __forceinline__ unsigned h2_pack(_Float16 a, _Float16 b) { return h2_bits(a) | (h2_bits(b) << 16); }
__device__ __forceinline__ float h2_half(unsigned bits16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16); }
// stored value of element 0 / 1 of a (hi pair, lo pair)
__device__ __forceinline__ float h2_join0(unsigned hi2, unsigned lo2) { return h2_half(hi2 & 0xffffu) + h2_half(lo2 & 0xffffu); }
__device__ __forceinline__ float h2_join1(unsigned hi2, unsigned lo2) { return h2_half(hi2 >> 16) + h2_half(lo2 >> 16); }

// max over the wave, result in every lane
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ void h2_publish_amax(H2Meta* m, float wave_amax, int lane) {
  if (lane == 0 && wave_amax > 0.f) atomicMax(&m->amax, __float_as_uint(wave_amax));
}
// the same for a whole workgroup through `scratch` (one float per wave, in LDS): ONE atomic per workgroup -- atomics on one
// address serialise at ~10 ns each, so 12 k of them (a wave each, 6 jobs) are 0.1 ms at the tail of a launch
__device__ __forceinline__ void h2_publish_amax_block(H2Meta* m, float v, float* scratch, int tid, int nwaves) {
  const float w = wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = w;
  __syncthreads();
  if (tid == 0) {
    float a = scratch[0];
    for (int k = 1; k < nwaves; ++k) a = fmaxf(a, scratch[k]);
    if (a > 0.f) atomicMax(&m->amax, __float_as_uint(a));
  }
}

// global -> LDS, 16 B per lane (LDS address = M0 + lane * 16); see wino_common.h dma16 for why this is inline asm
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  // (the destination is wave-uniform by construction; readfirstlane makes that provable where the compiler's divergence analysis
  //  gives up -- it folds away when the value already lives in an SGPR)
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// One launch serves up to kMaxJobs convolutions of one shape (frame-level layer + set-level twin, every modality).
constexpr int kMaxJobs = 6;
constexpr int kGrid = 256;
struct MmJob {
  const uint16_t* in;        // H2 [n][hw][hw][2][kc]   (pooled input: [n][hw/2][hw/2][2][kc])
  const uint8_t* in_idx;     // pooled input: argmax bytes [n][hw/2][hw/2][kc]
  const H2Meta* in_meta;
  const uint16_t* wpk;       // packed filter halves (mm_pack)
  const WMeta* wmeta;
  uint16_t* out;             // H2 [n][ho][ho][2][nc]
  uint8_t* out_idx;          // pooled epilogue: argmax bytes
  H2Meta* out_meta;
  const uint16_t* act;       // data gradient: the layer's input activation (LeakyReLU' from the sign of its H plane)
};
struct MmJobs {
  MmJob job[kMaxJobs];
  int start[kMaxJobs + 1];
};
__device__ __forceinline__ int mm_job_of(const MmJobs& jt, int it) {
  int jb = 0;
#pragma unroll
  for (int j = 1; j < kMaxJobs; ++j) jb += it >= jt.start[j] ? 1 : 0;
  return jb;
}
inline int make_mm_table(MmJobs& jt, const MmJob* jobs, const int* n, int njobs, int per_img) {
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

// output channel <-> (32-column MFMA block nb, column j) of a workgroup that owns nc channels: with two or more blocks the
// blocks 2m, 2m+1 hold the even / odd channels of group m, so a lane owns ADJACENT channels 64m + 2j, 64m + 2j + 1 and packs
// their halves into whole dwords without cross-lane traffic.
__host__ __device__ constexpr int mm_block_of(int n, int nc) { return nc == 32 ? 0 : 2 * (n >> 6) + (n & 1); }
__host__ __device__ constexpr int mm_col_of(int n, int nc) { return nc == 32 ? n : (n & 63) >> 1; }

const void* zero_block();   // 256 B of zeros in HBM (conv3x3_mm.hip)
// Persistent workgroups per launch (ugn_set_persistent_wgs; default kGrid = one per CU).  Every persistent launch of the library
// that owns its CUs' LDS sizes its grid from it: f16x2 and bf16 forward / data gradient (items stride over the grid), weight gradients
// (groups per block combination on the grid, a multiple of 8; the launch's shares stay fixed), the 5x5 forward (4 workgroups per CU).
int persistent_wgs();
// groups per block combination of a weight-gradient launch: the largest multiple of 8 with groups * ncombo <= persistent_wgs()
inline int persistent_groups(int ng_full, int ncombo) {
  int g = (persistent_wgs() / ncombo) / 8 * 8;
  return g < 8 ? 8 : (g > ng_full ? ng_full : g);
}

}  // namespace ugn_mm
