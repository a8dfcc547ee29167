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

// Output channel n (within 32) of a wave that owns two 16-column accumulator blocks <-> (block cb, MFMA column lj).
// UGN_EPI_PAIR (default): n = 2*lj + cb -- the two values a lane holds for one pixel are adjacent channels.
#ifndef UGN_EPI_PAIR
#define UGN_EPI_PAIR 1
#endif
__host__ __device__ constexpr int pair_lj(int n) { return UGN_EPI_PAIR ? (n >> 1) & 15 : n & 15; }
__host__ __device__ constexpr int pair_cb(int n) { return UGN_EPI_PAIR ? n & 1 : (n >> 4) & 1; }

// filter layouts (decided from the GEMM dimensions alone, so that packer and kernels agree):
//   tall  : 32 output channels            -> conv3x3_wino_tall.hip
//   wide  : >= 64 output and K channels   -> wino_kernel, wide variant
//   narrow: everything else               -> wino_kernel, narrow variant
__host__ __device__ constexpr bool wino_tall(int kc, int nc) { return nc == 32; }
// (128 -> 64 channels at 16x16 stays narrow: wide would leave 600 frame-level items for 256 persistent workgroups, 2.34
//  rounds of which the last is a third full; narrow makes 1200 items = 4.7 rounds, and 48 instead of 24 for the set-level twin)
// Measured on merged 3-modality launches (tools/ab_ops.py, 1872 frames): 32 -> 64 forward wide 448 us against 473 narrow; the
// 128 -> 64 data gradient wide 392 against 432 (1872 wide items over 256 workgroups quantise to 91 %, but an item is that
// much cheaper; with one modality per launch -- 600 items, 2.34 rounds -- narrow was the better of the two).
__host__ __device__ constexpr bool wino_wide(int kc, int nc) { return nc >= 64 && kc >= 32; }
// bf16 operands: the data gradient of a POOLED 64 -> 64 layer (a4 / b2) runs narrow -- the wide pooled variant sits at 254
// registers in fp32 and spills with the zero-padded operand pairs of the bf16 form
// (bf16 operands also keep the two shapes narrow whose wide form spills with the zero-padded operand pairs)
__host__ __device__ constexpr bool wino_wide_ex(int kc, int nc, bool bf, bool pooled_dgrad) {
  return wino_wide(kc, nc) && !(bf && pooled_dgrad && kc == 64 && nc == 64) && !(bf && (kc == 32 || (kc == 128 && nc == 64)));
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  // lane l: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D reg r of lane l = D[i = 4*(l>>4) + r][j = l&15]
#if defined(UGN_ABLATE) && (UGN_ABLATE & 8)
  c[0] += a * b;      // (timing-only ablation: one VALU keeps the operands alive)
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
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

// Packed fp32 arithmetic.  The Winograd transforms are sums and differences of register PAIRS (a lane owns two channels of
// every patch element); written on a 2-vector type they become v_pk_add_f32 (both halves in one full-rate instruction, the
// subtraction as a neg modifier) -- half the vector instructions of the transforms, whose issue time adds to the MFMA time.
#ifndef UGN_PK
#define UGN_PK 1
#endif
// (2-vector arithmetic that the compiler turns into the packed instruction; it splits part of it back into scalar
// instructions in the tall kernel.  Forcing the instruction with inline asm is NOT an option next to MFMAs: the wait states
// of the MFMA hazards -- a vector instruction reading a fresh MFMA result, or overwriting the accumulator input of an MFMA
// in flight -- are inserted by the compiler for instructions it can see only; the asm form computed garbage.)
typedef float v2f __attribute__((ext_vector_type(2)));
template <int PK = UGN_PK>
__device__ __forceinline__ float2 pk_add(float2 a, float2 b) {
  if constexpr (PK) {
    const v2f r = v2f{a.x, a.y} + v2f{b.x, b.y};
    return make_float2(r.x, r.y);
  } else {
    return make_float2(a.x + b.x, a.y + b.y);
  }
}
template <int PK = UGN_PK>
__device__ __forceinline__ float2 pk_sub(float2 a, float2 b) {
  if constexpr (PK) {
    const v2f r = v2f{a.x, a.y} - v2f{b.x, b.y};
    return make_float2(r.x, r.y);
  } else {
    return make_float2(a.x - b.x, a.y - b.y);
  }
}

// Output transform Y = A^T M A of the lane's accumulators: acc[cb][pt] holds point pt of the lane's 4 tiles (one register
// each) for channel block cb -> y[cb][tile r][2x2 output, row-major].  The tiles r = 0..3 are the 4 registers of an
// accumulator, so the packed form handles two tiles per instruction.
template <int NB, int PK>
__device__ __forceinline__ void wino_out_transform(const f32x4 (&acc)[NB][16], float (&y)[NB][4][4]) {
#pragma unroll
  for (int rp = 0; rp < 2; ++rp)
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      auto m = [&](int pt) { return make_float2(acc[cb][pt][2 * rp], acc[cb][pt][2 * rp + 1]); };
      float2 sm[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sm[0][c] = pk_add<PK>(pk_add<PK>(m(0 * 4 + c), m(1 * 4 + c)), m(2 * 4 + c));
        sm[1][c] = pk_sub<PK>(pk_sub<PK>(m(1 * 4 + c), m(2 * 4 + c)), m(3 * 4 + c));
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const float2 y0 = pk_add<PK>(pk_add<PK>(sm[a][0], sm[a][1]), sm[a][2]);
        const float2 y1 = pk_sub<PK>(pk_sub<PK>(sm[a][1], sm[a][2]), sm[a][3]);
        y[cb][2 * rp][a * 2 + 0] = y0.x;
        y[cb][2 * rp + 1][a * 2 + 0] = y0.y;
        y[cb][2 * rp][a * 2 + 1] = y1.x;
        y[cb][2 * rp + 1][a * 2 + 1] = y1.y;
      }
    }
}

// One ds_read_b64 that the compiler cannot fuse.  The halo tiles are laid out bank-exact for ds_read_b64 (64 banks, 32 lanes
// per pass), but hipcc fuses neighbouring 8-byte reads into ds_read2_b64 / ds_read2st64_b64, which bank mod 32 at half rate and
// hit this layout 2-way: 16 LDS cycles per pair of reads instead of 4, on the operand every transform waits for (removing the
// patch reads altogether saves 15-20 % of a forward kernel, tools/stamps.py / UGN_ABLATE).  The asm form keeps them apart with
// ONE address register and immediate offsets.  The compiler does not see the load: patch_wait() must come before the first use.
// (A volatile load keeps the reads apart as well, but pins all 16 in program order: 57-152 spilled registers in the tall kernel,
// +8 % on the pooled data gradient.)
#ifndef UGN_B64ASM
#define UGN_B64ASM 1
#endif
template <int OFF_BYTES>
__device__ __forceinline__ float2 lds_read_b64(unsigned lds_byte_addr) {
  float2 r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "i"(OFF_BYTES));
  return r;
}
// s_waitcnt lgkmcnt(0) that the 16 patch registers depend on (so that no use can be scheduled above it)
__device__ __forceinline__ void patch_wait(float2 (&d)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]),
                 "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// even pixel columns of a halo row first, then the odd ones (bank layout, see conv3x3_wino.hip)
__device__ __forceinline__ constexpr int colpos(int col) { return (col & 1) ? 9 + (col >> 1) : (col >> 1); }

// Epilogue shared by wino_kernel and wino_tall_kernel.  y[cb][r][q]: the lane's outputs -- block cb, tile r (of its 4),
// position q (row-major in the 2x2 tile); o[r][q]: element offset (inside the image) of the lane's FIRST channel at that
// pixel (pooled epilogue: o[r][0] = the pooled pixel).  With two blocks per lane and the paired channel mapping (pair_lj /
// pair_cb) the lane's two channels are adjacent: every access moves 8 bytes per lane, i.e. whole 128-byte lines per 16 lanes
// and half the memory instructions; otherwise block cb sits 16 channels further.
template <int NB, int EPI, int EFLAGS>
__device__ __forceinline__ void wino_epilogue(const float (&y)[NB][4][4], const unsigned (&o)[4][4], float* __restrict__ out,
                                              uint8_t* __restrict__ out_idx, const float* __restrict__ act,
                                              const float* __restrict__ addend, float* __restrict__ raw_out,
                                              const float* __restrict__ sm_m, const float* __restrict__ sm_g) {
  constexpr int W = (NB == 2 && UGN_EPI_PAIR) ? 2 : 1;      // channels per access
  constexpr int NBLK = NB / W;
  auto ld = [](const float* p, unsigned off, float (&v)[W]) {
    if constexpr (W == 2) { const float2 t = *reinterpret_cast<const float2*>(p + off); v[0] = t.x; v[1] = t.y; }
    else v[0] = p[off];
  };
  auto st = [](float* p, unsigned off, const float (&v)[W]) {
    if constexpr (W == 2) *reinterpret_cast<float2*>(p + off) = make_float2(v[0], v[1]);
    else p[off] = v[0];
  };
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    const unsigned cofs = W == 2 ? 0u : (unsigned)blk * 16u;
    if constexpr (EPI == EPI_LRELU_POOL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float best[W];
        unsigned bi[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
          // LeakyReLU is increasing, also after rounding: the window is pooled on the raw sums and the activation applied to
          // the winner only (a quarter of the activation work).  Equal sums stay equal, so exact ties route as before; only two
          // DIFFERENT negative sums whose products with 0.3 round to the same float now prefer the larger instead of the first.
          best[e] = y[blk * W + e][r][0];
          bi[e] = 0;
#pragma unroll
          for (int q = 1; q < 4; ++q) {
            const float v = y[blk * W + e][r][q];
            if (v > best[e]) { best[e] = v; bi[e] = q; }     // strict >: the FIRST maximum wins (TF MaxPoolGrad routing)
          }
          best[e] = ugn_lrelu(best[e]);
        }
        st(out, o[r][0] + cofs, best);
        if constexpr (W == 2) *reinterpret_cast<uint16_t*>(out_idx + o[r][0]) = (uint16_t)(bi[0] | (bi[1] << 8));
        else out_idx[o[r][0] + cofs] = (uint8_t)bi[0];
      }
    } else if constexpr (EPI == EPI_LRELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[W];
#pragma unroll
          for (int e = 0; e < W; ++e) v[e] = ugn_lrelu(y[blk * W + e][r][q]);
          st(out, o[r][q] + cofs, v);
        }
    } else {
      float av[4][4][W], dv[4][4][W], mv[4][4][W], gv[4][4][W];
      if constexpr (EFLAGS & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) ld(act, o[r][q] + cofs, av[r][q]);
      }
      if constexpr (EFLAGS & 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) ld(addend, o[r][q] + cofs, dv[r][q]);
      }
      if constexpr (EFLAGS & 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ld(sm_m, o[r][q] + cofs, mv[r][q]);
            ld(sm_g, o[r][q] + cofs, gv[r][q]);
          }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[W];
#pragma unroll
          for (int e = 0; e < W; ++e) {
            v[e] = y[blk * W + e][r][q];
            if constexpr (EFLAGS & 2) v[e] += dv[r][q][e];
            if constexpr (EFLAGS & 8) v[e] += av[r][q][e] == mv[r][q][e] ? gv[r][q][e] : 0.f;   // set-max gradient -> the frames holding the maximum
          }
          if constexpr (EFLAGS & 4) st(raw_out, o[r][q] + cofs, v);
          if constexpr (EFLAGS & 1) {
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] *= ugn_lrelu_slope(av[r][q][e]);
          }
          st(out, o[r][q] + cofs, v);
        }
    }
  }
}

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
