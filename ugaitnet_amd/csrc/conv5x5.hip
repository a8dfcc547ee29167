// First encoder layer: ZeroPadding2D(2) + Conv2D(32, 5x5, SAME, no bias) + LeakyReLU(0.3) on the raw 60x60 frames
// (reference nets/mj_uwyhNets_ba.py:428-430) and its weight gradient.  The input has 1 or 2 channels, so the
// GEMM K is only 25 or 50: the kernel is bound by writing a1 (13.1 MB per clip), not by arithmetic.  It still runs
// on v_mfma_f32_32x32x2_f32 so that the accumulator layout (32 consecutive output channels per half wave) gives
// 128-byte coalesced NHWC stores with no LDS transpose.
//
// Domain: the explicit 2-pixel zero ring makes the conv domain 64x64; input pixel (u,v) of that domain is raw pixel
// (u-2,v-2) or 0.  The 5x5 SAME conv then reads (y+dy-2, x+dx-2), dy,dx in 0..4.
#include "mm_common.h"
#include "x3_common.h"

namespace {

using ugn_mm::H2Meta;

#ifndef UGN_C5_WAVES
#define UGN_C5_WAVES 4     // waves per SIMD the forward kernel is compiled for (measured: 4 and 6 equal, 8 spills)
#endif
#ifndef UGN_WITH_H2
#define UGN_WITH_H2 0     // 1: also build the entry points of the opt-in f16x2 set (build.py --h2)
#endif
#if UGN_WITH_H2
#include "../../include/ugaitnet_hip_h2.h"
#endif
#ifndef UGN_C5_H2MM
#define UGN_C5_H2MM 1     // H2 output: multiply on the f16 matrix pipe (x and w split into f16 halves on the fly) instead of the fp32 MFMA
#endif
#ifndef UGN_C5_GRID
#define UGN_C5_GRID 1024   // persistent workgroups of the forward kernel (256 CUs x 4 workgroups of 256 threads)
#endif
#ifndef UGN_C5_PADK
#define UGN_C5_PADK(K_, mb_) ((mb_) == 0 ? 0 : (K_) - 1)
#endif
#ifndef UGN_C5_PITCH
#define UGN_C5_PITCH 20  // pixels per patch row in the weight-gradient kernel's LDS tile: 26 would be bank-conflict free
                         // for the five tap rows, but its extra LDS-DMA pieces cost more than the conflicts (+2-3 %)
#endif
#ifndef UGN_C5_DS
#define UGN_C5_DS 32     // floats per pixel of the gradient tile in LDS (36 = padded, the earlier layout)
#endif
// pixels per patch row of the weight-gradient kernel's LDS tile.  Two input channels: 21 -- the lanes of an MFMA row block are
// (tap, channel) pairs, four tap rows of ten floats each: a row stride of 42 floats puts them on forty distinct banks (40 is a
// 2-way conflict: SQ_LDS_BANK_CONFLICT 0.37 of the active cycles), for one more 256-byte DMA piece in 14.
constexpr int c5_pitch(int cin) { return cin == 2 ? 21 : UGN_C5_PITCH; }
constexpr int T5 = 16;          // 16x16 output tile
constexpr int P5 = T5 + 4;      // 20x20 input patch
constexpr int RAW = 60, DOM = 64;

template <int CIN>
__device__ __forceinline__ void stage_patch(float* sP, const float* __restrict__ x, int img, int ty0, int tx0, int tid) {
  // patch pixel (yy,xx) = domain (ty0-2+yy, tx0-2+xx) = raw (ty0-4+yy, tx0-4+xx)
  for (int e = tid; e < P5 * P5; e += 256) {
    const int yy = e / P5, xx = e % P5;
    const int ry = ty0 - 4 + yy, rx = tx0 - 4 + xx;
    const bool ok = ry >= 0 && ry < RAW && rx >= 0 && rx < RAW;
    const size_t o = (((size_t)img * RAW + ry) * RAW + rx) * CIN;
    if constexpr (CIN == 1) {
      sP[e] = ok ? x[o] : 0.f;
    } else {
      float2 v = make_float2(0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float2*>(x + o);
      *reinterpret_cast<float2*>(sP + 2 * e) = v;
    }
  }
}

// Persistent form: gridDim.x workgroups stride over the n * 16 tiles.  The filter is staged once per workgroup; the raw patch
// of the NEXT tile is fetched into registers before the current tile is multiplied and written to LDS after it, so the
// global-load latency of the staging hides behind the MFMAs and the stores of the tile before.
// H2OUT: a1 leaves as an H2 tensor (mm_common.h: f16 halves [pixel][2][32] + block exponent) for the f16-matrix-pipe 3x3
// kernels.  x_meta = {0, bits(max|x|)} (ugn_absmax_multi); the exponent comes from the bound max|x| * max_co sum_k |w[k][co]|,
// which every workgroup forms from the filter it has just staged; the stored maximum is gathered per workgroup.
// OFMT: 0 fp32 a1, 1 H2 (above), 2 bf16 (configs[4]: [pixel][32] bf16, no exponent), 3 fp32 a1 multiplied in the x3 arithmetic
// (x3_common.h: the default fp32-tensor set since round 5) -- the gathered patch fragment and the filter split into three bf16
// planes, six v_mfma_f32_32x32x16_bf16 per k-step of 16 taps: 12 / 24 matrix instructions of 8 issue cycles per 32-pixel block
// instead of 13 / 25 fp32 ones of 64 cycles that hold the vector pipe (the fp32 form measured 45-63 % MFMA busy: the layer was
// bound by its fp32 MFMAs, not by writing a1)
template <int CIN, bool SIGN, int OFMT = 0>
__global__ __launch_bounds__(256, (OFMT == 1 && UGN_C5_H2MM) ? 3 : UGN_C5_WAVES) void conv5x5_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                        float* __restrict__ a1, uint32_t* __restrict__ sign_out,
                                                                        int ntiles, const H2Meta* __restrict__ x_meta = nullptr,
                                                                        H2Meta* __restrict__ out_meta = nullptr) {
  constexpr int K = 25 * CIN, KP = (K + 1) / 2;  // k-pairs
  constexpr int PE = P5 * P5;                     // patch pixels
  constexpr int PPT = (PE + 255) / 256;           // patch pixels per thread (2)
  // Floats per patch row in LDS.  A lane half reads 2 rows x 16 pixels (x its channel): with the natural pitch (20 / 40 floats) the
  // second row falls on banks the first one uses (SQ_LDS_BANK_CONFLICT 0.33-0.36 of the active cycles); 48
  // (= 16 mod 32) and 41 (odd) put it on the other sixteen.
  constexpr int FP = CIN == 1 ? 48 : 41;
  __shared__ __attribute__((aligned(16))) float sP[P5 * FP];
  __shared__ __attribute__((aligned(16))) float sW[2 * KP * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  for (int e = tid; e < 2 * KP * 32; e += 256) sW[e] = e < K * 32 ? w[e] : 0.f;
  constexpr bool H2OUT = OFMT == 1;
  // H2MM (H2 output, default): the layer multiplies on v_mfma_f32_32x32x16_f16 like the 3x3 layers -- x * 2^ex and w * 2^ew split into
  // f16 halves (the patch once per tile while it is written to LDS, the filter once per workgroup into registers), three MFMAs per
  // product (hi*hi + hi*lo + lo*hi, fp32 accumulate): 6 / 12 MFMAs of 32 cycles per 32-pixel block instead of 13 / 25 fp32 MFMAs of
  // 64 cycles that also block the vector pipe.  The layer is then bound by writing a1, as its byte count says it should be.
  constexpr bool H2MM = H2OUT && UGN_C5_H2MM;
  float h2_factor = 1.f, h2_mx = 0.f, x_scale = 1.f, w_scale = 1.f;
  if constexpr (H2OUT) {
    __syncthreads();
    float l1 = 0.f, wmax = 0.f;
    for (int k = 0; k < K; ++k) {      // lane li = output channel (both lane halves alike)
      l1 += fabsf(sW[k * 32 + li]);
      wmax = fmaxf(wmax, fabsf(sW[k * 32 + li]));
    }
    l1 = ugn_mm::wave_max(l1) * 1.0001f;
    const float xmax = ugn_mm::h2_true_amax(x_meta->e, x_meta->amax);
    const int e_out = ugn_mm::h2_exp_for_bound(xmax * l1);
    h2_factor = ldexpf(1.f, e_out);
    if (blockIdx.x == 0 && tid == 0) out_meta->e = e_out;
    if constexpr (H2MM) {
      const int ex = ugn_mm::h2_exp_for_bound(xmax), ew = ugn_mm::h2_exp_for_bound(ugn_mm::wave_max(wmax));
      x_scale = ldexpf(1.f, ex);
      w_scale = ldexpf(1.f, ew);
      h2_factor = ldexpf(1.f, e_out - ex - ew);       // the accumulators hold x * w * 2^(ex + ew)
    }
  }
  // patch pixel e = tid + 256 * j of tile t: (yy, xx) = (e / 20, e % 20) -> raw pixel (ty0 - 4 + yy, tx0 - 4 + xx)
  int pyy[PPT], pxx[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + 256 * j;
    pyy[j] = e / P5;
    pxx[j] = e % P5;
  }
  float pv[PPT][CIN];
  auto fetch = [&](int tile) {
    const int img = tile >> 4, trem = tile & 15;
    const int ty0 = (trem >> 2) * T5, tx0 = (trem & 3) * T5;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int ry = ty0 - 4 + pyy[j], rx = tx0 - 4 + pxx[j];
      const bool ok = tid + 256 * j < PE && ry >= 0 && ry < RAW && rx >= 0 && rx < RAW;
      const size_t o = (((size_t)img * RAW + ry) * RAW + rx) * CIN;
      if constexpr (CIN == 1) {
        pv[j][0] = ok ? x[o] : 0.f;
      } else {
        float2 v = make_float2(0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float2*>(x + o);
        pv[j][0] = v.x;
        pv[j][1] = v.y;
      }
    }
  };
  const int py = (li >> 1) & 1, px = 2 * (li >> 2) + (li & 1);
  int pb[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) pb[m] = (2 * (wave * 2 + m) + py) * FP + px * CIN + (CIN == 2 ? lh : 0);

  // bf16 output (configs[4]): the layer runs on v_mfma_f32_32x32x16_bf16 -- operands rounded to bf16 like every other layer of that
  // path, fp32 accumulate.  K = 25 * CIN taps in k-steps of 16: a lane (pixel li, k half lh) gathers its 8 taps of a k-step from the
  // fp32 patch (offsets fixed per lane), the filter fragments (8 x bf16 per k-step) stay in registers.  2 / 4 MFMAs of 32 cycles per
  // 32-pixel block instead of 13 / 25 fp32 MFMAs of 64 cycles that block the vector pipe.
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  constexpr bool BFMM = OFMT == 2;
  constexpr bool X3MM = OFMT == 3;
  constexpr int KS = (K + 15) / 16;
  int goff[BFMM || H2MM || X3MM ? KS : 1][8];
  bf8 wfrag[BFMM ? KS : 1];
  // filter fragments as f16 halves of w * 2^ew: [k-step][plane][lane-linear 1 KB] in LDS (identical for the four waves; in registers
  // they are 8 KS VGPRs that push the two-channel kernel into scratch)
  __shared__ __attribute__((aligned(16))) uint4 sWf[(OFMT == 1 && UGN_C5_H2MM) ? KS * 2 * 64 : 1];
  // x3: the filter fragments as three bf16 planes, [k-step][plane][lane-linear 1 KB] (the same for the four waves)
  __shared__ __attribute__((aligned(16))) uint4 sWx[X3MM ? KS * 3 * 64 : 1];
  int pbb[2];
  if constexpr (BFMM || H2MM || X3MM) {
    __syncthreads();            // sW is complete
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float wq[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 16 * ks + 8 * lh + i;
        const int tap = k / CIN, ch = k % CIN;
        goff[ks][i] = k < K ? (tap / 5) * FP + (tap % 5) * CIN + ch : 0;
        const float wv = k < K ? sW[k * 32 + li] : 0.f;
        wq[i] = wv;
        if constexpr (BFMM) wfrag[ks][i] = (__bf16)wv;
        if constexpr (H2MM) {
          if (wave == 0) {
            _Float16 hi, lo;
            ugn_mm::h2_split(wv * w_scale, hi, lo);
            reinterpret_cast<_Float16*>(sWf)[((ks * 2 + 0) * 64 + lane) * 8 + i] = hi;
            reinterpret_cast<_Float16*>(sWf)[((ks * 2 + 1) * 64 + lane) * 8 + i] = lo;
          }
        }
      }
      if constexpr (X3MM) {
        if (wave == 0) {
          uint4 q0, q1, q2;
          ugn_x3::split8(make_float4(wq[0], wq[1], wq[2], wq[3]), make_float4(wq[4], wq[5], wq[6], wq[7]), q0, q1, q2);
          sWx[(ks * 3 + 0) * 64 + lane] = q0;
          sWx[(ks * 3 + 1) * 64 + lane] = q1;
          sWx[(ks * 3 + 2) * 64 + lane] = q2;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) pbb[m] = (2 * (wave * 2 + m) + py) * FP + px * CIN;
  }

  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();            // the previous tile's MFMAs have read sP
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      if (tid + 256 * j < PE) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          if constexpr (H2MM) {       // the patch as packed halves (H | L << 16) of x * 2^ex: split ONCE per value, not per gather
            _Float16 hi, lo;
            ugn_mm::h2_split(pv[j][c] * x_scale, hi, lo);
            reinterpret_cast<unsigned*>(sP)[pyy[j] * FP + pxx[j] * CIN + c] = ugn_mm::h2_pack(hi, lo);
          } else {
            sP[pyy[j] * FP + pxx[j] * CIN + c] = pv[j][c];
          }
        }
      }
    __syncthreads();
    const int nt = tile + (int)gridDim.x;
    if (nt < ntiles) fetch(nt);
    const int img = tile >> 4, trem = tile & 15;
    const int ty0 = (trem >> 2) * T5, tx0 = (trem & 3) * T5;
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    if constexpr (BFMM) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          bf8 a;
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = (__bf16)sP[pbb[m] + goff[ks][i]];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wfrag[ks], acc[m], 0, 0, 0);
        }
    } else if constexpr (X3MM) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 wp[3] = {sWx[(ks * 3 + 0) * 64 + lane], sWx[(ks * 3 + 1) * 64 + lane], sWx[(ks * 3 + 2) * 64 + lane]};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = sP[pbb[m] + goff[ks][i]];
          uint4 ap[3];
          ugn_x3::split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), ap[0], ap[1], ap[2]);
#pragma unroll
          for (int i = 0; i < ugn_x3::kProducts; ++i)      // the six largest partial products, smallest first (x3_common.h)
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, ap[ugn_x3::prod_x<ugn_x3::kProducts>(i)]),
                                                             __builtin_bit_cast(bf8, wp[ugn_x3::prod_w<ugn_x3::kProducts>(i)]), acc[m], 0, 0, 0);
        }
      }
    } else if constexpr (H2MM) {
      const unsigned* sPu = reinterpret_cast<const unsigned*>(sP);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          unsigned d[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) d[i] = sPu[pbb[m] + goff[ks][i]];
          // the lane's 8 taps: low halves -> the H fragment, high halves -> the L fragment (v_perm_b32)
          const uint4 ah = make_uint4(__builtin_amdgcn_perm(d[1], d[0], 0x05040100u), __builtin_amdgcn_perm(d[3], d[2], 0x05040100u),
                                      __builtin_amdgcn_perm(d[5], d[4], 0x05040100u), __builtin_amdgcn_perm(d[7], d[6], 0x05040100u));
          const uint4 al = make_uint4(__builtin_amdgcn_perm(d[1], d[0], 0x07060302u), __builtin_amdgcn_perm(d[3], d[2], 0x07060302u),
                                      __builtin_amdgcn_perm(d[5], d[4], 0x07060302u), __builtin_amdgcn_perm(d[7], d[6], 0x07060302u));
          const ugn_mm::h8 fah = __builtin_bit_cast(ugn_mm::h8, ah), fal = __builtin_bit_cast(ugn_mm::h8, al);
          const ugn_mm::h8 wh = __builtin_bit_cast(ugn_mm::h8, sWf[(ks * 2 + 0) * 64 + lane]);
          const ugn_mm::h8 wl = __builtin_bit_cast(ugn_mm::h8, sWf[(ks * 2 + 1) * 64 + lane]);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, wh, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, wl, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, wh, acc[m], 0, 0, 0);
        }
    } else
#pragma unroll
    for (int s = 0; s < KP; ++s) {
      int off;
      if constexpr (CIN == 2) {
        off = (s / 5) * FP + (s % 5) * 2;    // tap s, channel = lane half
      } else {
        const int k0 = 2 * s, k1 = (2 * s + 1 < K) ? 2 * s + 1 : K - 1;
        const int o0 = (k0 / 5) * FP + (k0 % 5), o1 = (k1 / 5) * FP + (k1 % 5);
        off = lh ? o1 : o0;
      }
      const float b = sW[(2 * s + lh) * 32 + li];
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = ugn_mfma(sP[pb[m] + off], b, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int mbi = wave * 2 + m;
      uint32_t myword = 0;   // SIGN: lane li < 16 of each half collects the word of register r = li
      float* __restrict__ orow = a1 + (((size_t)img * DOM + ty0 + 2 * mbi) * DOM + tx0 + 2 * lh) * 32 + li;
      // H2: the pixel's 128-byte record = 32 H halves + 32 L halves; lanes (li, li ^ 1) exchange halves so that the even lane
      // stores the H pair and the odd lane the L pair (as the one-block epilogue of conv3x3_mm.hip)
      char* __restrict__ hrow = reinterpret_cast<char*>(a1) + (((size_t)img * DOM + ty0 + 2 * mbi) * DOM + tx0 + 2 * lh) * 128 +
                                ((li & 1) ? 64 + (li - 1) * 2 : li * 2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // pixel (2 * mbi + ((r >> 1) & 1), 2 * (lh + 2 * (r >> 2)) + (r & 1)) of the tile: compile-time offset from orow
        if constexpr (H2OUT) {
          const float v = ugn_lrelu(acc[m][r]) * h2_factor;
          h2_mx = fmaxf(h2_mx, fabsf(v));
          _Float16 hi, lo;
          ugn_mm::h2_split(v, hi, lo);
          const unsigned own = ugn_mm::h2_pack(hi, lo);
          const unsigned oth = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
          *reinterpret_cast<unsigned*>(hrow + (((r >> 1) & 1) * DOM + 4 * (r >> 2) + (r & 1)) * 128) =
              __builtin_amdgcn_perm(oth, own, (li & 1) ? 0x03020706u : 0x05040100u);
        } else if constexpr (OFMT == 2) {     // bf16: lanes (li, li ^ 1) pair up, the even lane stores both channels
          const unsigned own = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)ugn_lrelu(acc[m][r]));
          const unsigned oth = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, false);
          char* brow = reinterpret_cast<char*>(a1) + (((size_t)img * DOM + ty0 + 2 * mbi) * DOM + tx0 + 2 * lh) * 64 + li * 2;
          if (!(li & 1)) *reinterpret_cast<unsigned*>(brow + (((r >> 1) & 1) * DOM + 4 * (r >> 2) + (r & 1)) * 64) = own | (oth << 16);
        } else {
          orow[(((r >> 1) & 1) * DOM + 4 * (r >> 2) + (r & 1)) * 32] = ugn_lrelu(acc[m][r]);
        }
        if constexpr (SIGN) {   // bit c of a pixel's word = (a1 > 0): all the layer's backward needs of a1 besides its values
          const unsigned long long bal = __ballot(acc[m][r] > 0.f);   // lanes 0..31: pixel of lh = 0, 32..63: lh = 1
          if (li == r) myword = (uint32_t)(bal >> (32 * lh));
        }
      }
      if constexpr (SIGN) {
        if (li < 16) {   // one store per half wave and row block instead of sixteen single-lane ones
          const int y = ty0 + 2 * mbi + ((li >> 1) & 1);
          const int xx = tx0 + 2 * (lh + 2 * (li >> 2)) + (li & 1);
          sign_out[((size_t)img * DOM + y) * DOM + xx] = myword;
        }
      }
    }
  }
  // one atomicMax per WORKGROUP (through sP; the helper's first barrier ends the last tile's reads of it): one per wave -- 4,096
  // atomics on one address -- was a fixed 50 us at the end of every launch of this kernel
  if constexpr (H2OUT) ugn_mm::h2_publish_amax_block(out_meta, h2_mx, sP, tid, 4);
}

// global -> LDS without a VGPR destination (LDS address = M0 + lane * 16 or + lane * 4); see conv3x3_wino.hip for why asm
__device__ __forceinline__ void dma16_c5(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst_uniform)
               : "memory");
}
__device__ __forceinline__ void dma4_c5(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst_uniform)
               : "memory");
}

// dw[k][co] = sum_pixels patch[pixel + tap_k][c_k] * dz1[pixel][co]; persistent workgroups, slabs [groups][25*CIN][32].
// The gradient tile (16x16 pixels x 32 channels, unpadded: a lane pair reads pixels p, p + 1 = the two bank halves) and the
// input patch of the NEXT tile stream into
// the second LDS buffer by LDS-DMA (16-byte pieces for the gradient, dwords for the unaligned patch; lanes outside the image
// or in the pad read a zero block) while the current tile is multiplied: the kernel runs at the rate dz1 can be read.
// H2DZ: dz1 is an H2 tensor ([pixel][2][32] halves: the same 128 bytes per pixel, so the LDS-DMA staging is unchanged); a
// gradient value is re-assembled from its two halves when it is read, the block exponent is undone by reduce5_kernel.
// DZFMT: 0 fp32 dz1, 1 H2, 2 bf16 ([pixel][32] bf16: 64-byte records, half the gradient tile), 3 H2 multiplied on the f16 matrix pipe
// (round 4): K = 16 pixels per v_mfma_f32_32x32x16_f16 step; the gradient fragments -- 8 consecutive pixels of one channel per lane,
// H and L halves -- by transposed reads of the [pixel][2][32] tile; the patch as packed halves of x * 2^ex (converted in place once
// per tile); LeakyReLU'(a1) WITHOUT re-splitting a scaled gradient: slope = 0.3 + 0.7 [a1 > 0], so the kernel keeps two sums,
// acc_all += x * g and acc_pos += x * (g AND mask) -- masking the halves is exact -- and leaves 0.3 acc_all + 0.7 acc_pos.
// 6 f16 MFMAs of 32 cycles per 16 pixels and row block instead of 8 fp32 MFMAs of 64 that block the vector pipe.
// DZFMT 4 (round 5): fp32 dz1 multiplied in the x3 arithmetic (x3_common.h) -- K = 16 pixels per v_mfma_f32_32x32x16_bf16 step; a lane
// gathers the patch values under its (tap, channel) row and the gradient of its channel at 8 consecutive pixels (fp32, the gradient
// times LeakyReLU' from the sign words), splits both into three bf16 planes in registers and issues the six products: 6 matrix
// instructions of 8 issue cycles per 16 pixels and row block instead of 8 fp32 ones of 64 cycles that hold the vector pipe (the fp32
// form measured 52-64 % MFMA busy on a kernel that should run at the rate dz1 can be read).
template <int CIN, bool SIGN, int DZFMT = 0>
__global__ __launch_bounds__(256) void conv5x5_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz1,
                                                            float* __restrict__ slab, const float* __restrict__ zeros,
                                                            const uint32_t* __restrict__ a1_sign, int tiles_total,
                                                            const H2Meta* __restrict__ x_meta = nullptr) {
  constexpr int K = 25 * CIN, MBK = (K + 31) / 32;
  constexpr int DS = UGN_C5_DS;
  constexpr int SDF = T5 * T5 * DS;                       // floats per gradient buffer (32 KB = 32 pieces of 1 KB)
  // patch rows of WP pixels (UGN_C5_PITCH)
  constexpr int WP = c5_pitch(CIN);
  constexpr int PE = P5 * WP * CIN, PPIECES = (PE + 63) / 64, SPF = PPIECES * 64;   // patch dwords, 256-B pieces
  constexpr bool H2X = DZFMT == 3, H2DZ = DZFMT == 1 || H2X, BFDZ = DZFMT == 2, X3W = DZFMT == 4;
  constexpr int DPW = BFDZ ? 4 : DS / 4, PPW = (PPIECES + 3) / 4;    // pieces per wave
  extern __shared__ __attribute__((aligned(16))) float smem5[];
  float* sD0 = smem5;                 // [2][SDF]
  float* sP0 = smem5 + 2 * SDF;       // [2][SPF]
  const uint32_t* sS0 = reinterpret_cast<const uint32_t*>(smem5 + 2 * SDF + 2 * SPF);   // [2][256] sign words (optional)
  const unsigned ss_bytes = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem5) +
                            (2u * SDF + 2u * SPF) * 4u;
  const unsigned sd_bytes = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem5);
  const unsigned sp_bytes = sd_bytes + 2u * SDF * 4u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[MBK];
  f32x16 accp[H2X && SIGN ? MBK : 1];      // H2X: the sum over the pixels with a1 > 0
  float x_scale = 1.f;
  if constexpr (H2X) {
    x_scale = ldexpf(1.f, ugn_mm::h2_exp_for_bound(ugn_mm::h2_true_amax(x_meta->e, x_meta->amax)));
#pragma unroll
    for (int mb = 0; mb < (SIGN ? MBK : 1); ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) accp[mb][r] = 0.f;
  }
  int abase[MBK];
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    int k = mb * 32 + li;
    if (k >= K) k = UGN_C5_PADK(K, mb);  // padded rows: a valid address (a broadcast of a lane of this block), result discarded
    const int tap = k / CIN, c = k % CIN;
    abase[mb] = ((tap / 5) * WP + (tap % 5)) * CIN + c + ((wave * 4) * WP + lh) * CIN;  // wave owns tile rows 4w..4w+3
  }
  const int bbase = (wave * 64 + lh) * DS + li;

  // per-lane geometry of this wave's DMA pieces (tile independent)
  int dgeo[DPW], pgeo[PPW];
#pragma unroll
  for (int j = 0; j < DPW; ++j) {
    const int slot = (wave * DPW + j) * 64 + lane;       // 4 waves x DPW pieces = all slots
    if constexpr (BFDZ) {                                // 4 slots of 16 bytes per pixel; dgeo in FLOAT units of the 64-byte records
      const int p = slot >> 2, c4 = slot & 3;
      dgeo[j] = ((p >> 4) * DOM + (p & 15)) * 16 + c4 * 4;
    } else {
      const int p = slot / (DS / 4), c4 = slot - p * (DS / 4);
      dgeo[j] = c4 < 8 ? ((p >> 4) * DOM + (p & 15)) * 32 + c4 * 4 : -1;
    }
  }
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    int inst = wave * PPW + j;
    inst = inst < PPIECES ? inst : PPIECES - 1;
    const int e = inst * 64 + lane;
    const int pe = e / CIN, ch = e - pe * CIN;
    pgeo[j] = (e < PE && pe % WP < P5) ? ((pe / WP) << 16) | ((pe % WP) << 8) | ch : -1;
  }
  auto issue_dma = [&](int tile, int buf) {
    const int img = tile >> 4, trem = tile & 15;
    const int ty0 = (trem >> 2) * T5, tx0 = (trem & 3) * T5;
    const float* dzt = dz1 + (((size_t)img * DOM + ty0) * DOM + tx0) * (BFDZ ? 16 : 32);
#pragma unroll
    for (int j = 0; j < DPW; ++j)
      dma16_c5(dgeo[j] >= 0 ? dzt + dgeo[j] : zeros, sd_bytes + (unsigned)buf * SDF * 4u + (unsigned)(wave * DPW + j) * 1024u);
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      int inst = wave * PPW + j;
      inst = inst < PPIECES ? inst : PPIECES - 1;
      const int yy = pgeo[j] >> 16, xx = (pgeo[j] >> 8) & 0xff, ch = pgeo[j] & 0xff;
      const int ry = ty0 - 4 + yy, rx = tx0 - 4 + xx;
      const bool ok = pgeo[j] >= 0 && ry >= 0 && ry < RAW && rx >= 0 && rx < RAW;
      dma4_c5(ok ? x + (((size_t)img * RAW + ry) * RAW + rx) * CIN + ch : zeros,
              sp_bytes + (unsigned)buf * SPF * 4u + (unsigned)inst * 256u);
    }
    if (SIGN && wave == 0)   // 256 sign words of the tile: lane -> row lane/4, words 4*(lane%4)..+3
      dma16_c5(a1_sign + ((size_t)img * DOM + ty0 + (lane >> 2)) * DOM + tx0 + (lane & 3) * 4, ss_bytes + (unsigned)buf * 1024u);
  };
  int tile = blockIdx.x, buf = 0;
  if (tile < tiles_total) issue_dma(tile, 0);
  for (; tile < tiles_total; tile += gridDim.x, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the current buffers have landed
    __syncthreads();                                    // everyone's have; the other buffers have no readers left
    const int nt = tile + (int)gridDim.x;
    issue_dma(nt < tiles_total ? nt : tile, buf ^ 1);   // branch-free: past the end the current tile is fetched again
    const float* sD = sD0 + buf * SDF;
    const float* sP = sP0 + buf * SPF;
    if constexpr (H2X) {
      // the patch of this tile -> packed halves (H | L << 16) of x * 2^ex, in place: once per value, not once per gather
      unsigned* sPu = reinterpret_cast<unsigned*>(sP0 + buf * SPF);
      for (int e = tid; e < PE; e += 256) {
        _Float16 hi, lo;
        ugn_mm::h2_split(sP[e] * x_scale, hi, lo);
        sPu[e] = ugn_mm::h2_pack(hi, lo);
      }
      __syncthreads();
      typedef short s4v __attribute__((ext_vector_type(4)));
      const int gh = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
      const __attribute__((address_space(3))) char* dzl = (const __attribute__((address_space(3))) char*)sD +
                                                          (wave * 64 + 8 * lh + tq) * 128 + (16 * gh + 4 * tp) * 2;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {      // tile row 4 wave + ks: 16 pixels
        const s4v h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(dzl + ks * 2048));
        const s4v h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(dzl + ks * 2048 + 512));
        const s4v l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(dzl + ks * 2048 + 64));
        const s4v l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(dzl + ks * 2048 + 512 + 64));
        const uint2 hv0 = __builtin_bit_cast(uint2, h0), hv1 = __builtin_bit_cast(uint2, h1);
        const uint2 lv0 = __builtin_bit_cast(uint2, l0), lv1 = __builtin_bit_cast(uint2, l1);
        const uint4 gh4 = make_uint4(hv0.x, hv0.y, hv1.x, hv1.y), gl4 = make_uint4(lv0.x, lv0.y, lv1.x, lv1.y);
        uint4 mh4 = gh4, ml4 = gl4;
        if constexpr (SIGN) {                 // halves of the pixels with a1[pixel][li] > 0, zero elsewhere
          unsigned m[4];
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const uint32_t w0 = sS0[buf * 256 + wave * 64 + ks * 16 + 8 * lh + 2 * d], w1 = sS0[buf * 256 + wave * 64 + ks * 16 + 8 * lh + 2 * d + 1];
            m[d] = (((w0 >> li) & 1u) ? 0x0000ffffu : 0u) | (((w1 >> li) & 1u) ? 0xffff0000u : 0u);
          }
          mh4 = make_uint4(gh4.x & m[0], gh4.y & m[1], gh4.z & m[2], gh4.w & m[3]);
          ml4 = make_uint4(gl4.x & m[0], gl4.y & m[1], gl4.z & m[2], gl4.w & m[3]);
        }
        const ugn_mm::h8 bh = __builtin_bit_cast(ugn_mm::h8, gh4), bl = __builtin_bit_cast(ugn_mm::h8, gl4);
        const ugn_mm::h8 ph = __builtin_bit_cast(ugn_mm::h8, mh4), pl = __builtin_bit_cast(ugn_mm::h8, ml4);
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) {
          unsigned d[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) d[i] = sPu[abase[mb] + (ks * WP + 8 * lh + i - lh) * CIN];
          const uint4 a_h = make_uint4(__builtin_amdgcn_perm(d[1], d[0], 0x05040100u), __builtin_amdgcn_perm(d[3], d[2], 0x05040100u),
                                       __builtin_amdgcn_perm(d[5], d[4], 0x05040100u), __builtin_amdgcn_perm(d[7], d[6], 0x05040100u));
          const uint4 a_l = make_uint4(__builtin_amdgcn_perm(d[1], d[0], 0x07060302u), __builtin_amdgcn_perm(d[3], d[2], 0x07060302u),
                                       __builtin_amdgcn_perm(d[5], d[4], 0x07060302u), __builtin_amdgcn_perm(d[7], d[6], 0x07060302u));
          const ugn_mm::h8 ah = __builtin_bit_cast(ugn_mm::h8, a_h), al = __builtin_bit_cast(ugn_mm::h8, a_l);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[mb], 0, 0, 0);
          if constexpr (SIGN) {
            accp[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ph, accp[mb], 0, 0, 0);
            accp[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, pl, accp[mb], 0, 0, 0);
            accp[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ph, accp[mb], 0, 0, 0);
          }
        }
      }
    } else if constexpr (X3W) {
      typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
      unsigned lane_bit = 1u << li;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {      // tile row 4 wave + ks: 16 pixels, this lane half's 8
        float g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          g[i] = sD[(wave * 64 + ks * 16 + 8 * lh + i) * DS + li];
          if constexpr (SIGN) {
            const uint32_t wbits = sS0[buf * 256 + wave * 64 + ks * 16 + 8 * lh + i];
            g[i] *= (wbits & lane_bit) ? 1.f : UGN_LRELU_ALPHA;
          }
        }
        uint4 gp[3];
        ugn_x3::split8(make_float4(g[0], g[1], g[2], g[3]), make_float4(g[4], g[5], g[6], g[7]), gp[0], gp[1], gp[2]);
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = sP[abase[mb] + (ks * WP + 8 * lh + i - lh) * CIN];
          uint4 xp[3];
          ugn_x3::split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), xp[0], xp[1], xp[2]);
#pragma unroll
          for (int i = 0; i < ugn_x3::kProducts; ++i)      // (x plane, gradient plane): the six largest partial products, smallest first
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, xp[ugn_x3::prod_x<ugn_x3::kProducts>(i)]),
                                                              __builtin_bit_cast(bf8, gp[ugn_x3::prod_w<ugn_x3::kProducts>(i)]), acc[mb], 0, 0, 0);
        }
      }
    } else if constexpr (BFDZ) {
      // bf16 gradient (configs[4]): v_mfma_f32_32x32x16_bf16 with K = 16 pixels (one tile row) per step -- the wave's 64 pixels are
      // 4 MFMAs per row block instead of 32 fp32 ones.  A = the patch values under the lane's (tap, channel) at 8 consecutive
      // pixels, rounded to bf16; B = the gradient of 8 consecutive pixels of channel li: two transposed reads of the [pixel][32]
      // tile (as wgrad3x3_bf.hip), times LeakyReLU'(a1) from the sign words.
      typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
      typedef short s4v __attribute__((ext_vector_type(4)));
      const int gh = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
      const __attribute__((address_space(3))) char* dzl = (const __attribute__((address_space(3))) char*)sD +
                                                          (wave * 64 + 8 * lh + tq) * 64 + (16 * gh + 4 * tp) * 2;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const s4v t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(dzl + ks * 1024));
        const s4v t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(dzl + ks * 1024 + 256));
        bf8 bfrag;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float g = __uint_as_float((unsigned)(unsigned short)(i < 4 ? t0[i] : t1[i - 4]) << 16);
          if constexpr (SIGN) {
            const uint32_t wbits = sS0[buf * 256 + wave * 64 + ks * 16 + 8 * lh + i];
            g *= ((wbits >> li) & 1u) ? 1.f : UGN_LRELU_ALPHA;
          }
          bfrag[i] = (__bf16)g;
        }
#pragma unroll
        for (int mb = 0; mb < MBK; ++mb) {
          bf8 afrag;
#pragma unroll
          for (int i = 0; i < 8; ++i) afrag[i] = (__bf16)sP[abase[mb] + (ks * WP + 8 * lh + i - lh) * CIN];
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[mb], 0, 0, 0);
        }
      }
    } else
#pragma unroll
    for (int kp = 0; kp < 32; ++kp) {  // wave's 64 pixels: p = 2*kp + lh, row = p/16, col = p%16
      const int po = ((2 * kp) / 16) * WP + ((2 * kp) % 16);
      float b;
      if constexpr (H2DZ) {
        const uint16_t* rec = reinterpret_cast<const uint16_t*>(sD) + (size_t)(wave * 64 + lh + 2 * kp) * 64;
        b = ugn_mm::h2_half(rec[li]) + ugn_mm::h2_half(rec[32 + li]);
      } else if constexpr (BFDZ) {
        b = __uint_as_float((unsigned)reinterpret_cast<const uint16_t*>(sD)[(size_t)(wave * 64 + lh + 2 * kp) * 32 + li] << 16);
      } else {
        b = sD[bbase + 2 * kp * DS];
      }
      if constexpr (SIGN) {   // dz1 arrives as dL/da1: the LeakyReLU' factor of a1 is applied here, from one bit per element
        const uint32_t wbits = sS0[buf * 256 + wave * 64 + 2 * kp + lh];
        b *= ((wbits >> li) & 1u) ? 1.f : UGN_LRELU_ALPHA;
      }
#pragma unroll
      for (int mb = 0; mb < MBK; ++mb) acc[mb] = ugn_mfma(sP[abase[mb] + po * CIN], b, acc[mb]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (H2X) {        // LeakyReLU' = 0.3 + 0.7 [a1 > 0]; the x exponent is undone here, the gradient's by reduce5_kernel
    const float inv = 1.f / x_scale;
#pragma unroll
    for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[mb][r] = (SIGN ? UGN_LRELU_ALPHA * acc[mb][r] + (1.f - UGN_LRELU_ALPHA) * accp[SIGN ? mb : 0][r] : acc[mb][r]) * inv;
  }
  // cross-wave reduction through LDS (reuse sD: 4 waves x MBK x 16 x 64 floats <= 8192 floats = one gradient buffer)
  __syncthreads();
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) sD0[((wave * MBK + mb) * 16 + r) * 64 + lane] = acc[mb][r];
  __syncthreads();
  float* dst = slab + (size_t)blockIdx.x * K * 32;
  for (int e = tid; e < MBK * 16 * 64; e += 256) {
    const int l = e & 63, r = (e >> 6) & 15, mb = e >> 10;
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) s += sD0[((wv * MBK + mb) * 16 + r) * 64 + l];
    const int k = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    if (k < K) dst[k * 32 + (l & 31)] = s;
  }
}

// dst[e] = sum_g src[g][e]: 256 threads = 8 elements x 32 slab lanes (lane ly sums slabs ly, ly+32, ...), combined in a fixed
// order through LDS -> bitwise reproducible.  (800 / 1600 elements only: 32 lanes per element keep 100 / 200 workgroups busy.)
__global__ __launch_bounds__(256) void reduce5_kernel(const float* __restrict__ src, float* __restrict__ dst, int nelem,
                                                      int nin, const H2Meta* __restrict__ dz_meta = nullptr) {
  __shared__ float sR[32][9];
  const int le = threadIdx.x & 7, ly = threadIdx.x >> 3;
  const int e = blockIdx.x * 8 + le;
  float s = 0.f;
  if (e < nelem)
    for (int g = ly; g < nin; g += 32) s += src[(size_t)g * nelem + e];
  sR[ly][le] = s;
  __syncthreads();
  if (ly == 0 && e < nelem) {
#pragma unroll
    for (int k = 1; k < 32; ++k) s += sR[k][le];
    dst[e] = dz_meta ? ldexpf(s, -dz_meta->e) : s;
  }
}

constexpr int WG5_GROUPS = 512;

}  // namespace

extern "C" int ugn_conv5x5_in_fwd(const float* x, const float* w, float* a1, uint32_t* a1_sign, int n, int cin, void* stream) {
  UGN_REQUIRE(x && w && a1 && n > 0, "ugn_conv5x5_in_fwd: null pointer or n <= 0");
  UGN_REQUIRE(cin == 1 || cin == 2, "ugn_conv5x5_in_fwd: cin must be 1 or 2 (got %d)", cin);
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = n * 16;
  const int c5grid = UGN_C5_GRID / ugn_mm::kGrid * ugn_mm::persistent_wgs();     // (4 workgroups per CU of those left to this library)
  const int grid = ntiles < c5grid ? ntiles : c5grid;
#define UGN_C5F(C_, S_) hipLaunchKernelGGL((conv5x5_fwd_kernel<C_, S_>), dim3(grid), dim3(256), 0, st, x, w, a1, a1_sign, ntiles)
  if (cin == 1) {
    if (a1_sign) UGN_C5F(1, true); else UGN_C5F(1, false);
  } else {
    if (a1_sign) UGN_C5F(2, true); else UGN_C5F(2, false);
  }
#undef UGN_C5F
  UGN_CHECK_LAUNCH("conv5x5_fwd");
  return 0;
}

/* the same layer, fp32 in and out, multiplied in the x3 arithmetic (three-way bf16 split of the patch and the filter, six products
 * per fp32 product on v_mfma_f32_32x32x16_bf16: csrc/x3_common.h) -- the first layer of the default fp32-tensor set */
extern "C" int ugn_x3_conv5x5_in_fwd(const float* x, const float* w, float* a1, uint32_t* a1_sign, int n, int cin, void* stream) {
  UGN_REQUIRE(x && w && a1 && n > 0, "ugn_x3_conv5x5_in_fwd: null pointer or n <= 0");
  UGN_REQUIRE(cin == 1 || cin == 2, "ugn_x3_conv5x5_in_fwd: cin must be 1 or 2 (got %d)", cin);
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = n * 16;
  const int c5grid = UGN_C5_GRID / ugn_mm::kGrid * ugn_mm::persistent_wgs();     // (4 workgroups per CU of those left to this library)
  const int grid = ntiles < c5grid ? ntiles : c5grid;
#define UGN_C5F(C_, S_) hipLaunchKernelGGL((conv5x5_fwd_kernel<C_, S_, 3>), dim3(grid), dim3(256), 0, st, x, w, a1, a1_sign, ntiles)
  if (cin == 1) {
    if (a1_sign) UGN_C5F(1, true); else UGN_C5F(1, false);
  } else {
    if (a1_sign) UGN_C5F(2, true); else UGN_C5F(2, false);
  }
#undef UGN_C5F
  UGN_CHECK_LAUNCH("conv5x5_fwd_x3");
  return 0;
}

#if UGN_WITH_H2      /* the opt-in f16x2 set (python -m ugaitnet_amd.build --h2): not part of the default library */
/* the same layer with a1 as an H2 tensor [n][64][64][2][32] (+ its ugn_h2meta, zero on entry); x_meta = {0, bits(max|x|)} */
extern "C" int ugn_conv5x5_in_fwd_h2(const float* x, const void* x_meta, const float* w, uint16_t* a1, void* a1_meta,
                                     uint32_t* a1_sign, int n, int cin, void* stream) {
  UGN_REQUIRE(x && x_meta && w && a1 && a1_meta && n > 0, "ugn_conv5x5_in_fwd_h2: null pointer or n <= 0");
  UGN_REQUIRE(cin == 1 || cin == 2, "ugn_conv5x5_in_fwd_h2: cin must be 1 or 2 (got %d)", cin);
  UGN_REQUIRE(UGN_C5_DS == 32, "ugn_conv5x5_in_fwd_h2: built with a padded gradient tile");
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = n * 16;
  const int c5grid = UGN_C5_GRID / ugn_mm::kGrid * ugn_mm::persistent_wgs();     // (4 workgroups per CU of those left to this library)
  const int grid = ntiles < c5grid ? ntiles : c5grid;
#define UGN_C5F(C_, S_) hipLaunchKernelGGL((conv5x5_fwd_kernel<C_, S_, 1>), dim3(grid), dim3(256), 0, st, x, w, (float*)a1, a1_sign, \
                                           ntiles, (const H2Meta*)x_meta, (H2Meta*)a1_meta)
  if (cin == 1) {
    if (a1_sign) UGN_C5F(1, true); else UGN_C5F(1, false);
  } else {
    if (a1_sign) UGN_C5F(2, true); else UGN_C5F(2, false);
  }
#undef UGN_C5F
  UGN_CHECK_LAUNCH("conv5x5_fwd_h2");
  return 0;
}
#endif

extern "C" size_t ugn_conv5x5_in_wgrad_ws(int n, int cin) {
  if (n <= 0 || (cin != 1 && cin != 2)) return 0;
  const long tiles = (long)n * 16;
  const long groups = tiles < WG5_GROUPS ? tiles : WG5_GROUPS;
  return (size_t)groups * 25 * cin * 32 * sizeof(float);
}

static int conv5x5_wgrad_any(const float* x, const float* dz1, const H2Meta* dz_meta, const uint32_t* a1_sign, float* dw, int n,
                             int cin, void* ws, size_t ws_bytes, void* stream, bool bf = false, const H2Meta* x_meta = nullptr,
                             bool x3 = false);
/* fp32 dz1, multiplied in the x3 arithmetic (csrc/x3_common.h): the weight gradient of the first layer in the default fp32-tensor set */
extern "C" int ugn_x3_conv5x5_in_wgrad(const float* x, const float* dz1, const uint32_t* a1_sign, float* dw, int n, int cin,
                                       void* ws, size_t ws_bytes, void* stream) {
  return conv5x5_wgrad_any(x, dz1, nullptr, a1_sign, dw, n, cin, ws, ws_bytes, stream, false, nullptr, true);
}
extern "C" int ugn_conv5x5_in_wgrad(const float* x, const float* dz1, const uint32_t* a1_sign, float* dw, int n, int cin,
                                    void* ws, size_t ws_bytes, void* stream) {
  return conv5x5_wgrad_any(x, dz1, nullptr, a1_sign, dw, n, cin, ws, ws_bytes, stream);
}
#if UGN_WITH_H2
/* dz1 as an H2 tensor [n][64][64][2][32] with its ugn_h2meta */
extern "C" int ugn_conv5x5_in_wgrad_h2(const float* x, const uint16_t* dz1, const void* dz1_meta, const uint32_t* a1_sign, float* dw,
                                       int n, int cin, void* ws, size_t ws_bytes, void* stream) {
  UGN_REQUIRE(dz1_meta, "ugn_conv5x5_in_wgrad_h2: null meta");
  return conv5x5_wgrad_any(x, reinterpret_cast<const float*>(dz1), (const H2Meta*)dz1_meta, a1_sign, dw, n, cin, ws, ws_bytes, stream);
}
/* the same on the f16 matrix pipe: x_meta = {0, bits(max|x|)} (ugn_absmax_multi) gives the exponent the input patch is split with */
extern "C" int ugn_conv5x5_in_wgrad_h2x(const float* x, const void* x_meta, const uint16_t* dz1, const void* dz1_meta,
                                        const uint32_t* a1_sign, float* dw, int n, int cin, void* ws, size_t ws_bytes, void* stream) {
  UGN_REQUIRE(dz1_meta && x_meta, "ugn_conv5x5_in_wgrad_h2x: null meta");
  return conv5x5_wgrad_any(x, reinterpret_cast<const float*>(dz1), (const H2Meta*)dz1_meta, a1_sign, dw, n, cin, ws, ws_bytes, stream,
                           false, (const H2Meta*)x_meta);
}
#endif
/* dz1 as a bf16 tensor [n][64][64][32] (configs[4]) */
extern "C" int ugn_conv5x5_in_wgrad_bf(const float* x, const uint16_t* dz1, const uint32_t* a1_sign, float* dw, int n, int cin, void* ws,
                                       size_t ws_bytes, void* stream) {
  return conv5x5_wgrad_any(x, reinterpret_cast<const float*>(dz1), nullptr, a1_sign, dw, n, cin, ws, ws_bytes, stream, true);
}
/* first layer with a1 written as bf16 [n][64][64][32] */
extern "C" int ugn_conv5x5_in_fwd_bf(const float* x, const float* w, uint16_t* a1, uint32_t* a1_sign, int n, int cin, void* stream) {
  UGN_REQUIRE(x && w && a1 && n > 0, "ugn_conv5x5_in_fwd_bf: null pointer or n <= 0");
  UGN_REQUIRE(cin == 1 || cin == 2, "ugn_conv5x5_in_fwd_bf: cin must be 1 or 2 (got %d)", cin);
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = n * 16;
  const int c5grid = UGN_C5_GRID / ugn_mm::kGrid * ugn_mm::persistent_wgs();     // (4 workgroups per CU of those left to this library)
  const int grid = ntiles < c5grid ? ntiles : c5grid;
#define UGN_C5F(C_, S_) hipLaunchKernelGGL((conv5x5_fwd_kernel<C_, S_, 2>), dim3(grid), dim3(256), 0, st, x, w, (float*)a1, a1_sign, ntiles)
  if (cin == 1) {
    if (a1_sign) UGN_C5F(1, true); else UGN_C5F(1, false);
  } else {
    if (a1_sign) UGN_C5F(2, true); else UGN_C5F(2, false);
  }
#undef UGN_C5F
  UGN_CHECK_LAUNCH("conv5x5_fwd_bf");
  return 0;
}
static int conv5x5_wgrad_any(const float* x, const float* dz1, const H2Meta* dz_meta, const uint32_t* a1_sign, float* dw, int n,
                             int cin, void* ws, size_t ws_bytes, void* stream, bool bf, const H2Meta* x_meta, bool x3) {
  UGN_REQUIRE(x && dz1 && dw && ws && n > 0, "ugn_conv5x5_in_wgrad: null pointer or n <= 0");
  UGN_REQUIRE(cin == 1 || cin == 2, "ugn_conv5x5_in_wgrad: cin must be 1 or 2 (got %d)", cin);
  UGN_REQUIRE(ws_bytes >= ugn_conv5x5_in_wgrad_ws(n, cin), "ugn_conv5x5_in_wgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int tiles = n * 16;
  // (not sized by ugn_set_persistent_wgs: a workgroup's slab is the sum over ITS tiles, so another grid is another grouping of the
  //  fp32 sums; at 2 workgroups of 70 KB LDS and 8 waves per CU this launch leaves RCCL's channels room on every CU anyway)
  const int groups = tiles < WG5_GROUPS ? tiles : WG5_GROUPS;
  static float* zeros = nullptr;   // LDS-DMA source for the pad and the out-of-image lanes
  if (!zeros) {
    float* p = nullptr;
    UGN_REQUIRE(hipMalloc((void**)&p, 256) == hipSuccess && hipMemset(p, 0, 256) == hipSuccess,
                "ugn_conv5x5_in_wgrad: cannot allocate the zero block");
    zeros = p;
  }
  const int lds = (2 * 256 * UGN_C5_DS + 2 * ((20 * c5_pitch(cin) * cin + 63) / 64) * 64 + 2 * 256) * 4;
  static bool attr_done[3] = {false, false, false};
  if (!attr_done[cin]) {
    // (DZFMT 1 and 3, the f16x2 forms, exist only in a build with the opt-in set)
    const void* fns[] = {(const void*)conv5x5_wgrad_kernel<1, false>, (const void*)conv5x5_wgrad_kernel<1, true>,
                         (const void*)conv5x5_wgrad_kernel<1, false, 2>, (const void*)conv5x5_wgrad_kernel<1, true, 2>,
                         (const void*)conv5x5_wgrad_kernel<1, false, 4>, (const void*)conv5x5_wgrad_kernel<1, true, 4>,
#if UGN_WITH_H2
                         (const void*)conv5x5_wgrad_kernel<1, false, 1>, (const void*)conv5x5_wgrad_kernel<1, true, 1>,
                         (const void*)conv5x5_wgrad_kernel<1, false, 3>, (const void*)conv5x5_wgrad_kernel<1, true, 3>,
#endif
                         (const void*)conv5x5_wgrad_kernel<2, false>, (const void*)conv5x5_wgrad_kernel<2, true>,
                         (const void*)conv5x5_wgrad_kernel<2, false, 2>, (const void*)conv5x5_wgrad_kernel<2, true, 2>,
                         (const void*)conv5x5_wgrad_kernel<2, false, 4>, (const void*)conv5x5_wgrad_kernel<2, true, 4>,
#if UGN_WITH_H2
                         (const void*)conv5x5_wgrad_kernel<2, false, 1>, (const void*)conv5x5_wgrad_kernel<2, true, 1>,
                         (const void*)conv5x5_wgrad_kernel<2, false, 3>, (const void*)conv5x5_wgrad_kernel<2, true, 3>,
#endif
    };
    constexpr int NV = (int)(sizeof(fns) / sizeof(fns[0])) / 2;
    for (int v = 0; v < NV; ++v) {
      hipError_t e = hipFuncSetAttribute(fns[(cin - 1) * NV + v], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      UGN_REQUIRE(e == hipSuccess, "ugn_conv5x5_in_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    attr_done[cin] = true;
  }
#if UGN_WITH_H2
#define UGN_C5W_H2(C_, S_)                                                                                                   \
    else if (dz_meta && x_meta)                                                                                               \
      hipLaunchKernelGGL((conv5x5_wgrad_kernel<C_, S_, 3>), dim3(groups), dim3(256), lds, st, x, dz1, (float*)ws,             \
                         (const float*)zeros, a1_sign, tiles, x_meta);                                                        \
    else if (dz_meta)                                                                                                         \
      hipLaunchKernelGGL((conv5x5_wgrad_kernel<C_, S_, 1>), dim3(groups), dim3(256), lds, st, x, dz1, (float*)ws,             \
                         (const float*)zeros, a1_sign, tiles);
#else
#define UGN_C5W_H2(C_, S_)
#endif
#define UGN_C5W(C_, S_)                                                                                                      \
  do {                                                                                                                        \
    if (x3)                                                                                                                   \
      hipLaunchKernelGGL((conv5x5_wgrad_kernel<C_, S_, 4>), dim3(groups), dim3(256), lds, st, x, dz1, (float*)ws,             \
                         (const float*)zeros, a1_sign, tiles);                                                                \
    else if (bf)                                                                                                              \
      hipLaunchKernelGGL((conv5x5_wgrad_kernel<C_, S_, 2>), dim3(groups), dim3(256), lds, st, x, dz1, (float*)ws,             \
                         (const float*)zeros, a1_sign, tiles);                                                                \
    UGN_C5W_H2(C_, S_)                                                                                                        \
    else                                                                                                                      \
      hipLaunchKernelGGL((conv5x5_wgrad_kernel<C_, S_>), dim3(groups), dim3(256), lds, st, x, dz1, (float*)ws,                \
                         (const float*)zeros, a1_sign, tiles);                                                                \
  } while (0)
  UGN_REQUIRE(!dz_meta || UGN_WITH_H2, "ugn_conv5x5_in_wgrad: this library was built without the f16x2 set");
  UGN_REQUIRE(!dz_meta || UGN_C5_DS == 32, "ugn_conv5x5_in_wgrad_h2: built with a padded gradient tile");
  if (cin == 1) {
    if (a1_sign) UGN_C5W(1, true); else UGN_C5W(1, false);
  } else {
    if (a1_sign) UGN_C5W(2, true); else UGN_C5W(2, false);
  }
#undef UGN_C5W
#undef UGN_C5W_H2
  UGN_CHECK_LAUNCH("conv5x5_wgrad");
  const int nelem = 25 * cin * 32;
  hipLaunchKernelGGL(reduce5_kernel, dim3((nelem + 7) / 8), dim3(256), 0, st, (const float*)ws, dw, nelem, groups, dz_meta);
  UGN_CHECK_LAUNCH("conv5x5_wgrad reduce");
  return 0;
}
