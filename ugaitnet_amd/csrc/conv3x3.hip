// 3x3 SAME convolutions of the GaitSet-style encoder as fp32 implicit GEMMs on v_mfma_f32_32x32x2_f32.
//
// Replaces the implicit TF Conv2D / Conv2DBackpropInput / Conv2DBackpropFilter + LeakyRelu(Grad) +
// MaxPool(Grad) ops behind reference nets/mj_uwyhNets_ba.py:431-462 (forward) and Keras autodiff (backward).
//
// Forward / data-gradient kernel (one template, `conv3x3_kernel`)
//   GEMM view: M = output pixels, N = output channels, K = 9 taps x K-channels.
//   A workgroup (256 threads = 4 waves) owns a TH x 16 pixel tile and all N channels.  K is walked as
//   (32-channel chunk) x (tap): the input halo tile [(TH+2) x 18 pixels][32 ch] of the chunk sits in LDS with a
//   36-float pixel stride (ds_read_b128 conflict-free), the [N][32] weight slice of the tap is double-buffered
//   in LDS and prefetched through registers one tap ahead, so there is ONE barrier per tap.
//   M-blocks are 2 rows x 16 columns in "pool order" (lane bit0 = x&1, bit1 = y&1, bits2-4 = 2x2 window), so the
//   four pixels of a pooling window land in registers 4q..4q+3 of ONE lane of the 32x32 accumulator and the
//   MaxPool(+argmax) epilogue needs no cross-lane traffic.
//   K order inside an 8-channel group is permuted (lane half h takes channels 4h..4h+3) so both operands are
//   fetched with one ds_read_b128 per 4 MFMAs; A and B use the same permutation, so the sum is unchanged.
//   The data gradient is the same kernel with the HWIO weights read as [tap'][n=cin][k=cout], tap' = 8 - tap.
//
// Weight-gradient kernel (`wgrad3x3_kernel`)
//   GEMM view: M = 32 input channels (one chunk), N = 32/64 output channels, K = pixels; 9 accumulators (one per
//   tap) per wave.  Persistent workgroups walk 8x16 pixel tiles and keep the 9x32x32 partial sums in registers;
//   partial slabs are summed by `reduce_slabs_kernel` (deterministic, no atomics).
#include <stdlib.h>
#include "common.h"
#ifndef UGN_C3_PART
#define UGN_C3_PART 0     // build.py compiles this file five times: 0 = everything but the data gradients, 1-4 = data-gradient groups
#endif

namespace {

// Two workgroups share a CU (one wave of each per SIMD) and every workgroup does identical work, so left alone they run
// in lockstep: both stage (matrix pipe idle), both compute (pipe shared), both store.  Delaying the workgroups of the
// first resident round that sit in an odd wave slot by about one compute phase puts the partners out of phase for the
// whole launch: one computes while the other stages/stores.  Placement only affects speed, never results.
constexpr int kResidentWgs = 512;
__device__ __forceinline__ void ugn_stagger_first_round(int sleeps) {
  if (sleeps > 0 && blockIdx.x < kResidentWgs) {
    const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID.WAVE_ID
    if (wave_slot & 1u)
      for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
  }
}

constexpr int TW = 16;       // tile width (pixels)
constexpr int PW = TW + 2;   // halo tile width
constexpr int CS = 36;       // LDS pixel stride of a 32-channel chunk (floats): 144 B, b128 conflict-free

enum { EPI_LRELU = 0, EPI_LRELU_POOL = 1, EPI_DGRAD = 2 };

// NCF = output channels of the layer, NC = output channels owned by one workgroup (NCF / NC workgroups share a tile)
template <int KC, int NCF, int NC, int HW, int TH>
struct ConvCfg {
  static constexpr int PH = TH + 2, NPIX = PH * PW;
  static constexpr int NSPLIT = NCF / NC;
  static constexpr int MB = TH / 2, NB = NC / 32, WB = MB * NB / 4;
  static constexpr int WN = (NB >= 2 && WB >= 2) ? 2 : 1, WM = WB / WN;
  static constexpr int WAVES_N = NB / WN, WAVES_M = MB / WM;
  static constexpr int NCHUNK = KC / 32;
  static constexpr int SIN = NPIX * CS, SW = NC * CS;
  static constexpr int LDS_USED = (SIN + 2 * SW) * 4;
  // request at least 56 KB so that exactly two workgroups share a CU: with 2400..9600 equal work items the
  // 512 resident workgroups then finish in ~4.7..18.75 rounds (>= 94 % full) instead of quantising at 3 per CU
  static constexpr int LDS_BYTES = LDS_USED < 57344 ? 57344 : LDS_USED;
  static constexpr int IN_ITERS = (NPIX * 8 + 255) / 256;
  static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
  static_assert(WB >= 1 && WM >= 1, "tile too small");
  static_assert(HW % TH == 0 && HW % TW == 0, "tile must divide the image");
};

// EFLAGS (data-gradient epilogue, compile time so that the loads are batched, not branched around):
//   bit 0 = multiply by LeakyReLU'(act), bit 1 = add `addend`, bit 2 = also store the un-multiplied sum to raw_out
template <int KC, int NCF, int NC, int HW, int TH, int IN_UNPOOL, int EPI, int EFLAGS>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const float* __restrict__ in,
                                                          const uint8_t* __restrict__ in_idx,
                                                          const float* __restrict__ w, int flip,
                                                          float* __restrict__ out, uint8_t* __restrict__ out_idx,
                                                          const float* __restrict__ act,
                                                          const float* __restrict__ addend,
                                                          float* __restrict__ raw_out, int stagger) {
  using C = ConvCfg<KC, NCF, NC, HW, TH>;
  ugn_stagger_first_round(stagger);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sIn = smem;
  float* sW0 = smem + C::SIN;
  float* sW1 = sW0 + C::SW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  constexpr int TPX = HW / TW, TPY = HW / TH, TPI = TPX * TPY;
  const int bid = blockIdx.x / C::NSPLIT, nsp = blockIdx.x % C::NSPLIT;
  const int img = bid / TPI, trem = bid % TPI;
  const int ty0 = (trem / TPX) * TH, tx0 = (trem % TPX) * TW;

  // lane's pixel inside an M-block (pool order)
  const int py = (li >> 1) & 1, px = 2 * (li >> 2) + (li & 1);
  int aoff[C::WM], boff[C::WN];
#pragma unroll
  for (int m = 0; m < C::WM; ++m) aoff[m] = ((2 * (wm * C::WM + m) + py) * PW + px) * CS + 4 * lh;
#pragma unroll
  for (int n = 0; n < C::WN; ++n) boff[n] = ((wn * C::WN + n) * 32 + li) * CS + 4 * lh;

  f32x16 acc[C::WM][C::WN];
#pragma unroll
  for (int m = 0; m < C::WM; ++m)
#pragma unroll
    for (int n = 0; n < C::WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int wrow = tid >> 3, wc4 = tid & 7;
  float4 wreg[C::NB];

  for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
    __syncthreads();  // everyone is done with sIn / sW of the previous chunk
    // ---- stage the input halo tile of this chunk (zero outside the image) ----
    {
      float4 v[C::IN_ITERS];
#pragma unroll
      for (int it = 0; it < C::IN_ITERS; ++it) {
        const int e = tid + it * 256;
        const int p = e >> 3, c4 = e & 7;
        const int yy = p / PW, xx = p - yy * PW;
        const int gy = ty0 - 1 + yy, gx = tx0 - 1 + xx;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < C::NPIX * 8 && gy >= 0 && gy < HW && gx >= 0 && gx < HW) {
          if constexpr (IN_UNPOOL) {
            constexpr int HP = HW / 2;
            const size_t o = (((size_t)img * HP + (gy >> 1)) * HP + (gx >> 1)) * KC + chunk * 32 + c4 * 4;
            const float4 d = *reinterpret_cast<const float4*>(in + o);
            const uchar4 id = *reinterpret_cast<const uchar4*>(in_idx + o);
            const int pos = ((gy & 1) << 1) | (gx & 1);
            v[it].x = id.x == pos ? d.x : 0.f;
            v[it].y = id.y == pos ? d.y : 0.f;
            v[it].z = id.z == pos ? d.z : 0.f;
            v[it].w = id.w == pos ? d.w : 0.f;
          } else {
            const size_t o = (((size_t)img * HW + gy) * HW + gx) * KC + chunk * 32 + c4 * 4;
            v[it] = *reinterpret_cast<const float4*>(in + o);
          }
        }
      }
#pragma unroll
      for (int it = 0; it < C::IN_ITERS; ++it) {
        const int e = tid + it * 256;
        if (e < C::NPIX * 8) *reinterpret_cast<float4*>(sIn + (e >> 3) * CS + (e & 7) * 4) = v[it];
      }
    }
    // ---- first weight slice of the chunk ----
    {
      const int t = flip ? 8 : 0;
#pragma unroll
      for (int q = 0; q < C::NB; ++q)
        wreg[q] = *reinterpret_cast<const float4*>(w + ((size_t)(t * NCF + nsp * NC + q * 32 + wrow)) * KC + chunk * 32 + wc4 * 4);
#pragma unroll
      for (int q = 0; q < C::NB; ++q) *reinterpret_cast<float4*>(sW0 + (q * 32 + wrow) * CS + wc4 * 4) = wreg[q];
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) {
        const int t = flip ? 7 - tap : tap + 1;
#pragma unroll
        for (int q = 0; q < C::NB; ++q)
          wreg[q] =
              *reinterpret_cast<const float4*>(w + ((size_t)(t * NCF + nsp * NC + q * 32 + wrow)) * KC + chunk * 32 + wc4 * 4);
      }
      __syncthreads();
      const float* sW = (tap & 1) ? sW1 : sW0;
      const int toff = ((tap / 3) * PW + (tap % 3)) * CS;
      // register double-buffered operand fragments: the ds_read_b128 of group g+1 are in flight under the MFMAs of
      // group g (left to itself hipcc reuses one register set and exposes the LDS latency after every 4 MFMAs)
      float4 a[2][C::WM], b[2][C::WN];
#pragma unroll
      for (int m = 0; m < C::WM; ++m) a[0][m] = *reinterpret_cast<const float4*>(sIn + aoff[m] + toff);
#pragma unroll
      for (int n = 0; n < C::WN; ++n) b[0][n] = *reinterpret_cast<const float4*>(sW + boff[n]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g < 3) {
#pragma unroll
          for (int m = 0; m < C::WM; ++m)
            a[nxt][m] = *reinterpret_cast<const float4*>(sIn + aoff[m] + toff + 8 * (g + 1));
#pragma unroll
          for (int n = 0; n < C::WN; ++n) b[nxt][n] = *reinterpret_cast<const float4*>(sW + boff[n] + 8 * (g + 1));
          __builtin_amdgcn_sched_group_barrier(0x100, C::WM + C::WN, 0);  // the next group's DS reads first ...
        }
#pragma unroll
        for (int m = 0; m < C::WM; ++m)
#pragma unroll
          for (int n = 0; n < C::WN; ++n) {
            acc[m][n] = ugn_mfma(a[cur][m].x, b[cur][n].x, acc[m][n]);
            acc[m][n] = ugn_mfma(a[cur][m].y, b[cur][n].y, acc[m][n]);
            acc[m][n] = ugn_mfma(a[cur][m].z, b[cur][n].z, acc[m][n]);
            acc[m][n] = ugn_mfma(a[cur][m].w, b[cur][n].w, acc[m][n]);
          }
        __builtin_amdgcn_sched_group_barrier(0x8, 4 * C::WM * C::WN, 0);   // ... then this group's MFMAs
      }
      if (tap < 8) {
        float* sWn = ((tap + 1) & 1) ? sW1 : sW0;
#pragma unroll
        for (int q = 0; q < C::NB; ++q) *reinterpret_cast<float4*>(sWn + (q * 32 + wrow) * CS + wc4 * 4) = wreg[q];
      }
    }
  }

  // ---- epilogue ----
#pragma unroll
  for (int m = 0; m < C::WM; ++m) {
    const int mbi = wm * C::WM + m;
#pragma unroll
    for (int n = 0; n < C::WN; ++n) {
      const int co = nsp * NC + (wn * C::WN + n) * 32 + li;
      if constexpr (EPI == EPI_LRELU_POOL) {
        constexpr int HP = HW / 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float best = ugn_lrelu(acc[m][n][4 * q]);
          int bi = 0;
#pragma unroll
          for (int r = 1; r < 4; ++r) {
            const float v = ugn_lrelu(acc[m][n][4 * q + r]);
            if (v > best) { best = v; bi = r; }
          }
          const int oy = ty0 / 2 + mbi, ox = tx0 / 2 + lh + 2 * q;
          const size_t o = (((size_t)img * HP + oy) * HP + ox) * NCF + co;
          out[o] = best;
          out_idx[o] = (uint8_t)bi;
        }
      } else {
        size_t o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int y = ty0 + 2 * mbi + ((r >> 1) & 1);
          const int x = tx0 + 2 * (lh + 2 * (r >> 2)) + (r & 1);
          o[r] = (((size_t)img * HW + y) * HW + x) * NCF + co;
        }
        if constexpr (EPI == EPI_LRELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) out[o[r]] = ugn_lrelu(acc[m][n][r]);
        } else {
          float av[16], dv[16];
          if constexpr (EFLAGS & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) av[r] = act[o[r]];
          }
          if constexpr (EFLAGS & 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dv[r] = addend[o[r]];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[m][n][r];
            if constexpr (EFLAGS & 2) v += dv[r];
            if constexpr (EFLAGS & 4) raw_out[o[r]] = v;
            if constexpr (EFLAGS & 1) v *= ugn_lrelu_slope(av[r]);
            out[o[r]] = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Weights-resident persistent variant for the layers whose GEMM K is one 32-channel chunk (a2 fwd/dgrad, a3/b1 fwd).
// All nine [N][32] weight slices stay in LDS for the whole launch; the halo tile is double-buffered and the NEXT tile's
// halo is fetched into registers while the current tile computes, so a tile costs ONE barrier and no exposed global
// latency.  One workgroup per CU (LDS 93..135 KB), 256 persistent workgroups striding over the tiles.
// ------------------------------------------------------------------------------------------------------
template <int NC, int HW, int TH>
struct ResCfg {
  static constexpr int PH = TH + 2, NPIX = PH * PW;
  static constexpr int MB = TH / 2, NB = NC / 32, WB = MB * NB / 4;
  static constexpr int WN = (NB >= 2 && WB >= 2) ? 2 : 1, WM = WB / WN;
  static constexpr int WAVES_N = NB / WN, WAVES_M = MB / WM;
  static constexpr int SIN = NPIX * CS, SW = 9 * NC * CS;
  static constexpr int LDS_BYTES = (2 * SIN + SW) * 4;
  static constexpr int IN_ITERS = (NPIX * 8 + 255) / 256;
  static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
};

template <int HW, int TH, int KCT, int IN_UNPOOL, int ITERS>
__device__ __forceinline__ void res_load_halo(float4 (&v)[ITERS], const float* __restrict__ in,
                                              const uint8_t* __restrict__ in_idx, int item, int tid) {
  constexpr int TPX = HW / TW, TPI = TPX * (HW / TH), NPIX = (TH + 2) * PW;
  const int img = item / TPI, trem = item % TPI;
  const int ty0 = (trem / TPX) * TH, tx0 = (trem % TPX) * TW;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int e = tid + it * 256;
    const int p = e >> 3, c4 = e & 7;
    const int yy = p / PW, xx = p - yy * PW;
    const int gy = ty0 - 1 + yy, gx = tx0 - 1 + xx;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < NPIX * 8 && gy >= 0 && gy < HW && gx >= 0 && gx < HW) {
      if constexpr (IN_UNPOOL) {
        constexpr int HP = HW / 2;
        const size_t o = (((size_t)img * HP + (gy >> 1)) * HP + (gx >> 1)) * KCT + c4 * 4;
        const float4 d = *reinterpret_cast<const float4*>(in + o);
        const uchar4 id = *reinterpret_cast<const uchar4*>(in_idx + o);
        const int pos = ((gy & 1) << 1) | (gx & 1);
        v[it].x = id.x == pos ? d.x : 0.f;
        v[it].y = id.y == pos ? d.y : 0.f;
        v[it].z = id.z == pos ? d.z : 0.f;
        v[it].w = id.w == pos ? d.w : 0.f;
      } else {
        v[it] = *reinterpret_cast<const float4*>(in + (((size_t)img * HW + gy) * HW + gx) * KCT + c4 * 4);
      }
    }
  }
}

template <int NC, int HW, int TH, int IN_UNPOOL, int EPI, int EFLAGS>
__global__ __launch_bounds__(256, 1) void conv3x3_resident_kernel(const float* __restrict__ in,
                                                                   const uint8_t* __restrict__ in_idx,
                                                                   const float* __restrict__ w, int flip,
                                                                   float* __restrict__ out, uint8_t* __restrict__ out_idx,
                                                                   const float* __restrict__ act,
                                                                   const float* __restrict__ addend,
                                                                   float* __restrict__ raw_out, int nitems) {
  using C = ResCfg<NC, HW, TH>;
  constexpr int KC = 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                 // [9][NC][36]
  float* sIn0 = smem + C::SW;       // two halo buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int TPX = HW / TW, TPI = TPX * (HW / TH);

  // ---- all weights, once ----
  for (int e = tid; e < 9 * NC * 8; e += 256) {
    const int row = e >> 3, c4 = e & 7;          // row = tap*NC + n
    const int tap = row / NC, n = row % NC;
    const int t = flip ? 8 - tap : tap;
    *reinterpret_cast<float4*>(sW + row * CS + c4 * 4) =
        *reinterpret_cast<const float4*>(w + ((size_t)(t * NC + n)) * KC + c4 * 4);
  }
  const int py = (li >> 1) & 1, px = 2 * (li >> 2) + (li & 1);
  int aoff[C::WM], boff[C::WN];
#pragma unroll
  for (int m = 0; m < C::WM; ++m) aoff[m] = ((2 * (wm * C::WM + m) + py) * PW + px) * CS + 4 * lh;
#pragma unroll
  for (int n = 0; n < C::WN; ++n) boff[n] = ((wn * C::WN + n) * 32 + li) * CS + 4 * lh;

  int item = blockIdx.x;
  float4 vin[C::IN_ITERS];
  if (item < nitems) {
    res_load_halo<HW, TH, KC, IN_UNPOOL>(vin, in, in_idx, item, tid);
#pragma unroll
    for (int it = 0; it < C::IN_ITERS; ++it) {
      const int e = tid + it * 256;
      if (e < C::NPIX * 8) *reinterpret_cast<float4*>(sIn0 + (e >> 3) * CS + (e & 7) * 4) = vin[it];
    }
  }
  int buf = 0;
  for (; item < nitems; item += gridDim.x, buf ^= 1) {
    const int next = item + gridDim.x;
    if (next < nitems) res_load_halo<HW, TH, KC, IN_UNPOOL>(vin, in, in_idx, next, tid);
    const int img = item / TPI, trem = item % TPI;
    const int ty0 = (trem / TPX) * TH, tx0 = (trem % TPX) * TW;
    // output offsets (+ early loads of the epilogue operands: they land while the tile computes)
    size_t oo[C::WM][C::WN][EPI == EPI_LRELU_POOL ? 4 : 16];
    float av[C::WM][C::WN][16], dv[C::WM][C::WN][16];
#pragma unroll
    for (int m = 0; m < C::WM; ++m)
#pragma unroll
      for (int n = 0; n < C::WN; ++n) {
        const int mbi = wm * C::WM + m, co = (wn * C::WN + n) * 32 + li;
        if constexpr (EPI == EPI_LRELU_POOL) {
          constexpr int HP = HW / 2;
#pragma unroll
          for (int q = 0; q < 4; ++q) oo[m][n][q] = (((size_t)img * HP + ty0 / 2 + mbi) * HP + tx0 / 2 + lh + 2 * q) * NC + co;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int y = ty0 + 2 * mbi + ((r >> 1) & 1), x = tx0 + 2 * (lh + 2 * (r >> 2)) + (r & 1);
            oo[m][n][r] = (((size_t)img * HW + y) * HW + x) * NC + co;
            if constexpr (EPI == EPI_DGRAD && (EFLAGS & 1)) av[m][n][r] = act[oo[m][n][r]];
            if constexpr (EPI == EPI_DGRAD && (EFLAGS & 2)) dv[m][n][r] = addend[oo[m][n][r]];
          }
        }
      }
    __syncthreads();   // sIn[buf] (and, first time, sW) visible; everyone finished reading sIn[buf^1]
    const float* sIn = sIn0 + buf * C::SIN;
    f32x16 acc[C::WM][C::WN];
#pragma unroll
    for (int m = 0; m < C::WM; ++m)
#pragma unroll
      for (int n = 0; n < C::WN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float4 a[2][C::WM], b[2][C::WN];
#pragma unroll
    for (int m = 0; m < C::WM; ++m) a[0][m] = *reinterpret_cast<const float4*>(sIn + aoff[m]);
#pragma unroll
    for (int n = 0; n < C::WN; ++n) b[0][n] = *reinterpret_cast<const float4*>(sW + boff[n]);
#pragma unroll
    for (int s = 0; s < 36; ++s) {     // s = tap*4 + group
      const int cur = s & 1, nxt = cur ^ 1;
      if (s < 35) {
        const int tap = (s + 1) >> 2, g = (s + 1) & 3;
        const int toff = ((tap / 3) * PW + (tap % 3)) * CS + 8 * g;
#pragma unroll
        for (int m = 0; m < C::WM; ++m) a[nxt][m] = *reinterpret_cast<const float4*>(sIn + aoff[m] + toff);
#pragma unroll
        for (int n = 0; n < C::WN; ++n) b[nxt][n] = *reinterpret_cast<const float4*>(sW + boff[n] + tap * NC * CS + 8 * g);
        __builtin_amdgcn_sched_group_barrier(0x100, C::WM + C::WN, 0);
      }
#pragma unroll
      for (int m = 0; m < C::WM; ++m)
#pragma unroll
        for (int n = 0; n < C::WN; ++n) {
          acc[m][n] = ugn_mfma(a[cur][m].x, b[cur][n].x, acc[m][n]);
          acc[m][n] = ugn_mfma(a[cur][m].y, b[cur][n].y, acc[m][n]);
          acc[m][n] = ugn_mfma(a[cur][m].z, b[cur][n].z, acc[m][n]);
          acc[m][n] = ugn_mfma(a[cur][m].w, b[cur][n].w, acc[m][n]);
        }
      __builtin_amdgcn_sched_group_barrier(0x8, 4 * C::WM * C::WN, 0);
    }
    // next tile's halo into the other buffer (its last readers passed this iteration's barrier)
    if (next < nitems) {
      float* sNext = sIn0 + (buf ^ 1) * C::SIN;
#pragma unroll
      for (int it = 0; it < C::IN_ITERS; ++it) {
        const int e = tid + it * 256;
        if (e < C::NPIX * 8) *reinterpret_cast<float4*>(sNext + (e >> 3) * CS + (e & 7) * 4) = vin[it];
      }
    }
    // ---- epilogue ----
#pragma unroll
    for (int m = 0; m < C::WM; ++m)
#pragma unroll
      for (int n = 0; n < C::WN; ++n) {
        if constexpr (EPI == EPI_LRELU_POOL) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float best = ugn_lrelu(acc[m][n][4 * q]);
            int bi = 0;
#pragma unroll
            for (int r = 1; r < 4; ++r) {
              const float v = ugn_lrelu(acc[m][n][4 * q + r]);
              if (v > best) { best = v; bi = r; }
            }
            out[oo[m][n][q]] = best;
            out_idx[oo[m][n][q]] = (uint8_t)bi;
          }
        } else if constexpr (EPI == EPI_LRELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) out[oo[m][n][r]] = ugn_lrelu(acc[m][n][r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[m][n][r];
            if constexpr (EFLAGS & 2) v += dv[m][n][r];
            if constexpr (EFLAGS & 4) raw_out[oo[m][n][r]] = v;
            if constexpr (EFLAGS & 1) v *= ugn_lrelu_slope(av[m][n][r]);
            out[oo[m][n][r]] = v;
          }
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------------
constexpr int WG_TH = 8;                       // wgrad pixel tile: 8 rows x 16 columns
constexpr int WG_NPIX = (WG_TH + 2) * PW;      // halo tile pixels
constexpr int WG_TPIX = WG_TH * TW;            // 128 pixels = GEMM K per tile

template <int CI, int CO, int HW, int COC, int DZ_UNPOOL>
struct WgradCfg {
  static constexpr int DS = COC + 4;           // dz pixel stride in LDS
  static constexpr int SIN = WG_NPIX * CS, SDZ = WG_TPIX * DS;
  static constexpr int NBW = COC / 32, PS = 4 / NBW, PPW = WG_TPIX / PS;
  static constexpr int STAGE_BYTES = (SIN + SDZ) * 4;
  static constexpr int RED_BYTES = 9 * 16 * 64 * 4;  // one wave's accumulators
  static constexpr int LDS_BYTES = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
  static constexpr int IN_ITERS = (WG_NPIX * 8 + 255) / 256;
  static constexpr int DZ_ITERS = WG_TPIX * (COC / 4) / 256;
  static constexpr int NCOMBO = (CI / 32) * (CO / COC);
};

template <int CI, int CO, int HW, int COC, int DZ_UNPOOL>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(const float* __restrict__ in,
                                                           const float* __restrict__ dz,
                                                           const uint8_t* __restrict__ dz_idx,
                                                           float* __restrict__ slab, int tiles_total, int groups) {
  using C = WgradCfg<CI, CO, HW, COC, DZ_UNPOOL>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sIn = smem;
  float* sDz = smem + C::SIN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cb = wave % C::NBW, ps = wave / C::NBW;
  const int combo = blockIdx.x / groups, grp = blockIdx.x % groups;
  constexpr int NCO = CO / COC;
  const int cic = combo / NCO, coc = combo % NCO;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  constexpr int TPX = HW / TW, TPY = HW / WG_TH, TPI = TPX * TPY;
  // lane bases: pixel k = ps*PPW + 2*kp + lh
  const int k0 = ps * C::PPW + lh;
  const int abase = ((k0 / TW) * PW + (k0 % TW)) * CS + li;
  const int bbase = k0 * C::DS + cb * 32 + li;

  for (int tile = grp; tile < tiles_total; tile += groups) {
    const int img = tile / TPI, trem = tile % TPI;
    const int ty0 = (trem / TPX) * WG_TH, tx0 = (trem % TPX) * TW;
    __syncthreads();
    {
      float4 v[C::IN_ITERS];
#pragma unroll
      for (int it = 0; it < C::IN_ITERS; ++it) {
        const int e = tid + it * 256;
        const int p = e >> 3, c4 = e & 7;
        const int yy = p / PW, xx = p - yy * PW;
        const int gy = ty0 - 1 + yy, gx = tx0 - 1 + xx;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < WG_NPIX * 8 && gy >= 0 && gy < HW && gx >= 0 && gx < HW)
          v[it] = *reinterpret_cast<const float4*>(in + (((size_t)img * HW + gy) * HW + gx) * CI + cic * 32 + c4 * 4);
      }
      float4 d[C::DZ_ITERS];
#pragma unroll
      for (int it = 0; it < C::DZ_ITERS; ++it) {
        const int e = tid + it * 256;
        constexpr int Q = COC / 4;
        const int p = e / Q, c4 = e % Q;
        const int gy = ty0 + p / TW, gx = tx0 + p % TW;
        if constexpr (DZ_UNPOOL) {
          constexpr int HP = HW / 2;
          const size_t o = (((size_t)img * HP + (gy >> 1)) * HP + (gx >> 1)) * CO + coc * COC + c4 * 4;
          const float4 g = *reinterpret_cast<const float4*>(dz + o);
          const uchar4 id = *reinterpret_cast<const uchar4*>(dz_idx + o);
          const int pos = ((gy & 1) << 1) | (gx & 1);
          d[it].x = id.x == pos ? g.x : 0.f;
          d[it].y = id.y == pos ? g.y : 0.f;
          d[it].z = id.z == pos ? g.z : 0.f;
          d[it].w = id.w == pos ? g.w : 0.f;
        } else {
          d[it] = *reinterpret_cast<const float4*>(dz + (((size_t)img * HW + gy) * HW + gx) * CO + coc * COC + c4 * 4);
        }
      }
#pragma unroll
      for (int it = 0; it < C::IN_ITERS; ++it) {
        const int e = tid + it * 256;
        if (e < WG_NPIX * 8) *reinterpret_cast<float4*>(sIn + (e >> 3) * CS + (e & 7) * 4) = v[it];
      }
#pragma unroll
      for (int it = 0; it < C::DZ_ITERS; ++it) {
        const int e = tid + it * 256;
        constexpr int Q = COC / 4;
        *reinterpret_cast<float4*>(sDz + (e / Q) * C::DS + (e % Q) * 4) = d[it];
      }
    }
    __syncthreads();
    // operands of pixel pair kp+1 are fetched (into a second register set) under the 9 MFMAs of pixel pair kp
    float av[2][9], bv[2];
    bv[0] = sDz[bbase];
#pragma unroll
    for (int t = 0; t < 9; ++t) av[0][t] = sIn[abase + ((t / 3) * PW + (t % 3)) * CS];
#pragma unroll
    for (int kp = 0; kp < C::PPW / 2; ++kp) {
      // pixel k = ps*PPW + 2*kp + lh ; rows of 16 pixels
      const int cur = kp & 1, nxt = cur ^ 1;
      if (kp + 1 < C::PPW / 2) {
        const int ko = ((2 * (kp + 1)) / TW) * PW + ((2 * (kp + 1)) % TW);  // compile-time after unrolling
        bv[nxt] = sDz[bbase + 2 * (kp + 1) * C::DS];
#pragma unroll
        for (int t = 0; t < 9; ++t) av[nxt][t] = sIn[abase + (ko + (t / 3) * PW + (t % 3)) * CS];
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = ugn_mfma(av[cur][t], bv[cur], acc[t]);
      __builtin_amdgcn_sched_group_barrier(0x8, 9, 0);
    }
  }

  // ---- reduce the PS pixel-split waves that share an output block, then write the slab ----
  float* sRed = smem;
#pragma unroll 1
  for (int src = C::NBW; src < 4; ++src) {
    __syncthreads();
    if (wave == src) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sRed[(t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == src % C::NBW) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += sRed[(t * 16 + r) * 64 + lane];
    }
  }
  if (wave < C::NBW) {
    float* dst = slab + (size_t)grp * 9 * CI * CO;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = cic * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int co = coc * COC + cb * 32 + li;
        dst[((size_t)t * CI + ci) * CO + co] = acc[t][r];
      }
  }
}

// dst[e] = sum_g src[g][e] (e in float4 units): 256 threads = 16 elements x 16 slab lanes; each lane sums every 16th slab
// in a fixed order and the 16 partial sums are combined in a fixed order through LDS -> one launch, bitwise reproducible.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                           int nelem4, int nin) {
  __shared__ float4 sR[16][16];
  const int le = threadIdx.x & 15, lg = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + le;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < nelem4)
    for (int g = lg; g < nin; g += 16) {
      const float4 v = src[(size_t)g * nelem4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  sR[lg][le] = s;
  __syncthreads();
  if (lg == 0 && e < nelem4) {
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 v = sR[k][le];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dst[e] = s;
  }
}

__global__ void pack3x3_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin, int cout) {
  // w [9][cin][cout] -> wp [9][cout][cin]
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = 9 * cin * cout;
  if (e >= total) return;
  const int ci = e % cin, co = (e / cin) % cout, t = e / (cin * cout);
  wp[e] = w[((size_t)t * cin + ci) * cout + co];
}

inline bool ugn_stagger_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("UGN_STAGGER");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v != 0;
}

template <int KC, int NCF, int NC, int HW, int TH, int IN_UNPOOL, int EPI, int EFLAGS>
int launch_conv(const float* in, const uint8_t* in_idx, const float* w, int flip, float* out, uint8_t* out_idx,
                const float* act, const float* addend, float* raw_out, int n, hipStream_t st) {
  using C = ConvCfg<KC, NCF, NC, HW, TH>;
  auto kern = conv3x3_kernel<KC, NCF, NC, HW, TH, IN_UNPOOL, EPI, EFLAGS>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) { ugn_set_error("conv3x3: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const int grid = n * (HW / TH) * (HW / TW) * C::NSPLIT;
  // one workgroup's matrix work: NCHUNK * 9 taps * 16 k-pairs * WB blocks, 64 cycles each; s_sleep(127) = 8128 cycles
  int stagger = 0;
  if (ugn_stagger_enabled() && grid > kResidentWgs) stagger = (C::NCHUNK * 9 * 16 * C::WB * 64 + 4064) / 8128;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, st, in, in_idx, w, flip, out, out_idx, act, addend,
                     raw_out, stagger);
  UGN_CHECK_LAUNCH("conv3x3");
  return 0;
}

inline bool ugn_resident_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("UGN_RESIDENT");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v != 0;
}

constexpr int kResidentGrid = 256;   // one persistent workgroup per CU

template <int NC, int HW, int TH, int IN_UNPOOL, int EPI, int EFLAGS>
int launch_resident(const float* in, const uint8_t* in_idx, const float* w, int flip, float* out, uint8_t* out_idx,
                    const float* act, const float* addend, float* raw_out, int n, hipStream_t st) {
  using C = ResCfg<NC, HW, TH>;
  auto kern = conv3x3_resident_kernel<NC, HW, TH, IN_UNPOOL, EPI, EFLAGS>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) { ugn_set_error("conv3x3 resident: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const int nitems = n * (HW / TH) * (HW / TW);
  const int grid = nitems < kResidentGrid ? nitems : kResidentGrid;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, st, in, in_idx, w, flip, out, out_idx, act, addend,
                     raw_out, nitems);
  UGN_CHECK_LAUNCH("conv3x3 resident");
  return 0;
}

template <int NC, int HW, int TH, int IN_UNPOOL>
int launch_resident_dgrad(const float* in, const uint8_t* in_idx, const float* w, float* out, const float* act,
                          const float* addend, float* raw_out, int n, hipStream_t st) {
  const int flags = (act ? 1 : 0) | (addend ? 2 : 0) | (raw_out ? 4 : 0);
#define UGN_RDG(F_)                                                                                            \
  case F_:                                                                                                     \
    return launch_resident<NC, HW, TH, IN_UNPOOL, EPI_DGRAD, F_>(in, in_idx, w, 1, out, nullptr, act, addend,  \
                                                                 raw_out, n, st);
  switch (flags) {
    UGN_RDG(0) UGN_RDG(1) UGN_RDG(2) UGN_RDG(3) UGN_RDG(4) UGN_RDG(5) UGN_RDG(6) UGN_RDG(7)
  }
#undef UGN_RDG
  return UGN_EINVAL;
}

template <int KC, int NCF, int NC, int HW, int TH, int IN_UNPOOL>
int launch_dgrad(const float* in, const uint8_t* in_idx, const float* w, float* out, const float* act,
                 const float* addend, float* raw_out, int n, hipStream_t st) {
  const int flags = (act ? 1 : 0) | (addend ? 2 : 0) | (raw_out ? 4 : 0);
#define UGN_DG(F_)                                                                                              \
  case F_:                                                                                                      \
    return launch_conv<KC, NCF, NC, HW, TH, IN_UNPOOL, EPI_DGRAD, F_>(in, in_idx, w, 1, out, nullptr, act, addend, \
                                                                      raw_out, n, st);
  switch (flags) {
    UGN_DG(0) UGN_DG(1) UGN_DG(2) UGN_DG(3) UGN_DG(4) UGN_DG(5) UGN_DG(6) UGN_DG(7)
  }
#undef UGN_DG
  return UGN_EINVAL;
}

constexpr int WGRAD_WGS = 512;  // persistent workgroups (2 per CU)

template <int CI, int CO, int HW, int COC, int DZ_UNPOOL>
int launch_wgrad(const float* in, const float* dz, const uint8_t* dz_idx, float* dw, int n, float* ws,
                 size_t ws_floats, hipStream_t st) {
  using C = WgradCfg<CI, CO, HW, COC, DZ_UNPOOL>;
  auto kern = wgrad3x3_kernel<CI, CO, HW, COC, DZ_UNPOOL>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) { ugn_set_error("wgrad3x3: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const int tiles = n * (HW / WG_TH) * (HW / TW);
  int groups = WGRAD_WGS / C::NCOMBO;
  if (groups > tiles) groups = tiles;
  const size_t nelem = (size_t)9 * CI * CO;
  if (ws_floats < nelem * (size_t)groups) {
    ugn_set_error("wgrad3x3: workspace too small (%zu < %zu floats)", ws_floats, nelem * (size_t)groups);
    return UGN_EINVAL;
  }
  hipLaunchKernelGGL(kern, dim3(groups * C::NCOMBO), dim3(256), C::LDS_BYTES, st, in, dz, dz_idx, ws, tiles, groups);
  UGN_CHECK_LAUNCH("wgrad3x3");
  const int nelem4 = (int)(nelem / 4);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((nelem4 + 15) / 16), dim3(256), 0, st, (const float4*)ws, (float4*)dw, nelem4,
                     groups);
  UGN_CHECK_LAUNCH("wgrad3x3 reduce");
  return 0;
}

}  // namespace

#if UGN_C3_PART == 0
extern "C" int ugn_pack3x3(const float* w_hwio, float* w_packed, int cin, int cout, void* stream) {
  UGN_REQUIRE(w_hwio && w_packed && cin > 0 && cout > 0, "ugn_pack3x3: bad arguments");
  const int total = 9 * cin * cout;
  hipLaunchKernelGGL(pack3x3_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_hwio, w_packed,
                     cin, cout);
  UGN_CHECK_LAUNCH("pack3x3");
  return 0;
}

extern "C" int ugn_conv3x3_fwd(const float* in, const float* wp, float* out, uint8_t* out_idx, int n, int hw, int cin,
                               int cout, int pool, void* stream) {
  UGN_REQUIRE(in && wp && out && n > 0, "ugn_conv3x3_fwd: null pointer or n <= 0");
  UGN_REQUIRE(!pool || out_idx, "ugn_conv3x3_fwd: pool needs out_idx");
  hipStream_t st = (hipStream_t)stream;
  if (ugn_resident_enabled()) {
    if (cin == 32 && cout == 32 && hw == 64 && pool)
      return launch_resident<32, 64, 16, 0, EPI_LRELU_POOL, 0>(in, nullptr, wp, 0, out, out_idx, nullptr, nullptr, nullptr, n, st);
    if (cin == 32 && cout == 64 && hw == 32 && !pool)
      return launch_resident<64, 32, 8, 0, EPI_LRELU, 0>(in, nullptr, wp, 0, out, nullptr, nullptr, nullptr, nullptr, n, st);
  }
#define FWD(KC_, NCF_, NC_, HW_, TH_, P_)                                                                       \
  if (cin == KC_ && cout == NCF_ && hw == HW_ && (pool != 0) == (P_ != 0))                                        \
    return launch_conv<KC_, NCF_, NC_, HW_, TH_, 0, P_ ? EPI_LRELU_POOL : EPI_LRELU, 0>(                          \
        in, nullptr, wp, 0, out, out_idx, nullptr, nullptr, nullptr, n, st);
  FWD(32, 32, 32, 64, 16, 1)    // a2
  FWD(32, 64, 64, 32, 16, 0)    // a3, b1
  FWD(64, 64, 64, 32, 16, 1)    // a4, b2
  FWD(64, 128, 64, 16, 8, 0)    // a5, b3 : 2 workgroups per 8x16 tile (64 channels each) -> 2400 items / 600 frames
  FWD(128, 128, 64, 16, 8, 0)   // a6, b4
#undef FWD
  UGN_REQUIRE(false, "ugn_conv3x3_fwd: unsupported shape cin=%d cout=%d hw=%d pool=%d", cin, cout, hw, pool);
}

#endif  // UGN_C3_PART == 0 (pack, forward)

// The data gradient is 48 kernel instantiations (five shapes + the resident form, eight epilogue variants each): most of this file's
// compile time.  The file is therefore compiled as FIVE objects (build.py: -DUGN_C3_PART=0..4): part 0 holds everything but the data
// gradients, parts 1-4 one group of data-gradient shapes each behind ugn_c3::dgrad_part<k>.
namespace ugn_c3 {
int dgrad_part1(const float* dz, const uint8_t* dz_idx, const float* w, const float* act, const float* addend, float* out, float* raw_out,
                int n, int hw, int cin, int cout, hipStream_t st);
int dgrad_part2(const float* dz, const uint8_t* dz_idx, const float* w, const float* act, const float* addend, float* out, float* raw_out,
                int n, int hw, int cin, int cout, hipStream_t st);
int dgrad_part3(const float* dz, const uint8_t* dz_idx, const float* w, const float* act, const float* addend, float* out, float* raw_out,
                int n, int hw, int cin, int cout, hipStream_t st);
int dgrad_part4(const float* dz, const uint8_t* dz_idx, const float* w, const float* act, const float* addend, float* out, float* raw_out,
                int n, int hw, int cin, int cout, hipStream_t st);
}  // namespace ugn_c3

// kernel K channels = forward cout, kernel N channels = forward cin
#define DGR(CI_, NCW_, CO_, HW_, TH_, U_)                                                                 \
  if (cin == CI_ && cout == CO_ && hw == HW_ && unpool == U_)                                                \
    return launch_dgrad<CO_, CI_, NCW_, HW_, TH_, U_>(dz, dz_idx, w, out, act, addend, raw_out, n, st);
#if UGN_C3_PART == 1
int ugn_c3::dgrad_part1(const float* dz, const uint8_t* dz_idx, const float* w, const float* act, const float* addend, float* out,
                        float* raw_out, int n, int hw, int cin, int cout, hipStream_t st) {
  if (ugn_resident_enabled()) return launch_resident_dgrad<32, 64, 16, 1>(dz, dz_idx, w, out, act, addend, raw_out, n, st);
  return dgrad_part4(dz, dz_idx, w, act, addend, out, raw_out, n, hw, cin, cout, st);
}
#elif UGN_C3_PART == 4
int ugn_c3::dgrad_part4(const float* dz, const uint8_t* dz_idx, const float* w, const float* act, const float* addend, float* out,
                        float* raw_out, int n, int hw, int cin, int cout, hipStream_t st) {
  const int unpool = dz_idx != nullptr;
  DGR(32, 32, 32, 64, 16, 1)    // a2, the non-resident form (UGN_RESIDENT=0)
  return UGN_EINVAL;
}
#elif UGN_C3_PART == 2
int ugn_c3::dgrad_part2(const float* dz, const uint8_t* dz_idx, const float* w, const float* act, const float* addend, float* out,
                        float* raw_out, int n, int hw, int cin, int cout, hipStream_t st) {
  const int unpool = dz_idx != nullptr;
  DGR(32, 32, 64, 32, 16, 0)    // a3, b1
  DGR(64, 64, 64, 32, 16, 1)    // a4, b2
  return UGN_EINVAL;
}
#elif UGN_C3_PART == 3
int ugn_c3::dgrad_part3(const float* dz, const uint8_t* dz_idx, const float* w, const float* act, const float* addend, float* out,
                        float* raw_out, int n, int hw, int cin, int cout, hipStream_t st) {
  const int unpool = dz_idx != nullptr;
  DGR(64, 32, 128, 16, 8, 0)    // a5, b3 : 8x16 tiles x 2 channel halves -> 2400 items / 600 frames
  DGR(128, 64, 128, 16, 8, 0)   // a6, b4
  return UGN_EINVAL;
}
#else
extern "C" int ugn_conv3x3_dgrad(const float* dz, const uint8_t* dz_idx, const float* w, const float* act,
                                 const float* addend, float* out, float* raw_out, int n, int hw, int cin, int cout,
                                 void* stream) {
  UGN_REQUIRE(dz && w && out && n > 0, "ugn_conv3x3_dgrad: null pointer or n <= 0");
  hipStream_t st = (hipStream_t)stream;
  const int unpool = dz_idx != nullptr;
  if (cin == 32 && cout == 32 && hw == 64 && unpool) return ugn_c3::dgrad_part1(dz, dz_idx, w, act, addend, out, raw_out, n, hw, cin, cout, st);
  if (hw == 32 && ((cin == 32 && cout == 64 && !unpool) || (cin == 64 && cout == 64 && unpool)))
    return ugn_c3::dgrad_part2(dz, dz_idx, w, act, addend, out, raw_out, n, hw, cin, cout, st);
  if (hw == 16 && cout == 128 && (cin == 64 || cin == 128) && !unpool)
    return ugn_c3::dgrad_part3(dz, dz_idx, w, act, addend, out, raw_out, n, hw, cin, cout, st);
  UGN_REQUIRE(false, "ugn_conv3x3_dgrad: unsupported shape cin=%d cout=%d hw=%d unpool=%d", cin, cout, hw, unpool);
}
#endif
#undef DGR

#if UGN_C3_PART == 0
static bool wgrad_cfg(int hw, int cin, int cout, int* coc) {
  if ((hw == 64 && cin == 32 && cout == 32) || (hw == 32 && cin == 32 && cout == 64) ||
      (hw == 32 && cin == 64 && cout == 64) || (hw == 16 && cin == 64 && cout == 128) ||
      (hw == 16 && cin == 128 && cout == 128)) {
    *coc = cout >= 64 ? 64 : 32;
    return true;
  }
  return false;
}

extern "C" size_t ugn_conv3x3_wgrad_ws(int n, int hw, int cin, int cout) {
  int coc;
  if (!wgrad_cfg(hw, cin, cout, &coc) || n <= 0) return 0;
  const int ncombo = (cin / 32) * (cout / coc);
  const long tiles = (long)n * (hw / WG_TH) * (hw / TW);
  long groups = WGRAD_WGS / ncombo;
  if (groups > tiles) groups = tiles;
  return (size_t)9 * cin * cout * (size_t)groups * sizeof(float);
}

extern "C" int ugn_conv3x3_wgrad(const float* in, const float* dz, const uint8_t* dz_idx, float* dw, int n, int hw,
                                 int cin, int cout, void* ws, size_t ws_bytes, void* stream) {
  UGN_REQUIRE(in && dz && dw && ws && n > 0, "ugn_conv3x3_wgrad: null pointer or n <= 0");
  hipStream_t st = (hipStream_t)stream;
  const int unpool = dz_idx != nullptr;
#define WGR(CI_, CO_, HW_, COC_, U_)                                  \
  if (cin == CI_ && cout == CO_ && hw == HW_ && unpool == U_)          \
    return launch_wgrad<CI_, CO_, HW_, COC_, U_>(in, dz, dz_idx, dw, n, (float*)ws, ws_bytes / sizeof(float), st);
  WGR(32, 32, 64, 32, 1)    // a2
  WGR(32, 64, 32, 64, 0)    // a3, b1
  WGR(64, 64, 32, 64, 1)    // a4, b2
  WGR(64, 128, 16, 64, 0)   // a5, b3
  WGR(128, 128, 16, 64, 0)  // a6, b4
#undef WGR
  UGN_REQUIRE(false, "ugn_conv3x3_wgrad: unsupported shape cin=%d cout=%d hw=%d unpool=%d", cin, cout, hw, unpool);
}
#endif  // UGN_C3_PART == 0
