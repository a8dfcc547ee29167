// Direct 3x3 convolution (forward / data gradient) of H2 tensors on v_mfma_f32_32x32x16_f16 -- see mm_common.h for the
// format and why it exists.  Same operator contract as the Winograd kernels: replaces the implicit TF Conv2D /
// Conv2DBackpropInput + LeakyRelu(Grad) + MaxPool(Grad) of reference nets/mj_uwyhNets_ba.py:431-462.
//
// Implicit GEMM: M = pixels, N = output channels, K = 9 taps x input channels.  A 512-thread workgroup (8 waves, two per
// SIMD; 256 persistent workgroups stride over the items of all jobs) owns a 16x16-pixel region and ALL N channels (32, 64 or
// 128); wave w owns pixel rows 2w, 2w+1 of the region = one 32-row MFMA block in POOL ORDER (row 4*window + position), so a
// 2x2 pooling window is 4 consecutive accumulator registers of one lane and LeakyReLU + MaxPool + argmax are lane-local.
//   * K runs in 32-channel chunks.  The chunk's 18x18 halo tile sits in LDS as one 128-byte record per pixel -- the H plane
//     then the L plane of the 32 channels, i.e. exactly the bytes HBM holds -- fetched by LDS-DMA straight from the tensor;
//     9 slots of 16 B per pixel (8 + 1 pad) and 168 slots per row (18 x 9 + 6 pad): every A fragment is ONE ds_read_b128 at
//     lane base + immediate (tap, k-step, plane), and the 16-lane groups of ds_read_b128 hit 16 distinct slots for every tap.
//   * Filters are packed once per step (mm_pack) in the order the kernel streams them, as f16 halves with a block exponent; a
//     stage (1 tap x 128, or 3 taps x 32 / 64 output channels: 12-24 KB) is double-buffered in LDS by LDS-DMA, one barrier
//     per stage.  A B fragment is one ds_read_b128 at lane * 16 + immediate.
//   * Per (tap, 16-channel k-step, 32-channel block): acc += AH*BH + AH*BL + AL*BH -- three MFMAs, no VALU work at all.
//   * Data gradient of a MaxPool'ed layer: the pooled gradient + argmax bytes are staged by LDS-DMA (10x10 pooled pixels) and
//     scattered LDS -> LDS into the halo tile of the next chunk while the current one is multiplied.
//   * Epilogue: block exponent, LeakyReLU (+ 2x2 MaxPool + argmax, first maximum wins) or LeakyReLU'(act), split into halves,
//     4-byte stores of adjacent channel pairs; the stored maximum goes to the output's H2Meta with one atomicMax per wave.
// MaxPool ties: identical input patches give bit-identical sums here (same operations in the same order), so EVERY exact tie
// -- axis-aligned flats and diagonal edges alike -- routes to the first maximum like TF's MaxPoolGrad.
#include <stdlib.h>
#include "mm_common.h"
#include "../../include/ugaitnet_hip_h2.h"

using namespace ugn_mm;

namespace {

enum { EPI_LRELU = 0, EPI_LRELU_POOL = 1, EPI_DGRAD = 2, EPI_DGRAD_ACT = 3 };

// timing-only ablations (WRONG results; tools/bench_mm.py --variants): 1 no filter DMA after the prologue, 2 no input-tile DMA
// after the prologue, 4 no epilogue stores, 8 no MFMA, 16 no per-stage wait for the filter DMA
#ifndef UGN_MM_ABLATE
#define UGN_MM_ABLATE 0
#endif
#if UGN_MM_ABLATE & 4
#define UGN_ST(T_, ptr_, val_) do { unsigned v__ = (unsigned)(val_); asm volatile("" :: "v"(v__)); (void)(ptr_); } while (0)
#else
#define UGN_ST(T_, ptr_, val_) *reinterpret_cast<T_*>(ptr_) = (T_)(val_)
#endif

#ifndef UGN_MM16_PIPE
#define UGN_MM16_PIPE 0      /* 1: fragment reads pinned a micro-step ahead of their MFMAs (round-4 experiment 3: equal in isolation,
                                1 % SLOWER in the step -- 5.77 against 5.83 ms, same box, both orders) */
#endif
// First item of a persistent workgroup.  Workgroup w runs on XCD w % 8 (round-robin dispatch), so with item = w the 4 / 16 regions of
// an image -- whose 18 x 18 halos overlap by 27 % -- are read on different XCDs, each through its own L2 (FETCH_SIZE of the 32 -> 32
// forward: 1131 MB for 944 MB of input).  Here an XCD's workgroups take a CONTIGUOUS range of a round's items: the regions of an image
// are in flight on one XCD at the same time.  (Later rounds: + gridDim.x, as before.)
#ifndef UGN_MM_XCD
#define UGN_MM_XCD 1
#endif
__device__ __forceinline__ int xcd_first_item() {
  const int w = blockIdx.x, g = gridDim.x;
  return (UGN_MM_XCD && (g & 7) == 0) ? (w & 7) * (g >> 3) + (w >> 3) : w;
}
constexpr int HROW = 168;                       // 16-byte slots per halo row (18 pixels x 9 + 6 pad; = 8 mod 16)
constexpr int HPIECES = 48;                     // 18 rows x 168 slots = 3024 -> 48 pieces of 64 slots (6 per wave)
constexpr int HALO_BYTES = HPIECES * 1024;      // 49,152
constexpr int W_OFF = 2 * HALO_BYTES;           // filter stages start here
constexpr int STG_PIECES = 16;                  // pooled staging: 100 pooled pixels x 10 slots (H 4, L 4, argmax 2)
constexpr int STG_BYTES = STG_PIECES * 1024;

template <int NC>
struct Geo {
  static constexpr int NB = NC / 32;                     // 32-column MFMA blocks
  static constexpr int TPS = NB == 4 ? 1 : 3;            // taps per filter stage
  static constexpr int NSTG = 9 / TPS;                   // stages per 32-channel chunk
  static constexpr int WSTAGE = TPS * NB * 4096;         // bytes: [tap][k-step 2][block][plane 2][1 KB]
  static constexpr int WPIECES = WSTAGE / 1024;
};
// 32 -> 32 layers (one K chunk, one block): the whole filter of a job -- 3 stages, 36 KB -- stays in LDS
template <int KC, int NC>
constexpr bool filter_resident() { return KC == 32 && NC == 32; }
// REGSTG (experiment, -DUGN_MM_REGSTG=1; 32 -> 32 pooled data gradient): the pooled gradient + argmax bytes of the next tile travel
// through REGISTERS (plain global loads one item ahead), not through a DMA'd staging tile; waves 0-3 scatter them into the next halo
// buffer behind tap 3, waves 4-7 behind tap 6 -- while the other wave of their SIMD multiplies -- and the item keeps ONE barrier.
// Bit-identical results, one barrier and 16 DMA pieces fewer per item -- and not faster: the item period in MICROSECONDS did not move
// (4.81 -> 4.91 us; in cycles 9,790 -> 10,770 at a clock that rose from 2.04 to 2.19 GHz), see DESIGN.md "what bounds the 3x3 kernels".
#ifndef UGN_MM_REGSTG
#define UGN_MM_REGSTG 0       /* measured: 1-3 % slower than the DMA'd staging tile + scatter pass (profiles/r04_kernel_experiments.txt) */
#endif
#ifndef UGN_MM_REGSTG_TAP0
#define UGN_MM_REGSTG_TAP0 3
#define UGN_MM_REGSTG_TAP1 6
#endif
template <int KC, int NC, int IN_POOLED>
constexpr bool reg_staged() { return filter_resident<KC, NC>() && IN_POOLED && UGN_MM_REGSTG; }
template <int KC, int NC, int IN_POOLED>
constexpr int lds_bytes() {
  return W_OFF + (filter_resident<KC, NC>() ? 3 : 2) * Geo<NC>::WSTAGE + (IN_POOLED && !reg_staged<KC, NC, IN_POOLED>() ? STG_BYTES : 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// filter statistics + packing
// ---------------------------------------------------------------------------------------------------------------------
// Which (layer, direction) pairs run on the 16x16x32 kernel (conv_mm16_kernel): 64 or 128 output columns and an un-pooled input.
// The only pooled-input data gradient with 64 columns is the 64 -> 64 one (a4 / b2): decided by shape, so that the filter
// packing (which does not know about pooling) and the launch agree.
#ifndef UGN_T16_MIN
#define UGN_T16_MIN 32
#endif
#ifndef UGN_D2P16
#define UGN_D2P16 1       /* the pooled 32 -> 32 data gradient on conv_d2_kernel<..., IN_POOLED> (16x16x32 tiles); its first form, conv32_d2p_kernel
                             on 32x32x16 (400 against 362 us in the step), was removed in round 5: 0 now selects conv_mm_kernel.  (The fused dgrad32_w5 kernel reads that filter in the 32-column block
                             layout: its caller packs a copy with flag bit 1 of ugn_mm_pack_multi.) */
#endif
#ifndef UGN_NR_POOLED
#define UGN_NR_POOLED 1   /* the pooled 64 -> 64 data gradient on conv_nr_kernel<..., IN_POOLED> (else conv_mm_kernel, 32x32x16) */
#endif
__host__ __device__ constexpr bool mm_tile16(int kc, int nc, int dgrad) {
  return nc >= UGN_T16_MIN && !(dgrad && kc == 64 && nc == 64 && !UGN_NR_POOLED) && !(dgrad && kc == 32 && nc == 32 && !UGN_D2P16);
}
// ... and which of those run on conv_nr_kernel (64 / 128 columns; filter tiles of 16 consecutive channels)
#ifndef UGN_MM_NR
#define UGN_MM_NR 1       /* 128-column launches on conv_nr_kernel: 5-10 % shorter than conv_mm16_kernel, 64-column ones equal or 3 % longer */
#endif
#ifndef UGN_NR_ACTPF
#define UGN_NR_ACTPF 4    /* 16-row waves: rows of LeakyReLU' operands fetched behind the last MFMAs (the rest at the top of the epilogue) */
#endif
#ifndef UGN_NR_APF
#define UGN_NR_APF 1      /* conv_nr_kernel: the A fragments of row j + this many are read before the MFMAs of row j (0: hipcc's order) */
#endif
#ifndef UGN_NR_CNT
#define UGN_NR_CNT 1      /* conv_nr_kernel: counted vmcnt at the top of an item (leaves the previous item's stores in flight) */
#endif
#ifndef UGN_NR_MINNC
#define UGN_NR_MINNC 128
#endif
__host__ __device__ constexpr bool mm_nr(int kc, int nc, int dgrad) {
  return UGN_MM_NR && mm_tile16(kc, nc, dgrad) && (nc >= UGN_NR_MINNC || (UGN_NR_POOLED && dgrad && kc == 64 && nc == 64));
}

constexpr int kPackJobs = 64;
struct PackTable {
  const float* w[kPackJobs];
  uint16_t* pk[kPackJobs];
  WMeta* meta[kPackJobs];
  unsigned* amax[kPackJobs];     // bits of max|w| (the third word of the filter's ugn_wmeta record)
  int cin[kPackJobs], cout[kPackJobs], dgrad[kPackJobs];
};

// Filter statistics, 16 blocks per job: block exponent from max|w| and the L1 bound of the direction
//   forward: an output channel sums over (tap, cin); data gradient: an input channel sums over (tap, cout).
// Block s of a job owns a 1/16 slice of the outputs; its 256 threads = (output of the slice) x (part of the sum), the parts
// are added through LDS in a fixed order.  Maxima leave by atomicMax on the bits of non-negative floats (order-independent).
constexpr int kStatSlices = 16;
__global__ __launch_bounds__(256) void mm_wstats_zero_kernel(PackTable t, int njobs) {
  if ((int)threadIdx.x < njobs) { t.meta[threadIdx.x]->l1 = 0.f; t.amax[threadIdx.x][0] = 0u; }
}
__global__ __launch_bounds__(256) void mm_wstats_kernel(PackTable t) {
  const int j = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x;
  const float* w = t.w[j];
  const int cin = t.cin[j], cout = t.cout[j], dgrad = t.dgrad[j] & 1;
  __shared__ float sp[256];
  float amax = 0.f;
  const int total = 9 * cin * cout, per = (total + kStatSlices - 1) / kStatSlices;
  for (int e = sl * per + tid; e < min(total, (sl + 1) * per); e += 256) amax = fmaxf(amax, fabsf(w[e]));
  const int nout = dgrad ? cin : cout, os = nout / kStatSlices;     // outputs of this block (2 ... 8)
  const int parts = 256 / os, o = sl * os + tid / parts, part = tid % parts;
  float s = 0.f;
  if (!dgrad) {       // terms r = (tap, ci): rows of the [9*cin][cout] matrix
    for (int r = part; r < 9 * cin; r += parts) s += fabsf(w[(size_t)r * cout + o]);
  } else {            // terms (tap, co) of row (tap, ci = o)
    for (int q = part; q < 9 * cout; q += parts) s += fabsf(w[((size_t)(q / cout) * cin + o) * cout + (q % cout)]);
  }
  sp[tid] = s;
  __syncthreads();
  float l1 = 0.f;
  if (part == 0) {
    for (int k = 0; k < parts; ++k) l1 += sp[tid + k];
  }
  __syncthreads();
  sp[tid] = part == 0 ? l1 : 0.f;
  __shared__ float sa[256];
  sa[tid] = amax;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if (tid < k) { sa[tid] = fmaxf(sa[tid], sa[tid + k]); sp[tid] = fmaxf(sp[tid], sp[tid + k]); }
    __syncthreads();
  }
  if (tid == 0) {
    atomicMax(t.amax[j], __float_as_uint(sa[0]));
    atomicMax(reinterpret_cast<unsigned*>(&t.meta[j]->l1), __float_as_uint(sp[0] * 1.0001f));   // (the sums round; keep the bound a bound)
  }
}

// element e of job j -> its two halves at [chunk][tap][k-step][block][plane][h][col][8]
//   forward      : g[tap][k = cin][n = cout] = w[tap][cin][cout]
//   data gradient: g[tap][k = cout][n = cin] = w[8 - tap][cin][cout]
__global__ void mm_pack_kernel(PackTable t) {
  const int j = blockIdx.y;
  const int cin = t.cin[j], cout = t.cout[j], dgrad = t.dgrad[j] & 1, blocks32 = t.dgrad[j] >> 1;
  const int kc = dgrad ? cout : cin, nc = dgrad ? cin : cout;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 9 * kc * nc) return;
  const int tap = e / (kc * nc), rem = e - tap * (kc * nc);
  // consecutive threads walk cout, the contiguous axis of the HWIO filter
  const int k = dgrad ? rem % kc : rem / nc, n = dgrad ? rem / kc : rem % nc;
  const float v = dgrad ? t.w[j][((size_t)(8 - tap) * cin + n) * cout + k] : t.w[j][((size_t)tap * cin + k) * cout + n];
  const int ew = h2_exp_for_bound(__uint_as_float(t.amax[j][0])) - 1;      // stored |w| < 2^14
  if (e == 0) t.meta[j]->e = ew;
  _Float16 hi, lo;
  h2_split(ldexpf(v, ew), hi, lo);
  uint16_t* pk = t.pk[j];
  if (mm_tile16(kc, nc, dgrad) && !blocks32) {
    // 16x16x32 kernels: [chunk][tap][16-column tile][plane][lane-linear 1 KB]; lane = k group (8 channels) * 16 + column; tiles 2m,
    // 2m + 1 hold the even / odd channels of the 32-channel group m (a lane then owns adjacent channels, as in the 32-column form)
    const int chunk = k >> 5, kg = (k >> 3) & 3, ee = k & 7;
    // (conv_nr_kernel: a wave owns tile ct = 16 CONSECUTIVE channels and pairs them up across lanes in its epilogue)
    const bool nr = mm_nr(kc, nc, dgrad);
    const int ct = nr ? n >> 4 : 2 * (n >> 5) + (n & 1), col = nr ? n & 15 : (n & 31) >> 1;
    const size_t base = (((size_t)chunk * 9 + tap) * (nc / 16) + ct) * 2;
    pk[(base + 0) * 512 + (kg * 16 + col) * 8 + ee] = (uint16_t)h2_bits(hi);
    pk[(base + 1) * 512 + (kg * 16 + col) * 8 + ee] = (uint16_t)h2_bits(lo);
    return;
  }
  const int nb_all = nc / 32;
  const int chunk = k >> 5, s = (k >> 4) & 1, h = (k >> 3) & 1, ee = k & 7;
  const int nb = mm_block_of(n, nc), col = mm_col_of(n, nc);
  const size_t base = ((((size_t)chunk * 9 + tap) * 2 + s) * nb_all + nb) * 2;
  pk[(base + 0) * 512 + (h * 32 + col) * 8 + ee] = (uint16_t)h2_bits(hi);
  pk[(base + 1) * 512 + (h * 32 + col) * 8 + ee] = (uint16_t)h2_bits(lo);
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA pieces
// ---------------------------------------------------------------------------------------------------------------------
// Piece `piece` (0..47) of the halo tile of (image base, region origin, chunk): slots 64 * piece .. + 63 of the 18 x 168
// slot image.  Slot (row, 9 * px + c): c < 4 -> 16 bytes of the pixel's H plane, c < 8 -> of its L plane, c = 8 / the 6 slots
// at the end of a row / rows >= 18: padding (never read; fetches the zero block).
template <int KC, int HW>
__device__ __forceinline__ void dma_halo_piece(const char* __restrict__ img_base, const void* __restrict__ zeros, int ry0,
                                               int rx0, int chunk, int piece, int lane, unsigned lds_byte_base) {
  const int g = piece * 64 + lane;
  const int row = (g * 6242) >> 20;                 // g / 168 for g < 3072
  const int rem = g - row * HROW;
  const int px = (rem * 57) >> 9;                   // rem / 9 for rem < 168
  const int c = rem - px * 9;
  const int gy = ry0 - 1 + row, gx = rx0 - 1 + px;
  const bool ok = rem < 162 && c < 8 && row < 18 && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
  const unsigned off = (unsigned)(gy * HW + gx) * (unsigned)(KC * 4) + (unsigned)(chunk * 64) +
                       (c < 4 ? (unsigned)(c * 16) : (unsigned)(KC * 2 + (c - 4) * 16));
  const void* src = ok ? (const void*)(img_base + off) : zeros;
  dma16(src, lds_byte_base + (unsigned)piece * 1024u);
}

// Piece `piece` (0..15) of the pooled staging tile: 10 x 10 pooled pixels, 10 slots each: 4 of the H plane, 4 of the L plane,
// 2 of argmax bytes (32 channels of the chunk).
template <int KC, int HW>
__device__ __forceinline__ void dma_pooled_piece(const char* __restrict__ dz_img, const char* __restrict__ idx_img,
                                                 const void* __restrict__ zeros, int ry0, int rx0, int chunk, int piece,
                                                 int lane, unsigned lds_byte_base) {
  constexpr int HP = HW / 2;
  const int g = piece * 64 + lane;
  const int pp = (g * 205) >> 11;                   // g / 10 for g < 1024
  const int part = g - pp * 10;
  const int prow = (pp * 205) >> 11, pcol = pp - prow * 10;
  const int pr = ry0 / 2 - 1 + prow, pc = rx0 / 2 - 1 + pcol;
  const bool ok = pp < 100 && (unsigned)pr < (unsigned)HP && (unsigned)pc < (unsigned)HP;
  const unsigned o = (unsigned)(pr * HP + pc);
  const char* vsrc = dz_img + o * (unsigned)(KC * 4) + (unsigned)(chunk * 64) +
                     (part < 4 ? (unsigned)(part * 16) : (unsigned)(KC * 2 + (part - 4) * 16));
  const char* isrc = idx_img + o * (unsigned)KC + (unsigned)(chunk * 32) + (unsigned)((part - 8) * 16);
  const void* src = !ok ? zeros : (part < 8 ? (const void*)vsrc : (const void*)isrc);
  dma16(src, lds_byte_base + (unsigned)piece * 1024u);
}

// MaxPool backward while staging: pooled pixel pp (10 x 10), channel group cg (8 channels) -> its four positions of the
// 18 x 18 halo tile receive the value where the argmax byte names the position, zero elsewhere.
__device__ __forceinline__ void scatter_pooled(const char* stg, char* halo, int u) {
  if (u >= 400) return;
  const int pp = u >> 2, cg = u & 3;
  const int prow = (pp * 205) >> 11, pcol = pp - prow * 10;
  const uint4 hi = *reinterpret_cast<const uint4*>(stg + pp * 160 + cg * 16);
  const uint4 lo = *reinterpret_cast<const uint4*>(stg + pp * 160 + 64 + cg * 16);
  const uint2 ix = *reinterpret_cast<const uint2*>(stg + pp * 160 + 128 + cg * 8);
  const unsigned hv[4] = {hi.x, hi.y, hi.z, hi.w}, lv[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
  for (int pos = 0; pos < 4; ++pos) {
    const int hy = 2 * prow - 1 + (pos >> 1), hx = 2 * pcol - 1 + (pos & 1);
    if ((unsigned)hy >= 18u || (unsigned)hx >= 18u) continue;
    unsigned m[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {            // dword d = channels 2d, 2d+1 of the group
      const unsigned w = d < 2 ? ix.x : ix.y;
      const unsigned b0 = (w >> (16 * (d & 1))) & 0xffu, b1 = (w >> (16 * (d & 1) + 8)) & 0xffu;
      m[d] = (b0 == (unsigned)pos ? 0x0000ffffu : 0u) | (b1 == (unsigned)pos ? 0xffff0000u : 0u);
    }
    char* dst = halo + (hy * HROW + hx * 9) * 16 + cg * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hv[0] & m[0], hv[1] & m[1], hv[2] & m[2], hv[3] & m[3]);
    *reinterpret_cast<uint4*>(dst + 64) = make_uint4(lv[0] & m[0], lv[1] & m[1], lv[2] & m[2], lv[3] & m[3]);
  }
}

// The same scatter with the pooled values and argmax bytes of the lane's unit already in registers (REGSTG: the 32 -> 32 pooled
// data gradient fetches them with plain global loads one item ahead -- no staging tile, no DMA, no barrier of its own).
__device__ __forceinline__ void scatter_regs(const uint4& hi, const uint4& lo, const uint2& ix, char* halo, int prow, int pcol, int cg) {
  const unsigned hv[4] = {hi.x, hi.y, hi.z, hi.w}, lv[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
  for (int pos = 0; pos < 4; ++pos) {
    const int hy = 2 * prow - 1 + (pos >> 1), hx = 2 * pcol - 1 + (pos & 1);
    if ((unsigned)hy >= 18u || (unsigned)hx >= 18u) continue;
    unsigned m[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {            // dword d = channels 2d, 2d+1 of the group
      const unsigned w = d < 2 ? ix.x : ix.y;
      const unsigned b0 = (w >> (16 * (d & 1))) & 0xffu, b1 = (w >> (16 * (d & 1) + 8)) & 0xffu;
      m[d] = (b0 == (unsigned)pos ? 0x0000ffffu : 0u) | (b1 == (unsigned)pos ? 0xffff0000u : 0u);
    }
    char* dst = halo + (hy * HROW + hx * 9) * 16 + cg * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hv[0] & m[0], hv[1] & m[1], hv[2] & m[2], hv[3] & m[3]);
    *reinterpret_cast<uint4*>(dst + 64) = make_uint4(lv[0] & m[0], lv[1] & m[1], lv[2] & m[2], lv[3] & m[3]);
  }
}

__device__ __forceinline__ f32x16 mfma_h(const uint4& a, const uint4& b, f32x16 c) {
#if UGN_MM_ABLATE & 8
  c[0] += __uint_as_float(a.x ^ b.y);     // (keeps the operands alive)
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------------
// KC: GEMM K channels (input channels of the convolution being evaluated), NC: its output channels, HW: image size of the
// OUTPUT of this kernel (= of the input, un-pooled).  IN_POOLED: `in` is a pooled gradient + argmax (MaxPool backward).
// What a lane needs per input-tile piece, computed once per kernel: the byte offset of its 16-byte slot relative to the region's
// first pixel and the slot's (row, column) in the 18 x 18 tile for the image-border test, packed off << 12 | row << 5 | column
// (pad slots: row 127, never inside an image).  Per item a piece then costs a dozen instructions instead of forty.
template <int KC, int HW>
__device__ __forceinline__ int halo_lane_of(int piece, int lane) {
  const int g = piece * 64 + lane;
  const int row = (g * 6242) >> 20;                 // g / 168 for g < 3072
  const int rem = g - row * HROW;
  const int px = (rem * 57) >> 9;                   // rem / 9 for rem < 168
  const int c = rem - px * 9;
  const bool valid = rem < 162 && c < 8 && row < 18;
  const int off = ((row - 1) * HW + (px - 1)) * (KC * 4) + (c < 4 ? c * 16 : KC * 2 + (c - 4) * 16);   // |off| < 2^19
  return valid ? (int)(((unsigned)off << 12) | (unsigned)(row << 5) | (unsigned)px) : (127 << 5);
}
template <int HW>
__device__ __forceinline__ void dma_halo_lane(const char* region_base, const void* zeros, int ry0, int rx0, int packed,
                                              unsigned lds_dst) {
  const int off = packed >> 12;
  const int gy = ry0 - 1 + ((packed >> 5) & 127), gx = rx0 - 1 + (packed & 31);
  const bool ok = (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
  const void* src = ok ? (const void*)(region_base + (ptrdiff_t)off) : zeros;
  dma16(src, lds_dst);
}

// diagnostic build only (-DUGN_MM_STAMP, tools/stamp_mm.py): clock stamps of every wave at the phases of a stage / an item
#ifdef UGN_MM_STAMP
__device__ unsigned long long* g_mm_stamp = nullptr;
constexpr int kStampStages = 60, kStampItems = 20;
constexpr int kStampPerWave = 4 + 5 * kStampStages + 4 * kStampItems;
#define STAMP_ST(k_) do { if (stamp && lane == 0 && nst < kStampStages) stamp[4 + 5 * nst + (k_)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_IT(k_) do { if (stamp && lane == 0 && nit < kStampItems) stamp[4 + 5 * kStampStages + 4 * nit + (k_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP_ST(k_) do { } while (0)
#define STAMP_IT(k_) do { } while (0)
#endif

template <int KC, int NC, int HW, int IN_POOLED, int EPI>
__global__ __launch_bounds__(512, 2) void conv_mm_kernel(const MmJobs jt, const void* __restrict__ zeros) {
  using G = Geo<NC>;
  constexpr int NB = G::NB, TPS = G::TPS, NSTG = G::NSTG, WSTAGE = G::WSTAGE, WPIECES = G::WPIECES;
  constexpr int NCHUNK = KC / 32;
  constexpr bool RES = filter_resident<KC, NC>();
  constexpr int STG_OFF = W_OFF + (RES ? 3 : 2) * WSTAGE;
  constexpr int RPX = HW / 16, RPI = RPX * RPX;
  static_assert(!IN_POOLED || NSTG >= 2, "the pooled scatter runs in the second stage of a chunk");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // A-side role of the lane: row r of the wave's 32-pixel block = window win (0..7), position q (0..3), k-half h
  const int r = lane & 31, h = lane >> 5, win = r >> 2, q = r & 3;
  const int a_lane = ((2 * wave + (q >> 1)) * HROW + (2 * win + (q & 1)) * 9) * 16 + h * 16;
  const int b_lane = W_OFF + lane * 16;

  int item = xcd_first_item();
  const int nitems = jt.start[kMaxJobs];
  if (item >= nitems) return;
#ifdef UGN_MM_STAMP
  unsigned long long* stamp = g_mm_stamp ? g_mm_stamp + ((size_t)blockIdx.x * 8 + wave) * kStampPerWave : nullptr;
  if (stamp && lane == 0) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[1] = __builtin_amdgcn_s_memrealtime(); }
  int nst = 0, nit = 0;
#endif
  int jb = mm_job_of(jt, item), lit = item - jt.start[jb];

  auto img_in = [&](const MmJob& J, int img) {      // byte base of image `img` of the input tensor (+ its argmax map)
    constexpr size_t IMG = IN_POOLED ? (size_t)(HW / 2) * (HW / 2) * KC : (size_t)HW * HW * KC;
    return reinterpret_cast<const char*>(J.in) + (size_t)img * IMG * 4;
  };
  auto img_idx = [&](const MmJob& J, int img) {
    return reinterpret_cast<const char*>(J.in_idx) + (size_t)img * (HW / 2) * (HW / 2) * KC;
  };
  // DMA roles.  Waves 4..7 fetch input tiles, waves 0..3 filter stages: vmcnt retires in order PER WAVE, so with separate
  // queues an input tile can stay in flight for a whole chunk (HBM latency) while every filter stage (L2) is awaited.
  const bool is_hw = wave >= 4;
  const int rw = wave & 3;
  constexpr int HSTG = NSTG == 3 ? 2 : 6;           // stages of a chunk that issue halo pieces (12 per wave)
  constexpr int HPER = 12 / HSTG;
  // halo pieces [first, first + count) of this wave for (item, chunk); pooled input: the staging tile instead (4 pieces, sg 0)
  auto stage_in = [&](const MmJob& J, int lit_, int chunk, unsigned halo_dst, int first, int count) {
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    if constexpr (IN_POOLED) {
      const char* vb = img_in(J, img);
      const char* ib = img_idx(J, img);
#pragma unroll
      for (int j = 0; j < STG_PIECES / 4; ++j)
        dma_pooled_piece<KC, HW>(vb, ib, zeros, ry0, rx0, chunk, rw * (STG_PIECES / 4) + j, lane, sbase + STG_OFF);
    } else {
      const char* vb = img_in(J, img);
#pragma unroll
      for (int j = 0; j < HPER; ++j)
        if (j < count) dma_halo_piece<KC, HW>(vb, zeros, ry0, rx0, chunk, rw * 12 + first + j, lane, halo_dst);
    }
  };
  auto stage_w = [&](const uint16_t* wpk, int st, unsigned dst) {     // filter stage `st` (chunk * NSTG + sg) of a job
    const char* src = reinterpret_cast<const char*>(wpk) + (size_t)st * WSTAGE;
#pragma unroll
    for (int j = 0; j < WPIECES / 4; ++j) {
      const int p = rw + 4 * j;
      dma16(src + p * 1024 + lane * 16, dst + (unsigned)p * 1024u);
    }
  };

  // RES: every wave fetches a sixth ... eighth of an input tile (6 pieces; pooled input: 2 staging pieces)
  int hpk[RES && !IN_POOLED ? 6 : 1];
  if constexpr (RES && !IN_POOLED) {
#pragma unroll
    for (int j = 0; j < 6; ++j) hpk[j] = halo_lane_of<KC, HW>(wave * 6 + j, lane);
  }
  auto res_piece = [&](const MmJob& J, int lit_, unsigned halo_dst, int j) {     // piece j (0..5; pooled 0..1) of this wave
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    if constexpr (IN_POOLED) {
      dma_pooled_piece<KC, HW>(img_in(J, img), img_idx(J, img), zeros, ry0, rx0, 0, wave * 2 + j, lane, sbase + STG_OFF);
    } else {
      dma_halo_lane<HW>(img_in(J, img) + (size_t)(ry0 * HW + rx0) * (KC * 4), zeros, ry0, rx0, hpk[RES && !IN_POOLED ? j : 0],
                        halo_dst + (unsigned)(wave * 6 + j) * 1024u);
    }
  };
  // REGSTG: this thread's unit of the 10 x 10 pooled pixels under a region's 18 x 18 halo -- pooled pixel spp, channel group scg
  // (8 channels): 16 B of H halves, 16 B of L halves, 8 argmax bytes, fetched into registers one item ahead
  constexpr bool REGSTG = reg_staged<KC, NC, IN_POOLED>();
  const int spp = tid >> 2, scg = tid & 3;
  const int sprow = (spp * 205) >> 11, spcol = spp - sprow * 10;
  uint4 shi = make_uint4(0u, 0u, 0u, 0u), slo = make_uint4(0u, 0u, 0u, 0u);
  uint2 six = make_uint2(0u, 0u);
  auto stg_load = [&](const MmJob& J, int lit_) {
    constexpr int HP = HW / 2;
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int pr = ((rrem / RPX) * 16) / 2 - 1 + sprow, pc = ((rrem % RPX) * 16) / 2 - 1 + spcol;
    const bool ok = tid < 400 && (unsigned)pr < (unsigned)HP && (unsigned)pc < (unsigned)HP;
    shi = make_uint4(0u, 0u, 0u, 0u);
    slo = shi;
    six = make_uint2(0u, 0u);           // (outside the image the gradient is zero: the scatter still overwrites the stale halo)
    if (ok) {
      const unsigned o = (unsigned)(pr * HP + pc);
      const char* v = img_in(J, img) + o * (unsigned)(KC * 4) + (unsigned)(scg * 16);
      shi = *reinterpret_cast<const uint4*>(v);
      slo = *reinterpret_cast<const uint4*>(v + KC * 2);
      six = *reinterpret_cast<const uint2*>(img_idx(J, img) + o * (unsigned)KC + (unsigned)(scg * 8));
    }
  };
  // ---- prologue: input tile of (item, chunk 0) -> halo buffer 0, filter stage 0 -> filter buffer 0 (RES: the whole filter)
  if constexpr (RES) {
    if constexpr (REGSTG) {
      stg_load(jt.job[jb], lit);
    } else {
#pragma unroll
      for (int j = 0; j < (IN_POOLED ? 2 : 6); ++j) res_piece(jt.job[jb], lit, sbase, j);
    }
    if (!is_hw) {
#pragma unroll
      for (int d = 0; d < 3; ++d) stage_w(jt.job[jb].wpk, d, sbase + W_OFF + (unsigned)d * WSTAGE);
    }
  } else if (is_hw) {
#pragma unroll
    for (int k = 0; k < HSTG; ++k) stage_in(jt.job[jb], lit, 0, sbase, k * HPER, HPER);
  } else {
    stage_w(jt.job[jb].wpk, 0, sbase + W_OFF);
  }
  int res_job = jb;                 // RES: the job whose filter is in LDS
  if constexpr (REGSTG) {
    if (tid < 400) scatter_regs(shi, slo, six, smem, sprow, spcol, scg);      // (visible behind the first item's barrier)
  } else if constexpr (IN_POOLED) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    scatter_pooled(smem + STG_OFF, smem, tid);
  }
  int hbuf = 0, wbuf = 0;
  bool first_item = true;
  // per-job epilogue constants (block exponents) and the running maximum of what this wave stored for the job
  int meta_jb = -1, e_out = 0;
  float factor = 1.f, mx = 0.f;

  for (; item < nitems; item += gridDim.x) {
    const int next_item = item + gridDim.x;
    const bool more = next_item < nitems;
    const int jn = more ? mm_job_of(jt, next_item) : jb, nlit = more ? next_item - jt.start[jn] : lit;
    if (jb != meta_jb) {          // (wave-uniform) first item of a job in this workgroup: its exponents, loaded under the MFMAs
      if (meta_jb >= 0) h2_publish_amax(jt.job[meta_jb].out_meta, wave_max(mx), lane);   // (a job change: at most 5 per workgroup)
      mx = 0.f;
      const MmJob& Jm = jt.job[jb];
      const int e_in = Jm.in_meta->e;
      const float amax_in = h2_true_amax(e_in, Jm.in_meta->amax);
      e_out = h2_exp_for_bound(amax_in * Jm.wmeta->l1);
      factor = ldexpf(1.f, e_out - e_in - Jm.wmeta->e);
      meta_jb = jb;
    }
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    const int img = lit / RPI, rrem = lit % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    // LeakyReLU' epilogue: the H halves of the layer's input at the lane's 16 pixels x channel pairs are fetched during the last
    // stage (as epilogue loads between its stores they were 32 dependent round trips to HBM: +0.2 ms on the 128 -> 128 layer)
    constexpr bool ACTPF = EPI == EPI_DGRAD_ACT && NB >= 2;
    unsigned actv[ACTPF ? NB / 2 : 1][16];

    if constexpr (RES) {
      // ---- one chunk, the filter resident: ONE barrier per item, no filter traffic, the next tile's pieces issued between
      // the taps (the three-stage form spent 55 % of an item waiting for 12-KB filter stages, at barriers and issuing pieces:
      // tools/stamp_mm.py, profiles/r03_stage_stamps.txt)
      bool w_fresh = false;
      if (jb != res_job) {            // (at most twice per workgroup) every wave has left the previous job's taps
        __syncthreads();
        if (!is_hw) {
#pragma unroll
          for (int d = 0; d < 3; ++d) stage_w(jt.job[jb].wpk, d, sbase + W_OFF + (unsigned)d * WSTAGE);
        }
        res_job = jb;
        w_fresh = true;
      }
      STAMP_ST(0);
      // The tile of this item was issued during the previous item's taps, BEFORE that item's epilogue stores: vmcnt retires in
      // issue order, so waiting until at most the stores are outstanding waits for the tile and for none of the stores (a
      // lower bound of their number is enough: more outstanding operations only make the wait stricter).  Pooled input: the
      // staging tile was awaited before its scatter.
      constexpr int EPI_STORES = EPI == EPI_LRELU_POOL ? 8 : 16;       // per wave, one block: 4 windows x (value + argmax) / 16 pixels
      if (first_item || w_fresh) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (!IN_POOLED) {
        static_assert(EPI_STORES == 8 || EPI_STORES == 16, "vmcnt immediate");
        if constexpr (EPI_STORES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      }
      STAMP_ST(1);
      __syncthreads();
      STAMP_ST(2);
      if constexpr (REGSTG) {
        if (more) stg_load(jt.job[jn], nlit);
      } else if constexpr (IN_POOLED) {
        if (more) {
#pragma unroll
          for (int j = 0; j < 2; ++j) res_piece(jt.job[jn], nlit, 0u, j);
        }
      }
      STAMP_ST(3);
      const int a_addr = a_lane + hbuf * HALO_BYTES;
      // (reads and MFMAs in hipcc's own order here: pinning the 8 fragment reads of tap + 1 before the 6 MFMAs of tap, as the staged
      //  loop below and conv_mm16_kernel do, measured 6 % SLOWER on this single-accumulator chain -- profiles/r04_kernel_experiments.txt)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int aoff = (dy * HROW + dx * 9) * 16 + s * 32;
          const uint4 ah = *reinterpret_cast<const uint4*>(smem + a_addr + aoff);
          const uint4 al = *reinterpret_cast<const uint4*>(smem + a_addr + aoff + 64);
          const uint4 bh = *reinterpret_cast<const uint4*>(smem + b_lane + ((tap * 2 + s) * 2 + 0) * 1024);
          const uint4 bl = *reinterpret_cast<const uint4*>(smem + b_lane + ((tap * 2 + s) * 2 + 1) * 1024);
          acc[0] = mfma_h(ah, bh, acc[0]);
          acc[0] = mfma_h(ah, bl, acc[0]);
          acc[0] = mfma_h(al, bh, acc[0]);
        }
        if constexpr (!IN_POOLED) {
          if (tap < 6 && more) {
            __builtin_amdgcn_sched_barrier(0);
            res_piece(jt.job[jn], nlit, sbase + (unsigned)(hbuf ^ 1) * HALO_BYTES, tap);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if constexpr (REGSTG) {
          // MaxPool backward of the next tile, registers -> the other halo buffer (no reader until the next item's barrier).  The two
          // waves of a SIMD (w, w + 4) take turns: one scatters (~100 vector instructions, 8 ds_write_b128) while the other multiplies.
          if (more && tap == (wave < 4 ? UGN_MM_REGSTG_TAP0 : UGN_MM_REGSTG_TAP1)) {
            __builtin_amdgcn_sched_barrier(0);
            STAMP_IT(2);
            if (tid < 400) scatter_regs(shi, slo, six, smem + (hbuf ^ 1) * HALO_BYTES, sprow, spcol, scg);
            STAMP_IT(3);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          if (tap == 4 && more) {     // MaxPool backward of the next tile: staging (issued above) -> the other halo buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            scatter_pooled(smem + STG_OFF, smem + (hbuf ^ 1) * HALO_BYTES, tid);
          }
        }
      }
      hbuf ^= 1;
      STAMP_ST(4);
#ifdef UGN_MM_STAMP
      ++nst;
#endif
    } else {
#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool last_chunk = chunk + 1 == NCHUNK;
      const bool next_tile = !last_chunk || more;             // there is a (chunk, item) after this one
      const bool to_next = last_chunk && more;
      const int n_chunk = last_chunk ? 0 : chunk + 1;
      const int nx_job = to_next ? jn : jb, n_lit = to_next ? nlit : lit;
      const int a_addr = a_lane + hbuf * HALO_BYTES;
#pragma unroll
      for (int sg = 0; sg < NSTG; ++sg) {
        // What this stage reads must have landed: the filter stage (issued a stage ago by waves 0..3), in the first stage of
        // a chunk the input tile (issued a chunk ago by waves 4..7), before the pooled scatter the staging tile.  The first
        // stage of every item but the first was awaited BEFORE the previous item's epilogue (see there).
        STAMP_ST(0);
        if (!(sg == 0 && chunk == 0 && !first_item)) {
          if ((!is_hw && !(UGN_MM_ABLATE & 16)) || sg == 0 || (IN_POOLED && sg == NSTG - 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        STAMP_ST(1);
        __syncthreads();                                     // ... and is visible; the other buffers have no readers left
        STAMP_ST(2);
        if constexpr (ACTPF) {
          if (sg == NSTG - 1 && last_chunk) {
            const char* act = reinterpret_cast<const char*>(jt.job[jb].act) + (size_t)img * HW * HW * NC * 4;
#pragma unroll
            for (int m = 0; m < NB / 2; ++m)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int g = r >> 2, i = r & 3;
                const unsigned pix = (unsigned)((ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * (2 * g + h) + (i & 1));
                actv[m][r] = *reinterpret_cast<const unsigned*>(act + pix * (unsigned)(NC * 4) + (unsigned)(64 * m + 2 * (lane & 31)) * 2u);
              }
          }
        }
        if (!is_hw && !(UGN_MM_ABLATE & 1)) {   // filter stage after this one -> the other filter buffer
          if (sg + 1 < NSTG) {
            stage_w(jt.job[jb].wpk, chunk * NSTG + sg + 1, sbase + W_OFF + (unsigned)(wbuf ^ 1) * WSTAGE);
          } else if (next_tile) {
            stage_w(jt.job[nx_job].wpk, n_chunk * NSTG, sbase + W_OFF + (unsigned)(wbuf ^ 1) * WSTAGE);
          }
        } else if (is_hw && next_tile && !(UGN_MM_ABLATE & 2)) {   // input tile of the next chunk / item -> the other halo buffer (pooled: the staging tile)
          if (IN_POOLED ? sg == 0 : sg < HSTG)
            stage_in(jt.job[nx_job], n_lit, n_chunk, sbase + (unsigned)(hbuf ^ 1) * HALO_BYTES, sg * HPER, HPER);
        }
        if constexpr (IN_POOLED) {   // MaxPool backward of the next tile, staging -> the other halo buffer
          if (sg == NSTG - 1 && next_tile) scatter_pooled(smem + STG_OFF, smem + (hbuf ^ 1) * HALO_BYTES, tid);
        }
        STAMP_ST(3);
        const int b_addr = b_lane + wbuf * WSTAGE;
#if UGN_MM16_PIPE
        // software pipeline over micro-steps u = (tap, k-step): the 2 + 2 NB fragment reads of u + 1 before the 3 NB MFMAs of u
        constexpr int NU = TPS * 2;
        uint4 fa[2][2], fb[2][NB][2];      // [register set][plane] / [register set][block][plane]
        auto load_u = [&](int set, int u) {
          const int t = u >> 1, s = u & 1;
          const int tap = sg * TPS + t, dy = tap / 3, dx = tap % 3;
          const int aoff = (dy * HROW + dx * 9) * 16 + s * 32;
          fa[set][0] = *reinterpret_cast<const uint4*>(smem + a_addr + aoff);
          fa[set][1] = *reinterpret_cast<const uint4*>(smem + a_addr + aoff + 64);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            fb[set][nb][0] = *reinterpret_cast<const uint4*>(smem + b_addr + (((t * 2 + s) * NB + nb) * 2 + 0) * 1024);
            fb[set][nb][1] = *reinterpret_cast<const uint4*>(smem + b_addr + (((t * 2 + s) * NB + nb) * 2 + 1) * 1024);
          }
        };
        load_u(0, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (u + 1 < NU) load_u((u + 1) & 1, u + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_h(fa[u & 1][0], fb[u & 1][nb][0], acc[nb]);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_h(fa[u & 1][0], fb[u & 1][nb][1], acc[nb]);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_h(fa[u & 1][1], fb[u & 1][nb][0], acc[nb]);
          __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
          const int tap = sg * TPS + t, dy = tap / 3, dx = tap % 3;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int aoff = (dy * HROW + dx * 9) * 16 + s * 32;
            const uint4 ah = *reinterpret_cast<const uint4*>(smem + a_addr + aoff);
            const uint4 al = *reinterpret_cast<const uint4*>(smem + a_addr + aoff + 64);
            uint4 bh[NB], bl[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              bh[nb] = *reinterpret_cast<const uint4*>(smem + b_addr + (((t * 2 + s) * NB + nb) * 2 + 0) * 1024);
              bl[nb] = *reinterpret_cast<const uint4*>(smem + b_addr + (((t * 2 + s) * NB + nb) * 2 + 1) * 1024);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_h(ah, bh[nb], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_h(ah, bl[nb], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_h(al, bh[nb], acc[nb]);
          }
        }
#endif
        wbuf ^= 1;
        STAMP_ST(4);
#ifdef UGN_MM_STAMP
        ++nst;
#endif
      }
      hbuf ^= 1;
    }
    }
    // The next item's first filter stage and input tile are awaited HERE, before this item's stores enter the queue: the
    // first stage of the next item then needs no wait, and the stores have a whole stage to be acknowledged.
    if (more && !(RES && !IN_POOLED) && !REGSTG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (RES: awaited at the top of the next item)
    STAMP_IT(0);
    first_item = false;

    // ---- epilogue.  acc[nb][4g + i] of lane (col c = lane & 31, half h): window 2g + h, position i of the wave's 8 windows,
    // i.e. pixel (2 * wave + (i >> 1), 2 * (2g + h) + (i & 1)) of the region; channel: mm_block_of / mm_col_of.
    const MmJob& J = jt.job[jb];
    if (lit == 0 && tid == 0) J.out_meta->e = e_out;
    constexpr bool POOL = EPI == EPI_LRELU_POOL;
    constexpr int HO = POOL ? HW / 2 : HW;
    char* out = reinterpret_cast<char*>(J.out) + (size_t)img * HO * HO * NC * 4;
    const int c = lane & 31;
    if constexpr (NB >= 2) {
#pragma unroll
      for (int m = 0; m < NB / 2; ++m) {
        const unsigned chb = (unsigned)(64 * m + 2 * c) * 2u;       // byte offset of the lane's channel pair in a plane
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int wx = 2 * g + h;
          if constexpr (POOL) {
            float best[2];
            unsigned bi[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              best[e] = acc[2 * m + e][4 * g];
              bi[e] = 0;
#pragma unroll
              for (int i = 1; i < 4; ++i) {
                const float v = acc[2 * m + e][4 * g + i];
                if (v > best[e]) { best[e] = v; bi[e] = i; }     // strict >: the FIRST maximum wins (TF MaxPoolGrad)
              }
              best[e] = ugn_lrelu(best[e] * factor);
              mx = fmaxf(mx, fabsf(best[e]));
            }
            const unsigned pix = (unsigned)((ry0 / 2 + wave) * HO + rx0 / 2 + wx);
            _Float16 h0, l0, h1, l1;
            h2_split(best[0], h0, l0);
            h2_split(best[1], h1, l1);
            UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + chb, h2_pack(h0, h1));
            UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + (unsigned)(NC * 2) + chb, h2_pack(l0, l1));
            uint8_t* oi = J.out_idx + (size_t)img * HO * HO * NC;
            UGN_ST(uint16_t, oi + pix * (unsigned)NC + (unsigned)(64 * m + 2 * c), bi[0] | (bi[1] << 8));
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const unsigned pix = (unsigned)((ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * wx + (i & 1));
              float v0 = acc[2 * m][4 * g + i] * factor, v1 = acc[2 * m + 1][4 * g + i] * factor;
              if constexpr (EPI == EPI_LRELU) {
                v0 = ugn_lrelu(v0);
                v1 = ugn_lrelu(v1);
              } else if constexpr (EPI == EPI_DGRAD_ACT) {
                const unsigned ah2 = actv[m][4 * g + i];
                v0 *= (short)(ah2 & 0xffffu) > 0 ? 1.f : UGN_LRELU_ALPHA;      // LeakyReLU' from the sign of the H half
                v1 *= (short)(ah2 >> 16) > 0 ? 1.f : UGN_LRELU_ALPHA;
              }
              mx = fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1)));
              _Float16 h0, l0, h1, l1;
              h2_split(v0, h0, l0);
              h2_split(v1, h1, l1);
              UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + chb, h2_pack(h0, h1));
              UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + (unsigned)(NC * 2) + chb, h2_pack(l0, l1));
            }
          }
        }
      }
    } else {
      // one block: lane c owns channel c; neighbouring lanes (c, c^1) exchange their halves so that the even lane stores the H
      // pair and the odd lane the L pair -- 32 lanes x 4 bytes = the pixel's 128-byte record
      const unsigned sel = (c & 1) ? 0x03020706u : 0x05040100u;
      const unsigned chb = (c & 1) ? (unsigned)(NC * 2 + (c - 1) * 2) : (unsigned)(c * 2);
      auto store1 = [&](unsigned pix, float v) {
        _Float16 hi, lo;
        h2_split(v, hi, lo);
        const unsigned own = h2_pack(hi, lo);
        const unsigned oth = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + chb, __builtin_amdgcn_perm(oth, own, sel));
      };
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int wx = 2 * g + h;
        if constexpr (POOL) {
          float best = acc[0][4 * g];
          unsigned bi = 0;
#pragma unroll
          for (int i = 1; i < 4; ++i) {
            const float v = acc[0][4 * g + i];
            if (v > best) { best = v; bi = i; }
          }
          best = ugn_lrelu(best * factor);
          mx = fmaxf(mx, fabsf(best));
          const unsigned pix = (unsigned)((ry0 / 2 + wave) * HO + rx0 / 2 + wx);
          store1(pix, best);
          uint8_t* oi = J.out_idx + (size_t)img * HO * HO * NC;
          UGN_ST(uint8_t, oi + pix * (unsigned)NC + (unsigned)c, bi);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned pix = (unsigned)((ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * wx + (i & 1));
            float v = acc[0][4 * g + i] * factor;
            if constexpr (EPI == EPI_LRELU) {
              v = ugn_lrelu(v);
            } else if constexpr (EPI == EPI_DGRAD_ACT) {
              const char* act = reinterpret_cast<const char*>(J.act) + (size_t)img * HW * HW * NC * 4;
              const unsigned short ah = *reinterpret_cast<const unsigned short*>(act + pix * (unsigned)(NC * 4) + (unsigned)(c * 2));
              v *= (short)ah > 0 ? 1.f : UGN_LRELU_ALPHA;
            }
            mx = fmaxf(mx, fabsf(v));
            store1(pix, v);
          }
        }
      }
    }
    STAMP_IT(1);
#ifdef UGN_MM_STAMP
    ++nit;
#endif
    jb = jn;
    lit = nlit;
  }
#ifdef UGN_MM_STAMP
  if (stamp && lane == 0) { stamp[2] = __builtin_amdgcn_s_memtime(); stamp[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
  // one atomicMax per WORKGROUP for the job it ends with (not per item or wave: atomics on one address serialise at ~10 ns
  // each; 230 k of them cost a 64x64 layer a millisecond).  No DMA is in flight any more: any LDS serves as scratch.
  h2_publish_amax_block(jt.job[meta_jb].out_meta, mx, reinterpret_cast<float*>(smem), tid, 8);
}


// ---------------------------------------------------------------------------------------------------------------------
// the same convolution on v_mfma_f32_16x16x32_f16 (64 / 128 output channels, un-pooled input)
// ---------------------------------------------------------------------------------------------------------------------
// One k-step = the whole 32-channel chunk; the wave's 32 pixels are two row tiles of 16 (windows 0..3 | 4..7 in pool order: a
// lane's four accumulator registers of a tile are the four positions of ONE pooling window), the output channels 16-column tiles.
// Same FLOPs, LDS reads, bytes and accumulator registers as the 32x32x16 form; the chip holds a higher clock on this shape and a
// tap's products are 2 x NC/16 independent accumulator chains instead of NC/32: 13-23 % on the weight gradients that use it
// (wgrad3x3_mm.hip).  A fragment: lane (row = lane & 15, k group kg = lane >> 4) reads the 16 bytes of channels 8 kg .. 8 kg + 7 of its
// pixel: with 10 slots per pixel (8 + 2 pad) and 184 per row the four 16-lane groups of a ds_read_b128 hit 16 distinct slots.
constexpr int PS16 = 10, HROW16 = 184;
constexpr int HPIECES16 = 52;                       // 18 rows x 184 slots = 3312 -> 51.75 pieces (13 per fetching wave)
constexpr int HALO16_BYTES = HPIECES16 * 1024;      // 53,248
constexpr int W_OFF16 = 2 * HALO16_BYTES;
template <int NC>
constexpr int lds_bytes16() { return W_OFF16 + 2 * Geo<NC>::WSTAGE; }

template <int KC, int HW>
__device__ __forceinline__ void dma_halo_piece16(const char* __restrict__ img_base, const void* __restrict__ zeros, int ry0,
                                                 int rx0, int chunk, int piece, int lane, unsigned lds_byte_base) {
  const int g = piece * 64 + lane;
  const int row = (g * 5699) >> 20;                 // g / 184 for g < 3328
  const int rem = g - row * HROW16;
  const int px = (rem * 205) >> 11;                 // rem / 10 for rem < 184
  const int c = rem - px * PS16;
  const int gy = ry0 - 1 + row, gx = rx0 - 1 + px;
  const bool ok = rem < 18 * PS16 && c < 8 && row < 18 && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
  const unsigned off = (unsigned)(gy * HW + gx) * (unsigned)(KC * 4) + (unsigned)(chunk * 64) +
                       (c < 4 ? (unsigned)(c * 16) : (unsigned)(KC * 2 + (c - 4) * 16));
  const void* src = ok ? (const void*)(img_base + off) : zeros;
  dma16(src, lds_byte_base + (unsigned)piece * 1024u);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16_h(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

template <int KC, int NC, int HW, int EPI>
__global__ __launch_bounds__(512, 2) void conv_mm16_kernel(const MmJobs jt, const void* __restrict__ zeros) {
  using G = Geo<NC>;
  constexpr int TPS = G::TPS, NSTG = G::NSTG, WSTAGE = G::WSTAGE, WPIECES = G::WPIECES;
  constexpr int NT = NC / 16, NG = NC / 32;           // 16-column tiles, 32-channel groups (a lane owns a channel pair of each)
  constexpr int NCHUNK = KC / 32;
  constexpr int RPX = HW / 16, RPI = RPX * RPX;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // A-side role: row r16 of a 16-row tile = window (r16 >> 2) of the tile, position q; k group kg
  const int r16 = lane & 15, kg = lane >> 4, q = r16 & 3;
  int a_lane[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
    a_lane[rt] = ((2 * wave + (q >> 1)) * HROW16 + (2 * (4 * rt + (r16 >> 2)) + (q & 1)) * PS16) * 16 + kg * 16;
  const int b_lane = W_OFF16 + lane * 16;
  // C-side role: column col of a tile, window rg of the row tile (registers = positions 0..3)
  const int col = lane & 15, rg = lane >> 4;

  int item = xcd_first_item();
  const int nitems = jt.start[kMaxJobs];
  if (item >= nitems) return;
  int jb = mm_job_of(jt, item), lit = item - jt.start[jb];

  auto img_in = [&](const MmJob& J, int img) { return reinterpret_cast<const char*>(J.in) + (size_t)img * HW * HW * KC * 4; };
  const bool is_hw = wave >= 4;
  const int rw = wave & 3;
  constexpr int HSTG = NSTG == 3 ? 2 : 7;           // stages of a chunk that issue halo pieces (13 per wave)
  constexpr int HPER = NSTG == 3 ? 7 : 2;
  auto stage_in = [&](const MmJob& J, int lit_, int chunk, unsigned halo_dst, int sgi) {
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    const char* vb = img_in(J, img);
#pragma unroll
    for (int j = 0; j < HPER; ++j) {
      const int k = sgi * HPER + j;
      if (k < 13) dma_halo_piece16<KC, HW>(vb, zeros, ry0, rx0, chunk, rw * 13 + k, lane, halo_dst);
    }
  };
  auto stage_w = [&](const uint16_t* wpk, int st, unsigned dst) {
    const char* src = reinterpret_cast<const char*>(wpk) + (size_t)st * WSTAGE;
#pragma unroll
    for (int j = 0; j < WPIECES / 4; ++j) {
      const int p = rw + 4 * j;
      dma16(src + p * 1024 + lane * 16, dst + (unsigned)p * 1024u);
    }
  };

  if (is_hw) {
#pragma unroll
    for (int k = 0; k < HSTG; ++k) stage_in(jt.job[jb], lit, 0, sbase, k);
  } else {
    stage_w(jt.job[jb].wpk, 0, sbase + W_OFF16);
  }
  int hbuf = 0, wbuf = 0;
  bool first_item = true;
  int meta_jb = -1, e_out = 0;
  float factor = 1.f, mx = 0.f;

  for (; item < nitems; item += gridDim.x) {
    const int next_item = item + gridDim.x;
    const bool more = next_item < nitems;
    const int jn = more ? mm_job_of(jt, next_item) : jb, nlit = more ? next_item - jt.start[jn] : lit;
    if (jb != meta_jb) {
      if (meta_jb >= 0) h2_publish_amax(jt.job[meta_jb].out_meta, wave_max(mx), lane);
      mx = 0.f;
      const MmJob& Jm = jt.job[jb];
      const int e_in = Jm.in_meta->e;
      const float amax_in = h2_true_amax(e_in, Jm.in_meta->amax);
      e_out = h2_exp_for_bound(amax_in * Jm.wmeta->l1);
      factor = ldexpf(1.f, e_out - e_in - Jm.wmeta->e);
      meta_jb = jb;
    }
    f32x4 acc[2][NT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int img = lit / RPI, rrem = lit % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    constexpr bool ACTPF = EPI == EPI_DGRAD_ACT;
    unsigned actv[ACTPF ? NG : 1][8];       // H halves of the layer's input at the lane's 8 pixels x channel pair of group m

#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool last_chunk = chunk + 1 == NCHUNK;
      const bool next_tile = !last_chunk || more;
      const bool to_next = last_chunk && more;
      const int n_chunk = last_chunk ? 0 : chunk + 1;
      const int nx_job = to_next ? jn : jb, n_lit = to_next ? nlit : lit;
#pragma unroll
      for (int sg = 0; sg < NSTG; ++sg) {
        if (!(sg == 0 && chunk == 0 && !first_item)) {
          if (!is_hw || sg == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if constexpr (ACTPF) {
          if (sg == NSTG - 1 && last_chunk) {
            const char* act = reinterpret_cast<const char*>(jt.job[jb].act) + (size_t)img * HW * HW * NC * 4;
#pragma unroll
            for (int m = 0; m < NG; ++m)
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const int rt = r >> 2, i = r & 3;
                const unsigned pix = (unsigned)((ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * (4 * rt + rg) + (i & 1));
                actv[m][r] = *reinterpret_cast<const unsigned*>(act + pix * (unsigned)(NC * 4) + (unsigned)(32 * m + 2 * col) * 2u);
              }
          }
        }
        if (!is_hw) {
          if (sg + 1 < NSTG) {
            stage_w(jt.job[jb].wpk, chunk * NSTG + sg + 1, sbase + W_OFF16 + (unsigned)(wbuf ^ 1) * WSTAGE);
          } else if (next_tile) {
            stage_w(jt.job[nx_job].wpk, n_chunk * NSTG, sbase + W_OFF16 + (unsigned)(wbuf ^ 1) * WSTAGE);
          }
        } else if (next_tile && sg < HSTG) {
          stage_in(jt.job[nx_job], n_lit, n_chunk, sbase + (unsigned)(hbuf ^ 1) * HALO16_BYTES, sg);
        }
        const int b_addr = b_lane + wbuf * WSTAGE;
#if UGN_MM16_PIPE
        // Software pipeline over micro-steps u = (tap, half of the column tiles): the fragments of u + 1 are READ (ds_read_b128,
        // into the other register set) before the MFMAs of u are issued, and the two groups are pinned with sched_barrier.  Left to
        // itself hipcc issues each read one or two instructions ahead of the MFMA that needs it (`s_waitcnt lgkmcnt(0|1)` before
        // every second MFMA in the round-3 ISA): with 8 waves reading, an LDS read takes 100-300 cycles to return and the matrix
        // pipe idled behind it (stage period 3,050 cycles for 1,536 of MFMA; profiles/r03_stage_stamps.txt).  A micro-step is
        // 12-24 MFMAs = 190-380 cycles of matrix work per wave: one of them hides the reads of the next.
        constexpr int HALVES = NT > 4 ? 2 : 1, NTH = NT / HALVES, NU = TPS * HALVES;
        uint4 fa[2][2][2], fb[2][NTH][2];        // [register set][row tile | column tile][plane]
        auto load_a = [&](int set, int t) {
          const int tap = sg * TPS + t, dy = tap / 3, dx = tap % 3;
          const int aoff = (dy * HROW16 + dx * PS16) * 16 + hbuf * HALO16_BYTES;
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            fa[set][rt][0] = *reinterpret_cast<const uint4*>(smem + a_lane[rt] + aoff);
            fa[set][rt][1] = *reinterpret_cast<const uint4*>(smem + a_lane[rt] + aoff + 64);
          }
        };
        auto load_b = [&](int set, int t, int hf) {
#pragma unroll
          for (int c = 0; c < NTH; ++c) {
            fb[set][c][0] = *reinterpret_cast<const uint4*>(smem + b_addr + ((t * NT + hf * NTH + c) * 2 + 0) * 1024);
            fb[set][c][1] = *reinterpret_cast<const uint4*>(smem + b_addr + ((t * NT + hf * NTH + c) * 2 + 1) * 1024);
          }
        };
        load_a(0, 0);
        load_b(0, 0, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int t = u / HALVES, hf = u % HALVES;
          if (u + 1 < NU) {
            const int tn = (u + 1) / HALVES, hn = (u + 1) % HALVES;
            if (hn == 0) load_a(tn & 1, tn);
            load_b((u + 1) & 1, tn, hn);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < NTH; ++c)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][hf * NTH + c] = mfma16_h(fa[t & 1][rt][0], fb[u & 1][c][0], acc[rt][hf * NTH + c]);
#pragma unroll
          for (int c = 0; c < NTH; ++c)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][hf * NTH + c] = mfma16_h(fa[t & 1][rt][0], fb[u & 1][c][1], acc[rt][hf * NTH + c]);
#pragma unroll
          for (int c = 0; c < NTH; ++c)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][hf * NTH + c] = mfma16_h(fa[t & 1][rt][1], fb[u & 1][c][0], acc[rt][hf * NTH + c]);
          __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
          const int tap = sg * TPS + t, dy = tap / 3, dx = tap % 3;
          const int aoff = (dy * HROW16 + dx * PS16) * 16 + hbuf * HALO16_BYTES;
          uint4 ah[2], al[2];
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            ah[rt] = *reinterpret_cast<const uint4*>(smem + a_lane[rt] + aoff);
            al[rt] = *reinterpret_cast<const uint4*>(smem + a_lane[rt] + aoff + 64);
          }
          // all B fragments of the tap first (NT x 2 x 4 registers: two waves per SIMD leave room), then the MFMAs: read one tile at
          // a time, every tile's six MFMAs waited for its own reads
          uint4 bh[NT], bl[NT];
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) {
            bh[ct] = *reinterpret_cast<const uint4*>(smem + b_addr + ((t * NT + ct) * 2 + 0) * 1024);
            bl[ct] = *reinterpret_cast<const uint4*>(smem + b_addr + ((t * NT + ct) * 2 + 1) * 1024);
          }
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][ct] = mfma16_h(ah[rt], bh[ct], acc[rt][ct]);
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][ct] = mfma16_h(ah[rt], bl[ct], acc[rt][ct]);
#pragma unroll
          for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][ct] = mfma16_h(al[rt], bh[ct], acc[rt][ct]);
        }
#endif
        wbuf ^= 1;
      }
      hbuf ^= 1;
    }
    if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first_item = false;

    // ---- epilogue.  acc[rt][ct][i] of lane (col, rg): window 4 rt + rg of the wave's 8, position i; tiles 2m, 2m + 1 = channels
    // 32 m + 2 col, 32 m + 2 col + 1
    const MmJob& J = jt.job[jb];
    if (lit == 0 && tid == 0) J.out_meta->e = e_out;
    constexpr bool POOL = EPI == EPI_LRELU_POOL;
    constexpr int HO = POOL ? HW / 2 : HW;
    char* out = reinterpret_cast<char*>(J.out) + (size_t)img * HO * HO * NC * 4;
#pragma unroll
    for (int m = 0; m < NG; ++m) {
      const unsigned chb = (unsigned)(32 * m + 2 * col) * 2u;       // byte offset of the lane's channel pair in a plane
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int wx = 4 * rt + rg;
        if constexpr (POOL) {
          float best[2];
          unsigned bi[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            best[e] = acc[rt][2 * m + e][0];
            bi[e] = 0;
#pragma unroll
            for (int i = 1; i < 4; ++i) {
              const float v = acc[rt][2 * m + e][i];
              if (v > best[e]) { best[e] = v; bi[e] = i; }     // strict >: the FIRST maximum wins (TF MaxPoolGrad)
            }
            best[e] = ugn_lrelu(best[e] * factor);
            mx = fmaxf(mx, fabsf(best[e]));
          }
          const unsigned pix = (unsigned)((ry0 / 2 + wave) * HO + rx0 / 2 + wx);
          _Float16 h0, l0, h1, l1;
          h2_split(best[0], h0, l0);
          h2_split(best[1], h1, l1);
          UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + chb, h2_pack(h0, h1));
          UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + (unsigned)(NC * 2) + chb, h2_pack(l0, l1));
          uint8_t* oi = J.out_idx + (size_t)img * HO * HO * NC;
          UGN_ST(uint16_t, oi + pix * (unsigned)NC + (unsigned)(32 * m + 2 * col), bi[0] | (bi[1] << 8));
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned pix = (unsigned)((ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * wx + (i & 1));
            float v0 = acc[rt][2 * m][i] * factor, v1 = acc[rt][2 * m + 1][i] * factor;
            if constexpr (EPI == EPI_LRELU) {
              v0 = ugn_lrelu(v0);
              v1 = ugn_lrelu(v1);
            } else if constexpr (EPI == EPI_DGRAD_ACT) {
              const unsigned ah2 = actv[m][4 * rt + i];
              v0 *= (short)(ah2 & 0xffffu) > 0 ? 1.f : UGN_LRELU_ALPHA;      // LeakyReLU' from the sign of the H half
              v1 *= (short)(ah2 >> 16) > 0 ? 1.f : UGN_LRELU_ALPHA;
            }
            mx = fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1)));
            _Float16 h0, l0, h1, l1;
            h2_split(v0, h0, l0);
            h2_split(v1, h1, l1);
            UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + chb, h2_pack(h0, h1));
            UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + (unsigned)(NC * 2) + chb, h2_pack(l0, l1));
          }
        }
      }
    }
    jb = jn;
    lit = nlit;
  }
  h2_publish_amax_block(jt.job[meta_jb].out_meta, mx, reinterpret_cast<float*>(smem), tid, 8);
}



// ---------------------------------------------------------------------------------------------------------------------
// 32 -> 32 layers, TWO workgroups per CU ("D2"): dense swizzled halo tile + resident filter = 78 KB of LDS
// ---------------------------------------------------------------------------------------------------------------------
// The 32-channel layers (a2: 64 x 64 frames, a quarter of the step's 3x3 time) have ONE K chunk and ONE 32-column group: an item is
// 108 MFMAs per wave between a tile wait, two barriers and an epilogue, and with one workgroup per CU nothing runs beside those.
// Two workgroups per CU need <= 80 KB each.  (i) The halo tile is DENSE -- 18 x 18 pixels x the 128-byte record of HBM, no pad
// slots: 41,472 B instead of 52,992 -- and conflict-free by a SWIZZLE instead of a pitch: record quarter jj of pixel (row, col) lives
// in slot jj ^ g, g = ((col >> 1) & 3) << 1 | (row & 1) (the DMA fetches quarter slot ^ g into each slot: free).  The 16 lanes of a
// ds_read_b128 group read slot (plane * 4 + k group) of 16 pixels = 4 windows x 4 positions; 8 of them have an even pixel index and 8
// an odd one (which half of the 16 slots = 256 B a record sits in), and inside each half the 8 values of g -- 4 windows (distinct
// mod 4) x 2 rows -- are distinct, for every tap.  A lane's address is lane base(dx, dy parity) + immediate.  (ii) ONE halo buffer:
// the next tile is fetched behind a second barrier, while the epilogue stores are issued -- the other workgroup of the CU multiplies
// meanwhile.  (iii) The whole filter of a job (36 KB, 16x16x32 tile order) stays in LDS.
#ifndef UGN_MM_D2
#define UGN_MM_D2 3       /* bit 0: the 32 -> 32 forward kernel, bit 1: its pooled data gradient; bit 2 (experiment, measured 2-17 %
                             SLOWER: profiles/r04_kernel_experiments.txt): the other un-pooled launches with <= 64 columns */
#endif
// The packed-filter layout (mm_pack_kernel: mm_tile16 / mm_nr, i.e. UGN_D2P16, UGN_NR_POOLED, UGN_MM_NR) and the kernel a pooled data
// gradient is dispatched to (UGN_MM_D2, mm_nr) must name the same kernel family: an ablation build that switches one without the
// other would read tile16-packed filters through the 32-column block loader and return wrong gradients without any error.
static_assert(!UGN_D2P16 || (UGN_MM_D2 & 2), "-DUGN_MM_D2 without bit 1 needs -DUGN_D2P16=0: the pooled 32 -> 32 data gradient would run "
                                            "launch_mm on filters packed for conv_d2_kernel");
static_assert(!UGN_NR_POOLED || UGN_MM_NR, "-DUGN_MM_NR=0 needs -DUGN_NR_POOLED=0: the pooled 64 -> 64 data gradient would run launch_mm on "
                                           "filters packed for conv_nr_kernel");
constexpr int D2_SLOTS = 18 * 18 * 8;              // 2592 slots of 16 B
constexpr int D2_PIECES = (D2_SLOTS + 63) / 64;    // 41 (the last one half used: the filter starts behind it)
constexpr int D2_W_OFF = 42 * 1024;
constexpr int D2_WBYTES = 9 * 2 * 2 * 1024;        // a 32-column chunk filter: [tap][16-column tile][plane][lane-linear 1 KB]
constexpr int D2_LDS = D2_W_OFF + D2_WBYTES;       // 79,872 B
static_assert(D2_PIECES * 1024 <= D2_W_OFF && 2 * D2_LDS <= 163840, "two workgroups per CU");
// filter bytes in LDS: 32 columns -> the chunk's whole filter (9 taps, 36 KB, fetched once per chunk: no barrier inside a chunk);
// 64 columns -> one tap at a time (8 KB), double-buffered, fetched one tap ahead
template <int NC>
constexpr int d2_lds_bytes() { return D2_W_OFF + (NC == 32 ? D2_WBYTES : 2 * (NC / 16) * 2048); }

// KC / NC: GEMM K and N channels (NC 32 or 64), HW: image size, EPI: EPI_LRELU | EPI_LRELU_POOL | EPI_DGRAD.  16x16x32 tiles as
// conv_mm16_kernel (same filter packing, same epilogue).
// IN_POOLED (-DUGN_D2P16: the 32 -> 32 pooled data gradient on the 16x16x32 tiles of this kernel instead of conv32_d2p_kernel's
// 32x32x16 block): the tile is built as conv32_d2p_kernel builds it -- pooled values + argmax bytes through registers an item ahead,
// scattered into the swizzled halo behind barrier B.
template <int KC, int NC, int HW, int EPI, int IN_POOLED = 0>
__global__ __launch_bounds__(512, 4) void conv_d2_kernel(const MmJobs jt, const void* __restrict__ zeros) {
  constexpr int NT = NC / 16, NG = NC / 32, NCHUNK = KC / 32;
  static_assert(!IN_POOLED || (KC == 32 && NC == 32), "pooled input: one K chunk, resident filter");
  constexpr bool FRES = NC == 32;                    // the chunk's filter resident (else tap by tap)
  constexpr int TAPB = NT * 2048;                    // filter bytes of one tap
  constexpr int RPX = HW / 16, RPI = RPX * RPX;
  static_assert(NC == 32 || NC == 64, "two workgroups per CU: 128 registers per lane");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // A-side role (as conv_mm16_kernel): row r16 of a 16-row tile = window (r16 >> 2) of the tile, position q; k group kg
  const int r16 = lane & 15, kg = lane >> 4, q = r16 & 3, wq = r16 >> 2;
  const int row0 = 2 * wave + (q >> 1), col0 = 2 * wq + (q & 1);          // halo pixel of row tile 0 under tap (0, 0); row tile 1: + 8 columns
  int abase[3][2];                                                         // [dx][parity of dy]: H plane; L plane = ^ 64
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int dp = 0; dp < 2; ++dp) {
      const int g = ((((col0 + dx) >> 1) & 3) << 1) | ((row0 + dp) & 1);
      abase[dx][dp] = (row0 * 18 + col0) * 128 + ((g ^ kg) << 4);
    }
  const int b_lane = D2_W_OFF + lane * 16;
  const int col = lane & 15, rg = lane >> 4;                               // C-side role

  int item = xcd_first_item();
  const int nitems = jt.start[kMaxJobs];
  if (item >= nitems) return;
  int jb = mm_job_of(jt, item), lit = item - jt.start[jb];

  // LDS-DMA pieces of this wave (pi = wave + 8 j): what a lane fetches into slot pi * 64 + lane, precomputed (offset << 12 | row << 5 | col)
  constexpr int NJ = (D2_PIECES + 7) / 8;      // 6
  int hpk[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int sg = (wave + 8 * j) * 64 + lane;
    const int row = sg / 144, rem = sg - row * 144, px = rem >> 3, js = rem & 7;
    const int g = (((px >> 1) & 3) << 1) | (row & 1);
    const int jj = js ^ g;                       // the record quarter this slot holds: 0..3 H plane, 4..7 L plane of the chunk
    const int off = ((row - 1) * HW + (px - 1)) * (KC * 4) + (jj < 4 ? jj * 16 : KC * 2 + (jj - 4) * 16);      // |off| < 2^19
    hpk[j] = sg < D2_SLOTS ? (int)(((unsigned)off << 12) | (unsigned)(row << 5) | (unsigned)px) : (127 << 5);
  }
  auto issue_tile = [&](const MmJob& J, int lit_, int chunk) {
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    const char* base = reinterpret_cast<const char*>(J.in) + ((size_t)img * HW * HW + (size_t)(ry0 * HW + rx0)) * (KC * 4) + chunk * 64;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (wave + 8 * j < D2_PIECES) dma_halo_lane<HW>(base, zeros, ry0, rx0, hpk[j], sbase + (unsigned)(wave + 8 * j) * 1024u);
  };
  // filter bytes [first, first + count) KB of a chunk's (tap-major) packed filter -> LDS at dst
  auto issue_w = [&](const uint16_t* wpk, int chunk, int tap0, int kb, unsigned dst) {
    const char* src = reinterpret_cast<const char*>(wpk) + ((size_t)chunk * 9 + tap0) * TAPB;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int p = wave + 8 * j;
      if (p < kb) dma16(src + p * 1024 + lane * 16, dst + (unsigned)p * 1024u);
    }
  };
  // IN_POOLED: this thread's unit of the 10 x 10 pooled pixels under a region's halo: pooled pixel spp, channel group scg (8 channels)
  const int spp = tid >> 2, scg = tid & 3;
  const int sprow = (spp * 205) >> 11, spcol = spp - sprow * 10;
  uint4 shi = make_uint4(0u, 0u, 0u, 0u), slo = shi;
  uint2 six = make_uint2(0u, 0u);
  auto stg_load = [&](const MmJob& J, int lit_) {
    constexpr int HP = HW / 2;
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int pr = ((rrem / RPX) * 16) / 2 - 1 + sprow, pc = ((rrem % RPX) * 16) / 2 - 1 + spcol;
    const bool ok = tid < 400 && (unsigned)pr < (unsigned)HP && (unsigned)pc < (unsigned)HP;
    shi = make_uint4(0u, 0u, 0u, 0u);
    slo = shi;
    six = make_uint2(0u, 0u);
    if (ok) {
      const size_t o = (size_t)img * HP * HP + (size_t)(pr * HP + pc);
      const char* v = reinterpret_cast<const char*>(J.in) + o * (KC * 4) + (unsigned)(scg * 16);
      shi = *reinterpret_cast<const uint4*>(v);
      slo = *reinterpret_cast<const uint4*>(v + KC * 2);
      six = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(J.in_idx) + o * KC + (unsigned)(scg * 8));
    }
  };
  auto scatter = [&]() {              // registers -> the halo tile: quarter jj of pixel (hy, hx) into slot jj ^ g(hy, hx)
    if (tid >= 400) return;
    const unsigned hv[4] = {shi.x, shi.y, shi.z, shi.w}, lv[4] = {slo.x, slo.y, slo.z, slo.w};
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      const int hy = 2 * sprow - 1 + (pos >> 1), hx = 2 * spcol - 1 + (pos & 1);
      if ((unsigned)hy >= 18u || (unsigned)hx >= 18u) continue;
      unsigned m[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned w = d < 2 ? six.x : six.y;
        const unsigned b0 = (w >> (16 * (d & 1))) & 0xffu, b1 = (w >> (16 * (d & 1) + 8)) & 0xffu;
        m[d] = (b0 == (unsigned)pos ? 0x0000ffffu : 0u) | (b1 == (unsigned)pos ? 0xffff0000u : 0u);
      }
      const int g = (((hx >> 1) & 3) << 1) | (hy & 1);
      char* rec = smem + (hy * 18 + hx) * 128;
      *reinterpret_cast<uint4*>(rec + ((scg ^ g) << 4)) = make_uint4(hv[0] & m[0], hv[1] & m[1], hv[2] & m[2], hv[3] & m[3]);
      *reinterpret_cast<uint4*>(rec + (((4 + scg) ^ g) << 4)) = make_uint4(lv[0] & m[0], lv[1] & m[1], lv[2] & m[2], lv[3] & m[3]);
    }
  };
  if constexpr (IN_POOLED) {
    stg_load(jt.job[jb], lit);
    scatter();
    const int ni = item + gridDim.x;
    if (ni < nitems) { const int j2 = mm_job_of(jt, ni); stg_load(jt.job[j2], ni - jt.start[j2]); }
  } else {
    issue_tile(jt.job[jb], lit, 0);
  }
  issue_w(jt.job[jb].wpk, 0, 0, FRES ? 36 : TAPB / 1024, sbase + D2_W_OFF);
  bool first_item = true, filter_pending = true;
  int wbuf = 0;
  int meta_jb = -1, e_out = 0;
  float factor = 1.f, mx = 0.f;

  for (; item < nitems; item += gridDim.x) {
    const int next_item = item + gridDim.x;
    const bool more = next_item < nitems;
    const int jn = more ? mm_job_of(jt, next_item) : jb, nlit = more ? next_item - jt.start[jn] : lit;
    if (jb != meta_jb) {
      if (meta_jb >= 0) h2_publish_amax(jt.job[meta_jb].out_meta, wave_max(mx), lane);
      mx = 0.f;
      const MmJob& Jm = jt.job[jb];
      const int e_in = Jm.in_meta->e;
      const float amax_in = h2_true_amax(e_in, Jm.in_meta->amax);
      e_out = h2_exp_for_bound(amax_in * Jm.wmeta->l1);
      factor = ldexpf(1.f, e_out - e_in - Jm.wmeta->e);
      meta_jb = jb;
    }
    f32x4 acc[2][NT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int EPI_STORES = EPI == EPI_LRELU_POOL ? NG * 6 : NG * 16;       // per wave and item

#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool last_chunk = chunk + 1 == NCHUNK;
      const bool next_tile = !last_chunk || more;
      const int n_chunk = last_chunk ? 0 : chunk + 1;
      const int nx_job = last_chunk ? jn : jb, n_lit = last_chunk ? nlit : lit;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap == 0 || !FRES) {
          // what this stage reads has landed: the tile and the (first) filter stage were issued behind barrier B of the previous
          // chunk -- for the first chunk of an item BEFORE the previous item's epilogue stores, so a counted wait covers them and
          // none of the stores; a later tap's filter one tap ago
          if (IN_POOLED && !first_item && !filter_pending) {
            // (pooled input: the tile came through registers -- the scatter waited for its loads -- and the loads of the tile after the
            //  next are in flight: nothing to wait for unless a filter was fetched)
          } else if (tap == 0 && chunk == 0 && !first_item && !IN_POOLED) {
            static_assert(EPI_STORES == 6 || EPI_STORES == 12 || EPI_STORES == 16 || EPI_STORES == 32, "vmcnt immediate");
            if constexpr (EPI_STORES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (EPI_STORES == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (EPI_STORES == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          filter_pending = false;
          __syncthreads();            // A: visible; (tap by tap:) the other filter buffer has no readers left
          if constexpr (!FRES) {
            if (tap + 1 < 9) issue_w(jt.job[jb].wpk, chunk, tap + 1, TAPB / 1024, sbase + D2_W_OFF + (unsigned)(wbuf ^ 1) * TAPB);
          }
        }
        const int dy = tap / 3, dx = tap % 3;
        const int aoff = (dy * 18 + dx) * 128;
        const int ab = abase[dx][dy & 1];
        const int b_addr = b_lane + (FRES ? tap * TAPB : wbuf * TAPB);
        uint4 ah[2], al[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          ah[rt] = *reinterpret_cast<const uint4*>(smem + ab + aoff + rt * 1024);
          al[rt] = *reinterpret_cast<const uint4*>(smem + (ab ^ 64) + aoff + rt * 1024);
        }
#pragma unroll
        for (int cp = 0; cp < NT; cp += 2) {        // two column tiles at a time: 16 fragment registers
          uint4 bh[2], bl[2];
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            bh[c2] = *reinterpret_cast<const uint4*>(smem + b_addr + ((cp + c2) * 2 + 0) * 1024);
            bl[c2] = *reinterpret_cast<const uint4*>(smem + b_addr + ((cp + c2) * 2 + 1) * 1024);
          }
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][cp + c2] = mfma16_h(ah[rt], bh[c2], acc[rt][cp + c2]);
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][cp + c2] = mfma16_h(ah[rt], bl[c2], acc[rt][cp + c2]);
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][cp + c2] = mfma16_h(al[rt], bh[c2], acc[rt][cp + c2]);
        }
        if constexpr (!FRES) wbuf ^= 1;
      }
      __syncthreads();                // B: every wave has read its last fragment of the chunk: tile and filter buffers are free
      if (next_tile) {
        if constexpr (IN_POOLED) {
          scatter();                  // the next item's tile (its loads were issued an item ago) ...
          const int n2 = item + 2 * (int)gridDim.x;       // ... and the loads of the one after it
          if (n2 < nitems) { const int j2 = mm_job_of(jt, n2); stg_load(jt.job[j2], n2 - jt.start[j2]); }
        } else {
          issue_tile(jt.job[nx_job], n_lit, n_chunk);
        }
        // (one chunk of 32 columns: the filter in LDS is the whole filter of the job and stays until the job changes)
        if (!(FRES && NCHUNK == 1 && nx_job == jb)) {
          issue_w(jt.job[nx_job].wpk, n_chunk, 0, FRES ? 36 : TAPB / 1024, sbase + D2_W_OFF + (FRES ? 0u : (unsigned)wbuf * TAPB));
          filter_pending = true;
        }
      }
    }
    first_item = false;

    // ---- epilogue (conv_mm16_kernel's): acc[rt][ct][i] of lane (col, rg): window 4 rt + rg of the wave's 8, position i; tiles 2m,
    // 2m + 1 = channels 32 m + 2 col, 32 m + 2 col + 1
    const int img = lit / RPI, rrem = lit % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    const MmJob& J = jt.job[jb];
    if (lit == 0 && tid == 0) J.out_meta->e = e_out;
    constexpr bool POOL = EPI == EPI_LRELU_POOL;
    constexpr int HO = POOL ? HW / 2 : HW;
    char* out = reinterpret_cast<char*>(J.out) + (size_t)img * HO * HO * NC * 4;
#pragma unroll
    for (int m = 0; m < NG; ++m) {
      const unsigned chb = (unsigned)(32 * m + 2 * col) * 2u;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int wx = 4 * rt + rg;
        if constexpr (POOL) {
          float best[2];
          unsigned bi[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            best[e] = acc[rt][2 * m + e][0];
            bi[e] = 0;
#pragma unroll
            for (int i = 1; i < 4; ++i) {
              const float v = acc[rt][2 * m + e][i];
              if (v > best[e]) { best[e] = v; bi[e] = i; }     // strict >: the FIRST maximum wins (TF MaxPoolGrad)
            }
            best[e] = ugn_lrelu(best[e] * factor);
            mx = fmaxf(mx, fabsf(best[e]));
          }
          const unsigned pix = (unsigned)((ry0 / 2 + wave) * HO + rx0 / 2 + wx);
          _Float16 h0, l0, h1, l1;
          h2_split(best[0], h0, l0);
          h2_split(best[1], h1, l1);
          UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + chb, h2_pack(h0, h1));
          UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + (unsigned)(NC * 2) + chb, h2_pack(l0, l1));
          uint8_t* oi = J.out_idx + (size_t)img * HO * HO * NC;
          UGN_ST(uint16_t, oi + pix * (unsigned)NC + (unsigned)(32 * m + 2 * col), bi[0] | (bi[1] << 8));
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned pix = (unsigned)((ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * wx + (i & 1));
            float v0 = acc[rt][2 * m][i] * factor, v1 = acc[rt][2 * m + 1][i] * factor;
            if constexpr (EPI == EPI_LRELU) {
              v0 = ugn_lrelu(v0);
              v1 = ugn_lrelu(v1);
            } else if constexpr (EPI == EPI_DGRAD_ACT) {      // LeakyReLU' from the sign of the H half of the layer's input
              const unsigned ah2 = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(J.act) + (size_t)img * HW * HW * NC * 4 +
                                                                      pix * (unsigned)(NC * 4) + chb);
              v0 *= (short)(ah2 & 0xffffu) > 0 ? 1.f : UGN_LRELU_ALPHA;
              v1 *= (short)(ah2 >> 16) > 0 ? 1.f : UGN_LRELU_ALPHA;
            }
            mx = fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1)));
            _Float16 h0, l0, h1, l1;
            h2_split(v0, h0, l0);
            h2_split(v1, h1, l1);
            UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + chb, h2_pack(h0, h1));
            UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + (unsigned)(NC * 2) + chb, h2_pack(l0, l1));
          }
        }
      }
    }
    jb = jn;
    lit = nlit;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  h2_publish_amax_block(jt.job[meta_jb].out_meta, mx, reinterpret_cast<float*>(smem), tid, 8);
}

// ---------------------------------------------------------------------------------------------------------------------
// "NR": the N channels split over the waves, the filter fragments in REGISTERS (64 / 128 output columns, un-pooled input)
// ---------------------------------------------------------------------------------------------------------------------
// conv_mm16_kernel gives a wave 32 pixels x ALL columns, so every wave reads the whole filter stage from LDS: 16 B fragments per
// tap, 2 row tiles to use each on (0.42-0.5 ds_read_b128 per MFMA: the LDS array 83 % busy if the matrix pipe were 100 %), and the
// stage buffers cost a barrier per tap (128 columns) or per three taps -- 36 / 12 lock-steps of all 8 waves per item.
// Here a wave owns ONE 16-column tile and ROWS = 16 / 8 rows of 16 pixels (row-major 16-pixel tiles):
//   * its B fragments (2 KB per tap: 32 k x 16 columns x {H, L}) come straight from L2 into registers by plain global loads, the
//     three taps of a column dx one dx ahead (24 registers in use + 24 in flight): no filter in LDS, no filter barrier at all;
//     the L2 -> CU filter traffic is what the LDS-DMA moved before (144 KB per K chunk and workgroup).
//   * an A fragment = 16 pixels of ONE halo row shifted by dx, and row j serves the taps dy = 0, 1, 2 of the output rows j, j - 1,
//     j - 2: ROWS + 2 reads x {H, L} per dx feed ROWS x 3 x 3 MFMAs = 0.25 ds_read_b128 per MFMA (0.28 for 8 rows).
//   * ONE barrier per K chunk (the next chunk's tile, fetched a chunk ahead into the other halo buffer, is visible).
// Halo tile: dense, 18 x 18 pixels x the 128-byte record.  ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} (+ 32) (MI355X_MICROARCH.md, LDS): a group reads 16 consecutive pixels x of a row -- 8 of each parity,
// i.e. of each 128-byte half of the 256 bytes the banks span -- with k group kg for x in {0-3, 12-15} and kg ^ 1 for x in 4-11.
// Quarter jj = plane * 4 + kg of the pixel in column col sits in slot (jj & 1) | (((jj >> 1) ^ (col >> 1)) & 3) << 1: the 8
// pixels of one parity take the 4 values of (col >> 1) & 3 twice, at x and x + 8 -- which are always in DIFFERENT kg classes, so
// bit 0 tells them apart: conflict-free for every dx.  (XOR-ing the whole of (col >> 1) & 7 into jj, the first attempt, puts x = 3
// and x = 5 of a group on one slot for odd dx: SQ_LDS_BANK_CONFLICT 0.40 per LDS cycle.)  A lane's address is lane base(dx) +
// row * 2304 (immediate); L plane = base ^ 64; the second buffer = base ^ 65536.
// Epilogue: a lane holds ONE channel at 4 pixels of a row per tile; lanes (c, c ^ 1) exchange two of them (DPP) and then own the
// channel PAIR at 2 pixels -- dword stores as everywhere else.  2x2 MaxPool: rows (2p, 2p + 1) are two tiles of the same lane.
constexpr int NR_BUF = 65536;                      // stride of the two halo buffers
constexpr int NR_PIECES = 48;                      // 41.5 KB of tile, fetched as 6 pieces per wave (the rest: zeros, never read)
constexpr int NR_LDS = NR_BUF + NR_PIECES * 1024;  // 114,688 B

// slot of record quarter jj (plane * 4 + k group) of the pixel in tile column col
__host__ __device__ constexpr int nr_slot(int jj, int col) { return (jj & 1) | ((((jj >> 1) ^ (col >> 1)) & 3) << 1); }

__device__ __forceinline__ float dpp_swap1(float v) {       // value of lane ^ 1 (quad_perm [1, 0, 3, 2])
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
}

// IN_POOLED (the 64 -> 64 pooled data gradient): `in` is the POOLED gradient + argmax bytes; a thread's pooled pixel x 8 channels of
// the chunk after the next travel through registers (plain global loads a chunk ahead) and are scattered -- MaxPool backward: the value
// where the argmax byte names the position, zero elsewhere -- into the other halo buffer a few rows into the chunk.
template <int KC, int NC, int HW, int EPI, int IN_POOLED = 0>
__global__ __launch_bounds__(512, 2) void conv_nr_kernel(const MmJobs jt, const void* __restrict__ zeros) {
  constexpr int NT = NC / 16, MPARTS = 8 / NT > 0 ? 8 / NT : 1, ROWS = 16 / MPARTS, NCHUNK = KC / 32;
  constexpr int RPX = HW / 16, RPI = RPX * RPX;
  static_assert(NC == 64 || NC == 128, "a wave = one 16-column tile x 8 or 16 rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = wave % NT, mh = wave / NT;           // column tile, row part
  // A side: pixel x of a row, k group kg.  C side: column (= channel 16 ng + x), pixels 4 kg ... 4 kg + 3 of the row.
  const int x = lane & 15, kg = lane >> 4;
  int ab[3];                                          // [dx]: H plane in the current buffer
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) ab[dx] = ((mh * ROWS) * 18 + x + dx) * 128 + (nr_slot(kg, x + dx) << 4);

  int item = xcd_first_item();
  const int nitems = jt.start[kMaxJobs];
  if (item >= nitems) return;
  int jb = mm_job_of(jt, item), lit = item - jt.start[jb];

  // LDS-DMA pieces of this wave (wave + 8 j): what a lane fetches into slot piece * 64 + lane (offset << 12 | row << 5 | col)
  int hpk[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int sg = (wave + 8 * j) * 64 + lane;
    const int row = sg / 144, rem = sg - row * 144, px = rem >> 3, js = rem & 7;
    const int jj = nr_slot(js, px);                // the record quarter this slot holds (the map is an involution): 0..3 H plane, 4..7 L plane
    const int off = ((row - 1) * HW + (px - 1)) * (KC * 4) + (jj < 4 ? jj * 16 : KC * 2 + (jj - 4) * 16);      // |off| < 2^19
    hpk[j] = sg < D2_SLOTS ? (int)(((unsigned)off << 12) | (unsigned)(row << 5) | (unsigned)px) : (127 << 5);
  }
  auto issue_tile = [&](const MmJob& J, int lit_, int chunk, unsigned dst) {
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    const char* base = reinterpret_cast<const char*>(J.in) + ((size_t)img * HW * HW + (size_t)(ry0 * HW + rx0)) * (KC * 4) + chunk * 64;
#pragma unroll
    for (int j = 0; j < 6; ++j) dma_halo_lane<HW>(base, zeros, ry0, rx0, hpk[j], dst + (unsigned)(wave + 8 * j) * 1024u);
  };
  // the wave's B fragments of column dx of a chunk: [dy][plane]
  uint4 nb[3][2];
  auto load_b = [&](const uint16_t* wpk, int chunk, int dx) {
    const char* p = reinterpret_cast<const char*>(wpk) + (((size_t)chunk * 9 + dx) * NT + ng) * 2048 + lane * 16;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      nb[dy][0] = *reinterpret_cast<const uint4*>(p + dy * (3 * NT * 2048));
      nb[dy][1] = *reinterpret_cast<const uint4*>(p + dy * (3 * NT * 2048) + 1024);
    }
  };
  // IN_POOLED: this thread's unit of the 10 x 10 pooled pixels under a region's halo -- pooled pixel spp, channel group scg (8 channels)
  const int spp = tid >> 2, scg = tid & 3;
  const int sprow = (spp * 205) >> 11, spcol = spp - sprow * 10;
  uint4 shi = make_uint4(0u, 0u, 0u, 0u), slo = shi;
  uint2 six = make_uint2(0u, 0u);
  auto stg_load = [&](const MmJob& J, int lit_, int chunk) {
    constexpr int HP = HW / 2;
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int pr = ((rrem / RPX) * 16) / 2 - 1 + sprow, pc = ((rrem % RPX) * 16) / 2 - 1 + spcol;
    const bool ok = tid < 400 && (unsigned)pr < (unsigned)HP && (unsigned)pc < (unsigned)HP;
    shi = make_uint4(0u, 0u, 0u, 0u);
    slo = shi;
    six = make_uint2(0u, 0u);           // (outside the image the gradient is zero: the scatter still overwrites the stale halo)
    if (ok) {
      const size_t o = (size_t)img * HP * HP + (size_t)(pr * HP + pc);
      const char* v = reinterpret_cast<const char*>(J.in) + o * (KC * 4) + chunk * 64 + scg * 16;
      shi = *reinterpret_cast<const uint4*>(v);
      slo = *reinterpret_cast<const uint4*>(v + KC * 2);
      six = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(J.in_idx) + o * KC + chunk * 32 + scg * 8);
    }
  };
  auto scatter = [&](int buf) {       // registers -> halo buffer `buf`: quarter jj of pixel (hy, hx) into slot nr_slot(jj, hx)
    if (tid >= 400) return;
    const unsigned hv[4] = {shi.x, shi.y, shi.z, shi.w}, lv[4] = {slo.x, slo.y, slo.z, slo.w};
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      const int hy = 2 * sprow - 1 + (pos >> 1), hx = 2 * spcol - 1 + (pos & 1);
      if ((unsigned)hy >= 18u || (unsigned)hx >= 18u) continue;
      unsigned m[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned w = d < 2 ? six.x : six.y;
        const unsigned b0 = (w >> (16 * (d & 1))) & 0xffu, b1 = (w >> (16 * (d & 1) + 8)) & 0xffu;
        m[d] = (b0 == (unsigned)pos ? 0x0000ffffu : 0u) | (b1 == (unsigned)pos ? 0xffff0000u : 0u);
      }
      char* rec = smem + buf * NR_BUF + (hy * 18 + hx) * 128;
      *reinterpret_cast<uint4*>(rec + (nr_slot(scg, hx) << 4)) = make_uint4(hv[0] & m[0], hv[1] & m[1], hv[2] & m[2], hv[3] & m[3]);
      *reinterpret_cast<uint4*>(rec + (nr_slot(4 + scg, hx) << 4)) = make_uint4(lv[0] & m[0], lv[1] & m[1], lv[2] & m[2], lv[3] & m[3]);
    }
  };
  // tile after (item, chunk): (job, local item, chunk); valid = there is one
  if constexpr (IN_POOLED) {
    stg_load(jt.job[jb], lit, 0);
    scatter(0);
    if (NCHUNK > 1) {
      stg_load(jt.job[jb], lit, 1);
    } else {
      const int ni = item + gridDim.x;
      if (ni < nitems) { const int j2 = mm_job_of(jt, ni); stg_load(jt.job[j2], ni - jt.start[j2], 0); }
    }
  } else {
    issue_tile(jt.job[jb], lit, 0, sbase);
  }
  load_b(jt.job[jb].wpk, 0, 0);
  int hbuf = 0;
  bool first_item = true;
  int meta_jb = -1, e_out = 0;
  float factor = 1.f, mx = 0.f;
  const int odd = x & 1;

  for (; item < nitems; item += gridDim.x) {
    const int next_item = item + gridDim.x;
    const bool more = next_item < nitems;
    const int jn = more ? mm_job_of(jt, next_item) : jb, nlit = more ? next_item - jt.start[jn] : lit;
    if (jb != meta_jb) {
      if (meta_jb >= 0) h2_publish_amax(jt.job[meta_jb].out_meta, wave_max(mx), lane);
      mx = 0.f;
      const MmJob& Jm = jt.job[jb];
      const int e_in = Jm.in_meta->e;
      const float amax_in = h2_true_amax(e_in, Jm.in_meta->amax);
      e_out = h2_exp_for_bound(amax_in * Jm.wmeta->l1);
      factor = ldexpf(1.f, e_out - e_in - Jm.wmeta->e);
      meta_jb = jb;
    }
    f32x4 acc[ROWS];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int img = lit / RPI, rrem = lit % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    // what this lane stores after the exchange: the channel pair (16 ng + (x & ~1), + 1) at pixels 4 kg + 2 odd + t of each row
    const unsigned chb = (unsigned)(16 * ng + (x & ~1)) * 2u;
    const int px0 = rx0 + 4 * kg + 2 * odd, py0 = ry0 + mh * ROWS;
    constexpr bool ACT = EPI == EPI_DGRAD_ACT;
    constexpr int EPI_STORES = UGN_NR_CNT ? (EPI == EPI_LRELU_POOL ? (ROWS / 2) * 3 : ROWS * 4) : 0;      // per lane and item
    unsigned actv[ACT ? ROWS : 1][2];                 // H halves of the layer's input at those pixels x channel pair
    // (16 rows: the first 8 are fetched behind the last MFMAs, the rest at the top of the epilogue into the registers the B
    //  fragments leave -- all 32 at once spill)
    constexpr int APF = ROWS > 8 ? UGN_NR_ACTPF : ROWS;
    auto load_act = [&](int m0, int m1) {
      const char* act = reinterpret_cast<const char*>(jt.job[jb].act) + (size_t)img * HW * HW * NC * 4 + chb;
#pragma unroll
      for (int m = m0; m < m1; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          actv[ACT ? m : 0][t] = *reinterpret_cast<const unsigned*>(act + (unsigned)((py0 + m) * HW + px0 + t) * (unsigned)(NC * 4));
    };

#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool last_chunk = chunk + 1 == NCHUNK;
      const bool next_tile = !last_chunk || more;
      const int n_chunk = last_chunk ? 0 : chunk + 1;
      const int nx_job = last_chunk ? jn : jb, n_lit = last_chunk ? nlit : lit;
      // this wave's pieces of the chunk's tile (issued a chunk ago) have landed.  First chunk of a later item: they and the B fragments
      // in flight are OLDER than the previous item's epilogue stores, so a counted wait that leaves the stores in flight covers them
      if (chunk == 0 && !first_item) {
        if constexpr (EPI_STORES >= 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
        else if constexpr (EPI_STORES == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if constexpr (EPI_STORES == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();                                      // tile visible; nobody reads the other buffer any more
      // (the next tile is issued a few rows into the first column: the compiler's own counted wait for the B fragments of this
      //  column -- it does not see the DMA -- would otherwise wait for the DMA issued in front of it)
      auto tile_ahead = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (IN_POOLED) {
          if (next_tile) {
            scatter(hbuf ^ 1);              // the next tile (its values arrived a chunk ago) ...
            // ... and the loads of the one after it: chunk n_chunk + 1 of the same item, or chunk 0 of the item after nx
            if (n_chunk + 1 < NCHUNK) {
              stg_load(jt.job[nx_job], n_lit, n_chunk + 1);
            } else {
              const int base_item = last_chunk ? next_item : item;      // the item the next tile belongs to
              const int ni = base_item + gridDim.x;
              if (ni < nitems) { const int j2 = mm_job_of(jt, ni); stg_load(jt.job[j2], ni - jt.start[j2], 0); }
            }
          }
        } else {
          if (next_tile) issue_tile(jt.job[nx_job], n_lit, n_chunk, sbase + (unsigned)(hbuf ^ 1) * NR_BUF);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        uint4 cb[3][2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) { cb[dy][0] = nb[dy][0]; cb[dy][1] = nb[dy][1]; }
        // (unconditional -- behind the last chunk of the last item it re-reads chunk 0: a conditional load is its own basic block,
        //  which LLVM sinks below the column's MFMAs)
        if (dx < 2) load_b(jt.job[jb].wpk, chunk, dx + 1);
        else load_b(jt.job[nx_job].wpk, n_chunk, 0);
        if constexpr (ACT) {
          if (dx == 2 && last_chunk) load_act(0, APF);
        }
        // (left alone hipcc sinks the loads of the NEXT column behind this column's MFMAs -- into the registers they free -- and
        //  waits for them there)
        __builtin_amdgcn_sched_barrier(0);
        const int a_h = ab[dx], a_l = ab[dx] ^ 64;
#if UGN_NR_APF
        // the A fragments of row j + UGN_NR_APF are read before the MFMAs of row j (a ring of UGN_NR_APF + 1 register sets)
        constexpr int AD = UGN_NR_APF, AR = AD + 1;
        uint4 fa[AR][2];
#pragma unroll
        for (int j = 0; j < AD; ++j) {
          fa[j][0] = *reinterpret_cast<const uint4*>(smem + a_h + j * 2304);
          fa[j][1] = *reinterpret_cast<const uint4*>(smem + a_l + j * 2304);
        }
#pragma unroll
        for (int j = 0; j < ROWS + 2; ++j) {
          if (j + AD < ROWS + 2) {
            fa[(j + AD) % AR][0] = *reinterpret_cast<const uint4*>(smem + a_h + (j + AD) * 2304);
            fa[(j + AD) % AR][1] = *reinterpret_cast<const uint4*>(smem + a_l + (j + AD) * 2304);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              const int m = j - dy;
              if (m >= 0 && m < ROWS) acc[m] = mfma16_h(fa[j % AR][pass == 2 ? 1 : 0], cb[dy][pass == 1 ? 1 : 0], acc[m]);
            }
          __builtin_amdgcn_sched_barrier(0);
          if (dx == 0 && j == 2) tile_ahead();
        }
#else
#pragma unroll
        for (int j = 0; j < ROWS + 2; ++j) {
          const uint4 ah = *reinterpret_cast<const uint4*>(smem + a_h + j * 2304);
          const uint4 al = *reinterpret_cast<const uint4*>(smem + a_l + j * 2304);
#pragma unroll
          for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              const int m = j - dy;
              if (m >= 0 && m < ROWS) acc[m] = mfma16_h(pass == 2 ? al : ah, cb[dy][pass == 1 ? 1 : 0], acc[m]);
            }
          if (dx == 0 && j == 2) tile_ahead();
        }
#endif
      }
      hbuf ^= 1;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) ab[dx] ^= NR_BUF;
    }

    first_item = false;
    // ---- epilogue
    if constexpr (ACT && APF < ROWS) {
      __builtin_amdgcn_sched_barrier(0);
      load_act(APF, ROWS);
    }
    const MmJob& J = jt.job[jb];
    if (lit == 0 && tid == 0) J.out_meta->e = e_out;
    constexpr bool POOL = EPI == EPI_LRELU_POOL;
    constexpr int HO = POOL ? HW / 2 : HW;
    char* out = reinterpret_cast<char*>(J.out) + (size_t)img * HO * HO * NC * 4 + chb;
    if constexpr (POOL) {
      uint8_t* oi = J.out_idx + (size_t)img * HO * HO * NC + (unsigned)(16 * ng + (x & ~1));
#pragma unroll
      for (int m2 = 0; m2 < ROWS / 2; ++m2) {
        float best[2];
        unsigned bi[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {               // window w of the lane: pixels 4 kg + 2 w, + 1 of rows 2 m2, 2 m2 + 1
          best[w] = acc[2 * m2][2 * w];
          bi[w] = 0;
          const float cand[3] = {acc[2 * m2][2 * w + 1], acc[2 * m2 + 1][2 * w], acc[2 * m2 + 1][2 * w + 1]};
#pragma unroll
          for (int i = 0; i < 3; ++i)
            if (cand[i] > best[w]) { best[w] = cand[i]; bi[w] = i + 1; }       // strict >: the FIRST maximum wins (TF MaxPoolGrad)
          best[w] = ugn_lrelu(best[w] * factor);
          mx = fmaxf(mx, fabsf(best[w]));
        }
        // the even lane keeps window 0 and receives the odd channel's window 0; the odd lane keeps window 1
        const float keep = odd ? best[1] : best[0], recv = dpp_swap1(odd ? best[0] : best[1]);
        const unsigned kbi = odd ? bi[1] : bi[0];
        const unsigned rbi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? bi[0] : bi[1]), 0xB1, 0xf, 0xf, false);
        const float ve = odd ? recv : keep, vo = odd ? keep : recv;
        const unsigned be = odd ? rbi : kbi, bo = odd ? kbi : rbi;
        const unsigned pix = (unsigned)((py0 / 2 + m2) * HO + rx0 / 2 + 2 * kg + odd);
        _Float16 h0, l0, h1, l1;
        h2_split(ve, h0, l0);
        h2_split(vo, h1, l1);
        UGN_ST(unsigned, out + pix * (unsigned)(NC * 4), h2_pack(h0, h1));
        UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + (unsigned)(NC * 2), h2_pack(l0, l1));
        UGN_ST(uint16_t, oi + pix * (unsigned)NC, be | (bo << 8));
      }
    } else {
#pragma unroll
      for (int m = 0; m < ROWS; ++m) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float keep = odd ? acc[m][2 + t] : acc[m][t], recv = dpp_swap1(odd ? acc[m][t] : acc[m][2 + t]);
          float v0 = (odd ? recv : keep) * factor, v1 = (odd ? keep : recv) * factor;
          if constexpr (EPI == EPI_LRELU) {
            v0 = ugn_lrelu(v0);
            v1 = ugn_lrelu(v1);
          } else if constexpr (ACT) {
            const unsigned ah2 = actv[m][t];
            v0 *= (short)(ah2 & 0xffffu) > 0 ? 1.f : UGN_LRELU_ALPHA;      // LeakyReLU' from the sign of the H half
            v1 *= (short)(ah2 >> 16) > 0 ? 1.f : UGN_LRELU_ALPHA;
          }
          mx = fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1)));
          const unsigned pix = (unsigned)((py0 + m) * HW + px0 + t);
          _Float16 h0, l0, h1, l1;
          h2_split(v0, h0, l0);
          h2_split(v1, h1, l1);
          UGN_ST(unsigned, out + pix * (unsigned)(NC * 4), h2_pack(h0, h1));
          UGN_ST(unsigned, out + pix * (unsigned)(NC * 4) + (unsigned)(NC * 2), h2_pack(l0, l1));
        }
      }
    }
    jb = jn;
    lit = nlit;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  h2_publish_amax_block(jt.job[meta_jb].out_meta, mx, reinterpret_cast<float*>(smem), tid, 8);
}

// ---------------------------------------------------------------------------------------------------------------------
// data gradient of the pooled 32 -> 32 layer (a2) FUSED with the weight gradient of the 5x5 first layer
// ---------------------------------------------------------------------------------------------------------------------
// dz1 = dL/da1 (the output of this data gradient, 315 MB per 600 frames) has ONE consumer: the 5x5 layer's weight gradient
//   dW1[tap][c][co] = sum over pixels of x[pixel + tap][c] * dz1[pixel][co] * LeakyReLU'(a1[pixel][co])
// A lane of the 32-row MFMA block holds, after the taps, dz1 of ONE channel at 16 pixels of the region -- which is a B fragment
// (K = pixels, N = channel) as it stands.  So the workgroup multiplies it here with the matching input patch (A: M = the 25 * cin
// (tap, channel) rows, gathered from a 20 x 20 patch of x in LDS, split into f16 halves on the fly) and keeps dW1 in accumulator
// registers across all items of a job: dz1 is never written (0.94 GB per launch at 24 clips) or read back (the same again), and the
// three conv5x5_wgrad launches of a step disappear.  Structure otherwise = the filter_resident pooled path of conv_mm_kernel.
struct W5Job {
  const uint16_t* in;          // pooled gradient dL/dp2, H2 [n][32][32][2][32]
  const uint8_t* in_idx;       // argmax bytes of the a2 pooling
  const H2Meta* in_meta;
  const uint16_t* wpk;         // packed data-gradient filter of a2
  const WMeta* wmeta;
  const float* x;              // network input fp32 [n][60][60][cin]
  const H2Meta* x_meta;        // {0, bits(max|x|)}
  const uint32_t* sign;        // a1 sign words [n][64][64] (bit c: a1[..][c] > 0)
  H2Meta* scale;               // out: e = exponent to undo on the slab sums (e_dz + e_x)
  float* slab;                 // [workgroups][50][32] partial sums of this job
  int cin;                     // 1 or 2
};
struct W5Jobs {
  W5Job job[kMaxJobs];
  int start[kMaxJobs + 1];
};
constexpr int W5_PATCH_BYTES = 13 * 256;      // 20 x 20 x 2 floats = 3200 B -> 13 dword pieces of 256 B
constexpr int W5_LDS = W_OFF + 3 * Geo<32>::WSTAGE + STG_BYTES + 2 * W5_PATCH_BYTES;

__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_dst_uniform) {      // 4 B per lane (LDS address = M0 + lane * 4)
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

__global__ __launch_bounds__(512, 2) void dgrad32_w5_kernel(const W5Jobs jt, const void* __restrict__ zeros) {
  constexpr int KC = 32, HW = 64, WSTAGE = Geo<32>::WSTAGE, WPIECES = Geo<32>::WPIECES;
  constexpr int RPX = HW / 16, RPI = RPX * RPX;
  constexpr int STG_OFF = W_OFF + 3 * WSTAGE, PATCH_OFF = STG_OFF + STG_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, win = r >> 2, q = r & 3;
  const int a_lane = ((2 * wave + (q >> 1)) * HROW + (2 * win + (q & 1)) * 9) * 16 + h * 16;
  const int b_lane = W_OFF + lane * 16;
  const int c = lane & 31;

  int item = xcd_first_item();
  const int nitems = jt.start[kMaxJobs];
  if (item >= nitems) return;
  auto job_of = [&](int it) {
    int jb = 0;
#pragma unroll
    for (int j = 1; j < kMaxJobs; ++j) jb += it >= jt.start[j] ? 1 : 0;
    return jb;
  };
  int jb = job_of(item), lit = item - jt.start[jb];

  // staging tile (2 pieces per wave) and input patch (<= 2 dword pieces per wave) of item lit_ of job J
  auto fetch = [&](const W5Job& J, int lit_, int pb) {
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    const char* vb = reinterpret_cast<const char*>(J.in) + (size_t)img * 32 * 32 * KC * 4;
    const char* ib = reinterpret_cast<const char*>(J.in_idx) + (size_t)img * 32 * 32 * KC;
#pragma unroll
    for (int j = 0; j < 2; ++j) dma_pooled_piece<KC, HW>(vb, ib, zeros, ry0, rx0, 0, wave * 2 + j, lane, sbase + STG_OFF);
    const int cin = J.cin;
    const float* xb = J.x + (size_t)img * 3600 * cin;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = wave * 2 + j;
      if (p < 13) {
        const int e = p * 64 + lane, pe = cin == 2 ? e >> 1 : e, ch = cin == 2 ? (e & 1) : 0;
        const int yy = (pe * 3277) >> 16, xx = pe - yy * 20;              // pe / 20 for pe < 800
        const int ry = ry0 - 4 + yy, rx = rx0 - 4 + xx;
        const bool ok = pe < 400 && (unsigned)ry < 60u && (unsigned)rx < 60u;
        dma4(ok ? (const void*)(xb + (ry * 60 + rx) * cin + ch) : zeros, sbase + PATCH_OFF + (unsigned)pb * W5_PATCH_BYTES + (unsigned)p * 256u);
      }
    }
  };
  auto load_filter = [&](const uint16_t* wpk) {
    if (wave < 4) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const char* src = reinterpret_cast<const char*>(wpk) + (size_t)d * WSTAGE;
#pragma unroll
        for (int j = 0; j < WPIECES / 4; ++j) {
          const int p = wave + 4 * j;
          dma16(src + p * 1024 + lane * 16, sbase + W_OFF + (unsigned)d * WSTAGE + (unsigned)p * 1024u);
        }
      }
    }
  };
  // dW1 partial sums of the current job: rows (tap, channel) 0..31 and 32..63 (50 used with two input channels)
  f32x16 acc5[2];
  auto zero5 = [&]() {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc5[mb][i] = 0.f;
  };
  // add the eight waves' sums of job j through LDS (`scr`: 32 KB nobody else touches at that point) and write the workgroup's slab
  auto flush5 = [&](int j, float* scr) {
    const int rows = 25 * jt.job[j].cin;
    float* slab = jt.job[j].slab + (size_t)blockIdx.x * (50 * 32);
#pragma unroll 1
    for (int mb = 0; mb < 2; ++mb) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) scr[wave * 1024 + i * 64 + lane] = acc5[mb][i];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + 512 * k, reg = e >> 6, ln = e & 63;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += scr[w * 1024 + e];
        const int row = mb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5);
        if (row < rows) slab[row * 32 + (ln & 31)] = sum;
      }
    }
    __syncthreads();
  };

  // ---- prologue
  fetch(jt.job[jb], lit, 0);
  load_filter(jt.job[jb].wpk);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  scatter_pooled(smem + STG_OFF, smem, tid);
  int hbuf = 0, pbuf = 0, cur_job = -1, e_dz = 0, e_x = 0, cin = 1;
  float factor = 1.f, xs = 1.f;
  zero5();

  for (; item < nitems; item += gridDim.x) {
    const int next_item = item + gridDim.x;
    const bool more = next_item < nitems;
    const int jn = more ? job_of(next_item) : jb, nlit = more ? next_item - jt.start[jn] : lit;
    if (jb != cur_job) {               // (wave-uniform; at most kMaxJobs times per workgroup)
      if (cur_job >= 0) {
        // every wave has left the previous item's taps (the barrier inside flush5); the halo buffer that item read is free
        flush5(cur_job, reinterpret_cast<float*>(smem + (hbuf ^ 1) * HALO_BYTES));
        zero5();
        load_filter(jt.job[jb].wpk);
      }
      const W5Job& Jm = jt.job[jb];
      const int e_in = Jm.in_meta->e;
      e_dz = h2_exp_for_bound(h2_true_amax(e_in, Jm.in_meta->amax) * Jm.wmeta->l1);
      factor = ldexpf(1.f, e_dz - e_in - Jm.wmeta->e);
      e_x = h2_exp_for_bound(__uint_as_float(Jm.x_meta->amax));
      xs = ldexpf(1.f, e_x);
      cin = Jm.cin;
      if (lit == 0 && tid == 0) Jm.scale->e = e_dz + e_x;
      cur_job = jb;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int img = lit / RPI, rrem = lit % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    if (more) fetch(jt.job[jn], nlit, pbuf ^ 1);
    // LeakyReLU'(a1) of the lane's 16 pixels: the sign words (bit c), fetched under the taps
    uint32_t sw[16];
    {
      const uint32_t* sg = jt.job[jb].sign + (size_t)img * HW * HW;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int g = rr >> 2, i = rr & 3;
        sw[rr] = sg[(ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * (2 * g + h) + (i & 1)];
      }
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int a_addr = a_lane + hbuf * HALO_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int aoff = (dy * HROW + dx * 9) * 16 + s * 32;
        const uint4 ah = *reinterpret_cast<const uint4*>(smem + a_addr + aoff);
        const uint4 al = *reinterpret_cast<const uint4*>(smem + a_addr + aoff + 64);
        const uint4 bh = *reinterpret_cast<const uint4*>(smem + b_lane + ((tap * 2 + s) * 2 + 0) * 1024);
        const uint4 bl = *reinterpret_cast<const uint4*>(smem + b_lane + ((tap * 2 + s) * 2 + 1) * 1024);
        acc = mfma_h(ah, bh, acc);
        acc = mfma_h(ah, bl, acc);
        acc = mfma_h(al, bh, acc);
      }
      if (tap == 4 && more) {     // MaxPool backward of the next tile: staging -> the other halo buffer (its patch has landed too)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        scatter_pooled(smem + STG_OFF, smem + (hbuf ^ 1) * HALO_BYTES, tid);
      }
    }
    // ---- dz1 of channel c at the lane's 16 pixels = two B fragments (k = 8 registers each); A = the input patch under (tap, channel) row
    const float* patch = reinterpret_cast<const float*>(smem + PATCH_OFF + pbuf * W5_PATCH_BYTES);
    const int sh = cin - 1;           // (patch index of a pixel = pixel << sh | channel)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned bhw[4], blw[4];
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        _Float16 hh[2], ll[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const int rr = 8 * s + 2 * k2 + e2;
          const float v = acc[rr] * factor * (((sw[rr] >> c) & 1u) ? 1.f : UGN_LRELU_ALPHA);
          h2_split(v, hh[e2], ll[e2]);
        }
        bhw[k2] = h2_pack(hh[0], hh[1]);
        blw[k2] = h2_pack(ll[0], ll[1]);
      }
      const uint4 bh = make_uint4(bhw[0], bhw[1], bhw[2], bhw[3]), bl = make_uint4(blw[0], blw[1], blw[2], blw[3]);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        if (mb == 1 && cin == 1) break;          // (wave-uniform: 25 rows fit the first block)
        const int k5 = mb * 32 + (lane & 31);
        const int kk = k5 < 25 * cin ? k5 : 0;   // (padded rows: any valid address, the sums are discarded)
        const int tp = cin == 2 ? kk >> 1 : kk, ch = cin == 2 ? (kk & 1) : 0;
        const int ty = (tp * 13) >> 6, tx = tp - ty * 5;      // tp / 5 for tp < 25
        const int base = ((((2 * wave + ty) * 20 + tx + 2 * h + 8 * s)) << sh) + ch;
        unsigned ahw[4], alw[4];
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
          _Float16 hh[2], ll[2];
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            const int i8 = 2 * k2 + e2;
            const float xv = patch[base + (((((i8 >> 1) & 1) * 20) + 4 * (i8 >> 2) + (i8 & 1)) << sh)] * xs;
            h2_split(xv, hh[e2], ll[e2]);
          }
          ahw[k2] = h2_pack(hh[0], hh[1]);
          alw[k2] = h2_pack(ll[0], ll[1]);
        }
        const uint4 ah = make_uint4(ahw[0], ahw[1], ahw[2], ahw[3]), al = make_uint4(alw[0], alw[1], alw[2], alw[3]);
        acc5[mb] = mfma_h(ah, bh, acc5[mb]);
        acc5[mb] = mfma_h(ah, bl, acc5[mb]);
        acc5[mb] = mfma_h(al, bh, acc5[mb]);
      }
    }
    hbuf ^= 1;
    pbuf ^= 1;
    jb = jn;
    lit = nlit;
  }
  flush5(cur_job, reinterpret_cast<float*>(smem));      // (nothing is in flight any more)
}

struct W5Reduce {
  const float* slab[kMaxJobs];
  const H2Meta* scale[kMaxJobs];
  float* dw[kMaxJobs];
  int cin[kMaxJobs];
};
// dW1[k][co] = 2^-(e_dz + e_x) * sum over the workgroups' slabs, in order
__global__ __launch_bounds__(256) void w5_reduce_kernel(const W5Reduce rt, int nwg) {
  const int j = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 25 * rt.cin[j] * 32) return;
  float sum = 0.f;
  for (int g = 0; g < nwg; ++g) sum += rt.slab[j][(size_t)g * (50 * 32) + e];
  rt.dw[j][e] = ldexpf(sum, -rt.scale[j]->e);
}

// (the persistent grid: ugn_set_persistent_wgs, runtime.cpp)
#define g_persistent_wgs (ugn_mm::persistent_wgs())

template <int KC, int NC, int HW, int IN_POOLED, int EPI>
int launch_mm(const MmJob* jobs, const int* n, int njobs, hipStream_t st) {
  auto kern = conv_mm_kernel<KC, NC, HW, IN_POOLED, EPI>;
  constexpr int LDS = lds_bytes<KC, NC, IN_POOLED>();
  static_assert(LDS <= 163840, "LDS budget");
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { ugn_set_error("conv_mm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const void* zeros = zero_block();
  if (!zeros) { ugn_set_error("conv_mm: cannot allocate the zero block"); return UGN_EINVAL; }
  MmJobs jt;
  const int nitems = make_mm_table(jt, jobs, n, njobs, (HW / 16) * (HW / 16));
  const int grid = nitems < g_persistent_wgs ? nitems : g_persistent_wgs;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("conv_mm");
  return 0;
}

template <int KC, int NC, int HW, int EPI>
int launch_mm16(const MmJob* jobs, const int* n, int njobs, hipStream_t st) {
  auto kern = conv_mm16_kernel<KC, NC, HW, EPI>;
  constexpr int LDS = lds_bytes16<NC>();
  static_assert(LDS <= 163840, "LDS budget");
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { ugn_set_error("conv_mm16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const void* zeros = zero_block();
  if (!zeros) { ugn_set_error("conv_mm16: cannot allocate the zero block"); return UGN_EINVAL; }
  MmJobs jt;
  const int nitems = make_mm_table(jt, jobs, n, njobs, (HW / 16) * (HW / 16));
  const int grid = nitems < g_persistent_wgs ? nitems : g_persistent_wgs;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("conv_mm16");
  return 0;
}

template <int KC, int NC, int HW, int EPI, int IN_POOLED = 0>
int launch_d2(const MmJob* jobs, const int* n, int njobs, hipStream_t st) {
  auto kern = conv_d2_kernel<KC, NC, HW, EPI, IN_POOLED>;
  constexpr int LDS = d2_lds_bytes<NC>();
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { ugn_set_error("conv_d2: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const void* zeros = zero_block();
  if (!zeros) { ugn_set_error("conv_d2: cannot allocate the zero block"); return UGN_EINVAL; }
  MmJobs jt;
  const int nitems = make_mm_table(jt, jobs, n, njobs, (HW / 16) * (HW / 16));
  const int wgs = 2 * g_persistent_wgs;        // two workgroups per CU
  const int grid = nitems < wgs ? nitems : wgs;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("conv_d2");
  return 0;
}

template <int KC, int NC, int HW, int EPI, int IN_POOLED = 0>
int launch_nr(const MmJob* jobs, const int* n, int njobs, hipStream_t st) {
  auto kern = conv_nr_kernel<KC, NC, HW, EPI, IN_POOLED>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NR_LDS);
    if (e != hipSuccess) { ugn_set_error("conv_nr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const void* zeros = zero_block();
  if (!zeros) { ugn_set_error("conv_nr: cannot allocate the zero block"); return UGN_EINVAL; }
  MmJobs jt;
  const int nitems = make_mm_table(jt, jobs, n, njobs, (HW / 16) * (HW / 16));
  const int grid = nitems < g_persistent_wgs ? nitems : g_persistent_wgs;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), NR_LDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("conv_nr");
  return 0;
}


int dispatch_fwd(const MmJob* jobs, const int* n, int njobs, int hw, int cin, int cout, int pool, hipStream_t st) {
  if constexpr ((UGN_MM_D2 & 1) != 0) {
    if (cin == 32 && cout == 32 && hw == 64 && pool) return launch_d2<32, 32, 64, EPI_LRELU_POOL>(jobs, n, njobs, st);
  }
  if constexpr ((UGN_MM_D2 & 4) != 0) {
    if (cin == 32 && cout == 64 && hw == 32 && !pool) return launch_d2<32, 64, 32, EPI_LRELU>(jobs, n, njobs, st);
    if (cin == 64 && cout == 64 && hw == 32 && pool) return launch_d2<64, 64, 32, EPI_LRELU_POOL>(jobs, n, njobs, st);
  }

#define MF(KC_, NC_, HW_, P_)                                                \
  if (cin == KC_ && cout == NC_ && hw == HW_ && (pool != 0) == (P_ != 0)) {  \
    if constexpr (mm_nr(KC_, NC_, 0))                                        \
      return launch_nr<KC_, NC_ < 64 ? 64 : NC_, HW_, P_ ? EPI_LRELU_POOL : EPI_LRELU>(jobs, n, njobs, st); \
    else if constexpr (mm_tile16(KC_, NC_, 0))                               \
      return launch_mm16<KC_, NC_, HW_, P_ ? EPI_LRELU_POOL : EPI_LRELU>(jobs, n, njobs, st); \
    else                                                                     \
      return launch_mm<KC_, NC_, HW_, 0, P_ ? EPI_LRELU_POOL : EPI_LRELU>(jobs, n, njobs, st); \
  }
  MF(32, 32, 64, 1) MF(32, 64, 32, 0) MF(64, 64, 32, 1) MF(64, 128, 16, 0) MF(128, 128, 16, 0)
#undef MF
  ugn_set_error("ugn_mm_conv3x3_fwd: unsupported shape cin=%d cout=%d hw=%d pool=%d", cin, cout, hw, pool);
  return UGN_EINVAL;
}

// data gradient of the forward layer cin -> cout at hw x hw: K = cout, N = cin
int dispatch_dgrad(const MmJob* jobs, const int* n, int njobs, int hw, int cin, int cout, int unpool, bool act, hipStream_t st) {
  if constexpr ((UGN_MM_D2 & 2) != 0) {
    if (cin == 32 && cout == 32 && hw == 64 && unpool && !act)
      return launch_d2<32, 32, 64, EPI_DGRAD, 1>(jobs, n, njobs, st);
    if (UGN_D2P16 && cin == 32 && cout == 32 && hw == 64 && unpool && act) return launch_d2<32, 32, 64, EPI_DGRAD_ACT, 1>(jobs, n, njobs, st);
  }
  if constexpr ((UGN_MM_D2 & 4) != 0) {        // (GEMM K = cout, N = cin)
    if (cin == 32 && cout == 64 && hw == 32 && !unpool && !act) return launch_d2<64, 32, 32, EPI_DGRAD>(jobs, n, njobs, st);
    if (cin == 64 && cout == 128 && hw == 16 && !unpool && !act) return launch_d2<128, 64, 16, EPI_DGRAD>(jobs, n, njobs, st);
  }
  if constexpr (mm_nr(64, 64, 1)) {      // the pooled 64 -> 64 data gradient (a4 / b2) on conv_nr_kernel
    if (cin == 64 && cout == 64 && hw == 32 && unpool)
      return act ? launch_nr<64, 64, 32, EPI_DGRAD_ACT, 1>(jobs, n, njobs, st) : launch_nr<64, 64, 32, EPI_DGRAD, 1>(jobs, n, njobs, st);
  }
#define MD(CI_, CO_, HW_, U_)                                                                     \
  if (cin == CI_ && cout == CO_ && hw == HW_ && unpool == U_) {                                   \
    if constexpr (mm_nr(CO_, CI_, 1) && !U_)                                                      \
      return act ? launch_nr<CO_, CI_ < 64 ? 64 : CI_, HW_, EPI_DGRAD_ACT>(jobs, n, njobs, st)    \
                 : launch_nr<CO_, CI_ < 64 ? 64 : CI_, HW_, EPI_DGRAD>(jobs, n, njobs, st);       \
    else if constexpr (mm_tile16(CO_, CI_, 1) && !U_)                                             \
      return act ? launch_mm16<CO_, CI_, HW_, EPI_DGRAD_ACT>(jobs, n, njobs, st)                  \
                 : launch_mm16<CO_, CI_, HW_, EPI_DGRAD>(jobs, n, njobs, st);                     \
    else                                                                                          \
      return act ? launch_mm<CO_, CI_, HW_, U_, EPI_DGRAD_ACT>(jobs, n, njobs, st)                \
                 : launch_mm<CO_, CI_, HW_, U_, EPI_DGRAD>(jobs, n, njobs, st);                   \
  }
  MD(32, 32, 64, 1) MD(32, 64, 32, 0) MD(64, 64, 32, 1) MD(64, 128, 16, 0) MD(128, 128, 16, 0)
#undef MD
  ugn_set_error("ugn_mm_conv3x3_dgrad: unsupported shape cin=%d cout=%d hw=%d unpool=%d", cin, cout, hw, unpool);
  return UGN_EINVAL;
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 <-> H2 (tests, tools and the boundary of the H2 part of the path)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void absmax_kernel(const float* __restrict__ x, size_t n, H2Meta* meta) {
  __shared__ float red[4];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
  h2_publish_amax_block(meta, m, red, threadIdx.x, 4);
}
// x [npix][c] fp32 (meta: e = 0, amax = max|x|) -> y [npix][2][c]; the new exponent goes to meta_out (which may be meta_in:
// it is written by a separate one-thread kernel afterwards)
__global__ void h2_encode_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, const H2Meta* meta_in, size_t npix, int c) {
  const int e = h2_exp_for_bound(h2_true_amax(meta_in->e, meta_in->amax));
  const size_t total = npix * (size_t)c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i / c;
    const int ch = (int)(i - p * c);
    _Float16 hi, lo;
    h2_split(ldexpf(x[i], e), hi, lo);
    y[(p * 2 + 0) * c + ch] = (uint16_t)h2_bits(hi);
    y[(p * 2 + 1) * c + ch] = (uint16_t)h2_bits(lo);
  }
}
__global__ void h2_rescale_meta_kernel(H2Meta* meta) {
  const int e = h2_exp_for_bound(h2_true_amax(meta->e, meta->amax));
  const float a = ldexpf(__uint_as_float(meta->amax), e - meta->e);
  meta->e = e;
  meta->amax = __float_as_uint(a);
}
__global__ void h2_decode_kernel(const uint16_t* __restrict__ y, const H2Meta* meta, float* __restrict__ x, size_t npix, int c) {
  const int e = meta->e;
  const size_t total = npix * (size_t)c;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i / c;
    const int ch = (int)(i - p * c);
    x[i] = ldexpf(h2_half(y[(p * 2 + 0) * c + ch]) + h2_half(y[(p * 2 + 1) * c + ch]), -e);
  }
}

}  // namespace

// (ugn_mm::persistent_wgs, ugn_mm::zero_block and ugn_set_persistent_wgs: runtime.cpp -- every kernel set uses them)

#ifdef UGN_MM_STAMP
extern "C" int ugn_mm_debug_stamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mm_stamp), &p, sizeof(p));
}
#endif


extern "C" size_t ugn_mm_dgrad32_wgrad5_ws(int njobs) { return (size_t)njobs * kGrid * 50 * 32 * sizeof(float); }

/* data gradient of the pooled 32 -> 32 layer fused with the 5x5 first layer's weight gradient (see dgrad32_w5_kernel): dw5[j] =
 * dL/dW1 [5][5][cin][32] fp32; no dL/da1 tensor is produced.  scale[j]: an ugn_h2meta record the launch uses for its exponents. */
extern "C" int ugn_mm_dgrad32_wgrad5_multi(const uint16_t* const* dz, const void* const* dz_meta, const uint8_t* const* dz_idx,
                                           const uint16_t* const* wpk, const void* const* wmeta, const float* const* x,
                                           const void* const* x_meta, const uint32_t* const* a1_sign, float* const* dw5,
                                           void* const* scale, const int* n, const int* cin, int njobs, void* ws, size_t ws_bytes,
                                           void* stream) {
  UGN_REQUIRE(dz && dz_meta && dz_idx && wpk && wmeta && x && x_meta && a1_sign && dw5 && scale && n && cin && ws,
              "ugn_mm_dgrad32_wgrad5_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_mm_dgrad32_wgrad5_multi: njobs must be 1..%d (got %d)", kMaxJobs, njobs);
  UGN_REQUIRE(ws_bytes >= ugn_mm_dgrad32_wgrad5_ws(njobs), "ugn_mm_dgrad32_wgrad5_multi: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)dgrad32_w5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W5_LDS);
    if (e != hipSuccess) { ugn_set_error("dgrad32_w5: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const void* zeros = zero_block();
  if (!zeros) { ugn_set_error("dgrad32_w5: cannot allocate the zero block"); return UGN_EINVAL; }
  W5Jobs jt = {};
  W5Reduce rt = {};
  int total = 0;
  for (int j = 0; j < kMaxJobs; ++j) {
    const int jj = j < njobs ? j : njobs - 1;
    UGN_REQUIRE(dz[jj] && dz_meta[jj] && dz_idx[jj] && wpk[jj] && wmeta[jj] && x[jj] && x_meta[jj] && a1_sign[jj] && dw5[jj] && scale[jj] &&
                n[jj] > 0 && (cin[jj] == 1 || cin[jj] == 2), "ugn_mm_dgrad32_wgrad5_multi: bad job %d", jj);
    W5Job& J = jt.job[j];
    J.in = dz[jj]; J.in_idx = dz_idx[jj]; J.in_meta = (const H2Meta*)dz_meta[jj]; J.wpk = wpk[jj]; J.wmeta = (const WMeta*)wmeta[jj];
    J.x = x[jj]; J.x_meta = (const H2Meta*)x_meta[jj]; J.sign = a1_sign[jj]; J.scale = (H2Meta*)scale[jj];
    J.slab = (float*)ws + (size_t)jj * kGrid * 50 * 32; J.cin = cin[jj];
    jt.start[j] = total;
    if (j < njobs) {
      total += n[j] * 16;
      rt.slab[j] = J.slab; rt.scale[j] = J.scale; rt.dw[j] = dw5[j]; rt.cin[j] = cin[j];
    }
  }
  jt.start[kMaxJobs] = total;
  const int grid = total < g_persistent_wgs ? total : g_persistent_wgs;
  hipError_t me = hipMemsetAsync(ws, 0, ugn_mm_dgrad32_wgrad5_ws(njobs), st);     // (a workgroup writes only the jobs it met)
  if (me != hipSuccess) { ugn_set_error("dgrad32_w5: memset: %s", hipGetErrorString(me)); return (int)me; }
  hipLaunchKernelGGL(dgrad32_w5_kernel, dim3(grid), dim3(512), W5_LDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("dgrad32_w5");
  hipLaunchKernelGGL(w5_reduce_kernel, dim3((50 * 32 + 255) / 256, njobs), dim3(256), 0, st, rt, kGrid);
  UGN_CHECK_LAUNCH("w5_reduce");
  return 0;
}

extern "C" int ugn_mm_pack_multi(const float* const* w_hwio_host, uint16_t* const* wpk_host, void* const* wmeta_host,
                                 const int* cin_host, const int* cout_host, const int* dgrad_host, int njobs, void* stream) {
  UGN_REQUIRE(w_hwio_host && wpk_host && wmeta_host && cin_host && cout_host && dgrad_host, "ugn_mm_pack_multi: null pointer");
  UGN_REQUIRE(njobs >= 1 && njobs <= kPackJobs, "ugn_mm_pack_multi: njobs must be 1..%d (got %d)", kPackJobs, njobs);
  PackTable t = {};
  int maxe = 0;
  for (int j = 0; j < njobs; ++j) {
    const int ci = cin_host[j], co = cout_host[j];
    UGN_REQUIRE(w_hwio_host[j] && wpk_host[j] && wmeta_host[j], "ugn_mm_pack_multi: null pointer in job %d", j);
    UGN_REQUIRE(ci > 0 && co > 0 && ci % 32 == 0 && co % 32 == 0, "ugn_mm_pack_multi: channels must be multiples of 32 (job %d)", j);
    const int nc = (dgrad_host[j] & 1) ? ci : co;
    UGN_REQUIRE(nc == 32 || nc == 64 || nc == 128, "ugn_mm_pack_multi: %d output channels of the GEMM (32, 64 or 128; job %d)", nc, j);
    t.w[j] = w_hwio_host[j]; t.pk[j] = wpk_host[j]; t.meta[j] = (WMeta*)wmeta_host[j];
    t.amax[j] = reinterpret_cast<unsigned*>(wmeta_host[j]) + 2;
    // (bit 1 of a job's flag: the 32-column block layout of conv_mm_kernel / dgrad32_w5_kernel whatever the shape's default kernel)
    t.cin[j] = ci; t.cout[j] = co; t.dgrad[j] = (dgrad_host[j] & 1) | (dgrad_host[j] & 2);
    if (9 * ci * co > maxe) maxe = 9 * ci * co;
  }
  hipLaunchKernelGGL(mm_wstats_zero_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t, njobs);
  hipLaunchKernelGGL(mm_wstats_kernel, dim3(kStatSlices, njobs), dim3(256), 0, (hipStream_t)stream, t);
  UGN_CHECK_LAUNCH("mm_wstats");
  hipLaunchKernelGGL(mm_pack_kernel, dim3((maxe + 255) / 256, njobs), dim3(256), 0, (hipStream_t)stream, t);
  UGN_CHECK_LAUNCH("mm_pack");
  return 0;
}

extern "C" int ugn_mm_conv3x3_fwd_multi(const uint16_t* const* in, const void* const* in_meta, const uint16_t* const* wpk,
                                        const void* const* wmeta, uint16_t* const* out, uint8_t* const* out_idx,
                                        void* const* out_meta, const int* n, int njobs, int hw, int cin, int cout, int pool,
                                        void* stream) {
  UGN_REQUIRE(in && in_meta && wpk && wmeta && out && out_meta && n, "ugn_mm_conv3x3_fwd_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_mm_conv3x3_fwd_multi: njobs must be 1..%d (got %d)", kMaxJobs, njobs);
  MmJob jobs[kMaxJobs];
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(in[j] && in_meta[j] && wpk[j] && wmeta[j] && out[j] && out_meta[j] && n[j] > 0,
                "ugn_mm_conv3x3_fwd_multi: null pointer or n <= 0 in job %d", j);
    UGN_REQUIRE(!pool || (out_idx && out_idx[j]), "ugn_mm_conv3x3_fwd_multi: pool needs out_idx");
    jobs[j] = {in[j], nullptr, (const H2Meta*)in_meta[j], wpk[j], (const WMeta*)wmeta[j], out[j], pool ? out_idx[j] : nullptr,
               (H2Meta*)out_meta[j], nullptr};
  }
  return dispatch_fwd(jobs, n, njobs, hw, cin, cout, pool, (hipStream_t)stream);
}

extern "C" int ugn_mm_conv3x3_dgrad_multi(const uint16_t* const* dz, const uint8_t* const* dz_idx, const void* const* dz_meta,
                                          const uint16_t* const* wpk, const void* const* wmeta, const uint16_t* const* act,
                                          uint16_t* const* out, void* const* out_meta, const int* n, int njobs, int hw, int cin,
                                          int cout, void* stream) {
  UGN_REQUIRE(dz && dz_meta && wpk && wmeta && out && out_meta && n, "ugn_mm_conv3x3_dgrad_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_mm_conv3x3_dgrad_multi: njobs must be 1..%d (got %d)", kMaxJobs, njobs);
  MmJob jobs[kMaxJobs];
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(dz[j] && dz_meta[j] && wpk[j] && wmeta[j] && out[j] && out_meta[j] && n[j] > 0,
                "ugn_mm_conv3x3_dgrad_multi: null pointer or n <= 0 in job %d", j);
    jobs[j] = {dz[j], dz_idx ? dz_idx[j] : nullptr, (const H2Meta*)dz_meta[j], wpk[j], (const WMeta*)wmeta[j], out[j], nullptr,
               (H2Meta*)out_meta[j], act ? act[j] : nullptr};
    UGN_REQUIRE((jobs[0].in_idx != nullptr) == (jobs[j].in_idx != nullptr), "ugn_mm_conv3x3_dgrad_multi: dz_idx for all jobs or none");
    UGN_REQUIRE((jobs[0].act != nullptr) == (jobs[j].act != nullptr), "ugn_mm_conv3x3_dgrad_multi: act for all jobs or none");
  }
  return dispatch_dgrad(jobs, n, njobs, hw, cin, cout, jobs[0].in_idx != nullptr, jobs[0].act != nullptr, (hipStream_t)stream);
}

// meta <- {e = 0, amax = max|x|} of an fp32 tensor (meta must be zero on entry, like every H2Meta)
extern "C" int ugn_absmax(const float* x, size_t n, void* meta, void* stream) {
  UGN_REQUIRE(x && meta && n > 0, "ugn_absmax: null pointer or n == 0");
  const size_t blocks = (n + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, (hipStream_t)stream, x, n, (H2Meta*)meta);
  UGN_CHECK_LAUNCH("absmax");
  return 0;
}

extern "C" int ugn_h2_encode(const float* x, uint16_t* y, void* meta, size_t npix, int c, void* stream) {
  UGN_REQUIRE(x && y && meta && npix > 0 && c > 0, "ugn_h2_encode: null pointer or empty tensor");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(meta, 0, sizeof(H2Meta), st);
  if (e != hipSuccess) { ugn_set_error("ugn_h2_encode: memset: %s", hipGetErrorString(e)); return (int)e; }
  const size_t n = npix * (size_t)c;
  const size_t blocks = (n + 256 * 8 - 1) / (256 * 8);
  const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
  hipLaunchKernelGGL(absmax_kernel, dim3(grid < 256 ? grid : 256), dim3(256), 0, st, x, n, (H2Meta*)meta);
  hipLaunchKernelGGL(h2_encode_kernel, dim3(grid), dim3(256), 0, st, x, y, (const H2Meta*)meta, npix, c);
  hipLaunchKernelGGL(h2_rescale_meta_kernel, dim3(1), dim3(1), 0, st, (H2Meta*)meta);
  UGN_CHECK_LAUNCH("h2_encode");
  return 0;
}

extern "C" int ugn_h2_decode(const uint16_t* y, const void* meta, float* x, size_t npix, int c, void* stream) {
  UGN_REQUIRE(x && y && meta && npix > 0 && c > 0, "ugn_h2_decode: null pointer or empty tensor");
  const size_t n = npix * (size_t)c;
  const size_t blocks = (n + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(h2_decode_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, y,
                     (const H2Meta*)meta, x, npix, c);
  UGN_CHECK_LAUNCH("h2_decode");
  return 0;
}
